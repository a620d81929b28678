// Host entry of lvllm_paged_prefill_attention (kernel: prefill_mfma.h).
#include "../../include/lvllm_hip.h"
#include "prefill_mfma.h"

using namespace lvllm;

extern "C" int lvllm_paged_prefill_attention(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, const int32_t* query_start_loc,
    int max_query_len, int block_size, int max_num_blocks_per_seq, const float* alibi_slopes,
    int sliding_window, float softcap, int64_t q_stride, int64_t out_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int dtype, int kv_dtype, void* stream) {
  LV_CHECK(num_seqs >= 0 && num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
           "num_heads must be a positive multiple of num_kv_heads");
  LV_CHECK(dtype == LVLLM_F16 || dtype == LVLLM_BF16, "dtype must be float16 or bfloat16");
  LV_CHECK(kv_dtype == LVLLM_KV_AUTO, "fp8 kv cache is not built in this round (kv_cache_dtype must be 'auto')");
  LV_CHECK(block_size == 16 || block_size == 32, "Unsupported block size: " + std::to_string(block_size));
  LV_CHECK(max_query_len >= 0 && max_num_blocks_per_seq >= 0, "negative sizes");
  if (num_seqs == 0 || max_query_len == 0) return 0;
  LV_CHECK(max_num_blocks_per_seq > 0, "query tokens without a block table");
  LV_CHECK((((uintptr_t)query | (uintptr_t)out | (uintptr_t)key_cache | (uintptr_t)value_cache) & 15) == 0 &&
               (q_stride * 2) % 16 == 0 && (out_stride * 2) % 8 == 0 && (kv_block_stride * 2) % 16 == 0 &&
               (kv_head_stride * 2) % 16 == 0,
           "operands must be 16-byte aligned");
  PrefillParams p{};
  p.out = out; p.q = query; p.k_cache = key_cache; p.v_cache = value_cache;
  p.block_tables = block_tables; p.seq_lens = seq_lens; p.query_start_loc = query_start_loc;
  p.alibi_slopes = alibi_slopes;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.max_num_blocks_per_seq = max_num_blocks_per_seq;
  p.sliding_window = sliding_window; p.scale = scale; p.softcap = softcap;
  p.q_stride = q_stride; p.out_stride = out_stride;
  p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (dtype == LVLLM_BF16)
    rc = launch_prefill_hs<BF16>(p, head_size, block_size, num_seqs, max_query_len, s);
  else
    rc = launch_prefill_hs<F16>(p, head_size, block_size, num_seqs, max_query_len, s);
  if (rc) return rc;
  LV_LAUNCH_CHECK();
  return 0;
}
