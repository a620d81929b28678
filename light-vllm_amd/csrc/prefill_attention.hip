// Host entry of lvllm_paged_prefill_attention (kernel: prefill_mfma.h).
#include "../../include/lvllm_hip.h"
#include "prefill_mfma.h"

using namespace lvllm;

extern "C" int lvllm_paged_prefill_attention(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, const int32_t* query_start_loc,
    int max_query_len, int block_size, int max_num_blocks_per_seq, const float* alibi_slopes,
    int causal, int sliding_window, float softcap, int64_t q_stride, int64_t out_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int dtype, int kv_dtype, void* stream) {
  LV_CHECK(num_seqs >= 0 && num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
           "num_heads must be a positive multiple of num_kv_heads");
  LV_CHECK(dtype == LVLLM_F16 || dtype == LVLLM_BF16, "dtype must be float16 or bfloat16");
  LV_CHECK(kv_dtype == LVLLM_KV_AUTO, "fp8 kv cache is not built in this round (kv_cache_dtype must be 'auto')");
  LV_CHECK(block_size == 16 || block_size == 32, "Unsupported block size: " + std::to_string(block_size));
  LV_CHECK(max_query_len >= 0 && max_num_blocks_per_seq >= 0, "negative sizes");
  LV_CHECK(causal || (alibi_slopes == nullptr && sliding_window <= 0),
           "ALiBi and sliding windows are defined for causal attention only");
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
  p.causal = causal ? 1 : 0;
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

// ---- dense varlen attention: pack K/V into paged tiles in the workspace, then the kernel above ----
namespace lvllm {

constexpr int kVarlenBS = 32;  // block size of the scratch tiles

// One workgroup.  Phase 1: every sequence takes ceil(len / BS) consecutive scratch blocks from a
// shared counter (any disjoint placement is as good as any other: results do not depend on it).
// Phase 2: slot of every token.  Block tables are padded with the sequence's first block.
__global__ __launch_bounds__(1024) void varlen_setup_kernel(const int32_t* __restrict__ cu_seqlens,
                                                            int num_seqs, int num_tokens,
                                                            int max_blocks_per_seq,
                                                            int32_t* __restrict__ block_tables,
                                                            int32_t* __restrict__ seq_lens,
                                                            int64_t* __restrict__ slot_mapping) {
  __shared__ int next_block;
  if (threadIdx.x == 0) next_block = 0;
  __syncthreads();
  for (int s = threadIdx.x; s < num_seqs; s += blockDim.x) {
    const int len = cu_seqlens[s + 1] - cu_seqlens[s];
    const int nblk = (len + kVarlenBS - 1) / kVarlenBS;
    const int first = atomicAdd(&next_block, nblk);
    seq_lens[s] = len;
    int32_t* row = block_tables + (int64_t)s * max_blocks_per_seq;
    for (int i = 0; i < max_blocks_per_seq; ++i) row[i] = first + (i < nblk ? i : 0);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < num_tokens; t += blockDim.x) {
    int lo = 0, hi = num_seqs;  // last s with cu_seqlens[s] <= t
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (cu_seqlens[mid] <= t) lo = mid; else hi = mid;
    }
    const int pos = t - cu_seqlens[lo];
    const int len = cu_seqlens[lo + 1] - cu_seqlens[lo];
    const int first = block_tables[(int64_t)lo * max_blocks_per_seq];
    slot_mapping[t] = pos < len ? (int64_t)(first + pos / kVarlenBS) * kVarlenBS + pos % kVarlenBS : -1;
  }
}

struct VarlenLayout {
  int64_t cache_elems, off_v, off_tables, off_lens, off_slots, total;
  int max_blocks_per_seq, num_blocks;
};
static VarlenLayout varlen_layout(int num_tokens, int num_seqs, int max_seq_len, int num_kv_heads,
                                  int head_size) {
  VarlenLayout L{};
  L.max_blocks_per_seq = (max_seq_len + kVarlenBS - 1) / kVarlenBS;
  if (L.max_blocks_per_seq < 1) L.max_blocks_per_seq = 1;
  // sum of ceil(len_i / BS) <= T / BS + num_seqs
  L.num_blocks = num_tokens / kVarlenBS + num_seqs + 1;
  L.cache_elems = (int64_t)L.num_blocks * kVarlenBS * num_kv_heads * head_size;
  auto up = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
  L.off_v = up(L.cache_elems * 2);
  L.off_tables = L.off_v + up(L.cache_elems * 2);
  L.off_lens = L.off_tables + up((int64_t)num_seqs * L.max_blocks_per_seq * 4);
  L.off_slots = L.off_lens + up((int64_t)num_seqs * 4);
  L.total = L.off_slots + up((int64_t)num_tokens * 8);
  return L;
}

}  // namespace lvllm

extern "C" int64_t lvllm_varlen_attention_workspace_bytes(int num_tokens, int num_seqs, int max_seq_len,
                                                          int num_kv_heads, int head_size) {
  return varlen_layout(num_tokens, num_seqs, max_seq_len, num_kv_heads, head_size).total;
}

extern "C" int lvllm_varlen_attention(
    void* out, const void* query, const void* key, const void* value, const int32_t* cu_seqlens,
    int num_tokens, int num_seqs, int max_seq_len, int num_heads, int num_kv_heads, int head_size,
    float scale, int causal, const float* alibi_slopes, int sliding_window, float softcap,
    int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t out_stride, int dtype,
    void* workspace, int64_t workspace_bytes, void* stream) {
  LV_CHECK(num_tokens >= 0 && num_seqs >= 0 && max_seq_len >= 0, "negative sizes");
  LV_CHECK(dtype == LVLLM_F16 || dtype == LVLLM_BF16, "dtype must be float16 or bfloat16");
  LV_CHECK(head_size % 8 == 0, "head_size must be a multiple of 8");
  if (num_tokens == 0 || num_seqs == 0) return 0;
  const VarlenLayout L = varlen_layout(num_tokens, num_seqs, max_seq_len, num_kv_heads, head_size);
  LV_CHECK(workspace != nullptr && workspace_bytes >= L.total && ((uintptr_t)workspace & 255) == 0,
           "workspace too small or misaligned (lvllm_varlen_attention_workspace_bytes)");
  char* ws = (char*)workspace;
  void* k_cache = ws;
  void* v_cache = ws + L.off_v;
  int32_t* tables = (int32_t*)(ws + L.off_tables);
  int32_t* lens = (int32_t*)(ws + L.off_lens);
  int64_t* slots = (int64_t*)(ws + L.off_slots);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(varlen_setup_kernel, dim3(1), dim3(1024), 0, s, cu_seqlens, num_seqs, num_tokens,
                     L.max_blocks_per_seq, tables, lens, slots);
  LV_LAUNCH_CHECK();
  if (int rc = lvllm_reshape_and_cache(key, value, k_cache, v_cache, slots, num_tokens, num_kv_heads,
                                       head_size, kVarlenBS, 8, k_stride, v_stride, dtype,
                                       LVLLM_KV_AUTO, 1.f, 1.f, stream))
    return rc;
  const int64_t head_stride = (int64_t)head_size * kVarlenBS;
  return lvllm_paged_prefill_attention(
      out, query, k_cache, v_cache, num_seqs, num_heads, head_size, num_kv_heads, scale, tables, lens,
      cu_seqlens, max_seq_len, kVarlenBS, L.max_blocks_per_seq, alibi_slopes, causal, sliding_window,
      softcap, q_stride, out_stride, head_stride * num_kv_heads, head_stride, dtype, LVLLM_KV_AUTO, stream);
}
