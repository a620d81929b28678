// Host entry of lvllm_paged_prefill_attention (kernel: prefill_mfma.h).
#include <algorithm>

#include "../../include/lvllm_hip.h"
#include "prefill_mfma32.h"
#include "prefill_chunk.h"

using namespace lvllm;

extern "C" int lvllm_paged_prefill_attention(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, const int32_t* query_start_loc,
    int max_query_len, int block_size, int max_num_blocks_per_seq, const float* alibi_slopes,
    int causal, int sliding_window, float softcap, int64_t q_stride, int64_t out_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int dtype, int kv_dtype, void* stream) {
  return lvllm_paged_prefill_attention_ex(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                                          num_kv_heads, scale, block_tables, seq_lens, query_start_loc,
                                          max_query_len, block_size, max_num_blocks_per_seq, alibi_slopes, causal,
                                          sliding_window, softcap, q_stride, out_stride, kv_block_stride,
                                          kv_head_stride, dtype, kv_dtype, 1.f, 1.f, 0, stream);
}

extern "C" int lvllm_paged_prefill_attention_ex(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, const int32_t* query_start_loc,
    int max_query_len, int block_size, int max_num_blocks_per_seq, const float* alibi_slopes,
    int causal, int sliding_window, float softcap, int64_t q_stride, int64_t out_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale, float v_scale,
    int64_t kv_cache_bytes, void* stream) {
  return lvllm_paged_prefill_attention_ws(out, query, key_cache, value_cache, num_seqs, num_heads, head_size,
                                          num_kv_heads, scale, block_tables, seq_lens, query_start_loc,
                                          max_query_len, block_size, max_num_blocks_per_seq, alibi_slopes, causal,
                                          sliding_window, softcap, q_stride, out_stride, kv_block_stride,
                                          kv_head_stride, dtype, kv_dtype, k_scale, v_scale, kv_cache_bytes, 0, 0,
                                          nullptr, 0, stream);
}

extern "C" int64_t lvllm_paged_prefill_workspace_bytes(int num_seqs, int num_tokens, int max_query_len, int num_heads,
                                                       int num_kv_heads, int head_size, int max_seq_len) {
  if (num_seqs <= 0 || max_query_len <= 0 || num_heads <= 0 || num_kv_heads <= 0 || num_heads % num_kv_heads != 0)
    return 0;
  if (head_size != 64 && head_size != 128) return 0;  // (the bodies that can cut their key walk)
  lvllm::PrefillParams p{};
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads; p.num_tokens = num_tokens;
  if (lvllm::chunk_kernel_takes(p, head_size, num_seqs, max_query_len))
    return lvllm::chunk_plan(num_seqs, max_query_len, num_heads, num_kv_heads, head_size, max_seq_len, num_tokens)
        .ws_bytes;
  if (lvllm::takes_mfma32(p, num_seqs, max_query_len))
    return lvllm::prefill32_plan(num_seqs, max_query_len, num_heads, num_kv_heads, head_size, max_seq_len).ws_bytes;
  return 0;
}

extern "C" int lvllm_paged_prefill_attention_ws(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, const int32_t* query_start_loc,
    int max_query_len, int block_size, int max_num_blocks_per_seq, const float* alibi_slopes,
    int causal, int sliding_window, float softcap, int64_t q_stride, int64_t out_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale, float v_scale,
    int64_t kv_cache_bytes, int num_tokens, int max_seq_len, void* workspace, int64_t workspace_bytes, void* stream) {
  LV_CHECK(num_tokens >= 0 && max_seq_len >= 0 && workspace_bytes >= 0 &&
               (workspace != nullptr || workspace_bytes == 0),
           "bad num_tokens / max_seq_len / workspace");
  LV_CHECK(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  LV_CHECK(num_seqs >= 0 && num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
           "num_heads must be a positive multiple of num_kv_heads");
  LV_CHECK(dtype == LVLLM_F16 || dtype == LVLLM_BF16, "dtype must be float16 or bfloat16");
  LV_CHECK(kv_dtype == LVLLM_KV_AUTO || kv_dtype == LVLLM_KV_FP8_E4M3, "unsupported kv_cache_dtype");
  LV_CHECK(kv_dtype == LVLLM_KV_FP8_E4M3 ? (k_scale > 0.f && v_scale > 0.f) : (k_scale == 1.f && v_scale == 1.f),
           "k_scale / v_scale: positive with an fp8 cache, 1.0 otherwise");
  LV_CHECK(block_size == 16 || block_size == 32, "Unsupported block size: " + std::to_string(block_size));
  LV_CHECK(max_query_len >= 0 && max_num_blocks_per_seq >= 0, "negative sizes");
  LV_CHECK(causal || (alibi_slopes == nullptr && sliding_window <= 0),
           "ALiBi and sliding windows are defined for causal attention only");
  if (num_seqs == 0 || max_query_len == 0) return 0;
  LV_CHECK(max_num_blocks_per_seq > 0, "query tokens without a block table");
  LV_CHECK((((uintptr_t)query | (uintptr_t)out | (uintptr_t)key_cache | (uintptr_t)value_cache) & 15) == 0 &&
               (q_stride * 2) % 16 == 0 && (out_stride * 2) % 8 == 0 &&
               (kv_block_stride * (kv_dtype == LVLLM_KV_AUTO ? 2 : 1)) % 16 == 0 &&
               (kv_head_stride * (kv_dtype == LVLLM_KV_AUTO ? 2 : 1)) % 16 == 0,
           "operands must be 16-byte aligned");
  PrefillParams p{};
  p.out = out; p.q = query; p.k_cache = key_cache; p.v_cache = value_cache;
  p.block_tables = block_tables; p.seq_lens = seq_lens; p.query_start_loc = query_start_loc;
  p.alibi_slopes = alibi_slopes;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.max_num_blocks_per_seq = max_num_blocks_per_seq;
  p.max_block = 0x7fffffff;
  if (kv_cache_bytes > 0) {  // the extent the caller states: block numbers beyond it are clamped, not followed
    LV_CHECK(kv_block_stride > 0, "kv_block_stride must be positive");
    const int64_t nb = kv_cache_bytes / (kv_block_stride * (kv_dtype == LVLLM_KV_AUTO ? 2 : 1));
    LV_CHECK(nb >= 1, "kv_cache_bytes is smaller than one block of the stated strides and element size");
    p.max_block = (int)(nb - 1 < 0x7fffffff ? nb - 1 : 0x7fffffff);
  }
  p.causal = causal ? 1 : 0;
  p.kv_fp8 = kv_dtype == LVLLM_KV_FP8_E4M3; p.k_scale = k_scale; p.v_scale = v_scale;
  p.sliding_window = sliding_window; p.scale = scale; p.softcap = softcap;
  p.q_stride = q_stride; p.out_stride = out_stride;
  p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  p.max_seq_len = max_seq_len; p.num_tokens = num_tokens; p.workspace = workspace; p.workspace_bytes = workspace_bytes;
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

#ifndef LVLLM_VARLEN_BLOCK_PAD
#define LVLLM_VARLEN_BLOCK_PAD 1
#endif
constexpr int kVarlenBS = 16;  // block size of the scratch tiles (16: every K/V tile load of the kernel is one contiguous KiB)

// Sequence s owns the scratch blocks cu_seqlens[s] / BS + s + i, i < ceil(len_s / BS): consecutive
// sequences never overlap (floor((a + len) / BS) - floor(a / BS) + 1 >= ceil(len / BS)) and the
// total is at most T / BS + num_seqs, so the placement needs neither a scan nor a table.
//
// grid (block of the longest sequence, kv head, sequence), 256 threads: one (block, head)
// tile of K and of V per workgroup.  K chunks [tok][d8] -> [d8][tok][8] move as 16 bytes, written
// in destination order (the tile is BS*D*2 contiguous bytes).  V is a true transpose [tok][d] ->
// [d][tok]: through LDS, so that both the reads and the writes are 16-byte and contiguous (a
// direct scatter of 2-byte elements ran at a fifth of this speed).
__global__ __launch_bounds__(256) void varlen_pack_kernel(
    const uint16_t* __restrict__ key, const uint16_t* __restrict__ value, uint16_t* __restrict__ key_cache,
    uint16_t* __restrict__ value_cache, const int32_t* __restrict__ cu_seqlens, const int num_kv_heads,
    const int head_size, const int64_t key_stride, const int64_t value_stride, const int64_t block_stride) {
  extern __shared__ __attribute__((aligned(16))) uint16_t vt[];  // [head_size][kVarlenBS + 8] (padded rows)
  constexpr int ROW = kVarlenBS + 8;
  const int seq = blockIdx.z, head = blockIdx.y, blk = blockIdx.x;
  const int beg = cu_seqlens[seq];
  const int len = cu_seqlens[seq + 1] - beg;
  const int tok0 = blk * kVarlenBS;
  if (tok0 >= len) return;
  const int chunks_per_head = head_size >> 3;
  const int64_t block = beg / kVarlenBS + seq + blk;
  uint16_t* ktile = key_cache + block * block_stride + (int64_t)head * head_size * kVarlenBS;
  uint16_t* vtile = value_cache + block * block_stride + (int64_t)head * head_size * kVarlenBS;
  const int nchunks = chunks_per_head * kVarlenBS;
  // K: destination order (d8, tok)
  for (int i = threadIdx.x; i < nchunks; i += blockDim.x) {
    const int d8 = i / kVarlenBS, tok = i - d8 * kVarlenBS;
    uint4 kv = uint4{0, 0, 0, 0};
    if (tok0 + tok < len)
      kv = *reinterpret_cast<const uint4*>(key + (int64_t)(beg + tok0 + tok) * key_stride + head * head_size + d8 * 8);
    *reinterpret_cast<uint4*>(ktile + (int64_t)i * 8) = kv;
  }
  // V: source order (tok, d8) into LDS transposed ...
  for (int i = threadIdx.x; i < nchunks; i += blockDim.x) {
    const int tok = i / chunks_per_head, d8 = i - tok * chunks_per_head;
    uint4 vv = uint4{0, 0, 0, 0};
    if (tok0 + tok < len)
      vv = *reinterpret_cast<const uint4*>(value + (int64_t)(beg + tok0 + tok) * value_stride + head * head_size + d8 * 8);
    const uint16_t* ve = reinterpret_cast<const uint16_t*>(&vv);
#pragma unroll
    for (int e = 0; e < 8; ++e) vt[(d8 * 8 + e) * ROW + tok] = ve[e];
  }
  __syncthreads();
  // ... and out in destination order (d, 8 tokens)
  for (int i = threadIdx.x; i < nchunks; i += blockDim.x) {
    const int d = i / (kVarlenBS / 8), t8 = i - d * (kVarlenBS / 8);
    *reinterpret_cast<uint4*>(vtile + (int64_t)i * 8) = *reinterpret_cast<const uint4*>(vt + d * ROW + t8 * 8);
  }
}

struct VarlenLayout {
  int64_t cache_elems, off_v, total;
  int64_t block_stride;  // elements between scratch blocks: one block + 1/32 of it (see CacheConfig.block_pad_bytes:
                         // equal-length sequences start a power of two apart otherwise)
  int max_blocks_per_seq, num_blocks;
};
static VarlenLayout varlen_layout(int num_tokens, int num_seqs, int max_seq_len, int num_kv_heads,
                                  int head_size) {
  VarlenLayout L{};
  L.max_blocks_per_seq = (max_seq_len + kVarlenBS - 1) / kVarlenBS;
  if (L.max_blocks_per_seq < 1) L.max_blocks_per_seq = 1;
  L.num_blocks = num_tokens / kVarlenBS + num_seqs + 1;
  const int64_t block_elems = (int64_t)kVarlenBS * num_kv_heads * head_size;
  L.block_stride = block_elems + (LVLLM_VARLEN_BLOCK_PAD ? ((block_elems / 32 + 127) / 128) * 128 : 0);
  L.cache_elems = (int64_t)L.num_blocks * L.block_stride;
  auto up = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
  L.off_v = up(L.cache_elems * 2);
  L.total = L.off_v + up(L.cache_elems * 2);
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
  LV_CHECK(num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
           "num_heads must be a positive multiple of num_kv_heads");
  LV_CHECK(head_size % 8 == 0, "head_size must be a multiple of 8");
  LV_CHECK((((uintptr_t)key | (uintptr_t)value) & 15) == 0 && (k_stride * 2) % 16 == 0 && (v_stride * 2) % 16 == 0,
           "key/value must be 16-byte aligned");
  if (num_tokens == 0 || num_seqs == 0 || max_seq_len == 0) return 0;
  const VarlenLayout L = varlen_layout(num_tokens, num_seqs, max_seq_len, num_kv_heads, head_size);
  LV_CHECK(workspace != nullptr && workspace_bytes >= L.total && ((uintptr_t)workspace & 255) == 0,
           "workspace too small or misaligned (lvllm_varlen_attention_workspace_bytes)");
  hipStream_t s = (hipStream_t)stream;
  if (lvllm::tuning().varlen_dense) {
    // long plain sequences: the 32x32 body reads the caller's rows itself (prefill_mfma32.h, DENSE): no pack pass
    PrefillParams p{};
    p.out = out; p.q = query; p.k_cache = key; p.v_cache = value; p.query_start_loc = cu_seqlens;
    p.num_heads = num_heads; p.num_kv_heads = num_kv_heads; p.max_num_blocks_per_seq = 1; p.max_block = 0;
    p.causal = causal ? 1 : 0; p.k_scale = p.v_scale = 1.f;
    p.alibi_slopes = alibi_slopes; p.sliding_window = sliding_window; p.scale = scale; p.softcap = softcap;
    p.q_stride = q_stride; p.out_stride = out_stride; p.num_tokens = num_tokens;
    p.dense_k_stride = k_stride; p.dense_v_stride = v_stride;
    p.dense_k_bytes = ((int64_t)(num_tokens - 1) * k_stride + (int64_t)num_kv_heads * head_size) * 2;
    p.dense_v_bytes = ((int64_t)(num_tokens - 1) * v_stride + (int64_t)num_kv_heads * head_size) * 2;
    if (dense_mfma32_takes(p, head_size, num_seqs, max_seq_len)) {
      LV_CHECK((((uintptr_t)query | (uintptr_t)out) & 15) == 0 && (q_stride * 2) % 16 == 0 && (out_stride * 2) % 8 == 0,
               "operands must be 16-byte aligned");
      const int rc = dtype == LVLLM_BF16 ? launch_prefill_mfma32_dense<BF16>(p, head_size, num_seqs, max_seq_len, s)
                                         : launch_prefill_mfma32_dense<F16>(p, head_size, num_seqs, max_seq_len, s);
      if (rc) return rc;
      LV_LAUNCH_CHECK();
      return 0;
    }
  }
  char* ws = (char*)workspace;
  void* k_cache = ws;
  void* v_cache = ws + L.off_v;
  const size_t smem = (size_t)head_size * (kVarlenBS + 8) * 2;
  hipLaunchKernelGGL(varlen_pack_kernel, dim3(L.max_blocks_per_seq, num_kv_heads, num_seqs), dim3(256), smem, s,
                     (const uint16_t*)key, (const uint16_t*)value, (uint16_t*)k_cache, (uint16_t*)v_cache,
                     cu_seqlens, num_kv_heads, head_size, k_stride, v_stride, L.block_stride);
  LV_LAUNCH_CHECK();
  const int64_t head_stride = (int64_t)head_size * kVarlenBS;
  // block_tables = seq_lens = nullptr: arithmetic placement, context == chunk
  return lvllm_paged_prefill_attention(
      out, query, k_cache, v_cache, num_seqs, num_heads, head_size, num_kv_heads, scale, nullptr, nullptr,
      cu_seqlens, max_seq_len, kVarlenBS, L.max_blocks_per_seq, alibi_slopes, causal, sliding_window,
      softcap, q_stride, out_stride, L.block_stride, head_stride, dtype, LVLLM_KV_AUTO, stream);
}
