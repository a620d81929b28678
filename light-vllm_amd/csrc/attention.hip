// C-ABI entry points of paged_attention_v1 / paged_attention_v2 (see attention_mfma.h for
// the MI355X kernel design), the v2 partition reduce and the generic kernel
// for fp32 caches.  Reference: csrc/attention/attention_kernels.cu:86-997.
#include <float.h>
#include <stdlib.h>

#include "attention_params.h"
#include "common.h"

namespace lvllm {

// defined in attention_bf16.hip / attention_f16.hip (one translation unit per
// element type keeps the instantiation ladder compiling in parallel)
template <typename T>
int launch_mfma_hs(const AttnParams& p, int head_size, int block_size, int num_seqs,
                   int num_parts, int max_tokens_per_wg, hipStream_t stream);

// ---------------------------------------------------------------------------
// Generic kernel: any element type (fp32 included), any head size, any block
// size.  One workgroup (256 threads) per (query head, sequence, partition),
// logits in LDS, same math as above without MFMA.  It serves fp32 caches
// and shapes the MFMA kernel is not instantiated for.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void paged_attn_generic_kernel(const AttnParams p,
                                                                 const int head_size,
                                                                 const int block_size,
                                                                 const int chunk_tokens) {
  using S = typename T::store_t;
  constexpr int X = 16 / sizeof(S);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* logits = reinterpret_cast<float*>(smem_raw);  // [chunk_tokens]
  float* qs = logits + chunk_tokens;                   // [head_size]
  __shared__ float red[16];

  const int head = blockIdx.x, seq = blockIdx.y, part = blockIdx.z;
  const int G = p.num_heads / p.num_kv_heads;
  const int kvh = head / G;
  const int seq_len = p.seq_lens[seq];
  int t0 = 0, t1 = seq_len;
  if (p.partitioned && !split_range(seq_len, p.num_splits, part, &t0, &t1, p.split_tiles)) return;
  const int32_t* block_table = p.block_tables + (int64_t)seq * p.max_num_blocks_per_seq;
  const S* kc = (const S*)p.k_cache + (int64_t)kvh * p.kv_head_stride;
  const S* vc = (const S*)p.v_cache + (int64_t)kvh * p.kv_head_stride;
  const float alibi = p.alibi_slopes ? p.alibi_slopes[head] : 0.f;
  // 16-byte accesses when the layout allows them (caches, strides and blocks on 16-byte boundaries)
  const bool vec16 = head_size % X == 0 && block_size % X == 0 &&
                     ((((uintptr_t)p.k_cache | (uintptr_t)p.v_cache) & 15) == 0) &&
                     (p.kv_block_stride * sizeof(S)) % 16 == 0 && (p.kv_head_stride * sizeof(S)) % 16 == 0;

  const S* qrow = (const S*)p.q + (int64_t)seq * p.q_stride + (int64_t)head * head_size;
  for (int d = threadIdx.x; d < head_size; d += blockDim.x) qs[d] = T::to_float(qrow[d]);
  __syncthreads();

  // The context is walked in chunks of chunk_tokens logits (online softmax
  // across chunks) so LDS use does not grow with the sequence length.
  float m_run = -FLT_MAX, l_run = 0.f;
  // each thread owns output elements d = threadIdx.x, threadIdx.x + 256, ... (head_size <= 512)
  float o0 = 0.f, o1 = 0.f;
  for (int cs = t0; cs < t1; cs += chunk_tokens) {
    const int ce = min(t1, cs + chunk_tokens);
    float m_loc = -FLT_MAX;
    for (int tok = cs + threadIdx.x; tok < ce; tok += blockDim.x) {
      const int64_t bn = min((uint32_t)block_table[tok / block_size], (uint32_t)p.max_block);
      const int off = tok % block_size;
      const S* kb = kc + bn * p.kv_block_stride + off * X;
      float dot = 0.f;
      if (vec16) {  // one 16-byte chunk (X head-dim values of this token) per load
        for (int dx = 0; dx < head_size / X; ++dx) {
          const uint4 raw = *reinterpret_cast<const uint4*>(kb + (int64_t)dx * block_size * X);
          const S* ke = reinterpret_cast<const S*>(&raw);
#pragma unroll
          for (int e = 0; e < X; ++e) dot += qs[dx * X + e] * T::to_float(ke[e]);
        }
      } else {
        for (int d = 0; d < head_size; ++d)
          dot += qs[d] * T::to_float(kb[(d / X) * block_size * X + (d % X)]);
      }
      float x = dot * p.scale;
      x += (alibi != 0.f) ? alibi * (float)(tok - seq_len + 1) : 0.f;
      // a token of a block this head does not attend: logit -FLT_MAX, no contribution
      // (attention_kernels.cu:234-247)
      if (p.bs_vert_stride > 1 && !blocksparse_attended(p, tok, seq_len, head, kvh, block_size)) x = -FLT_MAX;
      logits[tok - cs] = x;
      m_loc = fmaxf(m_loc, x);
    }
    m_loc = wave_max(m_loc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m_loc;
    __syncthreads();
    float m_new = m_run;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m_new = fmaxf(m_new, red[w]);
    __syncthreads();
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    for (int i = threadIdx.x; i < ce - cs; i += blockDim.x) {
      const float e = logits[i] == -FLT_MAX ? 0.f : __expf(logits[i] - m_new);
      // probabilities are rounded to T before P.V (attention_kernels.cu:398-400)
      logits[i] = T::to_float(T::from_float(e));
      psum += e;
    }
    psum = block_sum(psum, red);
    l_run = l_run * alpha + psum;
    o0 *= alpha;
    o1 *= alpha;
    __syncthreads();
    if (vec16) {
      // a thread owns V rows d0 (and d1): per cache block it reads the row's block_size tokens as 16-byte
      // pieces (the row of a block is contiguous) instead of one element per token
      const int d0 = threadIdx.x, d1 = threadIdx.x + 256;
      for (int bstart = (cs / block_size) * block_size; bstart < ce; bstart += block_size) {
        const int64_t bn = min((uint32_t)block_table[bstart / block_size], (uint32_t)p.max_block);
        const S* vb = vc + bn * p.kv_block_stride;
        for (int o = 0; o < block_size; o += X) {
          const int tok = bstart + o;
          if (tok + X <= cs || tok >= ce) continue;
          uint4 r0 = uint4{0, 0, 0, 0}, r1 = uint4{0, 0, 0, 0};
          if (d0 < head_size) r0 = *reinterpret_cast<const uint4*>(vb + (int64_t)d0 * block_size + o);
          if (d1 < head_size) r1 = *reinterpret_cast<const uint4*>(vb + (int64_t)d1 * block_size + o);
          const S* e0 = reinterpret_cast<const S*>(&r0);
          const S* e1 = reinterpret_cast<const S*>(&r1);
#pragma unroll
          for (int e = 0; e < X; ++e) {
            const int t = tok + e;
            if (t < cs || t >= ce) continue;  // V beyond the context may hold anything: never multiplied
            const float pr = logits[t - cs];
            o0 += pr * T::to_float(e0[e]);
            o1 += pr * T::to_float(e1[e]);
          }
        }
      }
    } else {
      for (int tok = cs; tok < ce; ++tok) {
        const int64_t bn = min((uint32_t)block_table[tok / block_size], (uint32_t)p.max_block);
        const int off = tok % block_size;
        const S* vb = vc + bn * p.kv_block_stride + off;
        const float pr = logits[tok - cs];
        const int d0 = threadIdx.x, d1 = threadIdx.x + 256;
        if (d0 < head_size) o0 += pr * T::to_float(vb[(int64_t)d0 * block_size]);
        if (d1 < head_size) o1 += pr * T::to_float(vb[(int64_t)d1 * block_size]);
      }
    }
    __syncthreads();
  }
  const int P = p.partitioned ? p.max_num_partitions : 1;
  const float inv = __fdividef(1.f, l_run + 1e-6f);
  const int64_t row = ((int64_t)seq * p.num_heads + head) * P + (p.partitioned ? part : 0);
  const int d0 = threadIdx.x, d1 = threadIdx.x + 256;
  if (d0 < head_size) reinterpret_cast<S*>(p.out)[row * head_size + d0] = T::from_float(o0 * inv);
  if (d1 < head_size) reinterpret_cast<S*>(p.out)[row * head_size + d1] = T::from_float(o1 * inv);
  if (p.partitioned && threadIdx.x == 0) {
    p.max_logits[row] = m_run;
    p.exp_sums[row] = l_run;
  }
}

template <typename T>
__global__ void paged_attn_v2_reduce_generic_kernel(typename T::store_t* __restrict__ out,
                                                    const float* __restrict__ exp_sums,
                                                    const float* __restrict__ max_logits,
                                                    const typename T::store_t* __restrict__ tmp_out,
                                                    const int32_t* __restrict__ seq_lens,
                                                    const int max_num_partitions,
                                                    const int num_splits, const int split_tiles,
                                                    const int num_rows, const int num_heads,
                                                    const int D) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= num_rows) return;
  const int lane = threadIdx.x & 63;
  const int seq_len = seq_lens[row / num_heads];
  const int np = num_nonempty_splits(seq_len, num_splits, split_tiles);
  typename T::store_t* o = out + (int64_t)row * D;
  const typename T::store_t* tmp = tmp_out + (int64_t)row * max_num_partitions * D;
  if (np <= 1) {  // one share: copy through (attention_kernels.cu:582-594); empty context: zeros
    for (int d = lane; d < D; d += 64) o[d] = np == 1 ? tmp[d] : T::from_float(0.f);
    return;
  }
  const float* ml = max_logits + (int64_t)row * max_num_partitions;
  const float* es = exp_sums + (int64_t)row * max_num_partitions;
  float M = -FLT_MAX;
  for (int j = 0; j < np; ++j) M = fmaxf(M, ml[j]);
  float gsum = 0.f;
  for (int j = 0; j < np; ++j) gsum += es[j] * expf(ml[j] - M);
  const float inv = __fdividef(1.0f, gsum + 1e-6f);
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < np; ++j)
      acc += T::to_float(tmp[(int64_t)j * D + d]) * (es[j] * expf(ml[j] - M)) * inv;
    o[d] = T::from_float(acc);
  }
}

static bool mfma_head_size(int d) {
  return d == 64 || d == 80 || d == 96 || d == 112 || d == 120 || d == 128 || d == 192 ||
         d == 256;
}

template <typename T>
static int launch_generic(const AttnParams& p, int head_size, int block_size, int num_seqs,
                          int num_parts, hipStream_t stream) {
  LV_CHECK(head_size <= 512, "Unsupported head size: " + std::to_string(head_size));
  const int chunk = 2048;
  const size_t smem = (size_t)(chunk + head_size) * sizeof(float);
  hipLaunchKernelGGL((paged_attn_generic_kernel<T>), dim3(p.num_heads, num_seqs, num_parts),
                     dim3(256), smem, stream, p, head_size, block_size, chunk);
  return 0;
}

static int check_common(int num_seqs, int num_heads, int head_size, int num_kv_heads,
                        int block_size, int dtype, int kv_dtype, float k_scale, float v_scale,
                        int blocksparse_vert_stride) {
  LV_CHECK(num_seqs >= 0 && num_heads > 0 && num_kv_heads > 0 && num_heads % num_kv_heads == 0,
           "num_heads must be a positive multiple of num_kv_heads");
  LV_CHECK(dtype == LVLLM_F32 || dtype == LVLLM_F16 || dtype == LVLLM_BF16, "unsupported dtype");
  LV_CHECK(kv_dtype == LVLLM_KV_AUTO || kv_dtype == LVLLM_KV_FP8_E4M3, "unsupported kv_cache_dtype");
  if (kv_dtype == LVLLM_KV_AUTO) {
    LV_CHECK(k_scale == 1.0f && v_scale == 1.0f, "k_scale/v_scale must be 1.0 with kv_cache_dtype 'auto'");
  } else {
    LV_CHECK(dtype == LVLLM_F16 || dtype == LVLLM_BF16, "fp8 kv cache needs float16 or bfloat16 queries");
    LV_CHECK(block_size == 16 || block_size == 32, "fp8 kv cache: block size must be 16 or 32");
    LV_CHECK(head_size % 16 == 0, "fp8 kv cache: head size must be a multiple of 16");
    LV_CHECK(k_scale > 0.f && v_scale > 0.f, "fp8 kv cache: scales must be positive");
  }
  LV_CHECK(blocksparse_vert_stride <= 1 || kv_dtype == LVLLM_KV_AUTO,
           "block-sparse attention over an fp8 kv cache is not supported");
  LV_CHECK(block_size == 8 || block_size == 16 || block_size == 32,
           "Unsupported block size: " + std::to_string(block_size));
  LV_CHECK(mfma_head_size(head_size), "Unsupported head size: " + std::to_string(head_size));
  return 0;
}

// Largest block number the caches can hold, from the extent the caller states (kv_cache_bytes = bytes
// of the key cache == bytes of the value cache; 0: not stated, numbers are taken as they come).  Also
// rejects strides that put a kv head outside its block.
static int cache_extent(int64_t kv_cache_bytes, int64_t kv_block_stride, int64_t kv_head_stride,
                        int num_kv_heads, int head_size, int block_size, int elem_bytes, int* max_block) {
  LV_CHECK(kv_block_stride > 0 && kv_head_stride > 0, "kv cache strides must be positive");
  LV_CHECK((int64_t)(num_kv_heads - 1) * kv_head_stride + (int64_t)head_size * block_size <= kv_block_stride,
           "kv_head_stride / kv_block_stride do not describe [num_blocks, num_kv_heads, head_size * block_size]");
  *max_block = 0x7fffffff;
  if (kv_cache_bytes <= 0) return 0;
  const int64_t nb = kv_cache_bytes / (kv_block_stride * elem_bytes);
  LV_CHECK(nb >= 1, "kv_cache_bytes is smaller than one block of the stated strides and element size");
  *max_block = (int)(nb - 1 < 0x7fffffff ? nb - 1 : 0x7fffffff);
  return 0;
}

}  // namespace lvllm

using namespace lvllm;

extern "C" int lvllm_paged_attention_v1(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int block_size,
    int max_seq_len, int max_num_blocks_per_seq, const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int dtype, int kv_dtype, float k_scale, float v_scale, int tp_rank,
    int blocksparse_local_blocks, int blocksparse_vert_stride,
    int blocksparse_block_size, int blocksparse_head_sliding_step, int64_t kv_cache_bytes, void* stream) {
  if (int rc = check_common(num_seqs, num_heads, head_size, num_kv_heads, block_size, dtype,
                            kv_dtype, k_scale, v_scale, blocksparse_vert_stride))
    return rc;
  int max_block = 0;
  if (int rc = cache_extent(kv_cache_bytes, kv_block_stride, kv_head_stride, num_kv_heads, head_size, block_size,
                            kv_dtype == LVLLM_KV_FP8_E4M3 ? 1 : (dtype == LVLLM_F32 ? 4 : 2), &max_block))
    return rc;
  const bool bsparse = blocksparse_vert_stride > 1;
  LV_CHECK(!bsparse || (blocksparse_block_size > 0 && blocksparse_block_size % block_size == 0),
           "blocksparse_block_size must be a positive multiple of block_size");
  if (num_seqs == 0) return 0;
  if (max_num_blocks_per_seq <= 0) {
    // no block table at all: every context is empty, the result is zeros
    const size_t bytes = (size_t)num_seqs * num_heads * head_size * (dtype == LVLLM_F32 ? 4 : 2);
    if (hipMemsetAsync(out, 0, bytes, (hipStream_t)stream) != hipSuccess) LV_CHECK(false, "hipMemsetAsync failed");
    return 0;
  }
  AttnParams p{};
  p.bs_vert_stride = blocksparse_vert_stride; p.bs_local_blocks = blocksparse_local_blocks;
  p.bs_block_size = blocksparse_block_size; p.bs_head_sliding_step = blocksparse_head_sliding_step;
  p.tp_rank = tp_rank;
  p.out = out; p.exp_sums = nullptr; p.max_logits = nullptr;
  p.q = query; p.k_cache = key_cache; p.v_cache = value_cache;
  p.block_tables = block_tables; p.seq_lens = seq_lens; p.alibi_slopes = alibi_slopes;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.max_num_blocks_per_seq = max_num_blocks_per_seq; p.max_num_partitions = 1;
  p.partitioned = 0; p.scale = scale; p.max_block = max_block;
  p.q_stride = q_stride; p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  p.kv_fp8 = kv_dtype == LVLLM_KV_FP8_E4M3; p.k_scale = k_scale; p.v_scale = v_scale;
  hipStream_t s = (hipStream_t)stream;
  const int kvb = p.kv_fp8 ? 1 : 2;
  const bool vec_ok = (((uintptr_t)query | (uintptr_t)key_cache | (uintptr_t)value_cache) & 15) == 0 &&
                      (q_stride * 2) % 16 == 0 && (kv_block_stride * kvb) % 16 == 0 &&
                      (kv_head_stride * kvb) % 16 == 0 && block_size >= 8;
  LV_CHECK(!p.kv_fp8 || vec_ok, "fp8 kv cache: operands must be 16-byte aligned");
  int rc = 0;
  // block-sparse requests take the generic kernel (per-head masks; a niche of the reference's op)
  if (dtype == LVLLM_BF16 && vec_ok && !bsparse)
    rc = launch_mfma_hs<BF16>(p, head_size, block_size, num_seqs, 1, max_seq_len, s);
  else if (dtype == LVLLM_F16 && vec_ok && !bsparse)
    rc = launch_mfma_hs<F16>(p, head_size, block_size, num_seqs, 1, max_seq_len, s);
  else {
    LV_DISPATCH_DTYPE(dtype, rc = launch_generic<scalar_t>(p, head_size, block_size, num_seqs, 1, s));
  }
  if (rc) return rc;
  LV_LAUNCH_CHECK();
  return 0;
}

struct RopeArgs {
  const int64_t* positions;
  const void* cos_sin_cache;
  const void* k_new;
  const void* v_new;
  const int64_t* slot_mapping;
  int64_t k_new_stride, v_new_stride;
};

static int paged_attention_v2_impl(
    void* out, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int block_size,
    int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions,
    const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
    float v_scale, int tp_rank, int blocksparse_local_blocks,
    int blocksparse_vert_stride, int blocksparse_block_size,
    int blocksparse_head_sliding_step, int64_t kv_cache_bytes, int phases, void* stream,
    const RopeArgs* rope, void* out_fp8 = nullptr, const float* out_fp8_scale = nullptr) {
  if (int rc = check_common(num_seqs, num_heads, head_size, num_kv_heads, block_size, dtype,
                            kv_dtype, k_scale, v_scale, blocksparse_vert_stride))
    return rc;
  int max_block = 0;
  if (int rc = cache_extent(kv_cache_bytes, kv_block_stride, kv_head_stride, num_kv_heads, head_size, block_size,
                            kv_dtype == LVLLM_KV_FP8_E4M3 ? 1 : (dtype == LVLLM_F32 ? 4 : 2), &max_block))
    return rc;
  const bool bsparse = blocksparse_vert_stride > 1;
  LV_CHECK(!bsparse || (blocksparse_block_size > 0 && blocksparse_block_size % block_size == 0),
           "blocksparse_block_size must be a positive multiple of block_size");
  LV_CHECK((phases & ~3) == 0 && phases != 0, "phases must be 1 (partitions), 2 (reduce) or 3 (both)");
  LV_CHECK(max_num_partitions >= 1 &&
               (int64_t)max_num_partitions * kPartitionSize >= (int64_t)max_seq_len,
           "exp_sums.size(-1) must be >= ceil(max_seq_len / 512)");
  if (num_seqs == 0) return 0;
  if (max_num_blocks_per_seq <= 0) {
    // no block table at all: every context is empty, the result is zeros
    const size_t bytes = (size_t)num_seqs * num_heads * head_size * (dtype == LVLLM_F32 ? 4 : 2);
    if (hipMemsetAsync(out, 0, bytes, (hipStream_t)stream) != hipSuccess) LV_CHECK(false, "hipMemsetAsync failed");
    return 0;
  }
  AttnParams p{};
  p.bs_vert_stride = blocksparse_vert_stride; p.bs_local_blocks = blocksparse_local_blocks;
  p.bs_block_size = blocksparse_block_size; p.bs_head_sliding_step = blocksparse_head_sliding_step;
  p.tp_rank = tp_rank;
  p.out = tmp_out; p.exp_sums = exp_sums; p.max_logits = max_logits;
  p.q = query; p.k_cache = key_cache; p.v_cache = value_cache;
  p.block_tables = block_tables; p.seq_lens = seq_lens; p.alibi_slopes = alibi_slopes;
  p.num_heads = num_heads; p.num_kv_heads = num_kv_heads;
  p.max_num_blocks_per_seq = max_num_blocks_per_seq; p.max_num_partitions = max_num_partitions;
  p.partitioned = 1; p.scale = scale; p.max_block = max_block;
  p.q_stride = q_stride; p.kv_block_stride = kv_block_stride; p.kv_head_stride = kv_head_stride;
  p.kv_fp8 = kv_dtype == LVLLM_KV_FP8_E4M3; p.k_scale = k_scale; p.v_scale = v_scale;
  if (rope != nullptr) {
    p.positions = rope->positions; p.cos_sin_cache = rope->cos_sin_cache;
    p.k_new = rope->k_new; p.v_new = rope->v_new; p.slot_mapping = rope->slot_mapping;
    p.k_new_stride = rope->k_new_stride; p.v_new_stride = rope->v_new_stride;
    p.num_slots = kv_cache_bytes > 0 ? kv_cache_bytes / ((int64_t)num_kv_heads * head_size * (p.kv_fp8 ? 1 : 2))
                                     : INT64_MAX;
  }
  hipStream_t s = (hipStream_t)stream;
  const int kvb = p.kv_fp8 ? 1 : 2;
  const bool vec_ok = (((uintptr_t)query | (uintptr_t)key_cache | (uintptr_t)value_cache) & 15) == 0 &&
                      (q_stride * 2) % 16 == 0 && (kv_block_stride * kvb) % 16 == 0 &&
                      (kv_head_stride * kvb) % 16 == 0 && block_size >= 8;
  LV_CHECK(!p.kv_fp8 || vec_ok, "fp8 kv cache: operands must be 16-byte aligned");
  // How many equal shares each context is cut into.  The reference always cuts at 512
  // tokens; here the cut exists only to fill the GPU: with >= 8 waves per CU from
  // (sequence, kv head) pairs alone there is no cut at all (one pass, no reduce), otherwise
  // just enough shares, bounded by the caller's scratch (max_num_partitions slots per head)
  // and by a minimum of 4 tiles of work per share.
  const int G = num_heads / num_kv_heads;
  const bool use_mfma = (dtype == LVLLM_BF16 || dtype == LVLLM_F16) && vec_ok && !bsparse;
  const int64_t pairs = use_mfma ? (int64_t)num_seqs * num_kv_heads * ((G + 15) / 16)
                                 : (int64_t)num_seqs * num_heads;
  const int64_t want = use_mfma ? (2048 + 8 * pairs - 1) / (8 * pairs) : (1024 + pairs - 1) / pairs;
  const int max_tiles = (max_seq_len + 15) / 16;
  int nsplit = (int)(want < max_num_partitions ? want : max_num_partitions);
  if (nsplit > max_tiles / 4) nsplit = max_tiles / 4;
  if (nsplit < 1) nsplit = 1;
  // lvllm_set_tuning("attn_splits", n): n >= 1 forces n shares (tests, experiments); -1 = the reference's
  // partitioning, 512 tokens per share whatever the batch, so that the scratch slots hold exactly what
  // attention_kernels.cu:349-357,483-495 stores in them
  const int forced = tuning().attn_splits;
  if (forced >= 1) nsplit = forced < max_num_partitions ? forced : max_num_partitions;
  if (forced == -1) {
    nsplit = (max_seq_len + kPartitionSize - 1) / kPartitionSize;
    if (nsplit < 1) nsplit = 1;
    p.split_tiles = kPartitionSize / 16;
  }
  p.num_splits = nsplit;
  if (nsplit == 1 && forced != -1) {  // single share: write `out` directly, nothing to reduce
    p.out = out;
    p.partitioned = 0;
    p.max_num_partitions = 1;
  }
  if (out_fp8 != nullptr) {  // the fp8 twin is written by the MFMA kernel's merge, when that merge writes `out`
    if (!(nsplit == 1 && forced != -1 && use_mfma)) {
      set_error("paged_attention_v2 with an fp8 twin: single-pass MFMA launches only");
      return 3;
    }
    LV_CHECK(out_fp8_scale != nullptr, "out_fp8 needs out_fp8_scale");
    p.out_fp8 = (uint8_t*)out_fp8;
    p.out_fp8_scale = out_fp8_scale;
  }
  const int tokens_per_wg = (max_seq_len + nsplit - 1) / nsplit;
  int rc = 0;
  if (phases & 1) {
    if (dtype == LVLLM_BF16 && use_mfma)
      rc = launch_mfma_hs<BF16>(p, head_size, block_size, num_seqs, nsplit, tokens_per_wg, s);
    else if (dtype == LVLLM_F16 && use_mfma)
      rc = launch_mfma_hs<F16>(p, head_size, block_size, num_seqs, nsplit, tokens_per_wg, s);
    else {
      LV_DISPATCH_DTYPE(dtype, rc = launch_generic<scalar_t>(p, head_size, block_size, num_seqs, nsplit, s));
    }
    if (rc) return rc;
    LV_LAUNCH_CHECK();
  }
  if (nsplit == 1 && forced != -1) return 0;
  if (!(phases & 2)) return 0;

  const int num_rows = num_seqs * num_heads;
  const int waves_per_block = 4;
  const int grid = (num_rows + waves_per_block - 1) / waves_per_block;
#define LV_REDUCE(T_)                                                                              \
  hipLaunchKernelGGL((paged_attn_v2_reduce_generic_kernel<T_>), dim3(grid), dim3(waves_per_block * 64), \
                     0, s, (typename T_::store_t*)out, exp_sums, max_logits,                        \
                     (const typename T_::store_t*)tmp_out, seq_lens, max_num_partitions, nsplit,    \
                     p.split_tiles, num_rows, num_heads, head_size)
  if (dtype == LVLLM_BF16) LV_REDUCE(BF16);
  else if (dtype == LVLLM_F16) LV_REDUCE(F16);
  else LV_REDUCE(F32);
#undef LV_REDUCE
  LV_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvllm_paged_attention_v2_phases(
    void* out, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int block_size,
    int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions,
    const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
    float v_scale, int tp_rank, int blocksparse_local_blocks,
    int blocksparse_vert_stride, int blocksparse_block_size,
    int blocksparse_head_sliding_step, int64_t kv_cache_bytes, int phases, void* stream) {
  return paged_attention_v2_impl(
      out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
      head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
      max_num_blocks_per_seq, max_num_partitions, alibi_slopes, q_stride, kv_block_stride,
      kv_head_stride, dtype, kv_dtype, k_scale, v_scale, tp_rank, blocksparse_local_blocks,
      blocksparse_vert_stride, blocksparse_block_size, blocksparse_head_sliding_step, kv_cache_bytes, phases,
      stream, nullptr);
}

// Extension: one decode step's rotary_embedding (NeoX, rot_dim == head_size) + reshape_and_cache +
// paged_attention_v2 in ONE launch (plus the reduce pass when contexts are cut into shares).  Returns 3 when
// the arguments are outside the fused kernel's envelope (nothing was launched: run the three operators).
static int rope_cache_paged_attention_impl(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query, const void* key,
    const void* value, void* key_cache, void* value_cache, int num_seqs, int num_heads, int head_size,
    int num_kv_heads, float scale, const int32_t* block_tables, const int32_t* seq_lens,
    const int64_t* positions, const int64_t* slot_mapping, const void* cos_sin_cache, int rot_dim, int is_neox,
    int block_size, int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions, int64_t q_stride,
    int64_t key_stride, int64_t value_stride, int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
    int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes, void* stream, void* out_fp8,
    const float* out_fp8_scale) {
  const bool fp8 = kv_dtype == LVLLM_KV_FP8_E4M3;
  const bool ok =
      (dtype == LVLLM_BF16 || dtype == LVLLM_F16) && (kv_dtype == LVLLM_KV_AUTO || fp8) && is_neox &&
      rot_dim == head_size && (fp8 ? (head_size == 128 || head_size == 256) : (head_size == 64 || head_size == 128 || head_size == 256)) &&
      (fp8 ? (k_scale > 0.f && v_scale > 0.f) : (k_scale == 1.f && v_scale == 1.f)) &&
      (block_size == 16 || block_size == 32) &&
      num_kv_heads > 0 && num_heads % num_kv_heads == 0 && num_heads / num_kv_heads <= 16 &&
      (((uintptr_t)query | (uintptr_t)key | (uintptr_t)value | (uintptr_t)key_cache | (uintptr_t)value_cache |
        (uintptr_t)cos_sin_cache) & 15) == 0 &&
      q_stride % 8 == 0 && key_stride % 8 == 0 && value_stride % 8 == 0 && kv_block_stride % 16 == 0 &&
      kv_head_stride % 16 == 0 && max_num_blocks_per_seq > 0 && tuning().attn_splits != -1;
  if (!ok) {
    set_error("lvllm_rope_cache_paged_attention: arguments outside the fused kernel's envelope");
    return 3;
  }
  RopeArgs r{positions, cos_sin_cache, key, value, slot_mapping, key_stride, value_stride};
  return paged_attention_v2_impl(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs,
                                 num_heads, head_size, num_kv_heads, scale, block_tables, seq_lens, block_size,
                                 max_seq_len, max_num_blocks_per_seq, max_num_partitions, nullptr, q_stride,
                                 kv_block_stride, kv_head_stride, dtype, kv_dtype, k_scale, v_scale, 0, 0, 0, 64, 0,
                                 kv_cache_bytes, 3, stream, &r, out_fp8, out_fp8_scale);
}

extern "C" int lvllm_rope_cache_paged_attention(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query, const void* key,
    const void* value, void* key_cache, void* value_cache, int num_seqs, int num_heads, int head_size,
    int num_kv_heads, float scale, const int32_t* block_tables, const int32_t* seq_lens,
    const int64_t* positions, const int64_t* slot_mapping, const void* cos_sin_cache, int rot_dim, int is_neox,
    int block_size, int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions, int64_t q_stride,
    int64_t key_stride, int64_t value_stride, int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
    int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes, void* stream) {
  return rope_cache_paged_attention_impl(out, exp_sums, max_logits, tmp_out, query, key, value, key_cache, value_cache,
                                         num_seqs, num_heads, head_size, num_kv_heads, scale, block_tables, seq_lens,
                                         positions, slot_mapping, cos_sin_cache, rot_dim, is_neox, block_size, max_seq_len,
                                         max_num_blocks_per_seq, max_num_partitions, q_stride, key_stride, value_stride,
                                         kv_block_stride, kv_head_stride, dtype, kv_dtype, k_scale, v_scale,
                                         kv_cache_bytes, stream, nullptr, nullptr);
}

extern "C" int lvllm_rope_cache_paged_attention_q(
    void* out, void* out_fp8, const float* out_fp8_scale, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key, const void* value, void* key_cache, void* value_cache, int num_seqs,
    int num_heads, int head_size, int num_kv_heads, float scale, const int32_t* block_tables, const int32_t* seq_lens,
    const int64_t* positions, const int64_t* slot_mapping, const void* cos_sin_cache, int rot_dim, int is_neox,
    int block_size, int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions, int64_t q_stride,
    int64_t key_stride, int64_t value_stride, int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
    int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes, void* stream) {
  LV_CHECK(out_fp8 != nullptr && out_fp8_scale != nullptr, "out_fp8 and out_fp8_scale are required");
  return rope_cache_paged_attention_impl(out, exp_sums, max_logits, tmp_out, query, key, value, key_cache, value_cache,
                                         num_seqs, num_heads, head_size, num_kv_heads, scale, block_tables, seq_lens,
                                         positions, slot_mapping, cos_sin_cache, rot_dim, is_neox, block_size, max_seq_len,
                                         max_num_blocks_per_seq, max_num_partitions, q_stride, key_stride, value_stride,
                                         kv_block_stride, kv_head_stride, dtype, kv_dtype, k_scale, v_scale,
                                         kv_cache_bytes, stream, out_fp8, out_fp8_scale);
}

// paged_attention_v2 / the fused rope + cache + attention launch with an fp8 twin of the result for a W8A8 output
// projection (include/lvllm_hip.h).  3 = this launch would be cut into shares (or is not an MFMA launch): nothing was
// launched, call the plain entry and quantise.
extern "C" int lvllm_paged_attention_v2_q(
    void* out, void* out_fp8, const float* out_fp8_scale, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache, int num_seqs, int num_heads, int head_size,
    int num_kv_heads, float scale, const int32_t* block_tables, const int32_t* seq_lens, int block_size,
    int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes,
    void* stream) {
  LV_CHECK(out_fp8 != nullptr && out_fp8_scale != nullptr, "out_fp8 and out_fp8_scale are required");
  return paged_attention_v2_impl(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
                                 head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                                 max_num_blocks_per_seq, max_num_partitions, nullptr, q_stride, kv_block_stride,
                                 kv_head_stride, dtype, kv_dtype, k_scale, v_scale, 0, 0, 0, 64, 0, kv_cache_bytes, 3,
                                 stream, nullptr, out_fp8, out_fp8_scale);
}

extern "C" int lvllm_paged_attention_v2(
    void* out, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int block_size,
    int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions,
    const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
    float v_scale, int tp_rank, int blocksparse_local_blocks,
    int blocksparse_vert_stride, int blocksparse_block_size,
    int blocksparse_head_sliding_step, int64_t kv_cache_bytes, void* stream) {
  return lvllm_paged_attention_v2_phases(
      out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, num_seqs, num_heads,
      head_size, num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
      max_num_blocks_per_seq, max_num_partitions, alibi_slopes, q_stride, kv_block_stride,
      kv_head_stride, dtype, kv_dtype, k_scale, v_scale, tp_rank, blocksparse_local_blocks,
      blocksparse_vert_stride, blocksparse_block_size, blocksparse_head_sliding_step, kv_cache_bytes, 3, stream);
}
