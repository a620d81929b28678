// Partitioned key walks of the prompt kernels (prefill_mfma32.h, prefill_chunk.h): a launch that would leave CUs idle
// cuts the key range of its sequences into partitions across workgroups; every workgroup leaves (maximum, sum,
// normalised partial result) per row and partition in scratch, and one small kernel merges them -- paged_attention_v2's
// scheme (attention_kernels.cu:564-669) with one difference: a row of a causal chunk has its OWN horizon, so it merges
// only the partitions it reaches (the others were never written for it).
#pragma once
#include "prefill_mfma.h"

namespace lvllm {

struct ChunkScratch {   // (null tmp_out: single pass, results go to `out`)
  void* tmp_out;        // [num_tokens * num_heads][num_parts][D]  T, normalised inside the partition
  float* max_logits;    // [num_tokens * num_heads][num_parts]  (base-2 logits, as the kernel keeps them)
  float* exp_sums;      // [num_tokens * num_heads][num_parts]
  int num_parts;
  int part_tokens;      // keys per partition, a multiple of 16
};

inline int64_t chunk_up256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// scratch of `rows` x `parts` partial results (tmp_out | max_logits | exp_sums, each 256-byte aligned)
inline int64_t partition_scratch_bytes(int64_t rows, int parts, int head_size) {
  return chunk_up256(rows * parts * head_size * 2) + 2 * chunk_up256(rows * parts * 4);
}
inline ChunkScratch partition_scratch(void* workspace, int64_t rows, int parts, int part_tokens, int head_size) {
  ChunkScratch sc{};
  sc.num_parts = parts;
  sc.part_tokens = part_tokens;
  sc.tmp_out = workspace;
  sc.max_logits = (float*)((char*)workspace + chunk_up256(rows * parts * head_size * 2));
  sc.exp_sums = (float*)((char*)sc.max_logits + chunk_up256(rows * parts * 4));
  return sc;
}

// Merge: a workgroup of 256 threads per (sequence, query token, 4 heads); a wave per row, a lane per two d's.  Each
// row merges the partitions its own horizon reaches.  Every load of a row is issued before the first is used (the
// partition count is a kernel argument, the loops are unrolled over the 16 a launch may have and predicated): as a
// run-time loop of dependent waits the merge of 8 x 32 tokens x 32 heads took 12.6 us for 8 MB.
constexpr int kMaxPartitions = 16;
template <typename T, int D>
__global__ __launch_bounds__(256) void prefill_chunk_reduce_kernel(const PrefillParams p, const ChunkScratch sc) {
  using S = typename T::store_t;
  static_assert(D <= 128 && D % 2 == 0, "a lane merges two d's");
  const int t = blockIdx.y, seq = blockIdx.z;
  const int head = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int qbeg = p.query_start_loc[seq];
  const int qlen = p.query_start_loc[seq + 1] - qbeg;
  if (t >= qlen || head >= p.num_heads || 2 * lane >= D) return;
  const int tok = qbeg + t;
  const int visible = p.seq_lens[seq] - qlen + t + 1;
  const int np = min(sc.num_parts, (visible + sc.part_tokens - 1) / sc.part_tokens);
  const int64_t row = ((int64_t)tok * p.num_heads + head) * sc.num_parts;
  float m[kMaxPartitions], l[kMaxPartitions];
  uint32_t v[kMaxPartitions];
#pragma unroll
  for (int i = 0; i < kMaxPartitions; ++i) {
    m[i] = i < np ? sc.max_logits[row + i] : -FLT_MAX;
    l[i] = i < np ? sc.exp_sums[row + i] : 0.f;
    v[i] = i < np ? reinterpret_cast<const uint32_t*>(sc.tmp_out)[((row + i) * D) / 2 + lane] : 0u;
  }
  float M = -FLT_MAX;
#pragma unroll
  for (int i = 0; i < kMaxPartitions; ++i) M = fmaxf(M, m[i]);
  float L = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxPartitions; ++i) {
    const float w = l[i] * __builtin_amdgcn_exp2f(m[i] - M);  // (absent partitions: 0 * 2^(-huge) = 0)
    L += w;
    o0 += w * T::to_float((S)(v[i] & 0xffffu));
    o1 += w * T::to_float((S)(v[i] >> 16));
  }
  const float inv = L > 0.f ? __fdividef(1.f, L) : 0.f;
  reinterpret_cast<uint32_t*>(reinterpret_cast<S*>(p.out) + (int64_t)tok * p.out_stride + (int64_t)head * D)[lane] =
      pack2<T>(o0 * inv, o1 * inv);
}

template <typename T, int D>
static void launch_partition_reduce(const PrefillParams& p, const ChunkScratch& sc, int num_seqs, int max_query_len,
                                    hipStream_t stream) {
  hipLaunchKernelGGL((prefill_chunk_reduce_kernel<T, D>), dim3((p.num_heads + 3) / 4, max_query_len, num_seqs),
                     dim3(256), 0, stream, p, sc);
}

}  // namespace lvllm
