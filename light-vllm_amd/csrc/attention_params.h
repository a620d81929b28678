// Kernel argument block shared by the paged-attention translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lvllm {

constexpr int kPartitionSize = 512;  // csrc/attention/attention_kernels.cu:850

struct AttnParams {
  void* out;          // v1: [B,H,D]; v2: tmp_out [B,H,P,D]
  float* exp_sums;    // v2 only [B,H,P]
  float* max_logits;  // v2 only [B,H,P]
  const void* q;
  const void* k_cache;
  const void* v_cache;
  const int32_t* block_tables;
  const int32_t* seq_lens;
  const float* alibi_slopes;
  int num_heads, num_kv_heads, max_num_blocks_per_seq, max_num_partitions;
  int partitioned;  // 0: whole sequence per workgroup (v1)
  int num_splits;   // partitioned: every sequence is cut into at most this many equal shares
  int split_tiles;  // > 0: shares are exactly this many 16-token tiles (32 = the reference's 512-token
                    // partitions, attention_kernels.cu:850), not equal parts of the context
  int max_block;    // block numbers read from the table are clamped to [0, max_block] (unsigned min):
                    // a wrong table or a cache handed over with the wrong element size then reads
                    // wrong blocks of the allocation instead of faulting (kv_cache_bytes of the C-ABI)
  float scale;
  int64_t q_stride, kv_block_stride, kv_head_stride;  // kv strides in cache elements
  // kv_cache_dtype "fp8": the caches hold OCP e4m3fn bytes, layouts with x = 16; a dequantised
  // element is T(float(fp8) * scale) (csrc/quantization/fp8/nvidia/quant_utils.cuh:295-300)
  int kv_fp8;
  float k_scale, v_scale;
  // block-sparse attention (attention_kernels.cu:209-247): active when bs_vert_stride > 1.  A KV
  // cache block is attended by a head iff (k + offset) % vert_stride == 0 ("remote") or
  // k > q - local_blocks ("local"), k / q = the block-sparse block of the cache block / of the
  // query, offset = (tp_rank * heads + head) * sliding_step + 1 (kv heads for a negative step).
  int bs_vert_stride, bs_local_blocks, bs_block_size, bs_head_sliding_step, tp_rank;
  // ROPE instantiation (decode step, one new token per sequence): the kernel rotates q and the new k itself
  // (NeoX pairing, rot_dim == head size, arithmetic and roundings of pos_encoding.hip's rotate<T>), writes the
  // rotated k and v into the paged caches at slot_mapping[seq], and attends to them from registers: the
  // launches rotary_embedding + reshape_and_cache + paged_attention in one, bit for bit.
  const int64_t* positions;     // [num_seqs]
  const void* cos_sin_cache;    // [max_pos, head_size] = [cos | sin]
  const void* k_new;            // [num_seqs, num_kv_heads, D], row stride k_new_stride (NOT rotated in place)
  const void* v_new;            // [num_seqs, num_kv_heads, D], row stride v_new_stride
  const int64_t* slot_mapping;  // [num_seqs]; < 0 or beyond the caches: nothing is written
  int64_t k_new_stride, v_new_stride, num_slots;
  // fp8 twin of the result (single-pass launches only): out_fp8[seq, head, d] = static_scaled_fp8_quant(out as rounded
  // to T, *out_fp8_scale) -- the W8A8 output projection behind the attention then takes its activations as they are
  uint8_t* out_fp8;
  const float* out_fp8_scale;
};

__host__ __device__ inline bool blocksparse_attended(const AttnParams& p, int token, int seq_len, int head,
                                                     int kv_head, int cache_block_size) {
  const int q_bs = (seq_len - 1) / p.bs_block_size;
  const int off = p.bs_head_sliding_step >= 0
                      ? (p.tp_rank * p.num_heads + head) * p.bs_head_sliding_step + 1
                      : (p.tp_rank * p.num_kv_heads + kv_head) * (-p.bs_head_sliding_step) + 1;
  const int k_bs = (token / cache_block_size) * cache_block_size / p.bs_block_size;
  return ((k_bs + off) % p.bs_vert_stride == 0) || (k_bs > q_bs - p.bs_local_blocks);
}


// Share `s` of a context of `seq_len` tokens cut into at most `num_splits` equal shares of
// whole 16-token tiles: tokens [t0, t1).  Returns false for an empty share.  Both passes of
// paged_attention_v2 (and every kernel variant) use this one function, so the partition
// pass and the reduce pass always agree on which scratch slots hold data.
__host__ __device__ inline bool split_range(int seq_len, int num_splits, int s, int* t0, int* t1,
                                            int split_tiles = 0) {
  const int ntiles = (seq_len + 15) >> 4;
  const int chunk = split_tiles > 0 ? split_tiles : (ntiles + num_splits - 1) / num_splits;  // tiles per share
  const int first = s * chunk;
  if (first >= ntiles) return false;
  *t0 = first << 4;
  const int end = (first + chunk) << 4;
  *t1 = end < seq_len ? end : seq_len;
  return true;
}
__host__ __device__ inline int num_nonempty_splits(int seq_len, int num_splits, int split_tiles = 0) {
  const int ntiles = (seq_len + 15) >> 4;
  if (ntiles == 0) return 0;
  const int chunk = split_tiles > 0 ? split_tiles : (ntiles + num_splits - 1) / num_splits;
  return (ntiles + chunk - 1) / chunk;
}

}  // namespace lvllm
