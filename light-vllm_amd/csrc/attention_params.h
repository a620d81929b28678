// Kernel argument block shared by the paged-attention translation units.
#pragma once
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
  float scale;
  int64_t q_stride, kv_block_stride, kv_head_stride;
};

}  // namespace lvllm
