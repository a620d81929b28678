// advance_step: move a decode batch's input tensors one token forward on the device, so that the
// next step can be launched without a host round trip (csrc/prepare_inputs/advance_step.cu:14-57
// of the reference; index arithmetic only, bit-exact):
//   input_tokens[i]    = sampled_token_ids[i]
//   seq_lens[i]       += 1
//   input_positions[i] = seq_lens[i] - 1
//   slot_mapping[i]    = block_tables[i][pos / block_size] * block_size + pos % block_size
// for i < num_queries (rows num_queries .. num_seqs-1 of the batch are left alone).
#include "../../include/lvllm_hip.h"
#include "common.h"

namespace lvllm {

__global__ void advance_step_kernel(const int num_queries, const int block_size,
                                    int64_t* __restrict__ input_tokens,
                                    const int64_t* __restrict__ sampled_token_ids,
                                    int64_t* __restrict__ input_positions, int32_t* __restrict__ seq_lens,
                                    int64_t* __restrict__ slot_mapping,
                                    const int32_t* __restrict__ block_tables,
                                    const int64_t block_tables_stride, int64_t* __restrict__ token_log,
                                    const int skip_empty_rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_queries) return;
  if (token_log != nullptr) token_log[i] = sampled_token_ids[i];
  // a captured step runs at its padded batch size: rows without a sequence (length 0, slot -1) stay that way
  if (skip_empty_rows && seq_lens[i] <= 0) return;
  input_tokens[i] = sampled_token_ids[i];
  const int next_len = seq_lens[i] + 1;
  const int pos = next_len - 1;
  seq_lens[i] = next_len;
  input_positions[i] = pos;
  const int32_t* row = block_tables + block_tables_stride * i;
  slot_mapping[i] = (int64_t)row[pos / block_size] * block_size + pos % block_size;
}

}  // namespace lvllm

extern "C" int lvllm_advance_step(int num_seqs, int num_queries, int block_size, int64_t* input_tokens,
                                  const int64_t* sampled_token_ids, int64_t* input_positions,
                                  int32_t* seq_lens, int64_t* slot_mapping, const int32_t* block_tables,
                                  int64_t block_tables_stride, void* stream) {
  return lvllm_advance_step_ex(num_seqs, num_queries, block_size, input_tokens, sampled_token_ids, input_positions,
                               seq_lens, slot_mapping, block_tables, block_tables_stride, nullptr, 0, stream);
}

extern "C" int lvllm_advance_step_ex(int num_seqs, int num_queries, int block_size, int64_t* input_tokens,
                                     const int64_t* sampled_token_ids, int64_t* input_positions,
                                     int32_t* seq_lens, int64_t* slot_mapping, const int32_t* block_tables,
                                     int64_t block_tables_stride, int64_t* token_log, int skip_empty_rows,
                                     void* stream) {
  LV_CHECK(num_seqs >= 0 && num_queries >= 0 && num_queries <= num_seqs, "need 0 <= num_queries <= num_seqs");
  LV_CHECK(block_size > 0, "block_size must be positive");
  if (num_queries == 0) return 0;
  hipLaunchKernelGGL(lvllm::advance_step_kernel, dim3((num_queries + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, num_queries, block_size, input_tokens, sampled_token_ids,
                     input_positions, seq_lens, slot_mapping, block_tables, block_tables_stride, token_log,
                     skip_empty_rows);
  LV_LAUNCH_CHECK();
  return 0;
}
