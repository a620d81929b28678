// torch op registrations: `torch.ops._C.*`, `torch.ops._C_cache_ops.*`,
// `torch.ops._C_cuda_utils.*` with the schemas of the reference
// (csrc/torch_bindings.cpp:18-279), each implemented by unpacking tensors and
// forwarding to the C-ABI of include/lvllm_hip.h on the CURRENT stream of the
// tensor's device (csrc/attention/attention_kernels.cu:736-737 does the same).
//
// Host-only C++ (built with g++): no device code lives here.
#include <Python.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/all.h>
#include <torch/library.h>

#include <optional>
#include <string>
#include <vector>

#include "lvllm_hip.h"

namespace {

int dtype_code(const torch::Tensor& t, const char* what) {
  switch (t.scalar_type()) {
    case at::ScalarType::Float: return LVLLM_F32;
    case at::ScalarType::Half: return LVLLM_F16;
    case at::ScalarType::BFloat16: return LVLLM_BF16;
    default: TORCH_CHECK(false, what, ": unsupported dtype ", t.scalar_type());
  }
}

// kv_cache_dtype strings: csrc/quantization/fp8/amd/quant_utils.cuh:547-573
int kv_dtype_code(const std::string& s) {
  if (s == "auto") return LVLLM_KV_AUTO;
  if (s == "fp8" || s == "fp8_e4m3") return LVLLM_KV_FP8_E4M3;
  TORCH_CHECK(false, "Unsupported data type of kv cache: ", s);
}

void* current_stream(const torch::Tensor& t) {
  return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

void check(int rc) { TORCH_CHECK(rc == 0, lvllm_last_error()); }

#define LV_CHECK_DEVICE(t) TORCH_CHECK((t).is_cuda(), #t " must be a GPU tensor")

void paged_attention_v1(torch::Tensor& out, torch::Tensor& query, torch::Tensor& key_cache,
                        torch::Tensor& value_cache, int64_t num_kv_heads, double scale,
                        torch::Tensor& block_tables, torch::Tensor& seq_lens, int64_t block_size,
                        int64_t max_seq_len, const std::optional<torch::Tensor>& alibi_slopes,
                        const std::string& kv_cache_dtype, double k_scale, double v_scale,
                        const int64_t tp_rank, const int64_t blocksparse_local_blocks,
                        const int64_t blocksparse_vert_stride, const int64_t blocksparse_block_size,
                        const int64_t blocksparse_head_sliding_step) {
  LV_CHECK_DEVICE(query);
  TORCH_CHECK(out.is_contiguous(), "out must be contiguous");
  TORCH_CHECK(block_tables.scalar_type() == at::kInt && seq_lens.scalar_type() == at::kInt,
              "block_tables / seq_lens must be int32");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(query));
  const float* alibi = alibi_slopes ? alibi_slopes->data_ptr<float>() : nullptr;
  check(lvllm_paged_attention_v1(
      out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
      (int)query.size(0), (int)query.size(1), (int)query.size(2), (int)num_kv_heads, (float)scale,
      block_tables.data_ptr<int>(), seq_lens.data_ptr<int>(), (int)block_size, (int)max_seq_len,
      (int)block_tables.size(1), alibi, query.stride(0), key_cache.stride(0), key_cache.stride(1),
      dtype_code(query, "paged_attention_v1"), kv_dtype_code(kv_cache_dtype), (float)k_scale,
      (float)v_scale, (int)tp_rank, (int)blocksparse_local_blocks, (int)blocksparse_vert_stride,
      (int)blocksparse_block_size, (int)blocksparse_head_sliding_step, current_stream(query)));
}

void paged_attention_v2(torch::Tensor& out, torch::Tensor& exp_sums, torch::Tensor& max_logits,
                        torch::Tensor& tmp_out, torch::Tensor& query, torch::Tensor& key_cache,
                        torch::Tensor& value_cache, int64_t num_kv_heads, double scale,
                        torch::Tensor& block_tables, torch::Tensor& seq_lens, int64_t block_size,
                        int64_t max_seq_len, const std::optional<torch::Tensor>& alibi_slopes,
                        const std::string& kv_cache_dtype, double k_scale, double v_scale,
                        const int64_t tp_rank, const int64_t blocksparse_local_blocks,
                        const int64_t blocksparse_vert_stride, const int64_t blocksparse_block_size,
                        const int64_t blocksparse_head_sliding_step) {
  LV_CHECK_DEVICE(query);
  TORCH_CHECK(out.is_contiguous() && tmp_out.is_contiguous() && exp_sums.is_contiguous() &&
                  max_logits.is_contiguous(),
              "out / tmp_out / exp_sums / max_logits must be contiguous");
  TORCH_CHECK(block_tables.scalar_type() == at::kInt && seq_lens.scalar_type() == at::kInt,
              "block_tables / seq_lens must be int32");
  TORCH_CHECK(exp_sums.scalar_type() == at::kFloat && max_logits.scalar_type() == at::kFloat,
              "exp_sums / max_logits must be float32");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(query));
  const float* alibi = alibi_slopes ? alibi_slopes->data_ptr<float>() : nullptr;
  check(lvllm_paged_attention_v2(
      out.data_ptr(), exp_sums.data_ptr<float>(), max_logits.data_ptr<float>(), tmp_out.data_ptr(),
      query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), (int)query.size(0),
      (int)query.size(1), (int)query.size(2), (int)num_kv_heads, (float)scale,
      block_tables.data_ptr<int>(), seq_lens.data_ptr<int>(), (int)block_size, (int)max_seq_len,
      (int)block_tables.size(1), (int)exp_sums.size(-1), alibi, query.stride(0),
      key_cache.stride(0), key_cache.stride(1), dtype_code(query, "paged_attention_v2"),
      kv_dtype_code(kv_cache_dtype), (float)k_scale, (float)v_scale, (int)tp_rank,
      (int)blocksparse_local_blocks, (int)blocksparse_vert_stride, (int)blocksparse_block_size,
      (int)blocksparse_head_sliding_step, current_stream(query)));
}

void silu_and_mul(torch::Tensor& out, torch::Tensor& input) {
  LV_CHECK_DEVICE(input);
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "silu_and_mul needs contiguous tensors");
  const int64_t d = input.size(-1) / 2;
  const int64_t num_tokens = input.numel() / input.size(-1);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(input));
  check(lvllm_silu_and_mul(out.data_ptr(), input.data_ptr(), num_tokens, (int)d,
                           dtype_code(input, "silu_and_mul"), current_stream(input)));
}

void rms_norm(torch::Tensor& out, torch::Tensor& input, torch::Tensor& weight, double epsilon) {
  LV_CHECK_DEVICE(input);
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "rms_norm needs contiguous tensors");
  const int hidden = (int)input.size(-1);
  const int num_tokens = (int)(input.numel() / hidden);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(input));
  check(lvllm_rms_norm(out.data_ptr(), input.data_ptr(), weight.data_ptr(), (float)epsilon,
                       num_tokens, hidden, dtype_code(input, "rms_norm"), current_stream(input)));
}

void fused_add_rms_norm(torch::Tensor& input, torch::Tensor& residual, torch::Tensor& weight,
                        double epsilon) {
  LV_CHECK_DEVICE(input);
  TORCH_CHECK(input.is_contiguous() && residual.is_contiguous(),
              "fused_add_rms_norm needs contiguous tensors");
  const int hidden = (int)input.size(-1);
  const int num_tokens = (int)(input.numel() / hidden);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(input));
  check(lvllm_fused_add_rms_norm(input.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                 (float)epsilon, num_tokens, hidden,
                                 dtype_code(input, "fused_add_rms_norm"), current_stream(input)));
}

void rotary_embedding(torch::Tensor& positions, torch::Tensor& query, torch::Tensor& key,
                      int64_t head_size, torch::Tensor& cos_sin_cache, bool is_neox) {
  LV_CHECK_DEVICE(query);
  TORCH_CHECK(positions.scalar_type() == at::kLong, "positions must be int64");
  const int64_t num_tokens = query.numel() / query.size(-1);
  const int rot_dim = (int)cos_sin_cache.size(1);
  const int num_heads = (int)(query.size(-1) / head_size);
  const int num_kv_heads = (int)(key.size(-1) / head_size);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(query));
  check(lvllm_rotary_embedding(positions.data_ptr<int64_t>(), query.data_ptr(), key.data_ptr(),
                               (int)num_tokens, num_heads, num_kv_heads, (int)head_size, rot_dim,
                               query.stride(-2), key.stride(-2), cos_sin_cache.data_ptr(),
                               is_neox ? 1 : 0, dtype_code(query, "rotary_embedding"),
                               current_stream(query)));
}

void swap_blocks(torch::Tensor& src, torch::Tensor& dst, const torch::Tensor& block_mapping) {
  const bool src_dev = src.is_cuda(), dst_dev = dst.is_cuda();
  if (src_dev && dst_dev)
    TORCH_CHECK(src.device().index() == dst.device().index(), "src and dst must be on the same GPU");
  TORCH_CHECK(src_dev || dst_dev, "Invalid device combination");
  TORCH_CHECK(block_mapping.device().is_cpu(), "block_mapping must be on CPU");
  TORCH_CHECK(block_mapping.scalar_type() == at::kLong, "block_mapping must be int64");
  auto bm = block_mapping.contiguous();
  const int64_t block_bytes = src.element_size() * src[0].numel();
  const torch::Tensor& dev_t = src_dev ? src : dst;
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(dev_t));
  check(lvllm_swap_blocks(src.data_ptr(), dst.data_ptr(), bm.data_ptr<int64_t>(), (int)bm.size(0),
                          block_bytes, src_dev, dst_dev, current_stream(dev_t)));
}

// Device pointer tables for copy_blocks are cached per set of cache tensors, so
// the steady state does no host->device copy and never synchronises (the
// reference re-uploads them on every call, csrc/cache_kernels.cu:128-133).
struct PtrTable {
  std::vector<int64_t> key_ptrs, value_ptrs;
  torch::Tensor dev;  // [2, L] int64 on the cache device
};

void copy_blocks(std::vector<torch::Tensor> const& key_caches,
                 std::vector<torch::Tensor> const& value_caches,
                 const torch::Tensor& block_mapping) {
  const int num_layers = (int)key_caches.size();
  TORCH_CHECK(num_layers == (int)value_caches.size());
  if (num_layers == 0) return;
  const auto dev = key_caches[0].device();
  TORCH_CHECK(dev.is_cuda());
  const int num_pairs = (int)block_mapping.size(0);
  if (num_pairs == 0) return;
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(dev);

  static thread_local PtrTable table;
  std::vector<int64_t> kp(num_layers), vp(num_layers);
  for (int l = 0; l < num_layers; ++l) {
    kp[l] = (int64_t)key_caches[l].data_ptr();
    vp[l] = (int64_t)value_caches[l].data_ptr();
  }
  if (kp != table.key_ptrs || vp != table.value_ptrs || !table.dev.defined() ||
      table.dev.device() != dev) {
    auto host = torch::empty({2, num_layers}, torch::dtype(torch::kInt64));
    for (int l = 0; l < num_layers; ++l) {
      host[0][l] = kp[l];
      host[1][l] = vp[l];
    }
    table.dev = host.to(dev);
    table.key_ptrs = kp;
    table.value_ptrs = vp;
  }
  // the kernel reads block_mapping on the device (csrc/cache_kernels.cu:79-80)
  torch::Tensor bm = block_mapping.to(dev, torch::kInt64, /*non_blocking=*/true).contiguous();
  const int64_t block_bytes = key_caches[0].element_size() * key_caches[0][0].numel();
  const int64_t* base = table.dev.data_ptr<int64_t>();
  check(lvllm_copy_blocks((const void* const*)base, (const void* const*)(base + num_layers),
                          bm.data_ptr<int64_t>(), num_layers, num_pairs, block_bytes,
                          current_stream(key_caches[0])));
}

void reshape_and_cache(torch::Tensor& key, torch::Tensor& value, torch::Tensor& key_cache,
                       torch::Tensor& value_cache, torch::Tensor& slot_mapping,
                       const std::string& kv_cache_dtype, const double k_scale,
                       const double v_scale) {
  LV_CHECK_DEVICE(key);
  TORCH_CHECK(slot_mapping.scalar_type() == at::kLong, "slot_mapping must be int64");
  TORCH_CHECK(key_cache.is_contiguous() && value_cache.is_contiguous(),
              "key_cache / value_cache must be contiguous");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(key));
  check(lvllm_reshape_and_cache(key.data_ptr(), value.data_ptr(), key_cache.data_ptr(),
                                value_cache.data_ptr(), slot_mapping.data_ptr<int64_t>(),
                                (int)key.size(0), (int)key.size(1), (int)key.size(2),
                                (int)key_cache.size(3), (int)key_cache.size(4), key.stride(0),
                                value.stride(0), dtype_code(key, "reshape_and_cache"),
                                kv_dtype_code(kv_cache_dtype), (float)k_scale, (float)v_scale,
                                current_stream(key)));
}

void reshape_and_cache_flash(torch::Tensor& key, torch::Tensor& value, torch::Tensor& key_cache,
                             torch::Tensor& value_cache, torch::Tensor& slot_mapping,
                             const std::string& kv_cache_dtype, const double k_scale,
                             const double v_scale) {
  LV_CHECK_DEVICE(key);
  TORCH_CHECK(slot_mapping.scalar_type() == at::kLong, "slot_mapping must be int64");
  TORCH_CHECK(key_cache.stride(0) == value_cache.stride(0));
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(key));
  check(lvllm_reshape_and_cache_flash(
      key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
      slot_mapping.data_ptr<int64_t>(), (int)key.size(0), (int)key.size(1), (int)key.size(2),
      (int)key_cache.size(1), key_cache.stride(0), key.stride(0), value.stride(0),
      dtype_code(key, "reshape_and_cache_flash"), kv_dtype_code(kv_cache_dtype), (float)k_scale,
      (float)v_scale, current_stream(key)));
}

int64_t get_device_attribute(int64_t attribute, int64_t device_id) {
  return lvllm_get_device_attribute(attribute, device_id);
}
int64_t get_max_shared_memory_per_block_device_attribute(int64_t device_id) {
  return lvllm_get_max_shared_memory_per_block_device_attribute(device_id);
}

}  // namespace

// Same namespaces, names and schema strings as csrc/torch_bindings.cpp:18-279
// for the operators on the paged-attention decode path.
TORCH_LIBRARY(_C, ops) {
  ops.def(
      "paged_attention_v1("
      "    Tensor! out, Tensor query, Tensor key_cache,"
      "    Tensor value_cache, int num_kv_heads, float scale,"
      "    Tensor block_tables, Tensor seq_lens, int block_size,"
      "    int max_seq_len, Tensor? alibi_slopes,"
      "    str kv_cache_dtype, float k_scale, float v_scale,"
      "    int tp_rank, int blocksparse_local_blocks,"
      "    int blocksparse_vert_stride, int blocksparse_block_size,"
      "    int blocksparse_head_sliding_step) -> ()");
  ops.impl("paged_attention_v1", torch::kCUDA, &paged_attention_v1);

  ops.def(
      "paged_attention_v2("
      "    Tensor! out, Tensor exp_sums, Tensor max_logits,"
      "    Tensor tmp_out, Tensor query, Tensor key_cache,"
      "    Tensor value_cache, int num_kv_heads, float scale,"
      "    Tensor block_tables, Tensor seq_lens, int block_size,"
      "    int max_seq_len, Tensor? alibi_slopes,"
      "    str kv_cache_dtype, float k_scale, float v_scale,"
      "    int tp_rank, int blocksparse_local_blocks,"
      "    int blocksparse_vert_stride, int blocksparse_block_size,"
      "    int blocksparse_head_sliding_step) -> ()");
  ops.impl("paged_attention_v2", torch::kCUDA, &paged_attention_v2);

  ops.def("silu_and_mul(Tensor! out, Tensor input) -> ()");
  ops.impl("silu_and_mul", torch::kCUDA, &silu_and_mul);

  ops.def("rms_norm(Tensor! out, Tensor input, Tensor weight, float epsilon) -> ()");
  ops.impl("rms_norm", torch::kCUDA, &rms_norm);

  ops.def(
      "fused_add_rms_norm(Tensor! input, Tensor! residual, Tensor weight, "
      "float epsilon) -> ()");
  ops.impl("fused_add_rms_norm", torch::kCUDA, &fused_add_rms_norm);

  ops.def(
      "rotary_embedding(Tensor positions, Tensor! query,"
      "                 Tensor! key, int head_size,"
      "                 Tensor cos_sin_cache, bool is_neox) -> ()");
  ops.impl("rotary_embedding", torch::kCUDA, &rotary_embedding);
}

TORCH_LIBRARY(_C_cache_ops, cache_ops) {
  cache_ops.def("swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping) -> ()");
  cache_ops.impl("swap_blocks", torch::kCUDA, &swap_blocks);
  // a D2H swap dispatches on its CPU destination when src is listed first
  // only for CUDA; register CPU too so CPU->GPU and GPU->CPU both resolve
  cache_ops.impl("swap_blocks", torch::kCPU, &swap_blocks);

  cache_ops.def(
      "copy_blocks(Tensor[]! key_caches, Tensor[]! value_caches, Tensor "
      "block_mapping) -> ()");
  cache_ops.impl("copy_blocks", torch::kCUDA, &copy_blocks);

  cache_ops.def(
      "reshape_and_cache(Tensor key, Tensor value,"
      "                  Tensor! key_cache, Tensor! value_cache,"
      "                  Tensor slot_mapping,"
      "                  str kv_cache_dtype,"
      "                  float k_scale, float v_scale) -> ()");
  cache_ops.impl("reshape_and_cache", torch::kCUDA, &reshape_and_cache);

  cache_ops.def(
      "reshape_and_cache_flash(Tensor key, Tensor value,"
      "                        Tensor! key_cache,"
      "                        Tensor! value_cache,"
      "                        Tensor slot_mapping,"
      "                        str kv_cache_dtype,"
      "                        float k_scale, float v_scale) -> ()");
  cache_ops.impl("reshape_and_cache_flash", torch::kCUDA, &reshape_and_cache_flash);
}

TORCH_LIBRARY(_C_cuda_utils, cuda_utils) {
  cuda_utils.def("get_device_attribute", &get_device_attribute);
  cuda_utils.def("get_max_shared_memory_per_block_device_attribute",
                 &get_max_shared_memory_per_block_device_attribute);
}

// importable as a Python module (`import light_vllm_amd._C` style), as
// csrc/core/registration.h:18-22 does for the reference's extension
PyMODINIT_FUNC PyInit__C(void) {
  static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_C", nullptr, 0, nullptr};
  return PyModule_Create(&module);
}
