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

// "auto": the cache holds the query's element type; "fp8": one byte per element
void check_cache_dtype(const torch::Tensor& cache, const torch::Tensor& like, int kv_code, const char* op) {
  if (kv_code == LVLLM_KV_AUTO) {
    TORCH_CHECK(cache.scalar_type() == like.scalar_type(), op, ": kv_cache_dtype 'auto' needs a cache of dtype ",
                like.scalar_type(), ", got ", cache.scalar_type());
  } else {
    TORCH_CHECK(cache.element_size() == 1, op, ": kv_cache_dtype 'fp8' needs a one-byte cache dtype, got ",
                cache.scalar_type());
  }
}

// Bytes the kernels may touch behind key_cache.data_ptr() / value_cache.data_ptr() (the smaller of the two):
// the C-ABI clamps block numbers / skips slots beyond it instead of following them out of the allocation.
int64_t cache_extent_bytes(const torch::Tensor& key_cache, const torch::Tensor& value_cache) {
  auto bytes = [](const torch::Tensor& t) {
    return t.dim() == 0 ? (int64_t)0 : t.size(0) * t.stride(0) * (int64_t)t.element_size();
  };
  const int64_t k = bytes(key_cache), v = bytes(value_cache);
  return k < v ? k : v;
}

// A paged cache as the kernels address it: every block dense ([KVH, D/x, BS, x] or [KVH, D, BS] contiguous), blocks
// stride(0) elements apart -- the dense product, or more when the allocation pads its blocks.
bool blocks_are_dense(const torch::Tensor& t) {
  if (t.dim() < 2) return t.is_contiguous();
  int64_t expect = 1;
  for (int64_t d = t.dim() - 1; d >= 1; --d) {
    if (t.size(d) != 1 && t.stride(d) != expect) return false;
    expect *= t.size(d);
  }
  return t.stride(0) >= expect;
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
  check_cache_dtype(key_cache, query, kv_dtype_code(kv_cache_dtype), "paged_attention_v1");
  check_cache_dtype(value_cache, query, kv_dtype_code(kv_cache_dtype), "paged_attention_v1");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(query));
  const float* alibi = alibi_slopes ? alibi_slopes->data_ptr<float>() : nullptr;
  check(lvllm_paged_attention_v1(
      out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
      (int)query.size(0), (int)query.size(1), (int)query.size(2), (int)num_kv_heads, (float)scale,
      block_tables.data_ptr<int>(), seq_lens.data_ptr<int>(), (int)block_size, (int)max_seq_len,
      (int)block_tables.size(1), alibi, query.stride(0), key_cache.stride(0), key_cache.stride(1),
      dtype_code(query, "paged_attention_v1"), kv_dtype_code(kv_cache_dtype), (float)k_scale,
      (float)v_scale, (int)tp_rank, (int)blocksparse_local_blocks, (int)blocksparse_vert_stride,
      (int)blocksparse_block_size, (int)blocksparse_head_sliding_step, cache_extent_bytes(key_cache, value_cache),
      current_stream(query)));
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
  check_cache_dtype(key_cache, query, kv_dtype_code(kv_cache_dtype), "paged_attention_v2");
  check_cache_dtype(value_cache, query, kv_dtype_code(kv_cache_dtype), "paged_attention_v2");
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
      (int)blocksparse_head_sliding_step, cache_extent_bytes(key_cache, value_cache), current_stream(query)));
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
  // a block and its padding, if the allocation pads (both sides must pad alike: runs of blocks move as one copy)
  TORCH_CHECK(blocks_are_dense(src) && blocks_are_dense(dst) && src.dim() >= 1 && dst.dim() >= 1 &&
                  src.stride(0) == dst.stride(0) && src.element_size() == dst.element_size(),
              "swap_blocks: src and dst must have dense blocks and the same block stride");
  const int64_t block_bytes = src.element_size() * src.stride(0);
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
  // (a block and its padding, if the allocation pads: every layer's caches pad alike)
  for (int l = 0; l < num_layers; ++l)
    TORCH_CHECK(blocks_are_dense(key_caches[l]) && blocks_are_dense(value_caches[l]) &&
                    key_caches[l].stride(0) == key_caches[0].stride(0) &&
                    value_caches[l].stride(0) == key_caches[0].stride(0),
                "copy_blocks: dense blocks, one block stride for every cache");
  const int64_t block_bytes = key_caches[0].element_size() * key_caches[0].stride(0);
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
  TORCH_CHECK(blocks_are_dense(key_cache) && blocks_are_dense(value_cache) &&
                  key_cache.stride(0) == value_cache.stride(0),
              "key_cache / value_cache: dense blocks, the same block stride");
  check_cache_dtype(key_cache, key, kv_dtype_code(kv_cache_dtype), "reshape_and_cache");
  check_cache_dtype(value_cache, key, kv_dtype_code(kv_cache_dtype), "reshape_and_cache");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(key));
  check(lvllm_reshape_and_cache_strided(key.data_ptr(), value.data_ptr(), key_cache.data_ptr(),
                                        value_cache.data_ptr(), slot_mapping.data_ptr<int64_t>(),
                                        (int)key.size(0), (int)key.size(1), (int)key.size(2),
                                        (int)key_cache.size(3), (int)key_cache.size(4), key.stride(0),
                                        value.stride(0), dtype_code(key, "reshape_and_cache"),
                                        kv_dtype_code(kv_cache_dtype), (float)k_scale, (float)v_scale,
                                        cache_extent_bytes(key_cache, value_cache), key_cache.stride(0),
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

// Extension ops (not in the reference): F.linear for decode batches through the
// weight-streaming kernel.  `packed_shape` = [N, K] when `w` is in lvllm_pack_weight order.
torch::Tensor skinny_linear_impl(const torch::Tensor& x, const torch::Tensor& w,
                                 const std::optional<torch::Tensor>& bias, bool packed, int64_t N,
                                 int64_t K) {
  const int64_t M = x.size(0);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
  auto y = torch::empty({M, N}, x.options());
  const int64_t ws_bytes = lvllm_skinny_gemm_workspace_bytes((int)M, (int)N, (int)K);
  torch::Tensor ws;
  if (ws_bytes > 0) ws = torch::empty({ws_bytes / 4}, x.options().dtype(torch::kFloat));
  const int rc = lvllm_skinny_gemm(y.data_ptr(), x.data_ptr(), w.data_ptr(), bias ? bias->data_ptr() : nullptr,
                                   (int)M, (int)N, (int)K, x.stride(0), dtype_code(x, "skinny_linear"),
                                   packed ? 1 : 0, ws_bytes > 0 ? ws.data_ptr() : nullptr, ws_bytes,
                                   current_stream(x));
  if (rc == 3) return torch::Tensor();  // outside the envelope
  check(rc);
  return y;
}

bool skinny_ok(const torch::Tensor& x, const torch::Tensor& w, const std::optional<torch::Tensor>& bias) {
  return x.is_cuda() && x.dim() == 2 && (x.scalar_type() == at::kBFloat16 || x.scalar_type() == at::kHalf) &&
         w.scalar_type() == x.scalar_type() && w.is_contiguous() && x.stride(1) == 1 && x.size(0) <= 64 &&
         x.size(0) > 0 && (!bias || bias->is_contiguous());
}

torch::Tensor skinny_linear(const torch::Tensor& x, const torch::Tensor& w,
                            const std::optional<torch::Tensor>& bias) {
  TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && x.size(1) == w.size(1), "skinny_linear: x [M,K], w [N,K]");
  if (skinny_ok(x, w, bias)) {
    auto y = skinny_linear_impl(x, w, bias, false, w.size(0), w.size(1));
    if (y.defined()) return y;
  }
  return at::linear(x, w, bias);
}

torch::Tensor skinny_linear_packed(const torch::Tensor& x, const torch::Tensor& w_packed,
                                   const std::optional<torch::Tensor>& bias, int64_t N, int64_t K) {
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && w_packed.numel() == N * K, "skinny_linear_packed: bad shapes");
  TORCH_CHECK(skinny_ok(x, w_packed, bias), "skinny_linear_packed: M <= 64 rows of bf16/f16 on the GPU only");
  auto y = skinny_linear_impl(x, w_packed, bias, true, N, K);
  TORCH_CHECK(y.defined(), lvllm_last_error());
  return y;
}

// lm_head + greedy sampling: int64 [M] = argmax(x . w^T) over the rounded values, no logits tensor (in the
// projection's epilogue, or inside its split-K reduce pass when K is split over workgroups at this M).
torch::Tensor skinny_linear_packed_argmax(const torch::Tensor& x, const torch::Tensor& w_packed, int64_t N, int64_t K) {
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && w_packed.numel() == N * K, "skinny_linear_packed_argmax: bad shapes");
  TORCH_CHECK(skinny_ok(x, w_packed, std::nullopt), "skinny_linear_packed_argmax: M <= 64 rows of bf16/f16 on the GPU only");
  const int64_t M = x.size(0);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
  auto tokens = torch::empty({M}, x.options().dtype(torch::kLong));
  const int64_t ws_bytes = lvllm_skinny_gemm_argmax_workspace_bytes_ex((int)M, (int)N, (int)K);
  auto ws = torch::empty({ws_bytes / 4}, x.options().dtype(torch::kFloat));
  check(lvllm_skinny_gemm_argmax(tokens.data_ptr<int64_t>(), x.data_ptr(), w_packed.data_ptr(), (int)M, (int)N, (int)K,
                                 x.stride(0), dtype_code(x, "skinny_linear_packed_argmax"), ws.data_ptr(), ws_bytes,
                                 current_stream(x)));
  return tokens;
}

// 65..256 rows (large decode batches, prefill chunks): X through LDS, see lvllm_stream_gemm
torch::Tensor stream_linear_packed(const torch::Tensor& x, const torch::Tensor& w_packed,
                                   const std::optional<torch::Tensor>& bias, int64_t N, int64_t K) {
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && w_packed.numel() == N * K && x.stride(1) == 1,
              "stream_linear_packed: bad shapes");
  TORCH_CHECK(x.is_cuda() && w_packed.is_cuda() && w_packed.is_contiguous() && x.size(0) >= 1 && x.size(0) <= 256 &&
                  (x.scalar_type() == at::kBFloat16 || x.scalar_type() == at::kHalf) &&
                  w_packed.scalar_type() == x.scalar_type() && (!bias || bias->is_contiguous()),
              "stream_linear_packed: 1..256 rows of bf16/f16 on the GPU");
  const int64_t M = x.size(0);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
  auto y = torch::empty({M, N}, x.options());
  const int64_t ws_bytes = lvllm_stream_gemm_workspace_bytes((int)M, (int)N, (int)K);
  torch::Tensor ws;
  if (ws_bytes > 0) ws = torch::empty({ws_bytes / 4}, x.options().dtype(torch::kFloat));
  check(lvllm_stream_gemm(y.data_ptr(), x.data_ptr(), w_packed.data_ptr(), bias ? bias->data_ptr() : nullptr, (int)M,
                          (int)N, (int)K, x.stride(0), dtype_code(x, "stream_linear_packed"),
                          ws_bytes > 0 ? ws.data_ptr() : nullptr, ws_bytes, current_stream(x)));
  return y;
}

// gate_up projection + silu_and_mul in one launch: w_packed holds [gate rows | up rows] ([N, K]
// packed), the result is [M, N / 2].  Bit-identical to skinny_linear_packed followed by silu_and_mul.
torch::Tensor skinny_linear_packed_swiglu(const torch::Tensor& x, const torch::Tensor& w_packed,
                                          const std::optional<torch::Tensor>& bias, int64_t N, int64_t K) {
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && w_packed.numel() == N * K && N % 32 == 0,
              "skinny_linear_packed_swiglu: bad shapes");
  TORCH_CHECK(skinny_ok(x, w_packed, bias), "skinny_linear_packed_swiglu: M <= 64 rows of bf16/f16 on the GPU only");
  const int64_t M = x.size(0);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
  auto y = torch::empty({M, N / 2}, x.options());
  // K split over workgroups at this M: the fp32 partials are summed and activated by one reduce launch
  const int64_t ws_bytes = lvllm_skinny_gemm_workspace_bytes((int)M, (int)N, (int)K);
  torch::Tensor ws;
  if (ws_bytes > 0) ws = torch::empty({ws_bytes / 4}, x.options().dtype(torch::kFloat));
  check(lvllm_skinny_gemm_ex(y.data_ptr(), x.data_ptr(), w_packed.data_ptr(), bias ? bias->data_ptr() : nullptr,
                             (int)M, (int)N, (int)K, x.stride(0), dtype_code(x, "skinny_linear_packed_swiglu"), 1, 2,
                             0, nullptr, ws_bytes > 0 ? ws.data_ptr() : nullptr, ws_bytes, current_stream(x)));
  return y;
}

// A projection that leaves its fp32 split-K partials [S, M, N] for fused_add_rms_norm_splitk
// instead of reducing them itself.  swiglu: x = [gate | up] ([M, 2K]) and the activation is applied
// while loading x (bit-identical to silu_and_mul, but every workgroup redoes it for its K slice:
// measured 2x slower than the separate activation kernel at inter=14336 -- kept for small shapes).
torch::Tensor skinny_linear_packed_partials(const torch::Tensor& gate_up, const torch::Tensor& w_packed, int64_t N,
                                            int64_t K, bool swiglu) {
  TORCH_CHECK(gate_up.dim() == 2 && gate_up.size(1) == (swiglu ? 2 * K : K) && w_packed.numel() == N * K,
              "bad shapes");
  TORCH_CHECK(skinny_ok(gate_up, w_packed, std::nullopt), "M <= 64 rows of bf16/f16 on the GPU only");
  const int64_t M = gate_up.size(0);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(gate_up));
  int64_t ws_bytes = lvllm_skinny_gemm_workspace_bytes((int)M, (int)N, (int)K);
  if (ws_bytes == 0) ws_bytes = M * N * 4;
  int64_t S = ws_bytes / (M * N * 4);
  if (S < 16) S = 16;  // room for a forced finer split (tuning "gemm_partials_ksplit"); the slabs used are returned
  ws_bytes = S * M * N * 4;
  auto partials = torch::empty({S, M, N}, gate_up.options().dtype(torch::kFloat));
  int ksplit = 0;
  check(lvllm_skinny_gemm_ex(nullptr, gate_up.data_ptr(), w_packed.data_ptr(), nullptr, (int)M, (int)N, (int)K,
                             gate_up.stride(0), dtype_code(gate_up, "skinny_linear_packed_partials"), 1,
                             swiglu ? 1 : 0, 1, &ksplit,
                             partials.data_ptr(), ws_bytes, current_stream(gate_up)));
  TORCH_CHECK(ksplit >= 1 && ksplit <= S, "split count mismatch");
  return ksplit == S ? partials : partials.narrow(0, 0, ksplit);
}

void gelu(torch::Tensor& out, const torch::Tensor& x) {
  LV_CHECK_DEVICE(x);
  TORCH_CHECK(x.is_contiguous() && out.is_contiguous() && out.sizes() == x.sizes() && out.scalar_type() == x.scalar_type(),
              "gelu: contiguous out and x of the same shape and type");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
  check(lvllm_gelu(out.data_ptr(), x.data_ptr(), x.numel(), dtype_code(x, "gelu"), current_stream(x)));
}

void add_layer_norm(torch::Tensor& out, const torch::Tensor& x, const std::optional<torch::Tensor>& y,
                    const torch::Tensor& weight, const torch::Tensor& bias, double epsilon) {
  LV_CHECK_DEVICE(x);
  TORCH_CHECK(x.is_contiguous() && out.is_contiguous() && out.sizes() == x.sizes() && out.scalar_type() == x.scalar_type(),
              "add_layer_norm: contiguous out and x of the same shape and type");
  TORCH_CHECK(!y || (y->is_contiguous() && y->sizes() == x.sizes() && y->scalar_type() == x.scalar_type()),
              "add_layer_norm: y must match x");
  const int hidden = (int)x.size(-1);
  TORCH_CHECK(weight.numel() == hidden && bias.numel() == hidden && weight.is_contiguous() && bias.is_contiguous() &&
                  weight.scalar_type() == x.scalar_type() && bias.scalar_type() == x.scalar_type(),
              "add_layer_norm: weight and bias of [hidden] in x's type");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
  check(lvllm_add_layer_norm(out.data_ptr(), x.data_ptr(), y ? y->data_ptr() : nullptr, weight.data_ptr(),
                             bias.data_ptr(), (float)epsilon, (int)(x.numel() / hidden), hidden,
                             dtype_code(x, "add_layer_norm"), current_stream(x)));
}

void fused_add_rms_norm_splitk_scaled(torch::Tensor& out, torch::Tensor& residual, const torch::Tensor& partials,
                                      const torch::Tensor& weight, double epsilon,
                                      const c10::optional<torch::Tensor>& x_scale,
                                      const c10::optional<torch::Tensor>& w_scale) {
  TORCH_CHECK(partials.dim() == 3 && partials.scalar_type() == at::kFloat && partials.is_contiguous());
  TORCH_CHECK(out.is_contiguous() && residual.is_contiguous() && out.sizes() == residual.sizes());
  TORCH_CHECK(x_scale.has_value() == w_scale.has_value(), "x_scale and w_scale: both or none");
  if (x_scale.has_value())
    TORCH_CHECK(x_scale->scalar_type() == at::kFloat && w_scale->scalar_type() == at::kFloat && x_scale->is_cuda() &&
                w_scale->is_cuda() && x_scale->numel() == 1 && w_scale->numel() == 1, "scales: one float each, on the GPU");
  const int hidden = (int)out.size(-1);
  const int num_tokens = (int)(out.numel() / hidden);
  TORCH_CHECK(partials.size(1) == num_tokens && partials.size(2) == hidden);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(out));
  check(lvllm_fused_add_rms_norm_splitk_scaled(
      out.data_ptr(), residual.data_ptr(), partials.data_ptr<float>(), (int)partials.size(0), weight.data_ptr(),
      (float)epsilon, num_tokens, hidden, dtype_code(out, "fused_add_rms_norm_splitk"),
      x_scale.has_value() ? x_scale->data_ptr<float>() : nullptr,
      w_scale.has_value() ? w_scale->data_ptr<float>() : nullptr, current_stream(out)));
}

void fused_add_rms_norm_splitk(torch::Tensor& out, torch::Tensor& residual, const torch::Tensor& partials,
                               const torch::Tensor& weight, double epsilon) {
  fused_add_rms_norm_splitk_scaled(out, residual, partials, weight, epsilon, c10::nullopt, c10::nullopt);
}

// A W8A8 projection whose K is split over workgroups, leaving the raw fp32 partials [S, M, N] for
// fused_add_rms_norm_splitk_scaled.  Returns an EMPTY tensor when K is not split at this shape (nothing was done).
torch::Tensor skinny_linear_w8a8_partials(const torch::Tensor& x, const torch::Tensor& w_packed,
                                          const torch::Tensor& w_scale, const torch::Tensor& x_scale, int64_t N,
                                          int64_t K) {
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && x.is_cuda() && x.stride(1) == 1, "bad x");
  const int64_t M = x.size(0);
  const int64_t ws_bytes = lvllm_skinny_gemm_w8a8_workspace_bytes((int)M, (int)N, (int)K);
  if (ws_bytes == 0 || M == 0) return torch::empty({0}, x.options().dtype(torch::kFloat));
  const int64_t S = ws_bytes / (M * N * 4);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
  auto partials = torch::empty({S, M, N}, x.options().dtype(torch::kFloat));
  const int rc = lvllm_skinny_gemm_w8a8_ex(nullptr, x.data_ptr(), w_packed.data_ptr(), nullptr,
                                           x_scale.data_ptr<float>(), w_scale.data_ptr<float>(), (int)M, (int)N,
                                           (int)K, x.stride(0), dtype_code(x, "skinny_linear_w8a8_partials"), 4,
                                           partials.data_ptr(), ws_bytes, current_stream(x));
  if (rc == 3) return torch::empty({0}, x.options().dtype(torch::kFloat));
  check(rc);
  return partials;
}

// returns false when the arguments are outside the fused kernel's envelope (nothing was done)
bool rotary_embedding_and_cache(torch::Tensor& positions, torch::Tensor& query, torch::Tensor& key,
                                const torch::Tensor& value, int64_t head_size, torch::Tensor& cos_sin_cache,
                                bool is_neox, torch::Tensor& key_cache, torch::Tensor& value_cache,
                                const torch::Tensor& slot_mapping, const std::string& kv_cache_dtype,
                                double k_scale, double v_scale) {
  LV_CHECK_DEVICE(query);
  TORCH_CHECK(positions.scalar_type() == at::kLong && slot_mapping.scalar_type() == at::kLong);
  const int kv_code = kv_dtype_code(kv_cache_dtype);
  check_cache_dtype(key_cache, query, kv_code, "rotary_embedding_and_cache");
  check_cache_dtype(value_cache, query, kv_code, "rotary_embedding_and_cache");
  const int64_t num_tokens = query.numel() / query.size(-1);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(query));
  TORCH_CHECK(blocks_are_dense(key_cache) && blocks_are_dense(value_cache) &&
                  key_cache.stride(0) == value_cache.stride(0),
              "key_cache / value_cache: dense blocks, the same block stride");
  const int rc = lvllm_rotary_embedding_and_cache_strided(
      positions.data_ptr<int64_t>(), query.data_ptr(), key.data_ptr(), value.data_ptr(), (int)num_tokens,
      (int)(query.size(-1) / head_size), (int)(key.size(-1) / head_size), (int)head_size,
      (int)cos_sin_cache.size(1), query.stride(-2), key.stride(-2), value.stride(-2), cos_sin_cache.data_ptr(),
      is_neox ? 1 : 0, key_cache.data_ptr(), value_cache.data_ptr(), slot_mapping.data_ptr<int64_t>(),
      (int)value_cache.size(3), dtype_code(query, "rotary_embedding_and_cache"), kv_code, (float)k_scale,
      (float)v_scale, cache_extent_bytes(key_cache, value_cache), key_cache.stride(0), current_stream(query));
  if (rc == 3) return false;
  check(rc);
  return true;
}

// the QKV projection's split-K reduce + rope + cache write in one launch (include/lvllm_hip.h); false = outside the
// fused kernel's envelope (nothing was done)
bool rotary_embedding_and_cache_splitk(torch::Tensor& positions, torch::Tensor& qkv, const torch::Tensor& partials,
                                       const std::optional<torch::Tensor>& bias, int64_t num_heads, int64_t num_kv_heads,
                                       int64_t head_size, torch::Tensor& cos_sin_cache, bool is_neox,
                                       torch::Tensor& key_cache, torch::Tensor& value_cache,
                                       const torch::Tensor& slot_mapping, const std::string& kv_cache_dtype,
                                       double k_scale, double v_scale) {
  LV_CHECK_DEVICE(qkv);
  TORCH_CHECK(positions.scalar_type() == at::kLong && slot_mapping.scalar_type() == at::kLong);
  const int64_t row = (num_heads + 2 * num_kv_heads) * head_size;
  TORCH_CHECK(qkv.dim() == 2 && qkv.is_contiguous() && qkv.size(1) == row, "qkv: contiguous [tokens, (H + 2 KVH) * D]");
  TORCH_CHECK(partials.scalar_type() == at::kFloat && partials.is_contiguous() && partials.dim() == 3 &&
                  partials.size(1) == qkv.size(0) && partials.size(2) == row && partials.device() == qkv.device(),
              "partials: contiguous fp32 [slabs, tokens, (H + 2 KVH) * D] on qkv's device");
  TORCH_CHECK(!bias.has_value() || (bias->numel() == row && bias->scalar_type() == qkv.scalar_type() && bias->is_contiguous()),
              "bias: [(H + 2 KVH) * D] of qkv's type");
  const int kv_code = kv_dtype_code(kv_cache_dtype);
  check_cache_dtype(key_cache, qkv, kv_code, "rotary_embedding_and_cache_splitk");
  check_cache_dtype(value_cache, qkv, kv_code, "rotary_embedding_and_cache_splitk");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(qkv));
  TORCH_CHECK(blocks_are_dense(key_cache) && blocks_are_dense(value_cache) &&
                  key_cache.stride(0) == value_cache.stride(0),
              "key_cache / value_cache: dense blocks, the same block stride");
  const int rc = lvllm_rotary_embedding_and_cache_splitk(
      positions.data_ptr<int64_t>(), qkv.data_ptr(), partials.data_ptr<float>(), (int)partials.size(0),
      bias.has_value() ? bias->data_ptr() : nullptr, (int)qkv.size(0), (int)num_heads, (int)num_kv_heads, (int)head_size,
      (int)cos_sin_cache.size(1), cos_sin_cache.data_ptr(), is_neox ? 1 : 0, key_cache.data_ptr(), value_cache.data_ptr(),
      slot_mapping.data_ptr<int64_t>(), (int)value_cache.size(3), dtype_code(qkv, "rotary_embedding_and_cache_splitk"),
      kv_code, (float)k_scale, (float)v_scale, cache_extent_bytes(key_cache, value_cache), key_cache.stride(0),
      current_stream(qkv));
  if (rc == 3) return false;
  check(rc);
  return true;
}

// returns false when the arguments are outside the fused kernel's envelope (nothing was done)
bool rope_cache_paged_attention(torch::Tensor& out, torch::Tensor& exp_sums, torch::Tensor& max_logits,
                                torch::Tensor& tmp_out, const torch::Tensor& positions, const torch::Tensor& query,
                                const torch::Tensor& key, const torch::Tensor& value, int64_t head_size,
                                const torch::Tensor& cos_sin_cache, bool is_neox, torch::Tensor& key_cache,
                                torch::Tensor& value_cache, const torch::Tensor& slot_mapping, int64_t num_kv_heads,
                                double scale, const torch::Tensor& block_tables, const torch::Tensor& seq_lens,
                                int64_t block_size, int64_t max_seq_len, const std::string& kv_cache_dtype,
                                double k_scale, double v_scale, const c10::optional<torch::Tensor>& out_fp8,
                                const c10::optional<torch::Tensor>& out_fp8_scale) {
  LV_CHECK_DEVICE(query);
  TORCH_CHECK(positions.scalar_type() == at::kLong && slot_mapping.scalar_type() == at::kLong);
  TORCH_CHECK(out_fp8.has_value() == out_fp8_scale.has_value(), "out_fp8 and out_fp8_scale: both or none");
  if (out_fp8.has_value())
    TORCH_CHECK(out_fp8->is_cuda() && out_fp8->element_size() == 1 && out_fp8->is_contiguous() &&
                out_fp8->numel() == out.numel() && out_fp8_scale->is_cuda() && out_fp8_scale->scalar_type() == at::kFloat &&
                out_fp8_scale->numel() == 1, "out_fp8: bytes of out's shape; out_fp8_scale: one float on the device");
  TORCH_CHECK(query.dim() == 2 && key.dim() == 2 && value.dim() == 2, "query / key / value must be [tokens, heads * head_size]");
  TORCH_CHECK(block_tables.scalar_type() == at::kInt && seq_lens.scalar_type() == at::kInt,
              "block_tables / seq_lens must be int32");
  TORCH_CHECK(out.is_contiguous() && tmp_out.is_contiguous() && exp_sums.is_contiguous() && max_logits.is_contiguous());
  const int kv_code = kv_dtype_code(kv_cache_dtype);
  check_cache_dtype(key_cache, query, kv_code, "rope_cache_paged_attention");
  check_cache_dtype(value_cache, query, kv_code, "rope_cache_paged_attention");
  const int64_t num_seqs = query.numel() / query.size(-1);
  const int64_t num_heads = query.size(-1) / head_size;
  TORCH_CHECK(key.size(-1) == num_kv_heads * head_size && value.size(-1) == num_kv_heads * head_size);
  TORCH_CHECK(out.numel() == num_seqs * num_heads * head_size);
  TORCH_CHECK(positions.numel() >= num_seqs && slot_mapping.numel() >= num_seqs && seq_lens.numel() >= num_seqs &&
              block_tables.size(0) >= num_seqs);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(query));
  const int rc = out_fp8.has_value() ? lvllm_rope_cache_paged_attention_q(
      out.data_ptr(), out_fp8->data_ptr(), out_fp8_scale->data_ptr<float>(), exp_sums.data_ptr<float>(),
      max_logits.data_ptr<float>(), tmp_out.data_ptr(), query.data_ptr(),
      key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), (int)num_seqs, (int)num_heads,
      (int)head_size, (int)num_kv_heads, (float)scale, block_tables.data_ptr<int32_t>(), seq_lens.data_ptr<int32_t>(),
      positions.data_ptr<int64_t>(), slot_mapping.data_ptr<int64_t>(), cos_sin_cache.data_ptr(),
      (int)cos_sin_cache.size(1), is_neox ? 1 : 0, (int)block_size, (int)max_seq_len, (int)block_tables.size(1),
      (int)exp_sums.size(-1), query.stride(-2), key.stride(-2), value.stride(-2), key_cache.stride(0),
      key_cache.stride(1), dtype_code(query, "rope_cache_paged_attention"), kv_code, (float)k_scale, (float)v_scale,
      cache_extent_bytes(key_cache, value_cache), current_stream(query))
                                     : lvllm_rope_cache_paged_attention(
      out.data_ptr(), exp_sums.data_ptr<float>(), max_logits.data_ptr<float>(), tmp_out.data_ptr(), query.data_ptr(),
      key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), (int)num_seqs, (int)num_heads,
      (int)head_size, (int)num_kv_heads, (float)scale, block_tables.data_ptr<int32_t>(), seq_lens.data_ptr<int32_t>(),
      positions.data_ptr<int64_t>(), slot_mapping.data_ptr<int64_t>(), cos_sin_cache.data_ptr(),
      (int)cos_sin_cache.size(1), is_neox ? 1 : 0, (int)block_size, (int)max_seq_len, (int)block_tables.size(1),
      (int)exp_sums.size(-1), query.stride(-2), key.stride(-2), value.stride(-2), key_cache.stride(0),
      key_cache.stride(1), dtype_code(query, "rope_cache_paged_attention"), kv_code, (float)k_scale, (float)v_scale,
      cache_extent_bytes(key_cache, value_cache), current_stream(query));
  if (rc == 3) return false;
  check(rc);
  return true;
}

// paged_attention_v2 with an fp8 twin of its result (lvllm_paged_attention_v2_q); false = the launch would be cut into
// shares (nothing was done: call paged_attention_v2)
bool paged_attention_v2_q(torch::Tensor& out, torch::Tensor& out_fp8, const torch::Tensor& out_fp8_scale,
                          torch::Tensor& exp_sums, torch::Tensor& max_logits, torch::Tensor& tmp_out,
                          const torch::Tensor& query, const torch::Tensor& key_cache, const torch::Tensor& value_cache,
                          int64_t num_kv_heads, double scale, const torch::Tensor& block_tables,
                          const torch::Tensor& seq_lens, int64_t block_size, int64_t max_seq_len,
                          const std::string& kv_cache_dtype, double k_scale, double v_scale) {
  LV_CHECK_DEVICE(query);
  TORCH_CHECK(query.dim() == 3 && out.is_contiguous() && out.sizes() == query.sizes(), "query / out: [seqs, heads, head_size]");
  TORCH_CHECK(out_fp8.is_cuda() && out_fp8.element_size() == 1 && out_fp8.is_contiguous() && out_fp8.numel() == out.numel() &&
              out_fp8_scale.is_cuda() && out_fp8_scale.scalar_type() == at::kFloat && out_fp8_scale.numel() == 1,
              "out_fp8: bytes of out's shape; out_fp8_scale: one float on the device");
  TORCH_CHECK(block_tables.scalar_type() == at::kInt && seq_lens.scalar_type() == at::kInt, "block_tables / seq_lens must be int32");
  const int kv_code = kv_dtype_code(kv_cache_dtype);
  check_cache_dtype(key_cache, query, kv_code, "paged_attention_v2_q");
  check_cache_dtype(value_cache, query, kv_code, "paged_attention_v2_q");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(query));
  const int rc = lvllm_paged_attention_v2_q(
      out.data_ptr(), out_fp8.data_ptr(), out_fp8_scale.data_ptr<float>(), exp_sums.data_ptr<float>(),
      max_logits.data_ptr<float>(), tmp_out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
      (int)query.size(0), (int)query.size(1), (int)query.size(2), (int)num_kv_heads, (float)scale,
      block_tables.data_ptr<int32_t>(), seq_lens.data_ptr<int32_t>(), (int)block_size, (int)max_seq_len,
      (int)block_tables.size(1), (int)exp_sums.size(-1), query.stride(0), key_cache.stride(0), key_cache.stride(1),
      dtype_code(query, "paged_attention_v2_q"), kv_code, (float)k_scale, (float)v_scale,
      cache_extent_bytes(key_cache, value_cache), current_stream(query));
  if (rc == 3) return false;
  check(rc);
  return true;
}

void paged_prefill_attention(torch::Tensor& out, const torch::Tensor& query, const torch::Tensor& key_cache,
                             const torch::Tensor& value_cache, int64_t num_kv_heads, double scale,
                             const torch::Tensor& block_tables, const torch::Tensor& seq_lens,
                             const torch::Tensor& query_start_loc, int64_t max_query_len, int64_t block_size,
                             const c10::optional<torch::Tensor>& alibi_slopes, int64_t sliding_window,
                             double softcap, const std::string& kv_cache_dtype, bool causal, double k_scale,
                             double v_scale, int64_t max_seq_len) {
  LV_CHECK_DEVICE(query);
  LV_CHECK_DEVICE(out);
  check_cache_dtype(key_cache, query, kv_dtype_code(kv_cache_dtype), "paged_prefill_attention");
  check_cache_dtype(value_cache, query, kv_dtype_code(kv_cache_dtype), "paged_prefill_attention");
  TORCH_CHECK(query.dim() == 3 && out.dim() == 3, "paged_prefill_attention: query/out must be [T, H, D]");
  TORCH_CHECK(query.stride(2) == 1 && query.stride(1) == query.size(2) && out.stride(2) == 1 &&
                  out.stride(1) == out.size(2),
              "paged_prefill_attention: heads of a token must be contiguous");
  TORCH_CHECK(block_tables.scalar_type() == at::kInt && seq_lens.scalar_type() == at::kInt &&
                  query_start_loc.scalar_type() == at::kInt,
              "paged_prefill_attention: block_tables / seq_lens / query_start_loc must be int32");
  TORCH_CHECK(block_tables.dim() == 2 && block_tables.is_contiguous());
  const int64_t num_seqs = seq_lens.numel();
  TORCH_CHECK(query_start_loc.numel() == num_seqs + 1 && block_tables.size(0) >= num_seqs);
  TORCH_CHECK(value_cache.size(3) == block_size, "paged_prefill_attention: block_size does not match the cache");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(query));
  const float* alibi = alibi_slopes.has_value() ? alibi_slopes->data_ptr<float>() : nullptr;
  // short chunks with a known bound on seq_lens: scratch for a partitioned key walk (0 bytes: single pass)
  torch::Tensor workspace;
  const int64_t ws_bytes =
      max_seq_len > 0 ? lvllm_paged_prefill_workspace_bytes((int)num_seqs, (int)query.size(0), (int)max_query_len,
                                                            (int)query.size(1), (int)num_kv_heads, (int)query.size(2),
                                                            (int)max_seq_len)
                      : 0;
  if (ws_bytes > 0) workspace = torch::empty({ws_bytes}, query.options().dtype(torch::kUInt8));
  check(lvllm_paged_prefill_attention_ws(
      out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), (int)num_seqs,
      (int)query.size(1), (int)query.size(2), (int)num_kv_heads, (float)scale, block_tables.data_ptr<int32_t>(),
      seq_lens.data_ptr<int32_t>(), query_start_loc.data_ptr<int32_t>(), (int)max_query_len, (int)block_size,
      (int)block_tables.size(1), alibi, causal ? 1 : 0, (int)sliding_window, (float)softcap, query.stride(0),
      out.stride(0),
      key_cache.stride(0), key_cache.stride(1), dtype_code(query, "paged_prefill_attention"),
      kv_dtype_code(kv_cache_dtype), (float)k_scale, (float)v_scale, cache_extent_bytes(key_cache, value_cache),
      (int)query.size(0), (int)max_seq_len, ws_bytes > 0 ? workspace.data_ptr() : nullptr, ws_bytes,
      current_stream(query)));
}

int64_t paged_prefill_workspace_bytes(int64_t num_seqs, int64_t num_tokens, int64_t max_query_len, int64_t num_heads,
                                      int64_t num_kv_heads, int64_t head_size, int64_t max_seq_len) {
  return lvllm_paged_prefill_workspace_bytes((int)num_seqs, (int)num_tokens, (int)max_query_len, (int)num_heads,
                                             (int)num_kv_heads, (int)head_size, (int)max_seq_len);
}

int64_t varlen_attention_workspace_bytes(int64_t num_tokens, int64_t num_seqs, int64_t max_seq_len,
                                         int64_t num_kv_heads, int64_t head_size) {
  return lvllm_varlen_attention_workspace_bytes((int)num_tokens, (int)num_seqs, (int)max_seq_len,
                                                (int)num_kv_heads, (int)head_size);
}

void varlen_attention(torch::Tensor& out, const torch::Tensor& query, const torch::Tensor& key,
                      const torch::Tensor& value, const torch::Tensor& cu_seqlens, int64_t max_seq_len,
                      double scale, bool causal, const c10::optional<torch::Tensor>& alibi_slopes,
                      int64_t sliding_window, double softcap, torch::Tensor& workspace) {
  LV_CHECK_DEVICE(query);
  LV_CHECK_DEVICE(key);
  LV_CHECK_DEVICE(value);
  LV_CHECK_DEVICE(out);
  LV_CHECK_DEVICE(workspace);
  TORCH_CHECK(query.dim() == 3 && key.dim() == 3 && value.dim() == 3 && out.dim() == 3,
              "varlen_attention: query/key/value/out must be [T, heads, D]");
  for (const torch::Tensor* t : {&query, &key, &value, (const torch::Tensor*)&out})
    TORCH_CHECK(t->stride(2) == 1 && t->stride(1) == t->size(2), "varlen_attention: heads of a token must be contiguous");
  TORCH_CHECK(cu_seqlens.scalar_type() == at::kInt && cu_seqlens.is_cuda() && cu_seqlens.is_contiguous(),
              "varlen_attention: cu_seqlens must be a contiguous int32 device tensor");
  TORCH_CHECK(key.scalar_type() == query.scalar_type() && value.scalar_type() == query.scalar_type());
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(query));
  const float* alibi = alibi_slopes.has_value() ? alibi_slopes->data_ptr<float>() : nullptr;
  check(lvllm_varlen_attention(out.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(),
                               cu_seqlens.data_ptr<int32_t>(), (int)query.size(0), (int)cu_seqlens.numel() - 1,
                               (int)max_seq_len, (int)query.size(1), (int)key.size(1), (int)query.size(2),
                               (float)scale, causal ? 1 : 0, alibi, (int)sliding_window, (float)softcap,
                               query.stride(0), key.stride(0), value.stride(0), out.stride(0),
                               dtype_code(query, "varlen_attention"), workspace.data_ptr(),
                               (int64_t)workspace.nbytes(), current_stream(query)));
}

// csrc/prepare_inputs/advance_step.cu:59-125: same argument checks (verify_tensor)
void advance_step(int64_t num_seqs, int64_t num_queries, int64_t block_size, torch::Tensor& input_tokens,
                  torch::Tensor& sampled_token_ids, torch::Tensor& input_positions, torch::Tensor& seq_lens,
                  torch::Tensor& slot_mapping, torch::Tensor& block_tables) {
  auto verify = [](const char* name, const torch::Tensor& t, int64_t size_0, int64_t size_1, c10::ScalarType type) {
    const bool ok = (size_0 == -1 || t.size(0) == size_0) && (size_1 == -1 || (t.dim() > 1 && t.size(1) == size_1)) &&
                    t.is_contiguous() && t.scalar_type() == type && t.is_cuda();
    TORCH_CHECK(ok, "tensor: name = ", name, ", shape = ", t.sizes(), " is_cont = ", t.is_contiguous(),
                ", type = ", t.dtype(), " is not as expected: shape = [", size_0, ", ", size_1, "], type = ", type);
  };
  verify("input_tokens", input_tokens, num_seqs, -1, at::kLong);
  verify("sampled_token_ids", sampled_token_ids, num_queries, 1, at::kLong);
  verify("input_positions", input_positions, num_seqs, -1, at::kLong);
  verify("seq_lens", seq_lens, num_seqs, -1, at::kInt);
  verify("slot_mapping", slot_mapping, num_seqs, -1, at::kLong);
  verify("block_tables", block_tables, num_seqs, -1, at::kInt);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(sampled_token_ids));
  check(lvllm_advance_step((int)num_seqs, (int)num_queries, (int)block_size, input_tokens.data_ptr<int64_t>(),
                           sampled_token_ids.data_ptr<int64_t>(), input_positions.data_ptr<int64_t>(),
                           seq_lens.data_ptr<int32_t>(), slot_mapping.data_ptr<int64_t>(),
                           block_tables.data_ptr<int32_t>(), block_tables.stride(0),
                           current_stream(sampled_token_ids)));
}

// Extension: advance_step between the model steps of a multi-step decode.  Same arithmetic; the sampled
// ids are also logged into `token_log` and rows of the padded batch without a sequence are skipped.
void advance_step_logged(int64_t block_size, torch::Tensor& input_tokens, const torch::Tensor& sampled_token_ids,
                         torch::Tensor& input_positions, torch::Tensor& seq_lens, torch::Tensor& slot_mapping,
                         const torch::Tensor& block_tables, torch::Tensor& token_log) {
  const int64_t n = input_tokens.size(0);
  TORCH_CHECK(input_tokens.is_cuda() && input_tokens.scalar_type() == at::kLong && input_tokens.is_contiguous());
  TORCH_CHECK(sampled_token_ids.scalar_type() == at::kLong && sampled_token_ids.is_contiguous() &&
              sampled_token_ids.numel() >= n);
  TORCH_CHECK(input_positions.scalar_type() == at::kLong && input_positions.is_contiguous() && input_positions.size(0) == n);
  TORCH_CHECK(seq_lens.scalar_type() == at::kInt && seq_lens.is_contiguous() && seq_lens.size(0) == n);
  TORCH_CHECK(slot_mapping.scalar_type() == at::kLong && slot_mapping.is_contiguous() && slot_mapping.size(0) == n);
  TORCH_CHECK(block_tables.scalar_type() == at::kInt && block_tables.dim() == 2 && block_tables.size(0) == n &&
              block_tables.stride(1) == 1);
  TORCH_CHECK(token_log.scalar_type() == at::kLong && token_log.is_contiguous() && token_log.numel() >= n);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(input_tokens));
  check(lvllm_advance_step_ex((int)n, (int)n, (int)block_size, input_tokens.data_ptr<int64_t>(),
                              sampled_token_ids.data_ptr<int64_t>(), input_positions.data_ptr<int64_t>(),
                              seq_lens.data_ptr<int32_t>(), slot_mapping.data_ptr<int64_t>(),
                              block_tables.data_ptr<int32_t>(), block_tables.stride(0),
                              token_log.data_ptr<int64_t>(), 1, current_stream(input_tokens)));
}

// Device-side sampler (include/lvllm_hip.h: lvllm_sample_rows).  `params` uint8 [num_slots, 128], `counts` int32
// [num_slots, vocab], `scratch` float [rows, >= vocab]; all three absent = every row plain greedy.
void sample_rows(torch::Tensor& tokens_out, const torch::Tensor& logits, const std::optional<torch::Tensor>& state_slot,
                 const std::optional<torch::Tensor>& params, const std::optional<torch::Tensor>& counts,
                 const std::optional<torch::Tensor>& scratch, const std::optional<torch::Tensor>& processed_out,
                 bool update_state) {
  LV_CHECK_DEVICE(logits);
  TORCH_CHECK(logits.dim() == 2 && logits.stride(1) == 1, "sample_rows: logits [rows, vocab], unit inner stride");
  const int64_t rows = logits.size(0), vocab = logits.size(1);
  TORCH_CHECK(tokens_out.is_cuda() && tokens_out.scalar_type() == at::kLong && tokens_out.is_contiguous() &&
              tokens_out.numel() >= rows, "sample_rows: tokens_out int64 [rows]");
  const int32_t* slot = nullptr;
  void* pp = nullptr;
  int32_t* cc = nullptr;
  float* sc = nullptr;
  float* po = nullptr;
  int64_t cstride = 0, sstride = 0, pstride = 0, nslots = 0;
  if (state_slot.has_value()) {
    TORCH_CHECK(params.has_value() && counts.has_value() && scratch.has_value(),
                "sample_rows: state_slot needs params, counts and scratch");
    TORCH_CHECK(state_slot->is_cuda() && state_slot->scalar_type() == at::kInt && state_slot->is_contiguous() &&
                state_slot->numel() >= rows, "sample_rows: state_slot int32 [rows]");
    TORCH_CHECK(params->is_cuda() && params->scalar_type() == at::kByte && params->is_contiguous() && params->dim() == 2 &&
                params->size(1) == LVLLM_SAMPLER_PARAMS_BYTES, "sample_rows: params uint8 [slots, 128]");
    TORCH_CHECK(counts->is_cuda() && counts->scalar_type() == at::kInt && counts->dim() == 2 && counts->stride(1) == 1 &&
                counts->size(0) == params->size(0) && counts->size(1) >= vocab, "sample_rows: counts int32 [slots, vocab]");
    TORCH_CHECK(scratch->is_cuda() && scratch->scalar_type() == at::kFloat && scratch->dim() == 2 &&
                scratch->stride(1) == 1 && scratch->size(0) >= rows && scratch->size(1) >= vocab,
                "sample_rows: scratch float [rows, >= vocab]");
    slot = state_slot->data_ptr<int32_t>();
    pp = params->data_ptr();
    cc = counts->data_ptr<int32_t>();
    sc = scratch->data_ptr<float>();
    cstride = counts->stride(0);
    sstride = scratch->stride(0);
    nslots = params->size(0);
  }
  if (processed_out.has_value()) {
    TORCH_CHECK(processed_out->is_cuda() && processed_out->scalar_type() == at::kFloat && processed_out->dim() == 2 &&
                processed_out->stride(1) == 1 && processed_out->size(0) >= rows && processed_out->size(1) >= vocab,
                "sample_rows: processed_out float [rows, >= vocab]");
    po = processed_out->data_ptr<float>();
    pstride = processed_out->stride(0);
  }
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(logits));
  check(lvllm_sample_rows(tokens_out.data_ptr<int64_t>(), logits.data_ptr(), logits.stride(0),
                          dtype_code(logits, "sample_rows"), (int)rows, (int)vocab, slot, pp, cc, cstride, (int)nslots, sc,
                          sstride, po, pstride, update_state ? 1 : 0, current_stream(logits)));
}

void sampler_init_row(torch::Tensor& counts_row, const torch::Tensor& prompt_tokens, const torch::Tensor& output_tokens) {
  LV_CHECK_DEVICE(counts_row);
  TORCH_CHECK(counts_row.scalar_type() == at::kInt && counts_row.dim() == 1 && counts_row.stride(0) == 1);
  TORCH_CHECK(prompt_tokens.is_cuda() && prompt_tokens.scalar_type() == at::kLong && prompt_tokens.is_contiguous());
  TORCH_CHECK(output_tokens.is_cuda() && output_tokens.scalar_type() == at::kLong && output_tokens.is_contiguous());
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(counts_row));
  check(lvllm_sampler_init_row(counts_row.data_ptr<int32_t>(), (int)counts_row.size(0), prompt_tokens.data_ptr<int64_t>(),
                               (int)prompt_tokens.numel(), output_tokens.data_ptr<int64_t>(), (int)output_tokens.numel(),
                               current_stream(counts_row)));
}

// csrc/cache_kernels.cu:352-410
void convert_fp8(torch::Tensor& dst_cache, torch::Tensor& src_cache, const double scale,
                 const std::string& kv_cache_dtype) {
  TORCH_CHECK(src_cache.is_cuda(), "src must be on a GPU");
  TORCH_CHECK(dst_cache.is_cuda(), "dst must be on a GPU");
  TORCH_CHECK(src_cache.device().index() == dst_cache.device().index(), "src and dst must be on the same GPU");
  TORCH_CHECK(src_cache.is_contiguous() && dst_cache.is_contiguous() && src_cache.numel() == dst_cache.numel(),
              "convert_fp8: contiguous tensors of equal size");
  TORCH_CHECK(kv_cache_dtype == "fp8" || kv_cache_dtype == "fp8_e4m3", "Unsupported data type: ", kv_cache_dtype);
  const bool to_fp8 = dst_cache.element_size() == 1;
  TORCH_CHECK(to_fp8 != (src_cache.element_size() == 1), "convert_fp8: exactly one side must be a one-byte tensor");
  const torch::Tensor& wide = to_fp8 ? src_cache : dst_cache;
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(src_cache));
  check(lvllm_convert_fp8(dst_cache.data_ptr(), src_cache.data_ptr(), (float)scale, src_cache.numel(),
                          dtype_code(wide, "convert_fp8"), to_fp8 ? 1 : 0, LVLLM_KV_FP8_E4M3,
                          current_stream(src_cache)));
}

// fp8 activation quantisation, csrc/quantization/fp8/common.cu:226-292
void check_fp8_out(const torch::Tensor& out, const torch::Tensor& input, const char* op) {
  LV_CHECK_DEVICE(input);
  TORCH_CHECK(out.element_size() == 1, op, ": out must be a float8_e4m3fn (one-byte) tensor");
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), op, ": input / out must be contiguous");
}

void static_scaled_fp8_quant(torch::Tensor& out, torch::Tensor const& input, torch::Tensor const& scale) {
  check_fp8_out(out, input, "static_scaled_fp8_quant");
  TORCH_CHECK(scale.scalar_type() == at::kFloat && scale.is_cuda());
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(input));
  check(lvllm_static_scaled_fp8_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr<float>(), input.numel(),
                                      dtype_code(input, "static_scaled_fp8_quant"), current_stream(input)));
}

void dynamic_scaled_fp8_quant(torch::Tensor& out, torch::Tensor const& input, torch::Tensor& scale) {
  check_fp8_out(out, input, "dynamic_scaled_fp8_quant");
  TORCH_CHECK(scale.scalar_type() == at::kFloat && scale.is_cuda());
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(input));
  check(lvllm_dynamic_scaled_fp8_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr<float>(), input.numel(),
                                       dtype_code(input, "dynamic_scaled_fp8_quant"), current_stream(input)));
}

void dynamic_per_token_scaled_fp8_quant(torch::Tensor& out, torch::Tensor const& input, torch::Tensor& scales,
                                        std::optional<at::Tensor> const& scale_ub) {
  check_fp8_out(out, input, "dynamic_per_token_scaled_fp8_quant");
  TORCH_CHECK(scales.scalar_type() == at::kFloat && scales.is_cuda() && scales.is_contiguous());
  const int64_t hidden = input.size(-1);
  const int64_t tokens = input.numel() / hidden;
  TORCH_CHECK(scales.numel() >= tokens);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(input));
  check(lvllm_dynamic_per_token_scaled_fp8_quant(
      out.data_ptr(), scales.data_ptr<float>(), input.data_ptr(),
      scale_ub.has_value() ? scale_ub->data_ptr<float>() : nullptr, (int)tokens, (int)hidden,
      dtype_code(input, "dynamic_per_token_scaled_fp8_quant"), current_stream(input)));
}

torch::Tensor skinny_linear_w8a8(const torch::Tensor& x, const torch::Tensor& w_packed, const torch::Tensor& w_scale,
                                 const torch::Tensor& x_scale, int64_t N, int64_t K,
                                 const std::optional<torch::Tensor>& bias) {
  LV_CHECK_DEVICE(x);
  LV_CHECK_DEVICE(w_packed);
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && x.stride(1) == 1, "skinny_linear_w8a8: x must be [M, K]");
  TORCH_CHECK(w_packed.is_contiguous() && w_packed.numel() * w_packed.element_size() == N * K,
              "skinny_linear_w8a8: w_packed must hold N*K fp8 bytes");
  TORCH_CHECK(w_scale.scalar_type() == at::kFloat && x_scale.scalar_type() == at::kFloat && w_scale.is_cuda() &&
                  x_scale.is_cuda() && w_scale.numel() == 1 && x_scale.numel() == 1,
              "skinny_linear_w8a8: per-tensor float32 scales on the device");
  const int64_t M = x.size(0);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
  auto y = torch::empty({M, N}, x.options());
  const int64_t ws_bytes = lvllm_skinny_gemm_w8a8_workspace_bytes((int)M, (int)N, (int)K);
  torch::Tensor ws;
  if (ws_bytes > 0) ws = torch::empty({ws_bytes / 4}, x.options().dtype(torch::kFloat));
  check(lvllm_skinny_gemm_w8a8(y.data_ptr(), x.data_ptr(), w_packed.data_ptr(), bias ? bias->data_ptr() : nullptr,
                               x_scale.data_ptr<float>(), w_scale.data_ptr<float>(), (int)M, (int)N, (int)K,
                               x.stride(0), dtype_code(x, "skinny_linear_w8a8"),
                               ws_bytes > 0 ? ws.data_ptr() : nullptr, ws_bytes, current_stream(x)));
  return y;
}

// W8A8 gate_up projection + silu_and_mul in one launch ([M, N / 2]); falls back to the two launches
// when K is split over workgroups at this M.  Bit-identical to skinny_linear_w8a8 + silu_and_mul.
torch::Tensor skinny_linear_w8a8_swiglu(const torch::Tensor& x, const torch::Tensor& w_packed,
                                        const torch::Tensor& w_scale, const torch::Tensor& x_scale, int64_t N,
                                        int64_t K, const std::optional<torch::Tensor>& bias) {
  TORCH_CHECK(N % 32 == 0, "skinny_linear_w8a8_swiglu: N must be a multiple of 32");
  const int64_t M = x.size(0);
  auto y = torch::empty({M, N / 2}, x.options());
  if (lvllm_skinny_gemm_w8a8_workspace_bytes((int)M, (int)N, (int)K) > 0) {
    auto gate_up = skinny_linear_w8a8(x, w_packed, w_scale, x_scale, N, K, bias);
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
    check(lvllm_silu_and_mul(y.data_ptr(), gate_up.data_ptr(), M, (int)(N / 2), dtype_code(x, "silu_and_mul"),
                             current_stream(x)));
    return y;
  }
  LV_CHECK_DEVICE(x);
  LV_CHECK_DEVICE(w_packed);
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && x.stride(1) == 1, "skinny_linear_w8a8_swiglu: x must be [M, K]");
  TORCH_CHECK(w_packed.is_contiguous() && w_packed.numel() * w_packed.element_size() == N * K,
              "skinny_linear_w8a8_swiglu: w_packed must hold N*K fp8 bytes");
  TORCH_CHECK(w_scale.scalar_type() == at::kFloat && x_scale.scalar_type() == at::kFloat && w_scale.is_cuda() &&
                  x_scale.is_cuda() && w_scale.numel() == 1 && x_scale.numel() == 1,
              "skinny_linear_w8a8_swiglu: per-tensor float32 scales on the device");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
  check(lvllm_skinny_gemm_w8a8_ex(y.data_ptr(), x.data_ptr(), w_packed.data_ptr(), bias ? bias->data_ptr() : nullptr,
                                  x_scale.data_ptr<float>(), w_scale.data_ptr<float>(), (int)M, (int)N, (int)K,
                                  x.stride(0), dtype_code(x, "skinny_linear_w8a8_swiglu"), 2, nullptr, 0,
                                  current_stream(x)));
  return y;
}

// ---- the W8A8 decode step with activations quantised ONCE, by their producer (include/lvllm_hip.h: *_quant entries,
// lvllm_skinny_gemm_w8a8_q): every op below is bit-identical to the op it stands for followed by / preceded by
// static_scaled_fp8_quant
static void check_q_scale(const torch::Tensor& s, const char* op) {
  TORCH_CHECK(s.is_cuda() && s.scalar_type() == at::kFloat && s.numel() == 1, op, ": one float32 scale on the device");
}

// x_fp8 [T, hidden] uint8 = quant(rms_norm(input) * weight)
torch::Tensor rms_norm_fp8(const torch::Tensor& input, const torch::Tensor& weight, double epsilon,
                           const torch::Tensor& q_scale) {
  LV_CHECK_DEVICE(input);
  check_q_scale(q_scale, "rms_norm_fp8");
  TORCH_CHECK(input.dim() == 2 && input.is_contiguous() && weight.is_contiguous() && weight.numel() == input.size(1));
  auto out8 = torch::empty(input.sizes(), input.options().dtype(torch::kUInt8));
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(input));
  check(lvllm_rms_norm_quant(nullptr, out8.data_ptr(), q_scale.data_ptr<float>(), input.data_ptr(), weight.data_ptr(),
                             (float)epsilon, (int)input.size(0), (int)input.size(1), dtype_code(input, "rms_norm_fp8"),
                             current_stream(input)));
  return out8;
}

// residual <- input + residual; x_fp8 = quant(norm(residual) * weight); `input` is only read
torch::Tensor fused_add_rms_norm_fp8(const torch::Tensor& input, torch::Tensor& residual, const torch::Tensor& weight,
                                     double epsilon, const torch::Tensor& q_scale) {
  LV_CHECK_DEVICE(input);
  check_q_scale(q_scale, "fused_add_rms_norm_fp8");
  TORCH_CHECK(input.dim() == 2 && input.is_contiguous() && residual.is_contiguous() && input.sizes() == residual.sizes() &&
              input.scalar_type() == residual.scalar_type() && weight.is_contiguous() && weight.numel() == input.size(1));
  auto out8 = torch::empty(input.sizes(), input.options().dtype(torch::kUInt8));
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(input));
  check(lvllm_fused_add_rms_norm_quant(const_cast<void*>(input.data_ptr()), residual.data_ptr(), weight.data_ptr(),
                                       (float)epsilon, (int)input.size(0), (int)input.size(1),
                                       dtype_code(input, "fused_add_rms_norm_fp8"), out8.data_ptr(),
                                       q_scale.data_ptr<float>(), 0, current_stream(input)));
  return out8;
}

// the same behind a projection that left its fp32 split-K partials [S, T, hidden] (x_scale / w_scale: a W8A8 one)
torch::Tensor fused_add_rms_norm_splitk_fp8(torch::Tensor& residual, const torch::Tensor& partials,
                                            const torch::Tensor& weight, double epsilon,
                                            const c10::optional<torch::Tensor>& x_scale,
                                            const c10::optional<torch::Tensor>& w_scale, const torch::Tensor& q_scale) {
  LV_CHECK_DEVICE(residual);
  check_q_scale(q_scale, "fused_add_rms_norm_splitk_fp8");
  TORCH_CHECK(partials.dim() == 3 && partials.scalar_type() == at::kFloat && partials.is_contiguous());
  TORCH_CHECK(residual.dim() == 2 && residual.is_contiguous() && partials.size(1) == residual.size(0) &&
              partials.size(2) == residual.size(1));
  TORCH_CHECK(x_scale.has_value() == w_scale.has_value(), "x_scale and w_scale: both or none");
  if (x_scale.has_value()) {
    check_q_scale(*x_scale, "fused_add_rms_norm_splitk_fp8");
    check_q_scale(*w_scale, "fused_add_rms_norm_splitk_fp8");
  }
  auto out8 = torch::empty(residual.sizes(), residual.options().dtype(torch::kUInt8));
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(residual));
  check(lvllm_fused_add_rms_norm_splitk_quant(
      nullptr, residual.data_ptr(), partials.data_ptr<float>(), (int)partials.size(0), weight.data_ptr(), (float)epsilon,
      (int)residual.size(0), (int)residual.size(1), dtype_code(residual, "fused_add_rms_norm_splitk_fp8"),
      x_scale.has_value() ? x_scale->data_ptr<float>() : nullptr, w_scale.has_value() ? w_scale->data_ptr<float>() : nullptr,
      out8.data_ptr(), q_scale.data_ptr<float>(), current_stream(residual)));
  return out8;
}

static void check_w8a8_q(const torch::Tensor& x8, const torch::Tensor& w_packed, const torch::Tensor& w_scale,
                         const torch::Tensor& x_scale, int64_t N, int64_t K, const char* op) {
  LV_CHECK_DEVICE(x8);
  LV_CHECK_DEVICE(w_packed);
  TORCH_CHECK(x8.dim() == 2 && x8.size(1) == K && x8.stride(1) == 1 && x8.element_size() == 1 && x8.size(0) <= 32, op,
              ": x_fp8 must be [M <= 32, K] bytes");
  TORCH_CHECK(w_packed.is_contiguous() && w_packed.numel() * w_packed.element_size() == N * K, op,
              ": w_packed must hold N*K fp8 bytes");
  check_q_scale(w_scale, op);
  check_q_scale(x_scale, op);
}

// y [M, N] in `out_dtype` = the projection of pre-quantised activations (skinny_linear_w8a8 on the unquantised x)
torch::Tensor skinny_linear_w8a8_q(const torch::Tensor& x8, const torch::Tensor& w_packed, const torch::Tensor& w_scale,
                                   const torch::Tensor& x_scale, int64_t N, int64_t K,
                                   const std::optional<torch::Tensor>& bias, at::ScalarType out_dtype) {
  check_w8a8_q(x8, w_packed, w_scale, x_scale, N, K, "skinny_linear_w8a8_q");
  const int64_t M = x8.size(0);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x8));
  auto y = torch::empty({M, N}, x8.options().dtype(out_dtype));
  const int64_t ws_bytes = lvllm_skinny_gemm_w8a8_workspace_bytes((int)M, (int)N, (int)K);
  torch::Tensor ws;
  if (ws_bytes > 0) ws = torch::empty({ws_bytes / 4}, x8.options().dtype(torch::kFloat));
  check(lvllm_skinny_gemm_w8a8_q(y.data_ptr(), nullptr, nullptr, x8.data_ptr(), w_packed.data_ptr(),
                                 bias ? bias->data_ptr() : nullptr, x_scale.data_ptr<float>(), w_scale.data_ptr<float>(),
                                 (int)M, (int)N, (int)K, x8.stride(0), dtype_code(y, "skinny_linear_w8a8_q"), 0,
                                 ws_bytes > 0 ? ws.data_ptr() : nullptr, ws_bytes, current_stream(x8)));
  return y;
}

// the gate_up projection + silu_and_mul + static_scaled_fp8_quant(act, q_scale) in one launch: fp8 [M, N / 2]
torch::Tensor skinny_linear_w8a8_q_swiglu_fp8(const torch::Tensor& x8, const torch::Tensor& w_packed,
                                              const torch::Tensor& w_scale, const torch::Tensor& x_scale, int64_t N,
                                              int64_t K, const torch::Tensor& q_scale, at::ScalarType compute_dtype) {
  check_w8a8_q(x8, w_packed, w_scale, x_scale, N, K, "skinny_linear_w8a8_q_swiglu_fp8");
  check_q_scale(q_scale, "skinny_linear_w8a8_q_swiglu_fp8");
  TORCH_CHECK(N % 32 == 0 && lvllm_skinny_gemm_w8a8_workspace_bytes((int)x8.size(0), (int)N, (int)K) == 0,
              "skinny_linear_w8a8_q_swiglu_fp8: N % 32 == 0 and K within one workgroup");
  const int64_t M = x8.size(0);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x8));
  auto y8 = torch::empty({M, N / 2}, x8.options().dtype(torch::kUInt8));
  const int dt = compute_dtype == at::kBFloat16 ? LVLLM_BF16 : LVLLM_F16;
  TORCH_CHECK(compute_dtype == at::kBFloat16 || compute_dtype == at::kHalf, "compute_dtype: bfloat16 or float16");
  check(lvllm_skinny_gemm_w8a8_q(nullptr, y8.data_ptr(), q_scale.data_ptr<float>(), x8.data_ptr(), w_packed.data_ptr(),
                                 nullptr, x_scale.data_ptr<float>(), w_scale.data_ptr<float>(), (int)M, (int)N, (int)K,
                                 x8.stride(0), dt, 2, nullptr, 0, current_stream(x8)));
  return y8;
}

// raw fp32 split-K partials [S, M, N] of a projection of pre-quantised activations (empty: K is not split here)
torch::Tensor skinny_linear_w8a8_q_partials(const torch::Tensor& x8, const torch::Tensor& w_packed,
                                            const torch::Tensor& w_scale, const torch::Tensor& x_scale, int64_t N,
                                            int64_t K) {
  check_w8a8_q(x8, w_packed, w_scale, x_scale, N, K, "skinny_linear_w8a8_q_partials");
  const int64_t M = x8.size(0);
  const int64_t ws_bytes = lvllm_skinny_gemm_w8a8_workspace_bytes((int)M, (int)N, (int)K);
  if (ws_bytes == 0 || M == 0) return torch::empty({0}, x8.options().dtype(torch::kFloat));
  const int64_t S = ws_bytes / (M * N * 4);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x8));
  auto partials = torch::empty({S, M, N}, x8.options().dtype(torch::kFloat));
  check(lvllm_skinny_gemm_w8a8_q(nullptr, nullptr, nullptr, x8.data_ptr(), w_packed.data_ptr(), nullptr,
                                 x_scale.data_ptr<float>(), w_scale.data_ptr<float>(), (int)M, (int)N, (int)K,
                                 x8.stride(0), LVLLM_BF16, 4, partials.data_ptr(), ws_bytes, current_stream(x8)));
  return partials;
}

// W8A8 lm_head + greedy sampling in one launch (see skinny_linear_packed_argmax)
torch::Tensor skinny_linear_w8a8_argmax(const torch::Tensor& x, const torch::Tensor& w_packed, const torch::Tensor& w_scale,
                                        const torch::Tensor& x_scale, int64_t N, int64_t K) {
  const int64_t M = x.size(0);
  if (M > 32 || lvllm_skinny_gemm_w8a8_workspace_bytes((int)M, (int)N, (int)K) > 0)  // outside the epilogue's envelope
    return at::argmax(skinny_linear_w8a8(x, w_packed, w_scale, x_scale, N, K, std::nullopt), -1);
  LV_CHECK_DEVICE(x);
  LV_CHECK_DEVICE(w_packed);
  TORCH_CHECK(x.dim() == 2 && x.size(1) == K && x.stride(1) == 1 && M >= 1 && M <= 64,
              "skinny_linear_w8a8_argmax: x must be [M <= 64, K]");
  TORCH_CHECK(w_packed.is_contiguous() && w_packed.numel() * w_packed.element_size() == N * K,
              "skinny_linear_w8a8_argmax: w_packed must hold N*K fp8 bytes");
  TORCH_CHECK(w_scale.scalar_type() == at::kFloat && x_scale.scalar_type() == at::kFloat && w_scale.is_cuda() &&
                  x_scale.is_cuda() && w_scale.numel() == 1 && x_scale.numel() == 1,
              "skinny_linear_w8a8_argmax: per-tensor float32 scales on the device");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(x));
  auto tokens = torch::empty({M}, x.options().dtype(torch::kLong));
  const int64_t ws_bytes = lvllm_skinny_gemm_argmax_workspace_bytes((int)M);
  auto ws = torch::empty({ws_bytes / 4}, x.options().dtype(torch::kFloat));
  check(lvllm_skinny_gemm_w8a8_ex(tokens.data_ptr(), x.data_ptr(), w_packed.data_ptr(), nullptr,
                                  x_scale.data_ptr<float>(), w_scale.data_ptr<float>(), (int)M, (int)N, (int)K,
                                  x.stride(0), dtype_code(x, "skinny_linear_w8a8_argmax"), 3, ws.data_ptr(), ws_bytes,
                                  current_stream(x)));
  return tokens;
}

torch::Tensor pack_weight(const torch::Tensor& w) {
  TORCH_CHECK(w.is_cuda() && w.dim() == 2 && w.is_contiguous(), "pack_weight: contiguous [N,K] GPU tensor");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(device_of(w));
  auto out = torch::empty_like(w);
  check(lvllm_pack_weight(out.data_ptr(), w.data_ptr(), (int)w.size(0), (int)w.size(1),
                          dtype_code(w, "pack_weight"), current_stream(w)));
  return out;
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

  // prepare_inputs advance_step (torch_bindings.cpp:75-77 of the reference, schema inferred there)
  ops.def("advance_step(int num_seqs, int num_queries, int block_size, Tensor! input_tokens, "
          "Tensor sampled_token_ids, Tensor! input_positions, Tensor! seq_lens, Tensor! slot_mapping, "
          "Tensor block_tables) -> ()");
  ops.impl("advance_step", torch::kCUDA, &advance_step);

  // fp8 activation quantisation (torch_bindings.cpp:185-202 of the reference)
  ops.def("static_scaled_fp8_quant(Tensor! out, Tensor input, Tensor scale) -> ()");
  ops.impl("static_scaled_fp8_quant", torch::kCUDA, &static_scaled_fp8_quant);
  ops.def("dynamic_scaled_fp8_quant(Tensor! out, Tensor input, Tensor! scale) -> ()");
  ops.impl("dynamic_scaled_fp8_quant", torch::kCUDA, &dynamic_scaled_fp8_quant);
  ops.def("dynamic_per_token_scaled_fp8_quant(Tensor! out, Tensor input, Tensor! scale, Tensor? scale_ub) -> ()");
  ops.impl("dynamic_per_token_scaled_fp8_quant", torch::kCUDA, &dynamic_per_token_scaled_fp8_quant);

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
  // torch_bindings.cpp:261-264 of the reference
  cache_ops.def("convert_fp8(Tensor! dst_cache, Tensor src_cache, float scale, str kv_cache_dtype) -> ()");
  cache_ops.impl("convert_fp8", torch::kCUDA, &convert_fp8);

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

// operators this build adds beyond the reference's list
TORCH_LIBRARY(_C_amd, amd) {
  amd.def("skinny_linear(Tensor x, Tensor w, Tensor? bias) -> Tensor");
  amd.impl("skinny_linear", torch::kCUDA, &skinny_linear);
  amd.def("skinny_linear_packed(Tensor x, Tensor w_packed, Tensor? bias, int N, int K) -> Tensor");
  amd.impl("skinny_linear_packed", torch::kCUDA, &skinny_linear_packed);
  amd.def("pack_weight(Tensor w) -> Tensor");
  amd.impl("pack_weight", torch::kCUDA, &pack_weight);
  amd.def("gelu(Tensor! out, Tensor x) -> ()");
  amd.impl("gelu", torch::kCUDA, &gelu);
  amd.def("add_layer_norm(Tensor! out, Tensor x, Tensor? y, Tensor weight, Tensor bias, float epsilon) -> ()");
  amd.impl("add_layer_norm", torch::kCUDA, &add_layer_norm);
  amd.def("skinny_linear_packed_argmax(Tensor x, Tensor w_packed, int N, int K) -> Tensor");
  amd.impl("skinny_linear_packed_argmax", torch::kCUDA, &skinny_linear_packed_argmax);
  amd.def("stream_linear_packed(Tensor x, Tensor w_packed, Tensor? bias, int N, int K) -> Tensor");
  amd.impl("stream_linear_packed", torch::kCUDA, &stream_linear_packed);
  amd.def("skinny_linear_packed_swiglu(Tensor x, Tensor w_packed, Tensor? bias, int N, int K) -> Tensor");
  amd.impl("skinny_linear_packed_swiglu", torch::kCUDA, &skinny_linear_packed_swiglu);
  amd.def("skinny_linear_packed_partials(Tensor x, Tensor w_packed, int N, int K, bool swiglu) -> Tensor");
  amd.impl("skinny_linear_packed_partials", torch::kCUDA, &skinny_linear_packed_partials);
  amd.def("fused_add_rms_norm_splitk(Tensor! out, Tensor! residual, Tensor partials, Tensor weight, "
          "float epsilon) -> ()");
  amd.impl("fused_add_rms_norm_splitk", torch::kCUDA, &fused_add_rms_norm_splitk);
  amd.def("rotary_embedding_and_cache(Tensor positions, Tensor! query, Tensor! key, Tensor value, "
          "int head_size, Tensor cos_sin_cache, bool is_neox, Tensor! key_cache, Tensor! value_cache, "
          "Tensor slot_mapping, str kv_cache_dtype=\"auto\", float k_scale=1.0, float v_scale=1.0) -> bool");
  amd.impl("rotary_embedding_and_cache", torch::kCUDA, &rotary_embedding_and_cache);
  amd.def("rotary_embedding_and_cache_splitk(Tensor positions, Tensor! qkv, Tensor partials, Tensor? bias, "
          "int num_heads, int num_kv_heads, int head_size, Tensor cos_sin_cache, bool is_neox, Tensor! key_cache, "
          "Tensor! value_cache, Tensor slot_mapping, str kv_cache_dtype=\"auto\", float k_scale=1.0, "
          "float v_scale=1.0) -> bool");
  amd.impl("rotary_embedding_and_cache_splitk", torch::kCUDA, &rotary_embedding_and_cache_splitk);
  amd.def("paged_prefill_attention(Tensor! out, Tensor query, Tensor key_cache, Tensor value_cache, "
          "int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens, Tensor query_start_loc, "
          "int max_query_len, int block_size, Tensor? alibi_slopes, int sliding_window, float softcap, "
          "str kv_cache_dtype, bool causal=True, float k_scale=1.0, float v_scale=1.0, int max_seq_len=0) -> ()");
  amd.impl("paged_prefill_attention", torch::kCUDA, &paged_prefill_attention);
  amd.def("skinny_linear_w8a8(Tensor x, Tensor w_packed, Tensor w_scale, Tensor x_scale, int N, int K, "
          "Tensor? bias) -> Tensor");
  amd.impl("skinny_linear_w8a8", torch::kCUDA, &skinny_linear_w8a8);
  amd.def("skinny_linear_w8a8_swiglu(Tensor x, Tensor w_packed, Tensor w_scale, Tensor x_scale, int N, int K, "
          "Tensor? bias) -> Tensor");
  amd.impl("skinny_linear_w8a8_swiglu", torch::kCUDA, &skinny_linear_w8a8_swiglu);
  amd.def("skinny_linear_w8a8_argmax(Tensor x, Tensor w_packed, Tensor w_scale, Tensor x_scale, int N, int K) -> Tensor");
  amd.impl("skinny_linear_w8a8_argmax", torch::kCUDA, &skinny_linear_w8a8_argmax);
  amd.def("rope_cache_paged_attention(Tensor! out, Tensor! exp_sums, Tensor! max_logits, Tensor! tmp_out, "
          "Tensor positions, Tensor query, Tensor key, Tensor value, int head_size, Tensor cos_sin_cache, "
          "bool is_neox, Tensor! key_cache, Tensor! value_cache, Tensor slot_mapping, int num_kv_heads, "
          "float scale, Tensor block_tables, Tensor seq_lens, int block_size, int max_seq_len, "
          "str kv_cache_dtype, float k_scale=1.0, float v_scale=1.0, Tensor(a!)? out_fp8=None, "
          "Tensor? out_fp8_scale=None) -> bool");
  amd.impl("rope_cache_paged_attention", torch::kCUDA, &rope_cache_paged_attention);
  amd.def("paged_attention_v2_q(Tensor! out, Tensor! out_fp8, Tensor out_fp8_scale, Tensor! exp_sums, Tensor! max_logits, "
          "Tensor! tmp_out, Tensor query, Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale, "
          "Tensor block_tables, Tensor seq_lens, int block_size, int max_seq_len, str kv_cache_dtype, "
          "float k_scale=1.0, float v_scale=1.0) -> bool");
  amd.impl("paged_attention_v2_q", torch::kCUDA, &paged_attention_v2_q);
  amd.def("fused_add_rms_norm_splitk_scaled(Tensor! out, Tensor! residual, Tensor partials, Tensor weight, "
          "float epsilon, Tensor? x_scale, Tensor? w_scale) -> ()");
  amd.impl("fused_add_rms_norm_splitk_scaled", torch::kCUDA, &fused_add_rms_norm_splitk_scaled);
  amd.def("skinny_linear_w8a8_partials(Tensor x, Tensor w_packed, Tensor w_scale, Tensor x_scale, int N, int K) -> Tensor");
  amd.impl("skinny_linear_w8a8_partials", torch::kCUDA, &skinny_linear_w8a8_partials);
  amd.def("advance_step_logged(int block_size, Tensor! input_tokens, Tensor sampled_token_ids, "
          "Tensor! input_positions, Tensor! seq_lens, Tensor! slot_mapping, Tensor block_tables, "
          "Tensor! token_log) -> ()");
  amd.impl("advance_step_logged", torch::kCUDA, &advance_step_logged);
  amd.def("rms_norm_fp8(Tensor input, Tensor weight, float epsilon, Tensor q_scale) -> Tensor");
  amd.impl("rms_norm_fp8", torch::kCUDA, &rms_norm_fp8);
  amd.def("fused_add_rms_norm_fp8(Tensor input, Tensor! residual, Tensor weight, float epsilon, Tensor q_scale) -> Tensor");
  amd.impl("fused_add_rms_norm_fp8", torch::kCUDA, &fused_add_rms_norm_fp8);
  amd.def("fused_add_rms_norm_splitk_fp8(Tensor! residual, Tensor partials, Tensor weight, float epsilon, "
          "Tensor? x_scale, Tensor? w_scale, Tensor q_scale) -> Tensor");
  amd.impl("fused_add_rms_norm_splitk_fp8", torch::kCUDA, &fused_add_rms_norm_splitk_fp8);
  amd.def("skinny_linear_w8a8_q(Tensor x_fp8, Tensor w_packed, Tensor w_scale, Tensor x_scale, int N, int K, "
          "Tensor? bias, ScalarType out_dtype) -> Tensor");
  amd.impl("skinny_linear_w8a8_q", torch::kCUDA, &skinny_linear_w8a8_q);
  amd.def("skinny_linear_w8a8_q_swiglu_fp8(Tensor x_fp8, Tensor w_packed, Tensor w_scale, Tensor x_scale, int N, int K, "
          "Tensor q_scale, ScalarType compute_dtype) -> Tensor");
  amd.impl("skinny_linear_w8a8_q_swiglu_fp8", torch::kCUDA, &skinny_linear_w8a8_q_swiglu_fp8);
  amd.def("skinny_linear_w8a8_q_partials(Tensor x_fp8, Tensor w_packed, Tensor w_scale, Tensor x_scale, int N, int K) "
          "-> Tensor");
  amd.impl("skinny_linear_w8a8_q_partials", torch::kCUDA, &skinny_linear_w8a8_q_partials);
  amd.def("sample_rows(Tensor! tokens_out, Tensor logits, Tensor? state_slot, Tensor(a!)? params, Tensor(b!)? counts, "
          "Tensor(c!)? scratch, Tensor(d!)? processed_out, bool update_state) -> ()");
  amd.impl("sample_rows", torch::kCUDA, &sample_rows);
  amd.def("sampler_init_row(Tensor! counts_row, Tensor prompt_tokens, Tensor output_tokens) -> ()");
  amd.impl("sampler_init_row", torch::kCUDA, &sampler_init_row);
  amd.def("set_tuning(str key, int value) -> ()", [](const std::string& key, int64_t value) {
    check(lvllm_set_tuning(key.c_str(), (int)value));
  });
  amd.def("get_tuning(str key) -> int", [](const std::string& key) {
    int v = 0;
    check(lvllm_get_tuning(key.c_str(), &v));
    return (int64_t)v;
  });
  amd.def("paged_prefill_workspace_bytes(int num_seqs, int num_tokens, int max_query_len, int num_heads, "
          "int num_kv_heads, int head_size, int max_seq_len) -> int", &paged_prefill_workspace_bytes);
  amd.def("varlen_attention_workspace_bytes(int num_tokens, int num_seqs, int max_seq_len, int num_kv_heads, "
          "int head_size) -> int", &varlen_attention_workspace_bytes);
  amd.def("varlen_attention(Tensor! out, Tensor query, Tensor key, Tensor value, Tensor cu_seqlens, "
          "int max_seq_len, float scale, bool causal, Tensor? alibi_slopes, int sliding_window, "
          "float softcap, Tensor! workspace) -> ()");
  amd.impl("varlen_attention", torch::kCUDA, &varlen_attention);
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
