// RMSNorm kernels for gfx950: rms_norm and fused_add_rms_norm.
//
// Rounding points follow the reference kernels so results agree with the
// reference's GPU forward (csrc/layernorm_kernels.cu:21-45 and :254-287):
//   variance      : fp32 sum of squares of the (already rounded) T values
//   normalisation : t = T(x * rsqrt(var/hidden + eps))   -- rounded to T first
//   scaling       : out = T(float(t) * float(w))          -- a T x T multiply
//   fused add     : z = T(float(input) + float(residual)); residual = z
// One workgroup per token row; 16-byte loads; the row is kept in registers
// between the variance pass and the scaling pass (one HBM read per element).
#include "common.h"
#include "trace.h"

namespace lvllm {

constexpr int kMaxCached = 4;  // 16-byte chunks a thread keeps in registers

// out8 / q_scale (both or none): the normalised row ALSO leaves as fp8 -- e4m3(clamp(out * (1 / *q_scale))) on the
// value rounded to T, i.e. exactly what static_scaled_fp8_quant(out) would produce -- for a W8A8 projection that
// takes its activations already quantised (lvllm_skinny_gemm_w8a8_q); `out` itself may then be null (FUSED_ADD:
// `in` is only read).
template <typename T, bool FUSED_ADD>
__global__ void rms_norm_vec_kernel(typename T::store_t* out,  // == in when FUSED_ADD (no restrict)
                                    typename T::store_t* __restrict__ res,  // residual (FUSED_ADD)
                                    const typename T::store_t* in,
                                    const typename T::store_t* __restrict__ weight,
                                    const float epsilon, const int hidden_size,
                                    uint8_t* __restrict__ out8 = nullptr, const float* __restrict__ q_scale = nullptr) {
  LVLLM_TRACE_BEGIN();
  using V = Vec16<T>;
  constexpr int N = V::N;
  __shared__ float red[16];
  __shared__ float s_scale;
  const int nvec = hidden_size / N;
  const int64_t row = (int64_t)blockIdx.x * nvec;
  const V* in_v = reinterpret_cast<const V*>(in) + row;
  V* res_v = reinterpret_cast<V*>(res) + row;
  V* out_v = reinterpret_cast<V*>(out) + row;
  const V* w_v = reinterpret_cast<const V*>(weight);

  V cache[kMaxCached];
  float var = 0.f;
  int c = 0;
  for (int i = threadIdx.x; i < nvec; i += blockDim.x, ++c) {
    V x = in_v[i];
    if (FUSED_ADD) {
      V r = res_v[i];
#pragma unroll
      for (int j = 0; j < N; ++j)
        x.v[j] = T::from_float(T::to_float(x.v[j]) + T::to_float(r.v[j]));
      res_v[i] = x;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float f = T::to_float(x.v[j]);
      var += f * f;
    }
    // static indices only: a runtime-indexed register array would go to scratch
#pragma unroll
    for (int k = 0; k < kMaxCached; ++k)
      if (c == k) cache[k] = x;
  }
  var = block_sum(var, red);
  if (threadIdx.x == 0) s_scale = rsqrtf(var / hidden_size + epsilon);
  __syncthreads();
  const float s = s_scale;

  c = 0;
  for (int i = threadIdx.x; i < nvec; i += blockDim.x, ++c) {
    V x;
    if (c < kMaxCached) {
#pragma unroll
      for (int k = 0; k < kMaxCached; ++k)
        if (c == k) x = cache[k];
    } else {
      x = FUSED_ADD ? res_v[i] : in_v[i];
    }
    const V w = w_v[i];
    V o;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const typename T::store_t t = T::from_float(T::to_float(x.v[j]) * s);
      o.v[j] = T::from_float(T::to_float(t) * T::to_float(w.v[j]));
    }
    if (out != nullptr) out_v[i] = o;
    if (out8 != nullptr) {
      static_assert(N == 8 || N == 4, "16-byte vectors of 2- or 4-byte elements");
      const float inv = 1.0f / q_scale[0];
      uint8_t* dst = out8 + (int64_t)blockIdx.x * hidden_size + (int64_t)i * N;
      const uint32_t lo = fp8_act_quant4(T::to_float(o.v[0]), T::to_float(o.v[1]), T::to_float(o.v[2]), T::to_float(o.v[3]), inv);
      if constexpr (N == 8) {
        const uint32_t hi = fp8_act_quant4(T::to_float(o.v[4]), T::to_float(o.v[5]), T::to_float(o.v[6]), T::to_float(o.v[7]), inv);
        *reinterpret_cast<uint2*>(dst) = uint2{lo, hi};
      } else {
        *reinterpret_cast<uint32_t*>(dst) = lo;
      }
    }
  }
  LVLLM_TRACE_END(3);
}

// fused_add_rms_norm whose `input` arrives as fp32 split-K partials [S, M, hidden] of the
// preceding down projection (csrc/skinny_gemm.hip): x = T(sum_s partial_s) -- the value the GEMM's
// own reduce pass would have written -- then exactly fused_add_rms_norm.  One launch less per layer.
template <typename T>
__global__ void fused_add_rms_norm_splitk_kernel(typename T::store_t* out,  // [M, hidden] normed output
                                                 typename T::store_t* __restrict__ res,
                                                 const float* __restrict__ partials, const int num_partials,
                                                 const int64_t partial_stride,  // M * hidden
                                                 const typename T::store_t* __restrict__ weight,
                                                 const float epsilon, const int hidden_size,
                                                 const float* __restrict__ scale_a = nullptr,
                                                 const float* __restrict__ scale_b = nullptr,
                                                 uint8_t* __restrict__ out8 = nullptr,  // see rms_norm_vec_kernel
                                                 const float* __restrict__ q_scale = nullptr) {
  LVLLM_TRACE_BEGIN();
  using V = Vec16<T>;
  constexpr int N = V::N;  // 8
  __shared__ float red[16];
  __shared__ float s_scale;
  const int nvec = hidden_size / N;
  const int64_t row = (int64_t)blockIdx.x * nvec;
  V* res_v = reinterpret_cast<V*>(res) + row;
  V* out_v = reinterpret_cast<V*>(out) + row;
  const V* w_v = reinterpret_cast<const V*>(weight);
  const float* prow = partials + (int64_t)blockIdx.x * hidden_size;
  // W8A8 partials are raw fp8 x fp8 sums: the GEMM's reduce pass would have multiplied them by
  // x_scale * w_scale before rounding (skinny_gemm_reduce_kernel), so this kernel does
  const bool scaled = scale_a != nullptr;
  const float sc = scaled ? scale_a[0] * scale_b[0] : 1.f;

  V cache[kMaxCached];
  float var = 0.f;
  int c = 0;
  for (int i = threadIdx.x; i < nvec; i += blockDim.x, ++c) {
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    for (int sp = 0; sp < num_partials; ++sp) {
      const float4* p4 = reinterpret_cast<const float4*>(prow + sp * partial_stride + (int64_t)i * N);
      const float4 a = p4[0], b = p4[1];
      acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
      acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    }
    const V r = res_v[i];
    V x;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (scaled) acc[j] *= sc;
      const typename T::store_t y = T::from_float(acc[j]);  // the GEMM output, rounded to T
      x.v[j] = T::from_float(T::to_float(y) + T::to_float(r.v[j]));
      const float f = T::to_float(x.v[j]);
      var += f * f;
    }
    res_v[i] = x;
#pragma unroll
    for (int k = 0; k < kMaxCached; ++k)
      if (c == k) cache[k] = x;
  }
  var = block_sum(var, red);
  if (threadIdx.x == 0) s_scale = rsqrtf(var / hidden_size + epsilon);
  __syncthreads();
  const float s = s_scale;
  c = 0;
  for (int i = threadIdx.x; i < nvec; i += blockDim.x, ++c) {
    V x;
    if (c < kMaxCached) {
#pragma unroll
      for (int k = 0; k < kMaxCached; ++k)
        if (c == k) x = cache[k];
    } else {
      x = res_v[i];
    }
    const V w = w_v[i];
    V o;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const typename T::store_t t = T::from_float(T::to_float(x.v[j]) * s);
      o.v[j] = T::from_float(T::to_float(t) * T::to_float(w.v[j]));
    }
    if (out != nullptr) out_v[i] = o;
    if (out8 != nullptr) {
      const float inv = 1.0f / q_scale[0];
      const uint32_t lo = fp8_act_quant4(T::to_float(o.v[0]), T::to_float(o.v[1]), T::to_float(o.v[2]), T::to_float(o.v[3]), inv);
      const uint32_t hi = fp8_act_quant4(T::to_float(o.v[4]), T::to_float(o.v[5]), T::to_float(o.v[6]), T::to_float(o.v[7]), inv);
      *reinterpret_cast<uint2*>(out8 + (int64_t)blockIdx.x * hidden_size + (int64_t)i * N) = uint2{lo, hi};
    }
  }
  LVLLM_TRACE_END(4);
}

// element-wise path for rows that are not 16-byte friendly
template <typename T, bool FUSED_ADD>
__global__ void rms_norm_scalar_kernel(typename T::store_t* out,
                                       typename T::store_t* __restrict__ res,
                                       const typename T::store_t* in,
                                       const typename T::store_t* __restrict__ weight,
                                       const float epsilon, const int hidden_size) {
  __shared__ float red[16];
  __shared__ float s_scale;
  const int64_t row = (int64_t)blockIdx.x * hidden_size;
  float var = 0.f;
  for (int i = threadIdx.x; i < hidden_size; i += blockDim.x) {
    typename T::store_t x = in[row + i];
    if (FUSED_ADD) {
      x = T::from_float(T::to_float(x) + T::to_float(res[row + i]));
      res[row + i] = x;
    }
    const float f = T::to_float(x);
    var += f * f;
  }
  var = block_sum(var, red);
  if (threadIdx.x == 0) s_scale = rsqrtf(var / hidden_size + epsilon);
  __syncthreads();
  const float s = s_scale;
  for (int i = threadIdx.x; i < hidden_size; i += blockDim.x) {
    const typename T::store_t x = FUSED_ADD ? res[row + i] : in[row + i];
    const typename T::store_t t = T::from_float(T::to_float(x) * s);
    out[row + i] = T::from_float(T::to_float(t) * T::to_float(weight[i]));
  }
}

// LayerNorm of x (+ y) for the encoder models of the prefill-only workflow (bge-m3 / XLM-RoBERTa:
// `LayerNorm(hidden_states + input_tensor)` after the attention output and after the MLP,
// light_vllm/encode_only/modelzoo/xlm_roberta.py; the reference leaves both to torch).  One launch
// instead of an add and a layer_norm: z = T(float(x) + float(y)) -- the rounded sum torch's `x + y`
// materialises -- then out = T((z - mean) * rsqrt(var + eps) * w + b) with mean and the
// centred variance in fp32.  16-bit element types, hidden_size % 8 == 0, rows of at most
// 8 * kMaxCached * blockDim elements stay in registers between the passes.
template <typename T>
__global__ void add_layer_norm_kernel(typename T::store_t* out, const typename T::store_t* x,  // may alias
                                      const typename T::store_t* __restrict__ y,  // nullable
                                      const typename T::store_t* __restrict__ weight,
                                      const typename T::store_t* __restrict__ bias, const float epsilon,
                                      const int hidden_size) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  __shared__ float red[16];
  const int nvec = hidden_size / N;
  const int64_t row = (int64_t)blockIdx.x * nvec;
  const V* x_v = reinterpret_cast<const V*>(x) + row;
  const V* y_v = y ? reinterpret_cast<const V*>(y) + row : nullptr;
  V* out_v = reinterpret_cast<V*>(out) + row;
  const V* w_v = reinterpret_cast<const V*>(weight);
  const V* b_v = reinterpret_cast<const V*>(bias);
  V cache[kMaxCached];
  float sum = 0.f;
  int c = 0;
  for (int i = threadIdx.x; i < nvec; i += blockDim.x, ++c) {
    V z = x_v[i];
    if (y_v != nullptr) {
      const V r = y_v[i];
#pragma unroll
      for (int j = 0; j < N; ++j) z.v[j] = T::from_float(T::to_float(z.v[j]) + T::to_float(r.v[j]));
    }
#pragma unroll
    for (int j = 0; j < N; ++j) sum += T::to_float(z.v[j]);
#pragma unroll
    for (int k = 0; k < kMaxCached; ++k)
      if (c == k) cache[k] = z;
  }
  const float mean = block_sum(sum, red) / hidden_size;
  float var = 0.f;
  c = 0;
  for (int i = threadIdx.x; i < nvec; i += blockDim.x, ++c) {
    V z;
#pragma unroll
    for (int k = 0; k < kMaxCached; ++k)
      if (c == k) z = cache[k];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float d = T::to_float(z.v[j]) - mean;
      var += d * d;
    }
  }
  const float rstd = rsqrtf(block_sum(var, red) / hidden_size + epsilon);
  c = 0;
  for (int i = threadIdx.x; i < nvec; i += blockDim.x, ++c) {
    V z;
#pragma unroll
    for (int k = 0; k < kMaxCached; ++k)
      if (c == k) z = cache[k];
    const V w = w_v[i], b = b_v[i];
    V o;
#pragma unroll
    for (int j = 0; j < N; ++j)
      o.v[j] = T::from_float((T::to_float(z.v[j]) - mean) * rstd * T::to_float(w.v[j]) + T::to_float(b.v[j]));
    out_v[i] = o;
  }
}

// The same for rows of at most 8 * kMaxCached * 64 = 2048 elements: one WAVE per row, four rows per
// workgroup, wave-level reductions only -- no barrier between the three passes.
template <typename T>
__global__ void add_layer_norm_wave_kernel(typename T::store_t* out, const typename T::store_t* x,  // may alias
                                           const typename T::store_t* __restrict__ y,  // nullable
                                           const typename T::store_t* __restrict__ weight,
                                           const typename T::store_t* __restrict__ bias, const float epsilon,
                                           const int hidden_size, const int num_tokens) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  const int lane = threadIdx.x & 63;
  const int64_t token = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (token >= num_tokens) return;
  const int nvec = hidden_size / N;
  const V* x_v = reinterpret_cast<const V*>(x) + token * nvec;
  const V* y_v = y ? reinterpret_cast<const V*>(y) + token * nvec : nullptr;
  V* out_v = reinterpret_cast<V*>(out) + token * nvec;
  const V* w_v = reinterpret_cast<const V*>(weight);
  const V* b_v = reinterpret_cast<const V*>(bias);
  V cache[kMaxCached];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxCached; ++k) {
    const int i = lane + 64 * k;
    if (i < nvec) {
      V z = x_v[i];
      if (y_v != nullptr) {
        const V r = y_v[i];
#pragma unroll
        for (int j = 0; j < N; ++j) z.v[j] = T::from_float(T::to_float(z.v[j]) + T::to_float(r.v[j]));
      }
#pragma unroll
      for (int j = 0; j < N; ++j) sum += T::to_float(z.v[j]);
      cache[k] = z;
    }
  }
  const float mean = wave_sum(sum) / hidden_size;
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < kMaxCached; ++k)
    if (lane + 64 * k < nvec) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float d = T::to_float(cache[k].v[j]) - mean;
        var += d * d;
      }
    }
  const float rstd = rsqrtf(wave_sum(var) / hidden_size + epsilon);
#pragma unroll
  for (int k = 0; k < kMaxCached; ++k) {
    const int i = lane + 64 * k;
    if (i < nvec) {
      const V w = w_v[i], b = b_v[i];
      V o;
#pragma unroll
      for (int j = 0; j < N; ++j)
        o.v[j] = T::from_float((T::to_float(cache[k].v[j]) - mean) * rstd * T::to_float(w.v[j]) + T::to_float(b.v[j]));
      out_v[i] = o;
    }
  }
}

template <typename T, bool FUSED_ADD>
static int launch_rms(void* out, void* res, const void* in, const void* weight, float eps,
                      int num_tokens, int hidden_size, hipStream_t stream, uint8_t* out8 = nullptr,
                      const float* q_scale = nullptr) {
  using S = typename T::store_t;
  constexpr int N = Vec16<T>::N;
  const bool vec_ok = hidden_size % N == 0 && (((uintptr_t)out | (uintptr_t)res | (uintptr_t)in |
                                                (uintptr_t)weight) & 15) == 0;
  if (vec_ok) {
    const int nvec = hidden_size / N;
    int threads = ((nvec + 63) / 64) * 64;
    threads = threads > 1024 ? 1024 : threads;
    hipLaunchKernelGGL((rms_norm_vec_kernel<T, FUSED_ADD>), dim3(num_tokens), dim3(threads), 0,
                       stream, (S*)out, (S*)res, (const S*)in, (const S*)weight, eps, hidden_size, out8, q_scale);
  } else {
    if (out8 != nullptr) {
      set_error("rms_norm with an fp8 twin: 16-byte aligned rows of a multiple of 8 elements only");
      return 1;
    }
    int threads = ((hidden_size + 63) / 64) * 64;
    threads = threads > 1024 ? 1024 : threads;
    hipLaunchKernelGGL((rms_norm_scalar_kernel<T, FUSED_ADD>), dim3(num_tokens), dim3(threads), 0,
                       stream, (S*)out, (S*)res, (const S*)in, (const S*)weight, eps, hidden_size);
  }
  return 0;
}

}  // namespace lvllm

using namespace lvllm;

extern "C" int lvllm_rms_norm(void* out, const void* input, const void* weight, float epsilon,
                              int num_tokens, int hidden_size, int dtype, void* stream) {
  if (num_tokens == 0) return 0;
  LV_CHECK(hidden_size > 0, "hidden_size must be positive");
  LV_DISPATCH_DTYPE(dtype, (launch_rms<scalar_t, false>(out, nullptr, input, weight, epsilon,
                                                        num_tokens, hidden_size,
                                                        (hipStream_t)stream)));
  LV_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvllm_fused_add_rms_norm(void* input, void* residual, const void* weight,
                                        float epsilon, int num_tokens, int hidden_size,
                                        int dtype, void* stream) {
  if (num_tokens == 0) return 0;
  LV_CHECK(hidden_size > 0, "hidden_size must be positive");
  LV_DISPATCH_DTYPE(dtype, (launch_rms<scalar_t, true>(input, residual, input, weight, epsilon,
                                                       num_tokens, hidden_size,
                                                       (hipStream_t)stream)));
  LV_LAUNCH_CHECK();
  return 0;
}

// rms_norm / fused_add_rms_norm whose result ALSO (or only: out / write_normed) leaves as fp8 for the next W8A8
// projection: out_fp8 [num_tokens, hidden] = static_scaled_fp8_quant(normalised row, *q_scale), bit for bit.
extern "C" int lvllm_rms_norm_quant(void* out, void* out_fp8, const float* q_scale, const void* input, const void* weight,
                                    float epsilon, int num_tokens, int hidden_size, int dtype, void* stream) {
  if (num_tokens == 0) return 0;
  LV_CHECK(out_fp8 != nullptr && q_scale != nullptr, "out_fp8 and q_scale are required");
  LV_CHECK((dtype == LVLLM_BF16 || dtype == LVLLM_F16) && hidden_size > 0 && hidden_size % 8 == 0 &&
               (((uintptr_t)out_fp8) & 7) == 0, "16-bit rows of a multiple of 8 elements");
  int rc = 0;
  LV_DISPATCH_DTYPE(dtype, (rc = launch_rms<scalar_t, false>(out, nullptr, input, weight, epsilon, num_tokens, hidden_size,
                                                             (hipStream_t)stream, (uint8_t*)out_fp8, q_scale)));
  if (rc) return rc;
  LV_LAUNCH_CHECK();
  return 0;
}

// residual <- T(input + residual); normalised row -> out_fp8 (and -> input when write_normed != 0)
extern "C" int lvllm_fused_add_rms_norm_quant(void* input, void* residual, const void* weight, float epsilon,
                                              int num_tokens, int hidden_size, int dtype, void* out_fp8,
                                              const float* q_scale, int write_normed, void* stream) {
  if (num_tokens == 0) return 0;
  LV_CHECK(out_fp8 != nullptr && q_scale != nullptr, "out_fp8 and q_scale are required");
  LV_CHECK((dtype == LVLLM_BF16 || dtype == LVLLM_F16) && hidden_size > 0 && hidden_size % 8 == 0 &&
               (((uintptr_t)out_fp8) & 7) == 0, "16-bit rows of a multiple of 8 elements");
  int rc = 0;
  LV_DISPATCH_DTYPE(dtype, (rc = launch_rms<scalar_t, true>(write_normed ? input : nullptr, residual, input, weight, epsilon,
                                                            num_tokens, hidden_size, (hipStream_t)stream,
                                                            (uint8_t*)out_fp8, q_scale)));
  if (rc) return rc;
  LV_LAUNCH_CHECK();
  return 0;
}

// out = norm(T(sum_s partials[s]) + residual) * weight, residual updated in place; partials fp32
// [num_partials, num_tokens, hidden_size].  16-bit element types, hidden_size % 8 == 0.
extern "C" int lvllm_fused_add_rms_norm_splitk(void* out, void* residual, const float* partials,
                                               int num_partials, const void* weight, float epsilon,
                                               int num_tokens, int hidden_size, int dtype, void* stream) {
  return lvllm_fused_add_rms_norm_splitk_scaled(out, residual, partials, num_partials, weight, epsilon, num_tokens,
                                                hidden_size, dtype, nullptr, nullptr, stream);
}

// ... of a W8A8 projection: x = T(sum_s partials[s] * (*x_scale * *w_scale)) (scales: device pointers, both or none)
extern "C" int lvllm_fused_add_rms_norm_splitk_scaled(void* out, void* residual, const float* partials,
                                                      int num_partials, const void* weight, float epsilon,
                                                      int num_tokens, int hidden_size, int dtype,
                                                      const float* x_scale, const float* w_scale, void* stream) {
  return lvllm_fused_add_rms_norm_splitk_quant(out, residual, partials, num_partials, weight, epsilon, num_tokens,
                                               hidden_size, dtype, x_scale, w_scale, nullptr, nullptr, stream);
}

// ... whose normalised rows also (out != null) or only (out == null) leave as fp8 (see lvllm_rms_norm_quant)
extern "C" int lvllm_fused_add_rms_norm_splitk_quant(void* out, void* residual, const float* partials,
                                                     int num_partials, const void* weight, float epsilon,
                                                     int num_tokens, int hidden_size, int dtype,
                                                     const float* x_scale, const float* w_scale, void* out_fp8,
                                                     const float* q_scale, void* stream) {
  if (num_tokens == 0) return 0;
  LV_CHECK((out_fp8 == nullptr) == (q_scale == nullptr), "out_fp8 and q_scale: both or none");
  LV_CHECK(out != nullptr || out_fp8 != nullptr, "nowhere to write the normalised rows");
  LV_CHECK((((uintptr_t)out_fp8) & 7) == 0, "out_fp8 must be 8-byte aligned");
  LV_CHECK((x_scale == nullptr) == (w_scale == nullptr), "x_scale and w_scale: both or none");
  LV_CHECK(dtype == LVLLM_BF16 || dtype == LVLLM_F16, "16-bit element types only");
  LV_CHECK(hidden_size % 8 == 0 && num_partials >= 1, "hidden_size must be a multiple of 8");
  LV_CHECK((((uintptr_t)out | (uintptr_t)residual | (uintptr_t)partials | (uintptr_t)weight) & 15) == 0,
           "pointers must be 16-byte aligned");
  const int nvec = hidden_size / 8;
  int threads = ((nvec + 63) / 64) * 64;
  threads = threads > 1024 ? 1024 : threads;
  const int64_t stride = (int64_t)num_tokens * hidden_size;
  if (dtype == LVLLM_BF16)
    hipLaunchKernelGGL((fused_add_rms_norm_splitk_kernel<BF16>), dim3(num_tokens), dim3(threads), 0,
                       (hipStream_t)stream, (uint16_t*)out, (uint16_t*)residual, partials, num_partials, stride,
                       (const uint16_t*)weight, epsilon, hidden_size, x_scale, w_scale, (uint8_t*)out_fp8, q_scale);
  else
    hipLaunchKernelGGL((fused_add_rms_norm_splitk_kernel<F16>), dim3(num_tokens), dim3(threads), 0,
                       (hipStream_t)stream, (uint16_t*)out, (uint16_t*)residual, partials, num_partials, stride,
                       (const uint16_t*)weight, epsilon, hidden_size, x_scale, w_scale, (uint8_t*)out_fp8, q_scale);
  LV_LAUNCH_CHECK();
  return 0;
}

LVLLM_TRACE_READER(lvllm_trace_read_norm)

// out = LayerNorm(x + y) (y may be null: plain LayerNorm); see add_layer_norm_kernel.  out may alias x.
extern "C" int lvllm_add_layer_norm(void* out, const void* x, const void* y, const void* weight, const void* bias,
                                    float epsilon, int num_tokens, int hidden_size, int dtype, void* stream) {
  if (num_tokens == 0) return 0;
  LV_CHECK(dtype == LVLLM_BF16 || dtype == LVLLM_F16, "16-bit element types only");
  LV_CHECK(hidden_size > 0 && hidden_size % 8 == 0, "hidden_size must be a positive multiple of 8");
  LV_CHECK((((uintptr_t)out | (uintptr_t)x | (uintptr_t)y | (uintptr_t)weight | (uintptr_t)bias) & 15) == 0,
           "pointers must be 16-byte aligned");
  const int nvec = hidden_size / 8;
  if (nvec <= 64 * lvllm::kMaxCached) {  // a wave per row
    const int rows = 4;
    const dim3 grid((num_tokens + rows - 1) / rows), block(64 * rows);
    if (dtype == LVLLM_BF16)
      hipLaunchKernelGGL((lvllm::add_layer_norm_wave_kernel<lvllm::BF16>), grid, block, 0, (hipStream_t)stream,
                         (uint16_t*)out, (const uint16_t*)x, (const uint16_t*)y, (const uint16_t*)weight,
                         (const uint16_t*)bias, epsilon, hidden_size, num_tokens);
    else
      hipLaunchKernelGGL((lvllm::add_layer_norm_wave_kernel<lvllm::F16>), grid, block, 0, (hipStream_t)stream,
                         (uint16_t*)out, (const uint16_t*)x, (const uint16_t*)y, (const uint16_t*)weight,
                         (const uint16_t*)bias, epsilon, hidden_size, num_tokens);
    LV_LAUNCH_CHECK();
    return 0;
  }
  int threads = ((nvec + 63) / 64) * 64;
  threads = threads > 1024 ? 1024 : threads;
  LV_CHECK(nvec <= lvllm::kMaxCached * threads, "hidden_size beyond 32768 is not supported");
  if (dtype == LVLLM_BF16)
    hipLaunchKernelGGL((lvllm::add_layer_norm_kernel<lvllm::BF16>), dim3(num_tokens), dim3(threads), 0,
                       (hipStream_t)stream, (uint16_t*)out, (const uint16_t*)x, (const uint16_t*)y,
                       (const uint16_t*)weight, (const uint16_t*)bias, epsilon, hidden_size);
  else
    hipLaunchKernelGGL((lvllm::add_layer_norm_kernel<lvllm::F16>), dim3(num_tokens), dim3(threads), 0,
                       (hipStream_t)stream, (uint16_t*)out, (const uint16_t*)x, (const uint16_t*)y,
                       (const uint16_t*)weight, (const uint16_t*)bias, epsilon, hidden_size);
  LV_LAUNCH_CHECK();
  return 0;
}
