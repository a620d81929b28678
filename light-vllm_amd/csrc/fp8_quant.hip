// fp8 (OCP e4m3fn) activation quantisation: static_scaled_fp8_quant, dynamic_scaled_fp8_quant,
// dynamic_per_token_scaled_fp8_quant -- csrc/quantization/fp8/common.cu:24-38,46-83,164-224 of the
// reference.  Arithmetic kept (bit-exact against the oracle):
//   static / dynamic per tensor:  out = e4m3(clamp(x * (1.0f / scale), +-448))        (:171-176, is_scale_inverted)
//   dynamic per tensor:           scale = max over the tensor of |x|, / 448            (:46-83)
//   per token:                    scale = max(min(absmax, ub) / 448, 1 / (448 * 512)); out = e4m3(clamp(x / scale))
// clamp is fmax(-448, fmin(x, 448)): a NaN comes out as +448, as in the reference.
// Round to nearest even by the hardware (v_cvt_pk_fp8_f32); 16-byte loads, 8-byte stores.
#include <algorithm>

#include "../../include/lvllm_hip.h"
#include "common.h"

namespace lvllm {

constexpr float kFp8Max = 448.f;

template <bool INVERTED>
__device__ __forceinline__ float fp8_prep(float v, float scale) {
  const float x = INVERTED ? v * scale : v / scale;
  return fmaxf(-kFp8Max, fminf(x, kFp8Max));
}

template <bool INVERTED>
__device__ __forceinline__ uint32_t fp8_pack4(float a, float b, float c, float d, float scale) {
  uint32_t w = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_prep<INVERTED>(a, scale), fp8_prep<INVERTED>(b, scale), 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(fp8_prep<INVERTED>(c, scale), fp8_prep<INVERTED>(d, scale), w, true);
}

// quantise n elements (n % 8 == 0 handled vectorised, tail scalar)
template <typename T, bool INVERTED>
__device__ __forceinline__ void quant_span(uint8_t* __restrict__ out, const typename T::store_t* __restrict__ in,
                                           int64_t n, float scale, int64_t tid, int64_t step, bool vec_ok) {
  using S = typename T::store_t;
  constexpr int X = 16 / sizeof(S);  // elements per 16-byte load
  const int64_t nvec = vec_ok ? n / X : 0;
  for (int64_t i = tid; i < nvec; i += step) {
    S v[X];
    *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(in + i * X);
    uint32_t w[X / 4];
#pragma unroll
    for (int k = 0; k < X / 4; ++k)
      w[k] = fp8_pack4<INVERTED>(T::to_float(v[4 * k]), T::to_float(v[4 * k + 1]), T::to_float(v[4 * k + 2]),
                                 T::to_float(v[4 * k + 3]), scale);
    if constexpr (X == 8) *reinterpret_cast<uint2*>(out + i * X) = uint2{w[0], w[1]};
    else *reinterpret_cast<uint32_t*>(out + i * X) = w[0];
  }
  for (int64_t i = nvec * X + tid; i < n; i += step) {
    const uint32_t w = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_prep<INVERTED>(T::to_float(in[i]), scale), 0.f, 0, false);
    out[i] = (uint8_t)w;
  }
}

template <typename T>
__device__ __forceinline__ float absmax_span(const typename T::store_t* __restrict__ in, int64_t n, int64_t tid,
                                             int64_t step, bool vec_ok) {
  using S = typename T::store_t;
  constexpr int X = 16 / sizeof(S);
  const int64_t nvec = vec_ok ? n / X : 0;
  float m = 0.f;
  for (int64_t i = tid; i < nvec; i += step) {
    S v[X];
    *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(in + i * X);
#pragma unroll
    for (int k = 0; k < X; ++k) m = fmaxf(m, fabsf(T::to_float(v[k])));
  }
  for (int64_t i = nvec * X + tid; i < n; i += step) m = fmaxf(m, fabsf(T::to_float(in[i])));
  return m;
}

__device__ __forceinline__ float block_max(float v, float* smem) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  float r = threadIdx.x < nw ? smem[threadIdx.x] : 0.f;
  if (wave == 0) r = wave_max(r);
  if (threadIdx.x == 0) smem[0] = r;
  __syncthreads();
  return smem[0];
}

template <typename T>
__global__ __launch_bounds__(1024) void static_fp8_quant_kernel(uint8_t* __restrict__ out,
                                                                const typename T::store_t* __restrict__ in,
                                                                const float* __restrict__ scale, int64_t n,
                                                                bool vec_ok) {
  const float inv = 1.0f / (*scale);  // common.cu:171-173
  quant_span<T, true>(out, in, n, inv, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                      (int64_t)gridDim.x * blockDim.x, vec_ok);
}

// *scale must hold a value <= 0 on entry (the reference's contract, common.cu:40-45): every
// workgroup folds its maximum in with an integer atomic max on the bits of a non-negative float.
template <typename T>
__global__ __launch_bounds__(1024) void absmax_scale_kernel(float* __restrict__ scale,
                                                            const typename T::store_t* __restrict__ in, int64_t n,
                                                            bool vec_ok) {
  __shared__ float smem[16];
  const float m = block_max(absmax_span<T>(in, n, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                                           (int64_t)gridDim.x * blockDim.x, vec_ok),
                            smem);
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<int*>(scale), __float_as_int(m / kFp8Max));
}

template <typename T>
__global__ __launch_bounds__(1024) void per_token_fp8_quant_kernel(uint8_t* __restrict__ out,
                                                                   float* __restrict__ scales,
                                                                   const typename T::store_t* __restrict__ in,
                                                                   const float* __restrict__ scale_ub,
                                                                   int hidden_size, bool vec_ok) {
  __shared__ float smem[16];
  const int64_t token = blockIdx.x;
  const typename T::store_t* row = in + token * hidden_size;
  float m = block_max(absmax_span<T>(row, hidden_size, threadIdx.x, blockDim.x, vec_ok), smem);
  if (scale_ub != nullptr) m = fminf(m, *scale_ub);
  const float token_scale = fmaxf(m / kFp8Max, 1.0f / (kFp8Max * 512.f));  // common.cu:182-204
  if (threadIdx.x == 0) scales[token] = token_scale;
  quant_span<T, false>(out + token * hidden_size, row, hidden_size, token_scale, threadIdx.x, blockDim.x, vec_ok);
}

static bool span_vec_ok(const void* in, const void* out, int64_t row_elems, int esize) {
  return (((uintptr_t)in & 15) == 0) && (((uintptr_t)out & 7) == 0) && (row_elems * esize) % 16 == 0;
}

}  // namespace lvllm

using namespace lvllm;

extern "C" int lvllm_static_scaled_fp8_quant(void* out, const void* input, const float* scale, int64_t num_elems,
                                             int dtype, void* stream) {
  LV_CHECK(num_elems >= 0, "negative size");
  if (num_elems == 0) return 0;
  const bool vec_ok = (((uintptr_t)input & 15) == 0) && (((uintptr_t)out & 7) == 0);
  const int64_t want = (num_elems / 8 + 1023) / 1024;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, 2048));
  LV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((static_fp8_quant_kernel<scalar_t>), dim3(grid), dim3(1024), 0,
                                              (hipStream_t)stream, (uint8_t*)out,
                                              (const typename scalar_t::store_t*)input, scale, num_elems, vec_ok));
  LV_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvllm_dynamic_scaled_fp8_quant(void* out, const void* input, float* scale, int64_t num_elems,
                                              int dtype, void* stream) {
  LV_CHECK(num_elems >= 0, "negative size");
  if (num_elems == 0) return 0;
  const bool vec_ok = (((uintptr_t)input & 15) == 0) && (((uintptr_t)out & 7) == 0);
  const int64_t want = (num_elems / 8 + 1023) / 1024;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, 2048));
  LV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((absmax_scale_kernel<scalar_t>), dim3(grid), dim3(1024), 0,
                                              (hipStream_t)stream, scale,
                                              (const typename scalar_t::store_t*)input, num_elems, vec_ok));
  LV_LAUNCH_CHECK();
  LV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((static_fp8_quant_kernel<scalar_t>), dim3(grid), dim3(1024), 0,
                                              (hipStream_t)stream, (uint8_t*)out,
                                              (const typename scalar_t::store_t*)input, scale, num_elems, vec_ok));
  LV_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvllm_dynamic_per_token_scaled_fp8_quant(void* out, float* scales, const void* input,
                                                        const float* scale_ub, int num_tokens, int hidden_size,
                                                        int dtype, void* stream) {
  LV_CHECK(num_tokens >= 0 && hidden_size > 0, "bad sizes");
  if (num_tokens == 0) return 0;
  const int esize = dtype == LVLLM_F32 ? 4 : 2;
  const bool vec_ok = span_vec_ok(input, out, hidden_size, esize) && hidden_size % 8 == 0;
  const int per_thread = 16 / esize;
  int threads = ((hidden_size / per_thread + 63) / 64) * 64;
  threads = std::max(64, std::min(threads, 1024));
  LV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((per_token_fp8_quant_kernel<scalar_t>), dim3(num_tokens),
                                              dim3(threads), 0, (hipStream_t)stream, (uint8_t*)out, scales,
                                              (const typename scalar_t::store_t*)input, scale_ub, hidden_size,
                                              vec_ok));
  LV_LAUNCH_CHECK();
  return 0;
}
