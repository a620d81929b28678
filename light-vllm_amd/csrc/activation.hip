// silu_and_mul (SwiGLU gate) for gfx950.
//
// Semantics: csrc/activation_kernels.cu:9-30 of the reference:
//   out[t, i] = T( float(T(x / (1 + expf(-x)))) * float(y) ),
//   x = input[t, i], y = input[t, d + i]   (the activation is rounded to T
//   before the T x T multiply).
// Launch: flat grid over 16-byte chunks (not one block per token): at decode
// batch sizes a per-token grid leaves most of the 256 CUs idle.
#include "common.h"

namespace lvllm {

template <typename T>
__device__ inline typename T::store_t silu_mul(typename T::store_t x, typename T::store_t y) {
  const float xf = T::to_float(x);
  const typename T::store_t a = T::from_float(xf / (1.0f + expf(-xf)));
  return T::from_float(T::to_float(a) * T::to_float(y));
}

template <typename T>
__global__ void silu_and_mul_vec_kernel(typename T::store_t* __restrict__ out,
                                        const typename T::store_t* __restrict__ input,
                                        const int64_t num_chunks, const int chunks_per_row) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < num_chunks;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = idx / chunks_per_row;
    const int c = (int)(idx - t * chunks_per_row);
    const V* row = reinterpret_cast<const V*>(input) + t * 2 * chunks_per_row;
    const V x = row[c];
    const V y = row[chunks_per_row + c];
    V o;
#pragma unroll
    for (int j = 0; j < N; ++j) o.v[j] = silu_mul<T>(x.v[j], y.v[j]);
    reinterpret_cast<V*>(out)[idx] = o;
  }
}

template <typename T>
__global__ void silu_and_mul_scalar_kernel(typename T::store_t* __restrict__ out,
                                           const typename T::store_t* __restrict__ input,
                                           const int64_t n, const int d) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = idx / d;
    const int i = (int)(idx - t * d);
    out[idx] = silu_mul<T>(input[t * 2 * d + i], input[t * 2 * d + d + i]);
  }
}

template <typename T>
static int launch_silu(void* out, const void* input, int64_t num_tokens, int d,
                       hipStream_t stream) {
  using S = typename T::store_t;
  constexpr int N = Vec16<T>::N;
  const bool vec = d % N == 0 && (((uintptr_t)out | (uintptr_t)input) & 15) == 0;
  const int threads = 256;
  if (vec) {
    const int cpr = d / N;
    const int64_t chunks = num_tokens * cpr;
    const int64_t want = (chunks + threads - 1) / threads;
    hipLaunchKernelGGL((silu_and_mul_vec_kernel<T>), dim3((int)(want < 8192 ? want : 8192)),
                       dim3(threads), 0, stream, (S*)out, (const S*)input, chunks, cpr);
  } else {
    const int64_t n = num_tokens * d;
    const int64_t want = (n + threads - 1) / threads;
    hipLaunchKernelGGL((silu_and_mul_scalar_kernel<T>), dim3((int)(want < 8192 ? want : 8192)),
                       dim3(threads), 0, stream, (S*)out, (const S*)input, n, d);
  }
  return 0;
}

}  // namespace lvllm

using namespace lvllm;

extern "C" int lvllm_silu_and_mul(void* out, const void* input, int64_t num_tokens, int d,
                                  int dtype, void* stream) {
  if (num_tokens == 0 || d == 0) return 0;
  LV_DISPATCH_DTYPE(dtype, (launch_silu<scalar_t>(out, input, num_tokens, d, (hipStream_t)stream)));
  LV_LAUNCH_CHECK();
  return 0;
}

// Exact (erf) GELU of the encoder models' MLP (xlm_roberta.py `hidden_act = "gelu"`; torch.nn.functional.gelu in
// the reference): out = T(0.5 x (1 + erf(x / sqrt(2)))) in fp32, elementwise, out may alias x.
namespace lvllm {
template <typename T>
__global__ void gelu_vec_kernel(typename T::store_t* out, const typename T::store_t* x, const int64_t num_chunks) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < num_chunks;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const V a = reinterpret_cast<const V*>(x)[idx];
    V o;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float f = T::to_float(a.v[j]);
      o.v[j] = T::from_float(f * 0.5f * (1.0f + erff(f * 0.70710678118654752440f)));
    }
    reinterpret_cast<V*>(out)[idx] = o;
  }
}
}  // namespace lvllm

extern "C" int lvllm_gelu(void* out, const void* x, int64_t numel, int dtype, void* stream) {
  if (numel == 0) return 0;
  LV_CHECK(dtype == LVLLM_BF16 || dtype == LVLLM_F16, "16-bit element types only");
  LV_CHECK(numel % 8 == 0 && (((uintptr_t)out | (uintptr_t)x) & 15) == 0, "numel % 8 == 0 and 16-byte aligned pointers");
  const int64_t chunks = numel / 8;
  const int threads = 256;
  int64_t want = (chunks + threads - 1) / threads;
  const int grid = (int)(want < 16384 ? want : 16384);
  if (dtype == LVLLM_BF16)
    hipLaunchKernelGGL((lvllm::gelu_vec_kernel<lvllm::BF16>), dim3(grid), dim3(threads), 0, (hipStream_t)stream,
                       (uint16_t*)out, (const uint16_t*)x, chunks);
  else
    hipLaunchKernelGGL((lvllm::gelu_vec_kernel<lvllm::F16>), dim3(grid), dim3(threads), 0, (hipStream_t)stream,
                       (uint16_t*)out, (const uint16_t*)x, chunks);
  LV_LAUNCH_CHECK();
  return 0;
}
