// Shared device/host helpers for the gfx950 kernels of the paged-attention
// decode path.  CDNA4 only: wave = 64 lanes, no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/lvllm_hip.h"

namespace lvllm {

constexpr int kWave = 64;

// ---- error plumbing -------------------------------------------------------
void set_error(const std::string& msg);

// Launch-shape knobs a host can set for its concurrency level (lvllm_set_tuning):
//   gemm_workgroups  workgroups of the decode GEMM's N split (default 256 = one per CU; an engine
//                    that keeps two steps in flight on two streams sets 128: each GEMM then
//                    leaves half the CUs to the other stream's kernel, +10 % tokens/s measured)
//   gemm_balance     see Tuning
//   attn_waves       waves per workgroup of the long-context decode attention (8 or 4)
//   attn_splits      how paged_attention_v2 cuts contexts: 0 automatic, n >= 1 forced, -1 the reference's
//                    512-token partitions (scratch contents then equal the reference's)
//   prefill_lds      1 | 0, see Tuning
//   prefill_mfma32_min_query   see Tuning
//   prefill_chunk_max_query, prefill_chunk_max_avg_x8    see Tuning
//   varlen_dense     see Tuning
struct Tuning {
  int gemm_workgroups = 256;
  int gemm_workgroups_wide = 0;  // for projections with >= gemm_wide_min_tiles n-tiles; 0 = as above
  int gemm_wide_min_tiles = 1024;  // 1024: gate_up and lm_head of an 8B model; 4096: lm_head only
  int gemm_partials_ksplit = 0;  // > 0: a projection that leaves split-K partials splits K at least this many ways
  int gemm_balance = 1;  // 1: the launch takes the FEWEST workgroups (<= gemm_workgroups) that need no more rounds of n-tiles
                         // than gemm_workgroups would -- 384 tiles on 256 workgroups are two rounds with half the
                         // workgroups idle in the second, on 192 two full ones (round 4, profiles/r04_tuning.md); 0: all
  int attn_waves = 8;
  int attn_splits = 0;  // paged_attention_v2: 0 = shares chosen per call; n >= 1 = n shares; -1 = 512-token partitions
  int swap_kernel_min_runs = 3;  // swap_blocks: more contiguous runs than this (and a pinned host side) -> one kernel
  int cache_tile_min_tokens = 384;  // reshape_and_cache: >= this many tokens take the LDS-tiled kernel (consecutive
                                    // slots: 6.0 us against 6.8 at 512 tokens, 5.8 against 5.0 at 256; scattered
                                    // slots cost the tiled kernel 17 us at any small size)
  int varlen_dense = 1;  // lvllm_varlen_attention: launches the 32x32 body takes read the caller's K / V rows
                         // themselves (row-major LDS images, transposed reads of V); 0: always the pack pass first
  int varlen_dense_waves = 0;  // waves per workgroup of that launch: 8 (256 columns per tile stream), 4 (128), 0 = by
                               // the longest sequence (4 up to 256 tokens, causal up to 512)
  int prefill_lds = 1;  // prefill kernel: K/V tiles staged once per workgroup in LDS (0: per-wave loads)
  int prefill_mfma32_min_query = 64;  // launches whose longest chunk has at least this many query tokens take the
                                      // 32x32-MFMA body (prefill_mfma32.h; plain, head size 64 / 128, 16-bit cache) --
                                      // and launches with chunks of 16+ tokens whose grid fits the CUs at once; 0 = never
  int prefill_chunk_max_query = 64;   // launches that are mostly one-token sequences (mixed steps of chunked prefill:
  int prefill_chunk_max_avg_x8 = 16;  // at most max_avg_x8 / 8 = 2 query tokens per sequence on average, no chunk
                                      // longer than max_query) take the decode-style walk of prefill_chunk.h (plain
                                      // causal, head size 64 / 128, 16-bit cache); max_query 0 = never
};
Tuning& tuning();

#define LV_CHECK(cond, msg)                                   \
  do {                                                        \
    if (!(cond)) {                                            \
      ::lvllm::set_error(std::string(__func__) + ": " + msg); \
      return 1;                                               \
    }                                                         \
  } while (0)

#define LV_LAUNCH_CHECK()                                                    \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      ::lvllm::set_error(std::string(__func__) + ": " + hipGetErrorString(e_)); \
      return 2;                                                              \
    }                                                                        \
  } while (0)

// ---- element types --------------------------------------------------------
// Tags carry the storage type (16-bit patterns are moved as uint16_t) and the
// float conversions.  Rounding is round-to-nearest-even, the same rounding
// c10::BFloat16 / __half conversions perform in the reference.
struct F32 {
  using store_t = float;
  static constexpr int kDtype = LVLLM_F32;
  __device__ static inline float to_float(store_t v) { return v; }
  __device__ static inline store_t from_float(float f) { return f; }
};

struct F16 {
  using store_t = uint16_t;
  static constexpr int kDtype = LVLLM_F16;
  __device__ static inline float to_float(store_t v) {
    return (float)__builtin_bit_cast(_Float16, v);
  }
  // The rounding point is pinned: the fp32 value is materialised (empty asm) and THEN converted.  Left to
  // instruction selection, `(_Float16)(a * b)` may become one v_fma_mixlo_f16 in one kernel and v_mul_f32 +
  // v_cvt_f16_f32 in another -- which instruction sequence a kernel got changed with unrelated edits, and two kernels
  // that must agree bit for bit (fused_add_rms_norm vs its split-K twin, round 3) stopped agreeing on a few elements.
  // The reference's kernels round the fp32 result to half (`__float2half(x * s)`): two roundings, as here.
  __device__ static inline store_t from_float(float f) {
    asm("" : "+v"(f));
    return __builtin_bit_cast(uint16_t, (_Float16)f);
  }
};

struct BF16 {
  using store_t = uint16_t;
  static constexpr int kDtype = LVLLM_BF16;
  __device__ static inline float to_float(store_t v) {
    return __builtin_bit_cast(float, (uint32_t)v << 16);
  }
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 on gfx950 (RNE, NaN stays NaN)
  __device__ static inline store_t from_float(float f) {
    return __builtin_bit_cast(uint16_t, (__bf16)f);
  }
};

// 4 floats -> 4 fp8 (OCP e4m3fn) in one dword: fp8(x / scale), saturating at +-448, NaN -> 0x7f
// (the store side of an fp8 KV cache: csrc/cache_kernels.cu:194-202,
// fp8/nvidia/quant_utils.cuh:458-489; round to nearest even by v_cvt_pk_fp8_f32)
__device__ inline uint32_t fp8_kv_quant4(float a, float b, float c, float d, float scale) {
  auto sat = [scale](float v) {
    v = v / scale;
    return fabsf(v) > 448.f ? copysignf(448.f, v) : v;  // NaN compares false and passes through
  };
  const float sa = sat(a), sb = sat(b), sc = sat(c), sd = sat(d);
  uint32_t w = __builtin_amdgcn_cvt_pk_fp8_f32(sa, sb, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(sc, sd, w, true);
  if (sa != sa) w = (w & 0xffffff00u) | 0x0000007fu;
  if (sb != sb) w = (w & 0xffff00ffu) | 0x00007f00u;
  if (sc != sc) w = (w & 0xff00ffffu) | 0x007f0000u;
  if (sd != sd) w = (w & 0x00ffffffu) | 0x7f000000u;
  return w;
}

// 4 activations -> 4 fp8 (OCP e4m3fn) in one dword with the arithmetic of static_scaled_fp8_quant
// (csrc/quantization/fp8/common.cu:24-38,171-176): e4m3(clamp(x * inv_scale, +-448)), inv_scale = 1.0f / scale, the
// clamp as fmax(-448, fmin(x, 448)) (a NaN comes out as +448).  ONE definition for every place that quantises an
// activation -- the quant op, the W8A8 GEMM's own prologue, and the producers that hand the next GEMM its input
// already quantised (norm kernels, the SwiGLU epilogue) -- so that all of them produce the same bytes.
__device__ __forceinline__ uint32_t fp8_act_quant4(float a, float b, float c, float d, float inv_scale) {
  auto prep = [inv_scale](float v) { return fmaxf(-448.f, fminf(v * inv_scale, 448.f)); };
  const uint32_t w = __builtin_amdgcn_cvt_pk_fp8_f32(prep(a), prep(b), 0, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(prep(c), prep(d), w, true);
}

// 16-byte vector of storage elements
template <typename T>
struct Vec16 {
  static constexpr int N = 16 / sizeof(typename T::store_t);
  typename T::store_t v[N];
};

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// Block-wide sum for blocks of up to 16 waves; `red` is 16 floats of LDS.
__device__ inline float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = (lane < nw) ? red[lane] : 0.f;
  r = wave_sum(r);
  __syncthreads();
  return r;
}

#define LV_DISPATCH_DTYPE(dtype, ...)            \
  switch (dtype) {                               \
    case LVLLM_F32: {                            \
      using scalar_t = ::lvllm::F32;             \
      __VA_ARGS__;                               \
    } break;                                     \
    case LVLLM_F16: {                            \
      using scalar_t = ::lvllm::F16;             \
      __VA_ARGS__;                               \
    } break;                                     \
    case LVLLM_BF16: {                           \
      using scalar_t = ::lvllm::BF16;            \
      __VA_ARGS__;                               \
    } break;                                     \
    default:                                     \
      LV_CHECK(false, "unsupported dtype");      \
  }

}  // namespace lvllm
