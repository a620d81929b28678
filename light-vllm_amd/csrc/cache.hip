// KV-cache block operations for gfx950: reshape_and_cache, reshape_and_cache_flash,
// copy_blocks, swap_blocks.  Pure byte movement, bit-exact by construction.
//
// Semantics follow the reference kernels (index formulas only):
//   reshape_and_cache        csrc/cache_kernels.cu:152-204
//   reshape_and_cache_flash  csrc/cache_kernels.cu:206-247
//   copy_blocks              csrc/cache_kernels.cu:67-148
//   swap_blocks              csrc/cache_kernels.cu:24-63
// The launch shapes are not the reference's: work is cut into 16-byte chunks so
// that every global access is a dwordx4 wherever the layout allows it, and
// copy_blocks takes device pointer tables so it never synchronises the host.
#include <string.h>

#include <mutex>

#include "common.h"

namespace lvllm {

// One thread per x-element chunk of one (token, head): K chunk is contiguous
// in both the source row and the paged layout [.., D/x, BS, x]; V elements of
// the chunk go to x different rows of [.., D, BS].
template <typename store_t, int X>
__global__ void reshape_and_cache_kernel(
    const store_t* __restrict__ key, const store_t* __restrict__ value,
    store_t* __restrict__ key_cache, store_t* __restrict__ value_cache,
    const int64_t* __restrict__ slot_mapping, const int64_t num_chunks,
    const int chunks_per_head, const int num_heads, const int head_size,
    const int block_size, const int64_t key_stride, const int64_t value_stride,
    const bool vec_ok, const int64_t num_slots, const int64_t block_stride) {
  using vec_t = uint4;
  static_assert(sizeof(store_t) * X == 16, "chunk must be 16 bytes");
  const int chunks_per_token = chunks_per_head * num_heads;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < num_chunks; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t token = idx / chunks_per_token;
    const int rem = (int)(idx - token * chunks_per_token);
    const int64_t slot = slot_mapping[token];
    if (slot < 0 || slot >= num_slots) continue;  // padding token / a slot outside the stated extent
    const int head = rem / chunks_per_head;
    const int x_idx = rem - head * chunks_per_head;
    const int64_t block_idx = slot / block_size;
    const int64_t block_off = slot % block_size;

    const store_t* ksrc = key + token * key_stride + head * head_size + x_idx * X;
    const store_t* vsrc = value + token * value_stride + head * head_size + x_idx * X;
    // (block_stride: elements between blocks -- num_heads * head_size * block_size, or more when the caller pads)
    store_t* kdst = key_cache + block_idx * block_stride +
                    ((int64_t)head * chunks_per_head + x_idx) * (int64_t)block_size * X + block_off * X;
    store_t* vdst = value_cache + block_idx * block_stride +
                    ((int64_t)head * head_size + x_idx * X) * (int64_t)block_size + block_off;
    store_t kv[X], vv[X];
    if (vec_ok) {
      *reinterpret_cast<vec_t*>(kv) = *reinterpret_cast<const vec_t*>(ksrc);
      *reinterpret_cast<vec_t*>(vv) = *reinterpret_cast<const vec_t*>(vsrc);
      *reinterpret_cast<vec_t*>(kdst) = *reinterpret_cast<const vec_t*>(kv);
    } else {
#pragma unroll
      for (int i = 0; i < X; ++i) {
        kv[i] = ksrc[i];
        vv[i] = vsrc[i];
      }
#pragma unroll
      for (int i = 0; i < X; ++i) kdst[i] = kv[i];
    }
#pragma unroll
    for (int i = 0; i < X; ++i) vdst[(int64_t)i * block_size] = vv[i];
  }
}

// kv_cache_dtype "fp8": every element becomes e4m3fn(float(x) / scale), saturating at +-448, NaN
// kept (csrc/cache_kernels.cu:194-202, fp8/nvidia/quant_utils.cuh:458-489).  x = 16: one thread per
// 16-element chunk of one (token, head) -- a 16-byte K store, 16 V bytes to 16 rows of [.., D, BS].
template <typename T>
__global__ void reshape_and_cache_fp8_kernel(
    const typename T::store_t* __restrict__ key, const typename T::store_t* __restrict__ value,
    uint8_t* __restrict__ key_cache, uint8_t* __restrict__ value_cache,
    const int64_t* __restrict__ slot_mapping, const int64_t num_chunks, const int chunks_per_head,
    const int num_heads, const int head_size, const int block_size, const int64_t key_stride,
    const int64_t value_stride, const float k_scale, const float v_scale, const int64_t num_slots,
    const int64_t block_stride) {
  const int chunks_per_token = chunks_per_head * num_heads;
  auto quant4 = [](float a, float b, float c, float d, float scale) { return fp8_kv_quant4(a, b, c, d, scale); };
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < num_chunks;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t token = idx / chunks_per_token;
    const int rem = (int)(idx - token * chunks_per_token);
    const int64_t slot = slot_mapping[token];
    if (slot < 0 || slot >= num_slots) continue;  // padding token / outside the stated extent
    const int head = rem / chunks_per_head;
    const int x_idx = rem - head * chunks_per_head;
    const int64_t block_idx = slot / block_size;
    const int64_t block_off = slot % block_size;
    const typename T::store_t* ksrc = key + token * key_stride + head * head_size + x_idx * 16;
    const typename T::store_t* vsrc = value + token * value_stride + head * head_size + x_idx * 16;
    float kf[16], vf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      kf[i] = T::to_float(ksrc[i]);
      vf[i] = T::to_float(vsrc[i]);
    }
    uint4 kq;
    kq.x = quant4(kf[0], kf[1], kf[2], kf[3], k_scale);
    kq.y = quant4(kf[4], kf[5], kf[6], kf[7], k_scale);
    kq.z = quant4(kf[8], kf[9], kf[10], kf[11], k_scale);
    kq.w = quant4(kf[12], kf[13], kf[14], kf[15], k_scale);
    uint8_t* kdst = key_cache + block_idx * block_stride +
                    ((int64_t)head * chunks_per_head + x_idx) * (int64_t)block_size * 16 + block_off * 16;
    *reinterpret_cast<uint4*>(kdst) = kq;
    uint8_t* vdst = value_cache + block_idx * block_stride +
                    ((int64_t)head * head_size + x_idx * 16) * (int64_t)block_size + block_off;
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      const uint32_t w = quant4(vf[i], vf[i + 1], vf[i + 2], vf[i + 3], v_scale);
      vdst[(int64_t)(i + 0) * block_size] = (uint8_t)(w);
      vdst[(int64_t)(i + 1) * block_size] = (uint8_t)(w >> 8);
      vdst[(int64_t)(i + 2) * block_size] = (uint8_t)(w >> 16);
      vdst[(int64_t)(i + 3) * block_size] = (uint8_t)(w >> 24);
    }
  }
}

// convert_fp8 (csrc/cache_kernels.cu:334-410, "only for testing" there): elementwise
// T -> fp8(float(x) / scale) or fp8 -> T(float(fp8) * scale), the two scaled_convert directions.
template <typename T, bool TO_FP8>
__global__ void convert_fp8_kernel(void* __restrict__ dst, const void* __restrict__ src, const float scale,
                                   const int64_t n) {
  using S = typename T::store_t;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (TO_FP8) {
      float v = T::to_float(reinterpret_cast<const S*>(src)[i]) / scale;
      const bool nan = v != v;
      v = fabsf(v) > 448.f ? copysignf(448.f, v) : v;
      const uint32_t w = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
      reinterpret_cast<uint8_t*>(dst)[i] = nan ? (uint8_t)0x7f : (uint8_t)w;
    } else {
      const uint32_t w = reinterpret_cast<const uint8_t*>(src)[i];
      const float v = __builtin_amdgcn_cvt_f32_fp8(w, 0);
      reinterpret_cast<S*>(dst)[i] = T::from_float(v * scale);
    }
  }
}

// Flash layout [NB, BS, H, D]: both K and V rows are contiguous per token.
template <typename store_t, int X>
__global__ void reshape_and_cache_flash_kernel(
    const store_t* __restrict__ key, const store_t* __restrict__ value,
    store_t* __restrict__ key_cache, store_t* __restrict__ value_cache,
    const int64_t* __restrict__ slot_mapping, const int64_t num_chunks,
    const int chunks_per_token, const int row_elems, const int block_size,
    const int64_t block_stride, const int64_t key_stride,
    const int64_t value_stride, const bool vec_ok) {
  using vec_t = uint4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
       idx < num_chunks; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t token = idx / chunks_per_token;
    const int c = (int)(idx - token * chunks_per_token);
    const int64_t slot = slot_mapping[token];
    if (slot < 0) continue;
    const int64_t block_idx = slot / block_size;
    const int64_t block_off = slot % block_size;
    const int64_t dst = block_idx * block_stride + block_off * row_elems + (int64_t)c * X;
    const store_t* ksrc = key + token * key_stride + (int64_t)c * X;
    const store_t* vsrc = value + token * value_stride + (int64_t)c * X;
    if (vec_ok) {
      *reinterpret_cast<vec_t*>(key_cache + dst) = *reinterpret_cast<const vec_t*>(ksrc);
      *reinterpret_cast<vec_t*>(value_cache + dst) = *reinterpret_cast<const vec_t*>(vsrc);
    } else {
      const int n = min(X, row_elems - c * X);
      for (int i = 0; i < n; ++i) {
        key_cache[dst + i] = ksrc[i];
        value_cache[dst + i] = vsrc[i];
      }
    }
  }
}

// grid (layer, pair, 2 = K|V).  16-byte chunks, fully coalesced.
__global__ void copy_blocks_kernel(const void* const* __restrict__ key_cache_ptrs,
                                   const void* const* __restrict__ value_cache_ptrs,
                                   const int64_t* __restrict__ block_mapping,
                                   const int64_t block_bytes) {
  const int layer = blockIdx.x, pair = blockIdx.y;
  char* base = (char*)(blockIdx.z == 0 ? key_cache_ptrs[layer] : value_cache_ptrs[layer]);
  const int64_t src = block_mapping[2 * pair], dst = block_mapping[2 * pair + 1];
  const char* s = base + src * block_bytes;
  char* d = base + dst * block_bytes;
  if ((block_bytes & 15) == 0 && (((uintptr_t)base) & 15) == 0) {
    const int64_t n = block_bytes >> 4;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
      reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
  } else {
    for (int64_t i = threadIdx.x; i < block_bytes; i += blockDim.x) d[i] = s[i];
  }
}

// 16-bit caches, many tokens (prompt chunks): one workgroup per (tile of 64 consecutive tokens, kv head).
// The per-chunk kernel above writes a V chunk as 8 two-byte stores to 8 rows of [.., D, BS] and a K chunk as
// a 16-byte piece of a 256-byte-strided row: at 8 192 tokens it moved 1.2 TB/s (profiles/r01_bench_kernel_
// stats_v6.csv).  Here both tiles go through LDS and leave in DESTINATION order:
//   K  [tok][d8] -> [d8][tok]: lane (d8, tok) stores 16 bytes; consecutive tokens of a sequence sit in
//      consecutive slots, so 16 lanes cover one 256-byte row of the block;
//   V  [tok][d] -> [d][tok]: the tile's tokens are cut into groups of consecutive slots inside one aligned
//      8-slot window (group = at most 8 tokens, found with two ballots); a thread owns (d, group) and stores
//      its 8 elements as ONE 16-byte piece when the group is full, element by element at sequence edges.
// slot_mapping is arbitrary (slot < 0 or >= num_slots: skipped): every address comes from the token's own
// slot, only the speed depends on the slots being consecutive.  Pure byte movement: bit-exact.
constexpr int kTileTokens = 64;
__global__ __launch_bounds__(256) void reshape_and_cache_tile_kernel(
    const uint16_t* __restrict__ key, const uint16_t* __restrict__ value, uint16_t* __restrict__ key_cache,
    uint16_t* __restrict__ value_cache, const int64_t* __restrict__ slot_mapping, const int num_tokens,
    const int num_heads, const int head_size, const int block_size, const int64_t key_stride,
    const int64_t value_stride, const int64_t num_slots, const int64_t block_stride) {
  constexpr int TT = kTileTokens;
  constexpr int KROW = TT + 1;  // 16-byte units per d8 row of the K tile (odd: conflict-free transposed writes)
  constexpr int VROW = TT + 2;  // elements per d row of the V tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int cph = head_size >> 3;
  int64_t* slots = reinterpret_cast<int64_t*>(smem);                 // [TT]
  int* grp = reinterpret_cast<int*>(slots + TT);                     // [TT] start | len << 8
  int* ngrp = grp + TT;                                              // [4]
  uint4* kt = reinterpret_cast<uint4*>(ngrp + 4);                    // [cph][KROW]
  uint16_t* vt = reinterpret_cast<uint16_t*>(kt + cph * KROW);       // [head_size][VROW]
  const int tok0 = blockIdx.x * TT, head = blockIdx.y;
  const int nt = min(TT, num_tokens - tok0);
  const int tid = threadIdx.x;

  if (tid < TT) {  // wave 0: slots of the tile and the groups of its V stores
    int64_t s = tid < nt ? slot_mapping[tok0 + tid] : -1;
    if (s >= num_slots) s = -1;
    slots[tid] = s;
    const int64_t prev = (int64_t)__shfl_up((long long)s, 1);
    const bool valid = s >= 0;
    const bool first = valid && (tid == 0 || prev < 0 || s != prev + 1 || (s & 7) == 0);
    const uint64_t F = __ballot(first), V = __ballot(valid);
    if (first) {
      const uint64_t above = tid == 63 ? 0 : (F | ~V) >> (tid + 1);  // next group start or gap
      const int len = above ? __builtin_ctzll(above) + 1 : 64 - tid;
      grp[__builtin_popcountll(F & ((1ull << tid) - 1))] = tid | (len << 8);
    }
    if (tid == 0) ngrp[0] = __builtin_popcountll(F);
  }
  // source rows -> LDS, read in source order (whole 2*head_size-byte rows)
  for (int i = tid; i < nt * cph; i += 256) {
    const int tok = i / cph, j = i - tok * cph;
    const uint4 kv = *reinterpret_cast<const uint4*>(key + (int64_t)(tok0 + tok) * key_stride + head * head_size + j * 8);
    const uint4 vv = *reinterpret_cast<const uint4*>(value + (int64_t)(tok0 + tok) * value_stride + head * head_size + j * 8);
    kt[j * KROW + tok] = kv;
    const uint16_t* ve = reinterpret_cast<const uint16_t*>(&vv);
#pragma unroll
    for (int e = 0; e < 8; ++e) vt[(j * 8 + e) * VROW + tok] = ve[e];
  }
  __syncthreads();
  // K out: lane (d8, tok), tok fastest
  for (int i = tid; i < nt * cph; i += 256) {
    const int j = i / nt, tok = i - j * nt;
    const int64_t s = slots[tok];
    if (s < 0) continue;
    const int64_t b = s / block_size;
    const int o = (int)(s - b * block_size);
    uint16_t* dst = key_cache + b * block_stride + ((((int64_t)head * cph + j) * (int64_t)block_size + o) * 8);
    *reinterpret_cast<uint4*>(dst) = kt[j * KROW + tok];
  }
  // V out: thread (d, group), group fastest
  const int ng = ngrp[0];
  for (int i = tid; i < head_size * ng; i += 256) {
    const int d = i / ng, gi = i - d * ng;
    const int g = grp[gi];
    const int ts = g & 0xff, len = g >> 8;
    const int64_t s = slots[ts];
    const int64_t b = s / block_size;
    const int o = (int)(s - b * block_size);
    uint16_t* dst = value_cache + b * block_stride + (((int64_t)head * head_size + d) * block_size + o);
    const uint16_t* src = vt + d * VROW + ts;
    if (len == 8) {
      alignas(16) uint16_t e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = src[k];
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(e);
    } else {
      for (int k = 0; k < len; ++k) dst[k] = src[k];
    }
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace lvllm

using namespace lvllm;

extern "C" int lvllm_reshape_and_cache(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int x, int64_t key_stride, int64_t value_stride, int dtype,
    int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes, void* stream) {
  return lvllm_reshape_and_cache_strided(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads,
                                         head_size, block_size, x, key_stride, value_stride, dtype, kv_dtype, k_scale,
                                         v_scale, kv_cache_bytes, (int64_t)num_heads * head_size * block_size, stream);
}

extern "C" int lvllm_reshape_and_cache_strided(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int x, int64_t key_stride, int64_t value_stride, int dtype,
    int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes, int64_t kv_block_stride, void* stream) {
  LV_CHECK(kv_dtype == LVLLM_KV_AUTO || kv_dtype == LVLLM_KV_FP8_E4M3, "unsupported kv_cache_dtype");
  if (num_tokens == 0) return 0;
  const int esize = dtype == LVLLM_F32 ? 4 : 2;
  LV_CHECK(num_heads > 0 && head_size > 0 && block_size > 0, "num_heads, head_size, block_size must be positive");
  LV_CHECK(kv_block_stride >= (int64_t)num_heads * head_size * block_size,
           "kv_block_stride is smaller than one block");
  LV_CHECK((kv_block_stride * (kv_dtype == LVLLM_KV_FP8_E4M3 ? 1 : esize)) % 16 == 0,
           "kv_block_stride must be a multiple of 16 bytes");
  // slots the caches hold, from the extent the caller states (0: not stated): a slot beyond it is skipped
  // like a padding slot instead of being written outside the allocation
  const int64_t cache_esize = kv_dtype == LVLLM_KV_FP8_E4M3 ? 1 : esize;
  const int64_t num_slots = kv_cache_bytes > 0 ? kv_cache_bytes / (kv_block_stride * cache_esize) * block_size
                                               : INT64_MAX;
  LV_CHECK(dtype == LVLLM_F32 || dtype == LVLLM_F16 || dtype == LVLLM_BF16, "unsupported dtype");
  if (kv_dtype == LVLLM_KV_FP8_E4M3) {
    LV_CHECK(x == 16, "fp8 key_cache.size(4) must be 16");
    LV_CHECK(head_size % 16 == 0, "fp8 kv cache: head_size must be a multiple of 16");
    LV_CHECK(k_scale > 0.f && v_scale > 0.f, "fp8 kv cache: scales must be positive");
    LV_CHECK(aligned16(key_cache), "fp8 key_cache must be 16-byte aligned");
    const int chunks_per_head = head_size / 16;
    const int64_t num_chunks = (int64_t)num_tokens * num_heads * chunks_per_head;
    const int64_t want = (num_chunks + 255) / 256;
    const int grid = (int)(want < 4096 ? want : 4096);
    LV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(
        (reshape_and_cache_fp8_kernel<scalar_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
        (const typename scalar_t::store_t*)key, (const typename scalar_t::store_t*)value, (uint8_t*)key_cache,
        (uint8_t*)value_cache, slot_mapping, num_chunks, chunks_per_head, num_heads, head_size, block_size,
        key_stride, value_stride, k_scale, v_scale, num_slots, kv_block_stride));
    LV_LAUNCH_CHECK();
    return 0;
  }
  LV_CHECK(k_scale == 1.0f && v_scale == 1.0f, "k_scale/v_scale must be 1.0 with kv_cache_dtype 'auto'");
  LV_CHECK(x == 16 / esize, "key_cache.size(4) must be 16/sizeof(element)");
  LV_CHECK(head_size % x == 0, "head_size must be a multiple of x");
  const int chunks_per_head = head_size / x;
  const int64_t num_chunks = (int64_t)num_tokens * num_heads * chunks_per_head;
  const bool vec_ok = aligned16(key) && aligned16(value) && aligned16(key_cache) &&
                      (key_stride * esize) % 16 == 0 && (value_stride * esize) % 16 == 0;
  const int threads = 256;
  const int64_t want = (num_chunks + threads - 1) / threads;
  const int grid = (int)(want < 4096 ? want : 4096);
  hipStream_t s = (hipStream_t)stream;
  if (esize == 2 && vec_ok && aligned16(value_cache) && num_tokens >= tuning().cache_tile_min_tokens &&
      head_size <= 256 && block_size % 8 == 0 && num_heads <= 65535) {
    const int cph = head_size / 8;
    const size_t smem = kTileTokens * 8 + kTileTokens * 4 + 16 + (size_t)cph * (kTileTokens + 1) * 16 +
                        (size_t)head_size * (kTileTokens + 2) * 2;
    if (smem > 64 * 1024)  // head size 256: 67.8 KB of LDS -- say so, as the attention and prefill launches do
      (void)hipFuncSetAttribute((const void*)reshape_and_cache_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
    hipLaunchKernelGGL(reshape_and_cache_tile_kernel, dim3((num_tokens + kTileTokens - 1) / kTileTokens, num_heads),
                       dim3(256), smem, s, (const uint16_t*)key, (const uint16_t*)value, (uint16_t*)key_cache,
                       (uint16_t*)value_cache, slot_mapping, num_tokens, num_heads, head_size, block_size,
                       key_stride, value_stride, num_slots, kv_block_stride);
    LV_LAUNCH_CHECK();
    return 0;
  }
  if (esize == 2) {
    hipLaunchKernelGGL((reshape_and_cache_kernel<uint16_t, 8>), dim3(grid), dim3(threads), 0, s,
                       (const uint16_t*)key, (const uint16_t*)value, (uint16_t*)key_cache,
                       (uint16_t*)value_cache, slot_mapping, num_chunks, chunks_per_head,
                       num_heads, head_size, block_size, key_stride, value_stride, vec_ok, num_slots,
                       kv_block_stride);
  } else {
    hipLaunchKernelGGL((reshape_and_cache_kernel<float, 4>), dim3(grid), dim3(threads), 0, s,
                       (const float*)key, (const float*)value, (float*)key_cache,
                       (float*)value_cache, slot_mapping, num_chunks, chunks_per_head,
                       num_heads, head_size, block_size, key_stride, value_stride, vec_ok, num_slots,
                       kv_block_stride);
  }
  LV_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvllm_reshape_and_cache_flash(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int64_t block_stride, int64_t key_stride,
    int64_t value_stride, int dtype, int kv_dtype, float k_scale, float v_scale,
    void* stream) {
  LV_CHECK(kv_dtype == LVLLM_KV_AUTO, "fp8 kv cache is not built in this round (kv_cache_dtype must be 'auto')");
  (void)k_scale; (void)v_scale;
  if (num_tokens == 0) return 0;
  LV_CHECK(dtype == LVLLM_F32 || dtype == LVLLM_F16 || dtype == LVLLM_BF16, "unsupported dtype");
  const int esize = dtype == LVLLM_F32 ? 4 : 2;
  const int X = 16 / esize;
  const int row_elems = num_heads * head_size;
  const int chunks_per_token = (row_elems + X - 1) / X;
  const int64_t num_chunks = (int64_t)num_tokens * chunks_per_token;
  const bool vec_ok = row_elems % X == 0 && aligned16(key) && aligned16(value) &&
                      aligned16(key_cache) && aligned16(value_cache) &&
                      (key_stride * esize) % 16 == 0 && (value_stride * esize) % 16 == 0 &&
                      (block_stride * esize) % 16 == 0;
  const int threads = 256;
  const int64_t want = (num_chunks + threads - 1) / threads;
  const int grid = (int)(want < 4096 ? want : 4096);
  hipStream_t s = (hipStream_t)stream;
  if (esize == 2) {
    hipLaunchKernelGGL((reshape_and_cache_flash_kernel<uint16_t, 8>), dim3(grid), dim3(threads), 0, s,
                       (const uint16_t*)key, (const uint16_t*)value, (uint16_t*)key_cache,
                       (uint16_t*)value_cache, slot_mapping, num_chunks, chunks_per_token,
                       row_elems, block_size, block_stride, key_stride, value_stride, vec_ok);
  } else {
    hipLaunchKernelGGL((reshape_and_cache_flash_kernel<float, 4>), dim3(grid), dim3(threads), 0, s,
                       (const float*)key, (const float*)value, (float*)key_cache,
                       (float*)value_cache, slot_mapping, num_chunks, chunks_per_token,
                       row_elems, block_size, block_stride, key_stride, value_stride, vec_ok);
  }
  LV_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvllm_copy_blocks(const void* const* key_cache_ptrs,
                                 const void* const* value_cache_ptrs,
                                 const int64_t* block_mapping, int num_layers,
                                 int num_pairs, int64_t block_bytes, void* stream) {
  if (num_layers == 0 || num_pairs == 0) return 0;
  LV_CHECK(block_bytes > 0, "block_bytes must be positive");
  LV_CHECK(num_pairs <= 65535, "more than 65535 pairs in one call");
  const int64_t chunks = (block_bytes + 15) / 16;
  const int threads = (int)(chunks >= 1024 ? 1024 : ((chunks + 63) / 64) * 64);
  hipLaunchKernelGGL(copy_blocks_kernel, dim3(num_layers, num_pairs, 2), dim3(threads), 0,
                     (hipStream_t)stream, key_cache_ptrs, value_cache_ptrs, block_mapping,
                     block_bytes);
  LV_LAUNCH_CHECK();
  return 0;
}

namespace lvllm {

// Scattered swaps: one workgroup per (pair, 16 KiB slice of the block), 16 bytes per lane.  The host side of
// the transfer is PINNED memory mapped into the device's address space: the kernel reads / writes it over the
// host link itself, so n scattered blocks are one launch instead of n DMA submissions (10 us each: 3 GB/s for
// 32 KiB blocks against the link's 63 GB/s).  `mapping` is read from device-mapped pinned memory too.
__global__ __launch_bounds__(256) void swap_blocks_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                          const int64_t* __restrict__ mapping,
                                                          const int64_t block_bytes) {
  const int64_t s0 = mapping[2 * blockIdx.x], d0 = mapping[2 * blockIdx.x + 1];
  const uint4* sp = reinterpret_cast<const uint4*>(src + s0 * block_bytes);
  uint4* dp = reinterpret_cast<uint4*>(dst + d0 * block_bytes);
  const int64_t nvec = block_bytes >> 4;
  for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.y * blockDim.x)
    dp[i] = sp[i];
}

// Pinned, device-mapped ring for the pair lists of in-flight scattered swaps (the caller's list is pageable host
// memory and may change as soon as the call returns).  A slot is reused only after the launch that read it ended.
struct SwapRing {
  static constexpr int kSlots = 128;    // a swap of one sequence group is 2 * L calls: none of them waits for a slot
  static constexpr int kMaxPairs = 4096;  // 64 KiB per slot
  int64_t* host[kSlots] = {};
  int64_t* dev[kSlots] = {};
  hipEvent_t done[kSlots] = {};
  bool used[kSlots] = {};
  int next = 0;
  bool ok = false, tried = false;
  std::mutex mu;  // ONE ring per process (8 MiB of pinned memory, 128 events), whichever threads swap
  void release() {
    for (int i = 0; i < kSlots; ++i) {
      if (done[i] != nullptr) (void)hipEventDestroy(done[i]);
      if (host[i] != nullptr) (void)hipHostFree(host[i]);
      done[i] = nullptr;
      host[i] = dev[i] = nullptr;
    }
  }
  bool init() {  // (called with `mu` held)
    if (tried) return ok;
    tried = true;
    for (int i = 0; i < kSlots; ++i) {
      if (hipHostMalloc((void**)&host[i], (size_t)kMaxPairs * 16, hipHostMallocMapped) != hipSuccess ||
          hipHostGetDevicePointer((void**)&dev[i], host[i], 0) != hipSuccess ||
          hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        release();  // a partial ring is given back: the DMA path serves every call from now on
        return false;
      }
    }
    ok = true;
    return true;
  }
};
static SwapRing g_swap_ring;

// device-visible address of a host buffer, or nullptr when it is not pinned / mapped
static void* mapped_host_pointer(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory: not an error of ours
    return nullptr;
  }
  if (attr.type != hipMemoryTypeHost) return nullptr;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, const_cast<void*>(p), 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return d;
}

}  // namespace lvllm

extern "C" int lvllm_swap_blocks(const void* src, void* dst, const int64_t* block_mapping,
                                 int num_pairs, int64_t block_bytes, int src_is_device,
                                 int dst_is_device, void* stream) {
  LV_CHECK(src_is_device || dst_is_device, "Invalid device combination");
  hipMemcpyKind kind = src_is_device && dst_is_device ? hipMemcpyDeviceToDevice
                       : src_is_device               ? hipMemcpyDeviceToHost
                                                     : hipMemcpyHostToDevice;
  const char* s = (const char*)src;
  char* d = (char*)dst;
  if (num_pairs <= 0) return 0;
  // Runs where both block numbers advance by one move as ONE DMA (the reference issues one per block,
  // cache_kernels.cu:54-62; the bytes moved are identical).  When the mapping is scattered -- more than a few
  // runs -- and the host side is pinned, one kernel launch moves all of them (swap_blocks_kernel).
  int runs = 1;
  for (int i = 1; i < num_pairs; ++i)
    if (block_mapping[2 * i] != block_mapping[2 * (i - 1)] + 1 || block_mapping[2 * i + 1] != block_mapping[2 * (i - 1) + 1] + 1)
      ++runs;
  if (runs > tuning().swap_kernel_min_runs && (block_bytes & 15) == 0 && num_pairs <= SwapRing::kMaxPairs &&
      ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0)) {
    const void* ks = src_is_device ? src : mapped_host_pointer(src);
    void* kd = dst_is_device ? dst : mapped_host_pointer(dst);
    SwapRing& ring = g_swap_ring;
    std::unique_lock<std::mutex> lock(ring.mu);
    if (ks != nullptr && kd != nullptr && ring.init()) {
      const int slot = ring.next;
      ring.next = (ring.next + 1) % SwapRing::kSlots;
      if (ring.used[slot] && hipEventSynchronize(ring.done[slot]) != hipSuccess) LV_CHECK(false, "hipEventSynchronize failed");
      memcpy(ring.host[slot], block_mapping, (size_t)num_pairs * 16);
      const int slices = (int)((block_bytes + 16383) / 16384);
      hipLaunchKernelGGL(swap_blocks_kernel, dim3(num_pairs, slices < 1 ? 1 : (slices > 64 ? 64 : slices)), dim3(256), 0,
                         (hipStream_t)stream, (const char*)ks, (char*)kd, ring.dev[slot], block_bytes);
      LV_LAUNCH_CHECK();
      if (hipEventRecord(ring.done[slot], (hipStream_t)stream) != hipSuccess) LV_CHECK(false, "hipEventRecord failed");
      ring.used[slot] = true;
      return 0;
    }
    lock.unlock();
  }
  int i = 0;
  while (i < num_pairs) {
    const int64_t s0 = block_mapping[2 * i], d0 = block_mapping[2 * i + 1];
    int run = 1;
    while (i + run < num_pairs && block_mapping[2 * (i + run)] == s0 + run &&
           block_mapping[2 * (i + run) + 1] == d0 + run)
      ++run;
    hipError_t e = hipMemcpyAsync(d + d0 * block_bytes, s + s0 * block_bytes,
                                  (size_t)run * block_bytes, kind, (hipStream_t)stream);
    if (e != hipSuccess) {
      set_error(std::string("lvllm_swap_blocks: ") + hipGetErrorString(e));
      return 2;
    }
    i += run;
  }
  return 0;
}

extern "C" int lvllm_convert_fp8(void* dst, const void* src, float scale, int64_t num_elems, int dtype,
                                 int to_fp8, int kv_dtype, void* stream) {
  LV_CHECK(kv_dtype == LVLLM_KV_FP8_E4M3, "convert_fp8: kv_cache_dtype must be 'fp8' / 'fp8_e4m3'");
  LV_CHECK(scale > 0.f, "convert_fp8: scale must be positive");
  if (num_elems <= 0) return 0;
  const int64_t want = (num_elems + 255) / 256;
  const int grid = (int)(want < 8192 ? want : 8192);
  if (to_fp8) {
    LV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((convert_fp8_kernel<scalar_t, true>), dim3(grid), dim3(256), 0,
                                                (hipStream_t)stream, dst, src, scale, num_elems));
  } else {
    LV_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((convert_fp8_kernel<scalar_t, false>), dim3(grid), dim3(256), 0,
                                                (hipStream_t)stream, dst, src, scale, num_elems));
  }
  LV_LAUNCH_CHECK();
  return 0;
}
