// Rotary position embedding (in place on query and key) for gfx950.
//
// Semantics: csrc/pos_encoding_kernels.cu:10-92 of the reference.  Arithmetic
// is done in the element type T with a rounding after every operation, as the
// reference's `x * cos - y * sin` on scalar_t values does:
//   x' = T(T(x*cos) - T(y*sin));   y' = T(T(y*cos) + T(x*sin))
// NeoX pairing: (i, i + rot_dim/2); GPT-J pairing: (2i, 2i+1).
// cos_sin_cache row = [cos(0..rot_dim/2) | sin(0..rot_dim/2)].
//
// Launch: one workgroup per token; NeoX rows are processed in 16-byte chunks
// (8 rotation pairs of bf16 per thread: 4 dwordx4 loads, 2 dwordx4 stores).
#include "common.h"
#include "trace.h"

namespace lvllm {

template <typename T>
__device__ inline void rotate(typename T::store_t& x, typename T::store_t& y,
                              typename T::store_t c, typename T::store_t s) {
  const float xf = T::to_float(x), yf = T::to_float(y);
  const float cf = T::to_float(c), sf = T::to_float(s);
  const float xc = T::to_float(T::from_float(xf * cf));
  const float ys = T::to_float(T::from_float(yf * sf));
  const float yc = T::to_float(T::from_float(yf * cf));
  const float xs = T::to_float(T::from_float(xf * sf));
  // explicit single operations: keep hipcc from contracting them into FMAs
  x = T::from_float(__fsub_rn(xc, ys));
  y = T::from_float(__fadd_rn(yc, xs));
}

template <typename T, bool IS_NEOX, bool VEC>
__global__ void rotary_embedding_kernel(const int64_t* __restrict__ positions,
                                        typename T::store_t* __restrict__ query,
                                        typename T::store_t* __restrict__ key,
                                        const typename T::store_t* __restrict__ cos_sin_cache,
                                        const int rot_dim, const int64_t query_stride,
                                        const int64_t key_stride, const int num_heads,
                                        const int num_kv_heads, const int head_size) {
  using S = typename T::store_t;
  using V = Vec16<T>;
  constexpr int N = V::N;
  const int64_t token = blockIdx.x;
  const int64_t pos = positions[token];
  const S* cos_ptr = cos_sin_cache + pos * rot_dim;
  const int embed_dim = rot_dim / 2;
  const S* sin_ptr = cos_ptr + embed_dim;
  const int total_heads = num_heads + num_kv_heads;

  if constexpr (VEC && IS_NEOX) {
    const int cpe = embed_dim / N;  // chunks per head half
    const int n = total_heads * cpe;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int head = i / cpe, ch = i - head * cpe;
      S* base = head < num_heads
                    ? query + token * query_stride + (int64_t)head * head_size
                    : key + token * key_stride + (int64_t)(head - num_heads) * head_size;
      V x = *reinterpret_cast<const V*>(base + ch * N);
      V y = *reinterpret_cast<const V*>(base + embed_dim + ch * N);
      const V c = *reinterpret_cast<const V*>(cos_ptr + ch * N);
      const V s = *reinterpret_cast<const V*>(sin_ptr + ch * N);
#pragma unroll
      for (int j = 0; j < N; ++j) rotate<T>(x.v[j], y.v[j], c.v[j], s.v[j]);
      *reinterpret_cast<V*>(base + ch * N) = x;
      *reinterpret_cast<V*>(base + embed_dim + ch * N) = y;
    }
  } else if constexpr (VEC && !IS_NEOX) {
    // one 16-byte chunk = N/2 interleaved (x,y) pairs
    const int cph = rot_dim / N;
    const int n = total_heads * cph;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int head = i / cph, ch = i - head * cph;
      S* base = head < num_heads
                    ? query + token * query_stride + (int64_t)head * head_size
                    : key + token * key_stride + (int64_t)(head - num_heads) * head_size;
      V xy = *reinterpret_cast<const V*>(base + ch * N);
#pragma unroll
      for (int j = 0; j < N / 2; ++j) {
        const int r = ch * (N / 2) + j;
        rotate<T>(xy.v[2 * j], xy.v[2 * j + 1], cos_ptr[r], sin_ptr[r]);
      }
      *reinterpret_cast<V*>(base + ch * N) = xy;
    }
  } else {
    const int n = total_heads * embed_dim;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int head = i / embed_dim, r = i - head * embed_dim;
      S* base = head < num_heads
                    ? query + token * query_stride + (int64_t)head * head_size
                    : key + token * key_stride + (int64_t)(head - num_heads) * head_size;
      const int xi = IS_NEOX ? r : 2 * r;
      const int yi = IS_NEOX ? embed_dim + r : 2 * r + 1;
      rotate<T>(base[xi], base[yi], cos_ptr[r], sin_ptr[r]);
    }
  }
}

// Fused decode-step epilogue of the QKV projection (extension; the reference runs
// rotary_embedding and reshape_and_cache as two launches, qwen2.py:151-154 + paged_attn.py:65-85):
// RoPE in place on q and k with the roundings above, and in the same pass the rotated k row and
// the v row are scattered into the paged caches at slot_mapping[token] (layouts of
// csrc/cache_kernels.cu:184-192).  rot_dim == head_size, 16-bit types, 16-byte aligned rows.
// KV8: the caches hold fp8 (e4m3fn) with x = 16; the rotated key (already rounded to T, as the two
// separate launches would see it) and the value are quantised with fp8_kv_quant4 on the way in.
// SPLITK (the QKV projection of a step of 33..64 rows splits K over workgroups and leaves fp32 slabs, skinny_gemm.hip):
// q, k and v are not read from `query` / `key` / `value` but summed from the slabs -- partials[s][token][column], the
// projection's columns being [q | k | v] -- with the arithmetic of skinny_gemm_reduce_kernel (slab 0, then += slab 1 ...,
// + bias, rounded to T), and the three buffers receive what the reduce launch followed by this kernel's plain form
// would leave (rotated q and k, v as summed): one launch less per layer of a mixed step, bit for bit the same bytes.
struct RopeSplitK {
  const float* partials;   // [num_partials][num_tokens][row]  fp32
  int num_partials;
  int64_t partial_stride;  // floats between slabs
  int64_t row;             // floats per token row of a slab = (num_heads + 2 num_kv_heads) * head_size
  const void* bias;        // [row] of T or null
};

template <typename T, bool IS_NEOX, bool KV8, bool SPLITK = false>
__global__ void rotary_embedding_and_cache_kernel(
    const int64_t* __restrict__ positions, typename T::store_t* __restrict__ query,
    typename T::store_t* __restrict__ key, const typename T::store_t* __restrict__ value,
    const typename T::store_t* __restrict__ cos_sin_cache, void* __restrict__ key_cache_v,
    void* __restrict__ value_cache_v, const int64_t* __restrict__ slot_mapping,
    const int64_t query_stride, const int64_t key_stride, const int64_t value_stride, const int num_heads,
    const int num_kv_heads, const int head_size, const int block_size, const float k_scale, const float v_scale,
    const int64_t num_slots, const int64_t block_stride,  // block_stride: cache elements between blocks
    const RopeSplitK sk = RopeSplitK{}) {
  LVLLM_TRACE_BEGIN();
  using S = typename T::store_t;
  using V = Vec16<T>;
  constexpr int N = V::N;  // 8
  // 8 values of this token starting at column `col` of the projection: from memory, or summed from the slabs
  auto load8 = [&](const S* src, const int64_t col) -> V {
    if constexpr (!SPLITK) {
      (void)col;
      return *reinterpret_cast<const V*>(src);
    } else {
      (void)src;
      const float* p = sk.partials + (int64_t)blockIdx.x * sk.row + col;
      float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
      for (int sp = 1; sp < sk.num_partials; ++sp) {
        const float4* q4 = reinterpret_cast<const float4*>(p + sp * sk.partial_stride);
        const float4 c = q4[0], d = q4[1];
        a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
        b.x += d.x; b.y += d.y; b.z += d.z; b.w += d.w;
      }
      float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      V out;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (sk.bias != nullptr) f[e] += T::to_float(reinterpret_cast<const S*>(sk.bias)[col + e]);
        out.v[e] = T::from_float(f[e]);
      }
      return out;
    }
  };
  const int64_t k_col0 = (int64_t)num_heads * head_size, v_col0 = k_col0 + (int64_t)num_kv_heads * head_size;
  S* key_cache = reinterpret_cast<S*>(key_cache_v);
  S* value_cache = reinterpret_cast<S*>(value_cache_v);
  uint8_t* key_cache8 = reinterpret_cast<uint8_t*>(key_cache_v);
  uint8_t* value_cache8 = reinterpret_cast<uint8_t*>(value_cache_v);
  const int64_t token = blockIdx.x;
  const int64_t pos = positions[token];
  const int embed_dim = head_size / 2;
  const S* cos_ptr = cos_sin_cache + pos * head_size;
  const S* sin_ptr = cos_ptr + embed_dim;
  int64_t slot = slot_mapping[token];
  if (slot >= num_slots) slot = -1;  // outside the extent the caller stated: skipped like a padding slot
  const int64_t block_idx = slot >= 0 ? slot / block_size : 0;
  const int64_t block_off = slot >= 0 ? slot % block_size : 0;
  // 8 elements of head `head` starting at d (a multiple of 8) -> their 8 bytes inside the x = 16 chunk
  auto store_k8 = [&](const V& x, const int head, const int d) {
    uint8_t* dst = key_cache8 + block_idx * block_stride +
                   ((int64_t)head * (head_size / 16) + d / 16) * (int64_t)block_size * 16 + block_off * 16 + (d % 16);
    uint2 q;
    q.x = fp8_kv_quant4(T::to_float(x.v[0]), T::to_float(x.v[1]), T::to_float(x.v[2]), T::to_float(x.v[3]), k_scale);
    q.y = fp8_kv_quant4(T::to_float(x.v[4]), T::to_float(x.v[5]), T::to_float(x.v[6]), T::to_float(x.v[7]), k_scale);
    *reinterpret_cast<uint2*>(dst) = q;
  };
  const int cph = head_size / N;  // 16-byte chunks per head
  // work items: rotation units of q heads, then of k heads, then v chunks
  const int rot_units = IS_NEOX ? embed_dim / N : cph;  // per head
  const int nq = num_heads * rot_units, nk = num_kv_heads * rot_units, nv = num_kv_heads * cph;
  for (int i = threadIdx.x; i < nq + nk + nv; i += blockDim.x) {
    if (i < nq + nk) {
      const bool is_k = i >= nq;
      const int j = is_k ? i - nq : i;
      const int head = j / rot_units, u = j - head * rot_units;
      S* base = is_k ? key + token * key_stride + (int64_t)head * head_size
                     : query + token * query_stride + (int64_t)head * head_size;
      S* kc = key_cache + block_idx * block_stride + ((int64_t)head * cph) * (int64_t)block_size * N + block_off * N;
      const int64_t col = (is_k ? k_col0 : 0) + (int64_t)head * head_size;  // first column of this head
      if constexpr (IS_NEOX) {
        V x = load8(base + u * N, col + u * N);
        V y = load8(base + embed_dim + u * N, col + embed_dim + u * N);
        const V c = *reinterpret_cast<const V*>(cos_ptr + u * N);
        const V sn = *reinterpret_cast<const V*>(sin_ptr + u * N);
#pragma unroll
        for (int e = 0; e < N; ++e) rotate<T>(x.v[e], y.v[e], c.v[e], sn.v[e]);
        *reinterpret_cast<V*>(base + u * N) = x;
        *reinterpret_cast<V*>(base + embed_dim + u * N) = y;
        if (is_k && slot >= 0) {  // chunk d8 = u holds x', chunk d8 = D/16 + u holds y'
          if constexpr (KV8) {
            store_k8(x, head, u * N);
            store_k8(y, head, embed_dim + u * N);
          } else {
            *reinterpret_cast<V*>(kc + (int64_t)u * block_size * N) = x;
            *reinterpret_cast<V*>(kc + (int64_t)(embed_dim / N + u) * block_size * N) = y;
          }
        }
      } else {
        V xy = load8(base + u * N, col + u * N);
#pragma unroll
        for (int e = 0; e < N / 2; ++e) {
          const int r = u * (N / 2) + e;
          rotate<T>(xy.v[2 * e], xy.v[2 * e + 1], cos_ptr[r], sin_ptr[r]);
        }
        *reinterpret_cast<V*>(base + u * N) = xy;
        if (is_k && slot >= 0) {
          if constexpr (KV8) store_k8(xy, head, u * N);
          else *reinterpret_cast<V*>(kc + (int64_t)u * block_size * N) = xy;
        }
      }
    } else if (slot >= 0 || SPLITK) {
      const int j = i - nq - nk;
      const int head = j / cph, ch = j - head * cph;
      const S* vsrc = value + token * value_stride + (int64_t)head * head_size + ch * N;
      const V v = load8(vsrc, v_col0 + (int64_t)head * head_size + ch * N);
      if constexpr (SPLITK) {
        *reinterpret_cast<V*>(const_cast<S*>(vsrc)) = v;  // the row the reduce launch would have written
        if (slot < 0) continue;
      }
      if constexpr (KV8) {
        uint8_t* vdst = value_cache8 + block_idx * block_stride +
                        ((int64_t)head * head_size + ch * N) * (int64_t)block_size + block_off;
#pragma unroll
        for (int e = 0; e < N; e += 4) {
          const uint32_t w = fp8_kv_quant4(T::to_float(v.v[e]), T::to_float(v.v[e + 1]), T::to_float(v.v[e + 2]),
                                           T::to_float(v.v[e + 3]), v_scale);
          vdst[(int64_t)(e + 0) * block_size] = (uint8_t)w;
          vdst[(int64_t)(e + 1) * block_size] = (uint8_t)(w >> 8);
          vdst[(int64_t)(e + 2) * block_size] = (uint8_t)(w >> 16);
          vdst[(int64_t)(e + 3) * block_size] = (uint8_t)(w >> 24);
        }
      } else {
        S* vdst = value_cache + block_idx * block_stride + ((int64_t)head * head_size + ch * N) * (int64_t)block_size +
                  block_off;
#pragma unroll
        for (int e = 0; e < N; ++e) vdst[(int64_t)e * block_size] = v.v[e];
      }
    }
  }
  LVLLM_TRACE_END(5);
}

template <typename T>
static int launch_rope(const int64_t* positions, void* query, void* key, int num_tokens,
                       int num_heads, int num_kv_heads, int head_size, int rot_dim,
                       int64_t query_stride, int64_t key_stride, const void* cache, int is_neox,
                       hipStream_t stream) {
  using S = typename T::store_t;
  constexpr int N = Vec16<T>::N;
  const int esize = sizeof(S);
  const int embed_dim = rot_dim / 2;
  const bool ptr_ok = (((uintptr_t)query | (uintptr_t)key | (uintptr_t)cache) & 15) == 0 &&
                      (query_stride * esize) % 16 == 0 && (key_stride * esize) % 16 == 0 &&
                      (head_size * esize) % 16 == 0;
  const bool vec = ptr_ok && (is_neox ? (embed_dim % N == 0) : (rot_dim % N == 0));
  const int work = (num_heads + num_kv_heads) * (vec ? (is_neox ? embed_dim / N : rot_dim / N) : embed_dim);
  int threads = ((work + 63) / 64) * 64;
  threads = threads > 512 ? 512 : (threads < 64 ? 64 : threads);
#define LV_ROPE(NEOX, VEC)                                                                   \
  hipLaunchKernelGGL((rotary_embedding_kernel<T, NEOX, VEC>), dim3(num_tokens), dim3(threads), \
                     0, stream, positions, (S*)query, (S*)key, (const S*)cache, rot_dim,     \
                     query_stride, key_stride, num_heads, num_kv_heads, head_size)
  if (is_neox) {
    if (vec) LV_ROPE(true, true); else LV_ROPE(true, false);
  } else {
    if (vec) LV_ROPE(false, true); else LV_ROPE(false, false);
  }
#undef LV_ROPE
  return 0;
}

}  // namespace lvllm

using namespace lvllm;

extern "C" int lvllm_rotary_embedding(const int64_t* positions, void* query, void* key,
                                      int num_tokens, int num_heads, int num_kv_heads,
                                      int head_size, int rot_dim, int64_t query_stride,
                                      int64_t key_stride, const void* cos_sin_cache, int is_neox,
                                      int dtype, void* stream) {
  if (num_tokens == 0) return 0;
  LV_CHECK(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "bad rot_dim");
  LV_DISPATCH_DTYPE(dtype, (launch_rope<scalar_t>(positions, query, key, num_tokens, num_heads,
                                                  num_kv_heads, head_size, rot_dim, query_stride,
                                                  key_stride, cos_sin_cache, is_neox,
                                                  (hipStream_t)stream)));
  LV_LAUNCH_CHECK();
  return 0;
}

// Extension: RoPE (rot_dim == head_size) + paged-cache write of the rotated key and the value in
// one launch.  Returns 3 when the arguments are outside the fused kernel's envelope (the caller
// then runs lvllm_rotary_embedding and lvllm_reshape_and_cache).
extern "C" int lvllm_rotary_embedding_and_cache(
    const int64_t* positions, void* query, void* key, const void* value, int num_tokens, int num_heads,
    int num_kv_heads, int head_size, int rot_dim, int64_t query_stride, int64_t key_stride,
    int64_t value_stride, const void* cos_sin_cache, int is_neox, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int block_size, int dtype, void* stream) {
  return lvllm_rotary_embedding_and_cache_ex(positions, query, key, value, num_tokens, num_heads, num_kv_heads,
                                             head_size, rot_dim, query_stride, key_stride, value_stride,
                                             cos_sin_cache, is_neox, key_cache, value_cache, slot_mapping,
                                             block_size, dtype, LVLLM_KV_AUTO, 1.f, 1.f, 0, stream);
}

extern "C" int lvllm_rotary_embedding_and_cache_ex(
    const int64_t* positions, void* query, void* key, const void* value, int num_tokens, int num_heads,
    int num_kv_heads, int head_size, int rot_dim, int64_t query_stride, int64_t key_stride,
    int64_t value_stride, const void* cos_sin_cache, int is_neox, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int block_size, int dtype, int kv_dtype, float k_scale, float v_scale,
    int64_t kv_cache_bytes, void* stream) {
  return lvllm_rotary_embedding_and_cache_strided(positions, query, key, value, num_tokens, num_heads, num_kv_heads,
                                                  head_size, rot_dim, query_stride, key_stride, value_stride,
                                                  cos_sin_cache, is_neox, key_cache, value_cache, slot_mapping,
                                                  block_size, dtype, kv_dtype, k_scale, v_scale, kv_cache_bytes,
                                                  (int64_t)num_kv_heads * head_size * block_size, stream);
}

extern "C" int lvllm_rotary_embedding_and_cache_splitk(
    const int64_t* positions, void* qkv, const float* partials, int num_partials, const void* bias, int num_tokens,
    int num_heads, int num_kv_heads, int head_size, int rot_dim, const void* cos_sin_cache, int is_neox,
    void* key_cache, void* value_cache, const int64_t* slot_mapping, int block_size, int dtype, int kv_dtype,
    float k_scale, float v_scale, int64_t kv_cache_bytes, int64_t kv_block_stride, void* stream);

static int rope_cache_launch(
    const int64_t* positions, void* query, void* key, const void* value, int num_tokens, int num_heads,
    int num_kv_heads, int head_size, int rot_dim, int64_t query_stride, int64_t key_stride,
    int64_t value_stride, const void* cos_sin_cache, int is_neox, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int block_size, int dtype, int kv_dtype, float k_scale, float v_scale,
    int64_t kv_cache_bytes, int64_t kv_block_stride, void* stream, const RopeSplitK* sk);

extern "C" int lvllm_rotary_embedding_and_cache_strided(
    const int64_t* positions, void* query, void* key, const void* value, int num_tokens, int num_heads,
    int num_kv_heads, int head_size, int rot_dim, int64_t query_stride, int64_t key_stride,
    int64_t value_stride, const void* cos_sin_cache, int is_neox, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int block_size, int dtype, int kv_dtype, float k_scale, float v_scale,
    int64_t kv_cache_bytes, int64_t kv_block_stride, void* stream) {
  return rope_cache_launch(positions, query, key, value, num_tokens, num_heads, num_kv_heads, head_size, rot_dim,
                           query_stride, key_stride, value_stride, cos_sin_cache, is_neox, key_cache, value_cache,
                           slot_mapping, block_size, dtype, kv_dtype, k_scale, v_scale, kv_cache_bytes, kv_block_stride,
                           stream, nullptr);
}

// The QKV projection's reduce pass + rotary_embedding + reshape_and_cache in ONE launch (include/lvllm_hip.h).
extern "C" int lvllm_rotary_embedding_and_cache_splitk(
    const int64_t* positions, void* qkv, const float* partials, int num_partials, const void* bias, int num_tokens,
    int num_heads, int num_kv_heads, int head_size, int rot_dim, const void* cos_sin_cache, int is_neox,
    void* key_cache, void* value_cache, const int64_t* slot_mapping, int block_size, int dtype, int kv_dtype,
    float k_scale, float v_scale, int64_t kv_cache_bytes, int64_t kv_block_stride, void* stream) {
  LV_CHECK(partials != nullptr && num_partials >= 1 && num_partials <= 16, "partials: 1..16 fp32 slabs");
  LV_CHECK((((uintptr_t)partials) & 15) == 0, "partials must be 16-byte aligned");
  const int64_t row = (int64_t)(num_heads + 2 * num_kv_heads) * head_size;
  const int esz = 2;
  char* base = (char*)qkv;
  RopeSplitK sk{partials, num_partials, (int64_t)num_tokens * row, row, bias};
  return rope_cache_launch(positions, base, base + (int64_t)num_heads * head_size * esz,
                           base + (int64_t)(num_heads + num_kv_heads) * head_size * esz, num_tokens, num_heads,
                           num_kv_heads, head_size, rot_dim, row, row, row, cos_sin_cache, is_neox, key_cache, value_cache,
                           slot_mapping, block_size, dtype, kv_dtype, k_scale, v_scale, kv_cache_bytes, kv_block_stride,
                           stream, &sk);
}

static int rope_cache_launch(
    const int64_t* positions, void* query, void* key, const void* value, int num_tokens, int num_heads,
    int num_kv_heads, int head_size, int rot_dim, int64_t query_stride, int64_t key_stride,
    int64_t value_stride, const void* cos_sin_cache, int is_neox, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int block_size, int dtype, int kv_dtype, float k_scale, float v_scale,
    int64_t kv_cache_bytes, int64_t kv_block_stride, void* stream, const RopeSplitK* sk) {
  if (num_tokens == 0) return 0;
  LV_CHECK(kv_block_stride >= (int64_t)num_kv_heads * head_size * block_size &&
               (kv_block_stride * (kv_dtype == LVLLM_KV_FP8_E4M3 ? 1 : 2)) % 16 == 0,
           "kv_block_stride: at least one block, a multiple of 16 bytes");
  LV_CHECK(kv_dtype == LVLLM_KV_AUTO || kv_dtype == LVLLM_KV_FP8_E4M3, "unsupported kv_cache_dtype");
  const bool kv8 = kv_dtype == LVLLM_KV_FP8_E4M3;
  LV_CHECK(kv8 ? (k_scale > 0.f && v_scale > 0.f) : (k_scale == 1.f && v_scale == 1.f),
           "k_scale / v_scale: positive with an fp8 cache, 1.0 otherwise");
  const bool ok = (dtype == LVLLM_BF16 || dtype == LVLLM_F16) && rot_dim == head_size && head_size % 16 == 0 &&
                  (((uintptr_t)query | (uintptr_t)key | (uintptr_t)value | (uintptr_t)cos_sin_cache |
                    (uintptr_t)key_cache | (uintptr_t)value_cache) & 15) == 0 &&
                  (query_stride % 8) == 0 && (key_stride % 8) == 0 && (value_stride % 8) == 0;
  if (!ok) {
    set_error("lvllm_rotary_embedding_and_cache: arguments outside the fused kernel's envelope");
    return 3;
  }
  const int64_t num_slots = kv_cache_bytes > 0 ? kv_cache_bytes / (kv_block_stride * (kv8 ? 1 : 2)) * block_size
                                               : INT64_MAX;
  const int units = (num_heads + num_kv_heads) * (is_neox ? head_size / 16 : head_size / 8) + num_kv_heads * head_size / 8;
  int threads = ((units + 63) / 64) * 64;
  threads = threads > 512 ? 512 : threads;
#define LV_RC1(T_, NEOX_, KV8_, SK_)                                                                         \
  hipLaunchKernelGGL((rotary_embedding_and_cache_kernel<T_, NEOX_, KV8_, SK_>), dim3(num_tokens), dim3(threads), 0, \
                     (hipStream_t)stream, positions, (uint16_t*)query, (uint16_t*)key, (const uint16_t*)value, \
                     (const uint16_t*)cos_sin_cache, key_cache, value_cache, slot_mapping, query_stride,      \
                     key_stride, value_stride, num_heads, num_kv_heads, head_size, block_size, k_scale, v_scale, \
                     num_slots, kv_block_stride, sk != nullptr ? *sk : RopeSplitK{})
#define LV_RC(T_, NEOX_, KV8_)                                   \
  do {                                                           \
    if (sk != nullptr) LV_RC1(T_, NEOX_, KV8_, true); else LV_RC1(T_, NEOX_, KV8_, false); \
  } while (0)
#define LV_RC_N(T_)                                   \
  do {                                                \
    if (is_neox) {                                    \
      if (kv8) LV_RC(T_, true, true); else LV_RC(T_, true, false);   \
    } else {                                          \
      if (kv8) LV_RC(T_, false, true); else LV_RC(T_, false, false); \
    }                                                 \
  } while (0)
  if (dtype == LVLLM_BF16) LV_RC_N(BF16); else LV_RC_N(F16);
#undef LV_RC_N
#undef LV_RC
#undef LV_RC1
  LV_LAUNCH_CHECK();
  return 0;
}

LVLLM_TRACE_READER(lvllm_trace_read_rope)
