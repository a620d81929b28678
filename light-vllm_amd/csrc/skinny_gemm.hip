// Weight-streaming GEMM for decode batches on gfx950:  Y[M,N] = X[M,K] . W[N,K]^T (+ bias),
// M <= 64 rows (one decode step's tokens), bf16/f16 in, fp32 accumulate.
//
// At M <= 64 a projection is a stream of W through the chip exactly once: the roofline is HBM
// (2 bytes per weight), not MFMA.  hipBLASLt's M=32 kernels reach 2.5-4.4 TB/s on the shapes of
// an 8B model (profiles/r01_bench_kernel_stats.csv); this kernel is organised like the
// paged-attention kernel instead:
//   * the activations live in REGISTERS: wave w of a workgroup owns a fixed K slice (up to 16
//     k-steps of 32) and keeps its X fragments (B operand of v_mfma_f32_16x16x32) for the whole
//     launch; they are read once per workgroup from L2, in row order, and transposed into
//     fragment order through a wave-private LDS scratch (see the kernel);
//   * W goes HBM -> VGPR -> MFMA A operand with no LDS staging: lane (g, c) of a wave loads the
//     16 bytes W[n0 + c][k0 + 8g ..], which is its A fragment; 8 such loads per unit, two units in
//     flight per wave, 8 waves per CU;
//   * the 8 waves of a workgroup split K; their 16x(16*MT) partial tiles meet in LDS
//     (double-buffered, one barrier per n-tile) and MT waves write the bf16 result;
//   * K > 4096 is split over workgroups as well (blockIdx.y); partials go to an fp32 workspace and
//     a small second kernel adds them (and the bias).
// Loads outside the problem (k-steps past K, tiles past N) use an out-of-range buffer offset and
// return zeros without touching memory, so the loop is branch-free.
// Epilogues (act): 0 plain (+ bias); 2 SwiGLU over (gate, up) tile pairs -> [M, N/2]; 3 greedy arg-max
// over N -> int64 [M] (no logits written); act = 1 applies SwiGLU while loading X (kept for small shapes).
// The same file holds the W8A8 variant (fp8 weights, X quantised while the fragments are built) and,
// at the end, the 65..256-row kernel that moves X through LDS instead of registers.
//
// Weight layouts.  With row-major W[N,K] a wave-wide fragment load touches 16 rows x 64 bytes
// (half cache lines): correct, but the texture addresser works twice per byte and the stream tops
// out near 4 TB/s.  Weights are static, so they can be PACKED once at load time into the order the
// MFMA wants -- [N/16][K/32][g = 4][c = 16][8 elements], i.e. one contiguous 1 KiB per (n-tile,
// k-step), the same shape as a paged K-cache tile -- and then every wave load is one contiguous
// KiB (lvllm_pack_weight / `packed` below).
#include <stdlib.h>

#include "common.h"
#include "trace.h"

namespace lvllm {

typedef __bf16 g_bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 g_f16x8_t __attribute__((ext_vector_type(8)));
typedef float g_f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int g_u32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ g_f32x4_t gemm_mfma(g_u32x4_t a, g_u32x4_t b, g_f32x4_t c);
template <>
__device__ __forceinline__ g_f32x4_t gemm_mfma<BF16>(g_u32x4_t a, g_u32x4_t b, g_f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(g_bf16x8_t, a),
                                                 __builtin_bit_cast(g_bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ g_f32x4_t gemm_mfma<F16>(g_u32x4_t a, g_u32x4_t b, g_f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(g_f16x8_t, a),
                                                __builtin_bit_cast(g_f16x8_t, b), c, 0, 0, 0);
}

// fp8 x fp8 (OCP e4m3fn), K = 32 per instruction, 8 bytes per lane and operand
__device__ __forceinline__ g_f32x4_t gemm_mfma_fp8(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, g_f32x4_t c) {
  const long a = (long)(((unsigned long)a1 << 32) | a0), b = (long)(((unsigned long)b1 << 32) | b0);
  return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0);
}

constexpr int kGemmWaves = 8;
// diagnosis builds (tools/ab_gemm_tile_pad.py): bytes between the 16-row tiles of a packed weight beyond their size
#ifndef LVLLM_GEMM_TILE_PAD
#define LVLLM_GEMM_TILE_PAD 0
#endif
#ifdef LVLLM_GEMM_TRACE  // diagnosis build only (tools/trace_gemm.py): per-workgroup phase timestamps
__device__ unsigned long long g_gemm_trace[8 * 4096];
#define GEMM_TRACE(p)                                                                     \
  do {                                                                                    \
    if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.x < 1024)                  \
      g_gemm_trace[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 4 + (p)] = wall_clock64();     \
  } while (0)
#else
#define GEMM_TRACE(p)
#endif
constexpr unsigned kOutOfRange = 0xfffffff0u;  // >= any descriptor size: load returns 0
#ifndef LVLLM_GEMM_AUX
#define LVLLM_GEMM_AUX 2  // nt: weights are read once per launch
#endif
#ifndef LVLLM_GEMM_NT
#define LVLLM_GEMM_NT 2   // partial slabs (n-tiles x m-tiles) accumulated between two wave meetings
#endif

// W8 (W8A8): `w` holds fp8 (OCP e4m3fn) weights packed as if [N, K] bytes were [N, K/2] 16-bit
// elements, so every address below is unchanged with K := K/2; a 16-byte fragment is then 16
// consecutive k of one row and feeds TWO v_mfma_f32_16x16x32_fp8_fp8 (bytes 0-7, bytes 8-15; the
// contraction only needs W and X to agree on the order).  X arrives in T and is quantised while
// the fragments are built, with the arithmetic of static_scaled_fp8_quant (fp8_quant.hip):
// fp8(clamp(x * (1 / *x_scale))).  The result is T(acc * (*x_scale * *w_scale) + bias): what
// torch._scaled_mm(fp8(x), fp8(w), scale_a, scale_b) computes (w8a8_utils.py:147-156).
// XQ (W8 only): X arrives ALREADY quantised -- fp8 [M, K] bytes, rows ldx bytes apart, written by the producer (a norm
// kernel's fp8 twin, the SwiGLU epilogue below) with the same arithmetic this kernel's own prologue uses.  The
// prologue then only moves bytes (row order -> LDS transpose -> fragments): every one of the 128-256 workgroups of a
// W8A8 launch used to re-quantise all of X (1 280 vector instructions per wave at K = 4 096: 13 us for a projection
// whose 25 MB of weights stream in 5), profiles/r03_tuning.md section 5.
// y8 / y8_scale (SwiGLU epilogue): the activation leaves as fp8 for the down projection (and as T when y != null).
template <typename T, int MT, int KSTEPS, bool PACKED, int NT, bool W8, bool XQ = false>
__global__ __launch_bounds__(kGemmWaves * 64, 2) void skinny_gemm_kernel(
    typename T::store_t* __restrict__ y,   // [M, N]            (ksplit == 1)
    float* __restrict__ partial,           // [ksplit, M, N]    (ksplit > 1)
    const typename T::store_t* __restrict__ x, const typename T::store_t* __restrict__ w,
    const typename T::store_t* __restrict__ bias, const int M, const int N, const int K,
    const int64_t ldx, const int steps_per_wave, const int ntiles, const int act, const int stage_tiles,
    const float* __restrict__ x_scale, const float* __restrict__ w_scale, uint8_t* __restrict__ y8 = nullptr,
    const float* __restrict__ y8_scale = nullptr) {
  using S = typename T::store_t;
  static_assert(!XQ || (W8 && KSTEPS == 8 && MT <= 2), "pre-quantised activations: the W8A8 variants of <= 32 rows");
  LVLLM_TRACE_BEGIN();
  GEMM_TRACE(0);
  const float out_scale = W8 ? x_scale[0] * w_scale[0] : 1.f;
  constexpr int HALF = KSTEPS / 2;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  g_f32x4_t* red = reinterpret_cast<g_f32x4_t*>(smem_raw);  // [2][waves][NT * MT][64]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int total_steps = K >> 5;
  const int step0 = (blockIdx.y * kGemmWaves + wave) * steps_per_wave;  // first k-step of this wave
  int nvalid = total_steps - step0;                                     // k-steps that exist
  nvalid = nvalid < 0 ? 0 : (nvalid > steps_per_wave ? steps_per_wave : nvalid);

  g_u32x4_t xf[MT][KSTEPS];  // filled after the first weight loads have been issued (below)
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)w, 0, (int)((int64_t)N * K * 2 + (PACKED ? (int64_t)(N / 16) * LVLLM_GEMM_TILE_PAD : 0)), 0x00020000);
  // byte offset of this lane inside a 16-row tile at k-step step0, the byte stride between
  // k-steps, and between tiles
  const unsigned lane_off = PACKED ? (unsigned)(lane * 16 + step0 * 1024)
                                   : (unsigned)(((int64_t)c * K + (int64_t)step0 * 32 + 8 * g) * 2);
  const unsigned step_stride = PACKED ? 1024u : 64u;
  const unsigned tile_stride = (unsigned)((int64_t)16 * K * 2) + (PACKED ? LVLLM_GEMM_TILE_PAD : 0);

  // Which n-tiles this workgroup owns.  Plain: tile blockIdx.x + i * gridDim.x for local index i.
  // glu (act == 2, W = [gate rows | up rows]): the workgroup owns PAIRS -- local tiles 2j and 2j+1
  // are the gate tile and the up tile of pair blockIdx.x + j * gridDim.x -- so that the epilogue
  // can apply silu(gate) * up before anything is written (no [M, N] intermediate, no second kernel).
  const bool glu = act == 2;
  const int nunits = glu ? ntiles >> 1 : ntiles;  // tiles or pairs handed out round-robin
  const int my_units = nunits > (int)blockIdx.x ? (nunits - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int my_tiles_ld = glu ? 2 * my_units : my_units;
  auto tile_of = [&](const int i) __attribute__((always_inline)) {
    return glu ? (int)(blockIdx.x + (i >> 1) * gridDim.x) + (i & 1) * nunits : (int)(blockIdx.x + i * gridDim.x);
  };
  // Loads that must return zeros (k-steps past K, tiles past the workgroup's last) get an offset of
  // ~0: out of the descriptor's range, no memory access, no branch.  The masks are wave-uniform
  // and passed through an empty asm: when the compiler can see the conditions it turns the
  // selects into control flow -- two arms of loads writing the same registers with
  // s_waitcnt vmcnt(0) between them, which left two loads in flight per wave instead of sixteen.
  auto opaque = [](unsigned v) __attribute__((always_inline)) {
    v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    asm("" : "+s"(v));
    return v;
  };
  unsigned kmask[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) kmask[s] = opaque(s < nvalid ? 0u : ~0u);
  auto load_unit = [&](g_u32x4_t (&a)[HALF], const int i, const int h) __attribute__((always_inline)) {
    const int t = tile_of(i);  // n-tile (N is a multiple of 16: valid as a whole)
    const unsigned tmask = opaque(i < my_tiles_ld ? 0u : ~0u);
    const unsigned base = lane_off + (unsigned)t * tile_stride;
#pragma unroll
    for (int s = 0; s < HALF; ++s) {
      const int ks = h * HALF + s;  // h: which half of the wave's k-steps
#if defined(LVLLM_GEMM_EXP) && LVLLM_GEMM_EXP == 2  // diagnosis: no weight stream
      const unsigned off = kOutOfRange;
#else
      const unsigned off = (base + (unsigned)ks * step_stride) | tmask | kmask[ks];
#endif
      a[s] = __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, LVLLM_GEMM_AUX);
    }
  };

  // NT n-tiles are accumulated before the waves meet: one barrier per NT tiles, and the
  // NT*MT partial slabs are summed by different waves in parallel
  g_f32x4_t acc[NT][MT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[j][mt] = g_f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int my_tiles = my_tiles_ld;
  const int ngroups = (my_tiles + NT - 1) / NT;

  // Output staging.  A global store inside the streaming loop costs far more than its bytes:
  // vmcnt retires in issue order, so every load issued after a store waits for that store's
  // acknowledgement (measured: 2.4 us per wave meeting, 12 % of the kernel).  The summed tiles are
  // therefore parked in LDS (fp32, 1 KiB per (tile, m-tile)) and written out in one burst when the
  // stage is full or the workgroup is done -- no stores while weights are streaming.
  g_f32x4_t* stage = red + (size_t)2 * kGemmWaves * NT * MT * 64;  // [stage_tiles][MT][64]
  int stage_base = 0;  // first local tile held in the stage

  auto store_slab = [&](const int i, const int mt, g_f32x4_t sum) __attribute__((always_inline)) {
    // lane (g, c): rows n = n0 + 4g + r of column m = 16 mt + c
    const int t = tile_of(i);
    const int n0 = 16 * t + 4 * g;
    const int m = mt * 16 + c;
    if (m < M && n0 < N) {
      if (partial != nullptr) {
        float* dst = partial + ((int64_t)blockIdx.y * M + m) * N + n0;
        *reinterpret_cast<g_f32x4_t*>(dst) = sum;
      } else {
        if constexpr (W8) sum *= out_scale;
        if (bias != nullptr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sum[r] += T::to_float(bias[n0 + r]);
        }
        uint2 o;
        o.x = (uint32_t)T::from_float(sum[0]) | ((uint32_t)T::from_float(sum[1]) << 16);
        o.y = (uint32_t)T::from_float(sum[2]) | ((uint32_t)T::from_float(sum[3]) << 16);
        *reinterpret_cast<uint2*>(y + (int64_t)m * N + n0) = o;
      }
    }
  };

  // glu: one output slab per (pair, m-tile): y[m, 16 p + 4 g + r] = T(T(silu(gate)) * up) with gate and up
  // rounded to T first -- the rounding points of the projection followed by silu_and_mul
  // (activation.hip), so the fused launch is bit-identical to the two it replaces
  auto store_pair = [&](const int i, const int mt, g_f32x4_t gs, g_f32x4_t us) __attribute__((always_inline)) {
    const int pr = blockIdx.x + (i >> 1) * gridDim.x;
    const int n0 = 16 * pr + 4 * g, m = mt * 16 + c;
    const int half_n = N >> 1;
    if (m < M && n0 < half_n) {
      if constexpr (W8) { gs *= out_scale; us *= out_scale; }
      if (bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gs[r] += T::to_float(bias[n0 + r]);
          us[r] += T::to_float(bias[half_n + n0 + r]);
        }
      }
      S o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gf = T::to_float(T::from_float(gs[r]));
        const S a = T::from_float(gf / (1.0f + expf(-gf)));
        o[r] = T::from_float(T::to_float(a) * T::to_float(T::from_float(us[r])));
      }
      if (y != nullptr) {
        uint2 ov;
        ov.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        ov.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        *reinterpret_cast<uint2*>(y + (int64_t)m * half_n + n0) = ov;
      }
      if (y8 != nullptr)  // static_scaled_fp8_quant of the activation as rounded to T
        *reinterpret_cast<uint32_t*>(y8 + (int64_t)m * half_n + n0) =
            fp8_act_quant4(T::to_float(o[0]), T::to_float(o[1]), T::to_float(o[2]), T::to_float(o[3]), 1.0f / y8_scale[0]);
    }
  };

  // act == 3 (greedy sampling fused into the lm_head projection): no logits are written.  Every lane
  // keeps the best (value rounded to T, row n) it has seen per m-tile in a wave-private LDS slot --
  // the scratch of the activation transpose, idle after the prologue -- the workgroup merges its
  // lanes at the end and writes one candidate per row of X to the workspace behind `y`
  // ([gridDim.x][M] {float value, int n}); skinny_argmax_reduce_kernel picks the winner.  Ties go to
  // the smaller n, as torch.argmax does.
  const bool amax = act == 3;
  float* amax_v = reinterpret_cast<float*>(reinterpret_cast<char*>(stage + (size_t)stage_tiles * MT * 64) +
                                           wave * (W8 ? 16 * 512 : 16 * HALF * 64));  // [MT][64]
  int* amax_i = reinterpret_cast<int*>(amax_v + MT * 64);
  auto amax_update = [&](const int i, const int mt, g_f32x4_t sum) __attribute__((always_inline)) {
    const int n0 = 16 * tile_of(i) + 4 * g;
    if (mt * 16 + c < M && n0 < N) {
      if constexpr (W8) sum *= out_scale;
      float bv = amax_v[mt * 64 + lane];
      int bi = amax_i[mt * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = T::to_float(T::from_float(sum[r]));
        if (v > bv || (v == bv && n0 + r < bi)) {
          bv = v;
          bi = n0 + r;
        }
      }
      amax_v[mt * 64 + lane] = bv;
      amax_i[mt * 64 + lane] = bi;
    }
  };

  auto flush_stage = [&](const int end_tile) __attribute__((always_inline)) {
    __syncthreads();  // every reducer's slab is in the stage
    if (glu) {  // stage_base and end_tile are even: pairs are never split across two flushes
      const int nslabs = ((end_tile - stage_base) >> 1) * MT;
      for (int sl = wave; sl < nslabs; sl += kGemmWaves) {
        const int pl = sl / MT, mt = sl % MT;
        store_pair(stage_base + 2 * pl, mt, stage[((size_t)(2 * pl) * MT + mt) * 64 + lane],
                   stage[((size_t)(2 * pl + 1) * MT + mt) * 64 + lane]);
      }
    } else {
      const int nslabs = (end_tile - stage_base) * MT;
      for (int sl = wave; sl < nslabs; sl += kGemmWaves) {
        const int i = stage_base + sl / MT, mt = sl % MT;
        if (amax) amax_update(i, mt, stage[(size_t)sl * 64 + lane]);
        else store_slab(i, mt, stage[(size_t)sl * 64 + lane]);
      }
    }
    stage_base = end_tile;
    __syncthreads();  // the stage may be overwritten again
  };

  auto finish_group = [&](const int grp) __attribute__((always_inline)) {
#ifdef LVLLM_GEMM_NOREDUCE  // timing experiment only (wrong results): no cross-wave reduction
    if (wave == 0 && lane == 0 && acc[0][0][0] == 12345.f) y[grp] = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[j][mt] = g_f32x4_t{0.f, 0.f, 0.f, 0.f};
    return;
#endif
    if ((grp + 1) * NT - stage_base > stage_tiles) flush_stage(grp * NT);  // uniform: make room first
    // LDS buffer (grp & 1): [wave][slab = j * MT + mt][lane]
    g_f32x4_t* buf = red + (size_t)(grp & 1) * kGemmWaves * NT * MT * 64;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        buf[(wave * NT * MT + j * MT + mt) * 64 + lane] = acc[j][mt];
        acc[j][mt] = g_f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    __syncthreads();
    for (int slab = wave; slab < NT * MT; slab += kGemmWaves) {
      const int j = slab / MT, mt = slab - j * MT;
      const int i = grp * NT + j;
      if (i >= my_tiles) continue;
      g_f32x4_t sum = buf[(0 * NT * MT + slab) * 64 + lane];
#pragma unroll
      for (int w2 = 1; w2 < kGemmWaves; ++w2) {
        const g_f32x4_t v = buf[(w2 * NT * MT + slab) * 64 + lane];
        sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
      }
      stage[((size_t)(i - stage_base) * MT + mt) * 64 + lane] = sum;
    }
  };

  auto compute_unit = [&](const g_u32x4_t (&a)[HALF], const int h, g_f32x4_t (&ac)[MT])
                          __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < HALF; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const g_u32x4_t xb = h == 0 ? xf[mt][s] : xf[mt][HALF + s];
        if constexpr (W8) {
          ac[mt] = gemm_mfma_fp8(a[s].x, a[s].y, xb.x, xb.y, ac[mt]);
          ac[mt] = gemm_mfma_fp8(a[s].z, a[s].w, xb.z, xb.w, ac[mt]);
        } else {
          ac[mt] = gemm_mfma<T>(a[s], xb, ac[mt]);
        }
      }
  };

  // units: (n-tile, half of the wave's k-steps); two register sets.  Each set is refilled right
  // after it is consumed, so 8-16 KiB per wave stay in flight, 16 KiB across a wave meeting
  // (the meeting otherwise drains the memory pipeline: measured 2.4 us per meeting).
  g_u32x4_t a0[HALF], a1[HALF];
  // ---- X fragments: X[m = 16 mt + c][k = 32 (step0 + s) + 8 g ..], kept for the whole launch ----
  // Read in fragment order, lane (g, c) fetches 16 bytes of row c: a wave-wide load then touches 64
  // separate 16-byte pieces, the texture addresser takes ~64 clocks for it, and with 256 such
  // loads per workgroup the activations (256 KiB out of L2) cost as much as a 33 MB weight stream
  // (tools/trace_gemm.py: 6.5 us of a 15 us o_proj).  So the plain 16-bit path reads ROW order --
  // 64 lanes x 16 bytes = whole 256/512-byte row segments, all of a wave's loads in flight at once,
  // landing in the registers that will hold the fragments -- and turns each (m-tile, k-half) pass
  // into fragment order through a wave-private 4/8 KiB LDS scratch.  Chunk q of row r sits at
  // position q ^ r, so the 16 rows of a fragment read hit 16 different bank groups.
  bool staged = false;
  if constexpr (!W8) {
    if (act != 1) {
      staged = true;
      constexpr int NCH = HALF * 4;  // 16-byte chunks of one row in one pass
      constexpr int RPI = 64 / NCH;  // rows per wave-wide load
      __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
          (void*)x, 0, (int)(((int64_t)(M - 1) * ldx + K) * 2), 0x00020000);
      const int b = lane / NCH, pos = lane % NCH;
#if defined(LVLLM_GEMM_XORDER) && LVLLM_GEMM_XORDER == 1
      load_unit(a0, 0, 0);
      load_unit(a1, 0, 1);
#endif
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < HALF; ++i) {
            const int r = i * RPI + b, q = pos ^ r;
            const int ks = h * HALF + (q >> 2), m = mt * 16 + r;
            const unsigned off = (m < M && ks < nvalid)
                                     ? (unsigned)(((int64_t)m * ldx + (int64_t)(step0 + ks) * 32 + (q & 3) * 8) * 2)
                                     : kOutOfRange;
            xf[mt][h * HALF + i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
          }
#if !defined(LVLLM_GEMM_XORDER) || LVLLM_GEMM_XORDER == 0
      load_unit(a0, 0, 0);  // the first weights follow the activations into the queue
      load_unit(a1, 0, 1);
#endif
      char* xs = reinterpret_cast<char*>(stage + (size_t)stage_tiles * MT * 64) + wave * (16 * NCH * 16);
#ifdef LVLLM_GEMM_FAKE_NORM
      // Diagnosis build only (tools/diag_fake_norm.py; WRONG results): prices a [add + norm -> projection] fusion.
      // The SwiGLU projection normalises its own activations in the prologue -- T(T(x * s_row) * w[k]) on every
      // 16-byte chunk this wave holds, the norm weight read like the real one would be (16 bytes per lane at the
      // chunk's k) -- with the weight stream already requested, while the add + norm launch in front of it is skipped
      // by the caller.  What a real fusion would add on top (row statistics from the producer's partial sums) is not
      // included: the figure is an upper bound of the gain.
      if (glu) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < HALF; ++i) {
              const int r = i * RPI + b, q = pos ^ r;
              const int ks = h * HALF + (q >> 2), m = mt * 16 + r;
              const unsigned woff = ks < nvalid ? (unsigned)(((int64_t)(step0 + ks) * 32 + (q & 3) * 8) * 2) : kOutOfRange;
              const g_u32x4_t wv = __builtin_amdgcn_raw_buffer_load_b128(xr, woff, 0, 0);  // stand-in for the norm weight
              const float srow = 1.0f + 1e-6f * (float)m;
              g_u32x4_t v = xf[mt][h * HALF + i];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float a0f = T::to_float((S)(v[e] & 0xffffu)), a1f = T::to_float((S)(v[e] >> 16));
                const float w0f = T::to_float((S)(wv[e] & 0xffffu)), w1f = T::to_float((S)(wv[e] >> 16));
                const S n0 = T::from_float(a0f * srow), n1 = T::from_float(a1f * srow);
                const S o0 = T::from_float(T::to_float(n0) * w0f), o1 = T::from_float(T::to_float(n1) * w1f);
                v[e] = (uint32_t)o0 | ((uint32_t)o1 << 16);
              }
              xf[mt][h * HALF + i] = v;
            }
      }
#endif
#ifndef LVLLM_GEMM_NO_XPOSE  // diagnosis build (wrong results): what the LDS transpose of the activations costs
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < HALF; ++i)
            *reinterpret_cast<g_u32x4_t*>(xs + i * 1024 + lane * 16) = xf[mt][h * HALF + i];
#pragma unroll
          for (int s2 = 0; s2 < HALF; ++s2)
            xf[mt][h * HALF + s2] =
                *reinterpret_cast<const g_u32x4_t*>(xs + c * (NCH * 16) + (((4 * s2 + g) ^ c) * 16));
        }
#else
      (void)xs;
#endif
    }
  }
  if constexpr (XQ) {
    // fp8 activations: a pass is 16 rows x this wave's 8 k-steps = 512 bytes per row; 64 lanes x 16 bytes cover two
    // rows per load, all 8 MT loads in flight at once, then through the wave-private scratch (chunk q of row r at
    // position q ^ r) into fragment order: lane (g, c) reads chunk 4 s + g of row c = k-step s, k 16 g .. 16 g + 15
    staged = true;
    const uint8_t* x8 = reinterpret_cast<const uint8_t*>(x);
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)x8, 0, (int)((int64_t)(M - 1) * ldx + 2 * (int64_t)K), 0x00020000);
    const int b = lane >> 5, pos = lane & 31;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = 2 * i + b, q = pos ^ r;
        const int ks = q >> 2, m = mt * 16 + r;
        const unsigned off = (m < M && ks < nvalid)
                                 ? (unsigned)((int64_t)m * ldx + (int64_t)(step0 + ks) * 64 + (q & 3) * 16)
                                 : kOutOfRange;
        xf[mt][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
      }
    load_unit(a0, 0, 0);  // the first weights follow the activations into the queue
    load_unit(a1, 0, 1);
    char* xs = reinterpret_cast<char*>(stage + (size_t)stage_tiles * MT * 64) + wave * (16 * 512);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<g_u32x4_t*>(xs + i * 1024 + lane * 16) = xf[mt][i];
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2)
        xf[mt][s2] = *reinterpret_cast<const g_u32x4_t*>(xs + c * 512 + (((4 * s2 + g) ^ c) * 16));
    }
  }
  if constexpr (W8 && !XQ && MT * KSTEPS < 32) {
    // The same for W8A8 (variants whose fragments leave 64 registers for two landing sets; the
    // others -- K beyond 4096, more than 32 rows -- keep the fragment-order loads below): X arrives in T, 16 consecutive k (32 bytes) of a row per lane and k-step.
    // A pass is 16 rows x 4 k-steps (512 bytes per row: the geometry of the 16-bit path above), two
    // passes are in flight in two landing sets, and the values are quantised between the transpose
    // and the fragment registers with the arithmetic of static_scaled_fp8_quant.
    staged = true;
    constexpr int NPASS = MT * (KSTEPS / 4);
    static_assert(KSTEPS % 4 == 0, "passes are 4 k-steps");
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)x, 0, (int)(((int64_t)(M - 1) * ldx + 2 * (int64_t)K) * 2), 0x00020000);
    const int b = lane >> 5, pos = lane & 31;
    const float inv = 1.0f / x_scale[0];
    char* xs = reinterpret_cast<char*>(stage + (size_t)stage_tiles * MT * 64) + wave * (16 * 512);
    auto quant8 = [&](const g_u32x4_t v, uint32_t& lo, uint32_t& hi) __attribute__((always_inline)) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[2 * j] = fmaxf(-448.f, fminf(T::to_float((S)(v[j] & 0xffffu)) * inv, 448.f));
        f[2 * j + 1] = fmaxf(-448.f, fminf(T::to_float((S)(v[j] >> 16)) * inv, 448.f));
      }
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false), true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false), true);
    };
    // two landing sets of 8 loads (one pass each) are in flight; the first weights are requested
    // right behind the first activations
    auto stage_x = [&](auto& L0, auto& L1, const bool early_weights) __attribute__((always_inline)) {
      constexpr int LI = (int)(sizeof(L0) / sizeof(L0[0]));
      constexpr int SUB = 8 / LI;  // landing sets per pass
      constexpr int NUNITS = NPASS * SUB;
      auto issue = [&](const int u, auto& dst) __attribute__((always_inline)) {
        const int p = u / SUB, i0 = (u % SUB) * LI;
        const int mt = p / (KSTEPS / 4), sp = p % (KSTEPS / 4);
#pragma unroll
        for (int i = 0; i < LI; ++i) {
          const int r = 2 * (i0 + i) + b, q = pos ^ r;  // chunk q of the row's 512 bytes sits at position q ^ r
          const int ks = 4 * sp + (q >> 3), m = mt * 16 + r;
          const unsigned off = (m < M && ks < nvalid)
                                   ? (unsigned)(((int64_t)m * ldx + (int64_t)(step0 + ks) * 64 + (q & 7) * 8) * 2)
                                   : kOutOfRange;
          dst[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
        }
      };
      issue(0, L0);
      if constexpr (NUNITS > 1) issue(1, L1);
      if (early_weights) {
        load_unit(a0, 0, 0);
        load_unit(a1, 0, 1);
      }
#pragma unroll
      for (int u = 0; u < NUNITS; ++u) {
        const int p = u / SUB, i0 = (u % SUB) * LI;
        const int mt = p / (KSTEPS / 4), sp = p % (KSTEPS / 4);
#pragma unroll
        for (int i = 0; i < LI; ++i)
          *reinterpret_cast<g_u32x4_t*>(xs + (i0 + i) * 1024 + lane * 16) = (u & 1) ? L1[i] : L0[i];
        if (u + 2 < NUNITS) {
          if (u & 1) issue(u + 2, L1); else issue(u + 2, L0);
        }
        if ((u % SUB) == SUB - 1) {  // the pass's 16 rows are in the scratch
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            const int q0 = 8 * s2 + 2 * g;
            uint32_t o0, o1, o2, o3;
            const g_u32x4_t v0 = *reinterpret_cast<const g_u32x4_t*>(xs + c * 512 + ((q0 ^ c) * 16));
            quant8(v0, o0, o1);
            const g_u32x4_t v1 = *reinterpret_cast<const g_u32x4_t*>(xs + c * 512 + (((q0 + 1) ^ c) * 16));
            quant8(v1, o2, o3);
            xf[mt][4 * sp + s2] = g_u32x4_t{o0, o1, o2, o3};
          }
        }
      }
    };
    g_u32x4_t l0[8], l1[8];
    stage_x(l0, l1, true);
  }
  if (!staged) {
    load_unit(a0, 0, 0);
    load_unit(a1, 0, 1);
    // the first weight loads (HBM) are in flight before the activations (L2) are requested
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + c;
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        xf[mt][s] = g_u32x4_t{0, 0, 0, 0};
        if constexpr (W8) {
          if (m < M && s < nvalid) {
            // 16 consecutive k of row m: k = 64 (step0 + s) + 16 g ..  (K here is K/2, see above)
            const S* src = x + (int64_t)m * ldx + (int64_t)(step0 + s) * 64 + 16 * g;
            S e[16];
            *reinterpret_cast<g_u32x4_t*>(e) = *reinterpret_cast<const g_u32x4_t*>(src);
            *reinterpret_cast<g_u32x4_t*>(e + 8) = *reinterpret_cast<const g_u32x4_t*>(src + 8);
            const float inv = 1.0f / x_scale[0];
            uint32_t q[4];
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) {
              float f[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) f[r] = fmaxf(-448.f, fminf(T::to_float(e[4 * d4 + r]) * inv, 448.f));
              uint32_t wq = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
              q[d4] = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], wq, true);
            }
            xf[mt][s] = g_u32x4_t{q[0], q[1], q[2], q[3]};
          }
        } else if (m < M && s < nvalid) {
          const S* src = x + (int64_t)m * ldx + (int64_t)(step0 + s) * 32 + 8 * g;
          g_u32x4_t v = *reinterpret_cast<const g_u32x4_t*>(src);
          if (act == 1) {
            // fused SwiGLU gate: the row holds [gate (K) | up (K)]; X = T(T(silu(gate)) * up), the
            // roundings of silu_and_mul (activation.hip), so the fused path is bit-identical
            const g_u32x4_t u = *reinterpret_cast<const g_u32x4_t*>(src + K);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              uint32_t out = 0;
#pragma unroll
              for (int hlf = 0; hlf < 2; ++hlf) {
                const S gs = (S)((v[q] >> (16 * hlf)) & 0xffffu), us = (S)((u[q] >> (16 * hlf)) & 0xffffu);
                const float gf = T::to_float(gs);
                const S a = T::from_float(gf / (1.0f + expf(-gf)));
                out |= (uint32_t)T::from_float(T::to_float(a) * T::to_float(us)) << (16 * hlf);
              }
              v[q] = out;
            }
          }
          xf[mt][s] = v;
        }
      }
    }
  }

  if (amax) {  // this wave's transposes are done: its scratch becomes its arg-max slots
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      amax_v[mt * 64 + lane] = -__builtin_inff();
      amax_i[mt * 64 + lane] = 0x7fffffff;
    }
  }
#ifdef LVLLM_GEMM_TRACE
  __builtin_amdgcn_s_waitcnt(0);  // X and the first two units have arrived
  GEMM_TRACE(1);
#endif
  for (int grp = 0; grp < ngroups; ++grp) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int i = grp * NT + j;  // tiles past my_tiles load nothing (out-of-range offsets)
      compute_unit(a0, 0, acc[j]);
      load_unit(a0, i + 1, 0);
      compute_unit(a1, 1, acc[j]);
      load_unit(a1, i + 1, 1);
    }
    finish_group(grp);
  }
  GEMM_TRACE(2);
#ifndef LVLLM_GEMM_NOREDUCE
  flush_stage(my_tiles);  // all weights have been streamed: now the stores
#endif
  if (amax) {  // flush_stage ended with a barrier: every wave's slots are final
    const int m = threadIdx.x;
    if (m < M) {
      const char* region = reinterpret_cast<const char*>(stage + (size_t)stage_tiles * MT * 64);
      float bv = -__builtin_inff();
      int bi = 0x7fffffff;
      for (int w2 = 0; w2 < kGemmWaves; ++w2) {
        const float* v = reinterpret_cast<const float*>(region + w2 * (W8 ? 16 * 512 : 16 * HALF * 64));
        const int* ix = reinterpret_cast<const int*>(v + MT * 64);
#pragma unroll
        for (int g2 = 0; g2 < 4; ++g2) {
          const int e = (m >> 4) * 64 + g2 * 16 + (m & 15);
          if (v[e] > bv || (v[e] == bv && ix[e] < bi)) {
            bv = v[e];
            bi = ix[e];
          }
        }
      }
      float* dst = reinterpret_cast<float*>(y) + ((int64_t)blockIdx.x * M + m) * 2;
      dst[0] = bv;
      reinterpret_cast<int*>(dst)[1] = bi;
    }
  }
#ifdef LVLLM_GEMM_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  GEMM_TRACE(3);
#endif
  LVLLM_TRACE_END(1);
}

// out[m, n] = T(sum_s partial[s, m, n] + bias[n])
template <typename T>
__global__ void skinny_gemm_reduce_kernel(typename T::store_t* __restrict__ y,
                                          const float* __restrict__ partial,
                                          const typename T::store_t* __restrict__ bias,
                                          const int64_t MN, const int N, const int ksplit,
                                          const float* __restrict__ x_scale = nullptr,
                                          const float* __restrict__ w_scale = nullptr) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= MN) return;
  g_f32x4_t sum = *reinterpret_cast<const g_f32x4_t*>(partial + i);
  for (int s = 1; s < ksplit; ++s) {
    const g_f32x4_t v = *reinterpret_cast<const g_f32x4_t*>(partial + (int64_t)s * MN + i);
    sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
  }
  if (x_scale != nullptr) sum *= x_scale[0] * w_scale[0];  // W8A8: see skinny_gemm_kernel
  if (bias != nullptr) {
    const int n = (int)(i % N);
#pragma unroll
    for (int r = 0; r < 4; ++r) sum[r] += T::to_float(bias[n + r]);
  }
  uint2 o;
  o.x = (uint32_t)T::from_float(sum[0]) | ((uint32_t)T::from_float(sum[1]) << 16);
  o.y = (uint32_t)T::from_float(sum[2]) | ((uint32_t)T::from_float(sum[3]) << 16);
  *reinterpret_cast<uint2*>(y + i) = o;
}

// tokens[m] = the n of the best candidate over the workgroups' [groups][M] {value, n} (ties: smaller n)
__global__ void skinny_argmax_reduce_kernel(int64_t* __restrict__ tokens, const float* __restrict__ cand,
                                            const int groups, const int M) {
  const int m = blockIdx.x;
  __shared__ float sv[256];
  __shared__ int si[256];
  float bv = -__builtin_inff();
  int bi = 0x7fffffff;
  for (int gidx = threadIdx.x; gidx < groups; gidx += blockDim.x) {
    const float v = cand[((int64_t)gidx * M + m) * 2];
    const int ix = reinterpret_cast<const int*>(cand)[((int64_t)gidx * M + m) * 2 + 1];
    if (v > bv || (v == bv && ix < bi)) {
      bv = v;
      bi = ix;
    }
  }
  sv[threadIdx.x] = bv;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s2 = blockDim.x / 2; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) {
      const float v = sv[threadIdx.x + s2];
      const int ix = si[threadIdx.x + s2];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && ix < si[threadIdx.x])) {
        sv[threadIdx.x] = v;
        si[threadIdx.x] = ix;
      }
    }
    __syncthreads();
  }
  // a row of NaNs beats no candidate: answer token 0 rather than an index no embedding table has
  if (threadIdx.x == 0) tokens[m] = si[0] == 0x7fffffff ? 0 : si[0];
}

// out[m, n] = silu_and_mul of T(sum_s partial[s, m, n] + bias[n]) and T(sum_s partial[s, m, N/2 + n] + bias[N/2 + n]):
// the SwiGLU epilogue for shapes whose K is split over workgroups (roundings of the reduce kernel followed
// by silu_and_mul, so the result is bit-identical to the separate launches)
template <typename T>
__global__ void skinny_gemm_reduce_swiglu_kernel(typename T::store_t* __restrict__ y, const float* __restrict__ partial,
                                                 const typename T::store_t* __restrict__ bias, const int M,
                                                 const int N, const int ksplit) {
  using S = typename T::store_t;
  const int half_n = N >> 1;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;  // index into [M, N/2]
  if (i >= (int64_t)M * half_n) return;
  const int m = (int)(i / half_n), n = (int)(i - (int64_t)m * half_n);
  const int64_t MN = (int64_t)M * N;
  g_f32x4_t gs = g_f32x4_t{0.f, 0.f, 0.f, 0.f}, us = gs;
  for (int sp = 0; sp < ksplit; ++sp) {
    const float* row = partial + (int64_t)sp * MN + (int64_t)m * N;
    const g_f32x4_t a = *reinterpret_cast<const g_f32x4_t*>(row + n);
    const g_f32x4_t b = *reinterpret_cast<const g_f32x4_t*>(row + half_n + n);
    if (sp == 0) { gs = a; us = b; }
    else { gs[0] += a[0]; gs[1] += a[1]; gs[2] += a[2]; gs[3] += a[3]; us[0] += b[0]; us[1] += b[1]; us[2] += b[2]; us[3] += b[3]; }
  }
  if (bias != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gs[r] += T::to_float(bias[n + r]);
      us[r] += T::to_float(bias[half_n + n + r]);
    }
  }
  S o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float gf = T::to_float(T::from_float(gs[r]));
    const S a = T::from_float(gf / (1.0f + expf(-gf)));
    o[r] = T::from_float(T::to_float(a) * T::to_float(T::from_float(us[r])));
  }
  uint2 ov;
  ov.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
  ov.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
  *reinterpret_cast<uint2*>(y + i) = ov;
}

// The arg-max epilogue for shapes whose K is split over workgroups: block (b, m) sums the fp32 partials of
// its slice of row m, rounds to T as the reduce kernel would, and leaves one candidate; the kernel above
// picks the winner over the 64 slices.
template <typename T>
__global__ void skinny_gemm_reduce_argmax_kernel(float* __restrict__ cand, const float* __restrict__ partial,
                                                 const int M, const int N, const int ksplit) {
  const int m = blockIdx.y, nb = gridDim.x;
  const int per = ((N / 4 + nb - 1) / nb) * 4;  // columns of a slice, a multiple of 4
  const int n_begin = blockIdx.x * per, n_end = n_begin + per < N ? n_begin + per : N;
  const int64_t MN = (int64_t)M * N;
  __shared__ float sv[256];
  __shared__ int si[256];
  float bv = -__builtin_inff();
  int bi = 0x7fffffff;
  for (int n = n_begin + 4 * threadIdx.x; n < n_end; n += 4 * blockDim.x) {
    g_f32x4_t sum = *reinterpret_cast<const g_f32x4_t*>(partial + (int64_t)m * N + n);
    for (int sp = 1; sp < ksplit; ++sp) {
      const g_f32x4_t v = *reinterpret_cast<const g_f32x4_t*>(partial + (int64_t)sp * MN + (int64_t)m * N + n);
      sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = T::to_float(T::from_float(sum[r]));
      if (v > bv || (v == bv && n + r < bi)) {
        bv = v;
        bi = n + r;
      }
    }
  }
  sv[threadIdx.x] = bv;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s2 = blockDim.x / 2; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) {
      const float v = sv[threadIdx.x + s2];
      const int ix = si[threadIdx.x + s2];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && ix < si[threadIdx.x])) {
        sv[threadIdx.x] = v;
        si[threadIdx.x] = ix;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* dst = cand + ((int64_t)blockIdx.x * M + m) * 2;
    dst[0] = sv[0];
    reinterpret_cast<int*>(dst)[1] = si[0];
  }
}

template <typename T, int MT, int KSTEPS, bool W8 = false, bool XQ = false>
static void launch_skinny(void* y, float* partial, const void* x, const void* w, const void* bias, int M,
                          int N, int K, int64_t ldx, int steps_per_wave, int ntiles, int groups, int ksplit,
                          bool packed, int act, hipStream_t stream, const float* x_scale = nullptr,
                          const float* w_scale = nullptr, uint8_t* y8 = nullptr, const float* y8_scale = nullptr) {
  using S = typename T::store_t;
  constexpr int NT = LVLLM_GEMM_NT / MT > 0 ? LVLLM_GEMM_NT / MT : 1;  // NT * MT slabs per meeting
  const size_t red_bytes = (size_t)2 * kGemmWaves * NT * MT * 64 * sizeof(g_f32x4_t);
  // output stage: as many tiles as the workgroup owns, capped by what is left of the 160 KiB LDS
  const bool glu = act == 2;  // tiles are handed out in (gate, up) pairs
  const int tiles_per_wg = glu ? 2 * ((ntiles / 2 + groups - 1) / groups) : (ntiles + groups - 1) / groups;
  int stage_tiles = ((tiles_per_wg + NT - 1) / NT) * NT;
  // wave-private scratch that turns row-order activation loads into fragments (16-bit path only)
  const size_t xs_bytes = W8 ? (MT * KSTEPS < 32 ? (size_t)kGemmWaves * 16 * 512 : 0)
                             : (size_t)kGemmWaves * 16 * (KSTEPS / 2) * 64;
  const int cap = (int)((160 * 1024 - red_bytes - xs_bytes) / ((size_t)MT * 1024) / NT) * NT;
  if (stage_tiles > cap) stage_tiles = cap;
  if (stage_tiles < NT) stage_tiles = NT;
  if (glu) stage_tiles = stage_tiles < 2 ? 2 : (stage_tiles & ~1);  // a flush never splits a pair
  const size_t smem = red_bytes + (size_t)stage_tiles * MT * 64 * sizeof(g_f32x4_t) + xs_bytes;
  auto go = [&](auto kern) {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kern, dim3(groups, ksplit), dim3(kGemmWaves * 64), smem, stream, (S*)y, partial,
                       (const S*)x, (const S*)w, (const S*)bias, M, N, K, ldx, steps_per_wave, ntiles, act,
                       stage_tiles, x_scale, w_scale, y8, y8_scale);
  };
  if constexpr (W8 && XQ) {
    go(skinny_gemm_kernel<T, MT, KSTEPS, true, NT, true, true>);
  } else if constexpr (W8) {
    go(skinny_gemm_kernel<T, MT, KSTEPS, true, NT, true>);  // fp8 weights are always packed
  } else {
    if (packed) go(skinny_gemm_kernel<T, MT, KSTEPS, true, NT, false>);
    else go(skinny_gemm_kernel<T, MT, KSTEPS, false, NT, false>);
  }
}

#ifdef LVLLM_GEMM_TRACE
extern "C" int lvllm_gemm_trace_read(void* host_dst, int nwords) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_gemm_trace), (size_t)nwords * 8);
}
#endif


// ---------------------------------------------------------------------------------------------
// Weight-streaming GEMM for 65..256 rows (a decode step of a large batch, a prefill chunk):
//   Y[M,N] = X[M,K] . W[N,K]^T (+ bias), W packed as above, bf16/f16, fp32 accumulate.
// With this many rows the activations no longer fit a wave's registers, so the roles are turned
// round: the 8 waves of a workgroup split N -- wave w owns n-tile 8 G + w of tile group G and walks
// the whole K range of the workgroup -- and X goes through LDS, shared by the 8 waves: a chunk of
// KC k-steps x (16 MT) rows (32 KiB at M <= 128) is copied L2 -> LDS by the DMA path (buffer_load ...
// lds, row order, XOR-swizzled like the scratch of the small kernel) into a ring of three chunks
// (two at M > 128), one barrier per chunk, counted vmcnt waits so that the next chunk stays in
// flight across the barrier.  W goes HBM -> VGPR -> MFMA A operand, 1 KiB per wave load, in step
// with the ring.  No
// cross-wave reduction.  Small N (qkv, o, down: 32-48 tile groups) would leave most CUs idle, so K is
// also split over workgroups (blockIdx.y) there; the fp32 partials are summed by the reduce kernel
// above.  One pass over W.  Measured against hipBLASLt at M = 128 (tools/bench_stream_gemm.py): down
// 37 vs 76 us, o 21 vs 24, qkv 26 vs 27, gate_up 59 vs 55 -- every wave reads every activation from
// LDS (8 bytes of LDS per byte of W), which is what bounds the wide shapes; a 32x32x16-MFMA variant
// (half the LDS reads, W fetched by two waves) was correct and slower (gate_up 80 us).  The engine
// uses this kernel where it wins: K >= 8192 (the down projection).
template <typename T, int MT, int KC, int NST>  // NST: chunks in the ring (2 or 3)
__global__ __launch_bounds__(kGemmWaves * 64, 1) void stream_gemm_kernel(
    typename T::store_t* __restrict__ y, float* __restrict__ partial,
    const typename T::store_t* __restrict__ x, const typename T::store_t* __restrict__ w,
    const typename T::store_t* __restrict__ bias, const int M, const int N, const int K, const int64_t ldx,
    const int ntiles, const int steps_per_split) {
  LVLLM_TRACE_BEGIN();
  constexpr int ROWS = 16 * MT;       // rows of X held in LDS (rows >= M are zero)
  constexpr int RB = KC * 64;         // bytes of one row in one chunk
  constexpr int NCH = KC * 4;         // 16-byte chunks per row and chunk
  constexpr int RPI = 1024 / RB;      // rows one wave-wide DMA covers
  constexpr int DMA = ROWS / RPI / kGemmWaves;  // DMA instructions per wave and chunk
  static_assert(NCH >= 16 && ROWS % (RPI * kGemmWaves) == 0, "chunk geometry");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // [2][ROWS][RB]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int total_steps = K >> 5;
  const int s_begin = blockIdx.y * steps_per_split;
  int s_end = s_begin + steps_per_split;
  if (s_end > total_steps) s_end = total_steps;
  const int nchunks = (s_end - s_begin + KC - 1) / KC;

  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((int64_t)N * K * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(((int64_t)(M - 1) * ldx + K) * 2), 0x00020000);
  const unsigned tile_stride = (unsigned)((int64_t)16 * K * 2);
  auto opaque = [](unsigned v) __attribute__((always_inline)) {
    v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    asm("" : "+s"(v));
    return v;
  };

  // X chunk ci -> LDS stage ci % NST.  Wave-wide copy j of this wave covers rows RPI (DMA wave + j) ..;
  // chunk q of row r lands at position q ^ (r & 15)
  const int xb = lane / NCH, xpos = lane % NCH;
  auto issue_x = [&](const int ci) __attribute__((always_inline)) {
    char* buf = smem_raw + (size_t)(ci % NST) * ROWS * RB;
    const int step0 = s_begin + ci * KC;
#pragma unroll
    for (int j = 0; j < DMA; ++j) {
      const int i = wave * DMA + j;
      const int r = i * RPI + xb, q = xpos ^ (r & 15);
      const int ks = step0 + (q >> 2);
      const unsigned off = (r < M && ks < s_end) ? (unsigned)(((int64_t)r * ldx + (int64_t)ks * 32 + (q & 3) * 8) * 2)
                                                 : kOutOfRange;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(buf + i * 1024), 16, off,
                                               0, 0, 0);
    }
  };
  auto issue_w = [&](g_u32x4_t (&a)[KC], const int t, const unsigned tmask, const int ci) __attribute__((always_inline)) {
    const int step0 = s_begin + ci * KC;
    const unsigned base = (unsigned)(lane * 16) + (unsigned)t * tile_stride;
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const unsigned kmask = opaque(step0 + s < s_end ? 0u : ~0u);
      a[s] = __builtin_amdgcn_raw_buffer_load_b128(wr, (base + (unsigned)(step0 + s) * 1024u) | tmask | kmask, 0,
                                                   LVLLM_GEMM_AUX);
    }
  };
  auto compute = [&](const g_u32x4_t (&a)[KC], g_f32x4_t (&acc)[MT], const int ci) __attribute__((always_inline)) {
    const char* buf = smem_raw + (size_t)(ci % NST) * ROWS * RB;
#pragma unroll
    for (int s = 0; s < KC; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const g_u32x4_t xb4 =
            *reinterpret_cast<const g_u32x4_t*>(buf + (16 * mt + c) * RB + (((4 * s + g) ^ c) * 16));
        acc[mt] = gemm_mfma<T>(a[s], xb4, acc[mt]);
      }
  };

  const int ngroups_total = (ntiles + kGemmWaves - 1) / kGemmWaves;
  for (int grp = blockIdx.x; grp < ngroups_total; grp += gridDim.x) {
    const int t = grp * kGemmWaves + wave;
    const unsigned tmask = opaque(t < ntiles ? 0u : ~0u);
    g_f32x4_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = g_f32x4_t{0.f, 0.f, 0.f, 0.f};
    // A ring of NST chunks: chunk ci is consumed while chunks ci+1 .. ci+NST-2 are in flight.  (With two
    // stages the wait for chunk ci+1 starts one chunk's MFMAs after its loads were issued: the memory
    // pipeline drains at every boundary, 3.9 us per 128 KiB chunk where the stream alone needs 2.)
    g_u32x4_t a[NST][KC];
    constexpr int kOps = DMA + KC;  // vector-memory operations one chunk costs a wave
    static_assert((NST - 2) * kOps < 64, "vmcnt range");
    __syncthreads();  // the previous group's readers are done with the LDS ring
#pragma unroll
    for (int k = 0; k < NST - 1; ++k)
      if (k < nchunks) {
        issue_x(k);
        issue_w(a[k], t, tmask, k);
      }
    auto step = [&](const int ci, const g_u32x4_t (&cur)[KC], g_u32x4_t (&nxt)[KC]) __attribute__((always_inline)) {
      // vmcnt retires in order: once only the younger chunks' operations are outstanding, chunk ci
      // (its DMA copies, which the compiler does not track, and its weight registers) has landed
      const int younger = (nchunks - 1 - ci) < (NST - 2) ? (nchunks - 1 - ci) : (NST - 2);
      if (younger <= 0) {
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
      } else {
        constexpr int n1 = kOps;
        __builtin_amdgcn_s_waitcnt((n1 & 15) | (7 << 4) | (15 << 8) | ((n1 >> 4) << 14));
      }
      __syncthreads();  // chunk ci is in LDS for everyone; everyone is past chunk ci - 1
      if (ci + NST - 1 < nchunks) {
        issue_x(ci + NST - 1);
        issue_w(nxt, t, tmask, ci + NST - 1);
      }
      compute(cur, acc, ci);
    };
    static_assert(NST == 2 || NST == 3, "the wait above covers one younger chunk at most");
    for (int ci = 0; ci < nchunks; ci += NST) {
#pragma unroll
      for (int k = 0; k < NST; ++k)
        if (ci + k < nchunks) step(ci + k, a[k], a[(k + NST - 1) % NST]);
    }
    // lane (g, c): rows n = 16 t + 4 g + r of column m = 16 mt + c
    const int n0 = 16 * t + 4 * g;
    if (t < ntiles) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = 16 * mt + c;
        if (m < M) {
          g_f32x4_t sum = acc[mt];
          if (partial != nullptr) {
            *reinterpret_cast<g_f32x4_t*>(partial + ((int64_t)blockIdx.y * M + m) * N + n0) = sum;
          } else {
            if (bias != nullptr) {
#pragma unroll
              for (int r = 0; r < 4; ++r) sum[r] += T::to_float(bias[n0 + r]);
            }
            uint2 o;
            o.x = (uint32_t)T::from_float(sum[0]) | ((uint32_t)T::from_float(sum[1]) << 16);
            o.y = (uint32_t)T::from_float(sum[2]) | ((uint32_t)T::from_float(sum[3]) << 16);
            *reinterpret_cast<uint2*>(y + (int64_t)m * N + n0) = o;
          }
        }
      }
    }
  }
  LVLLM_TRACE_END(6);
}

LVLLM_TRACE_READER(lvllm_trace_read_gemm)

// W[N,K] row-major -> packed [N/16][K/32][4][16][8]; one thread per 16-byte chunk
__global__ void pack_weight_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, const int64_t nchunks,
                                   const int K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // index of the destination chunk
  if (i >= nchunks) return;
  const int steps = K >> 5;
  const int c = (int)(i & 15), g = (int)((i >> 4) & 3);
  const int64_t ts = i >> 6;
  const int64_t t = ts / steps;
  const int s = (int)(ts - t * steps);
  const int64_t n = t * 16 + c, k = (int64_t)s * 32 + g * 8;
  dst[i + t * (LVLLM_GEMM_TILE_PAD / 16)] = src[(n * K + k) >> 3];
}

}  // namespace lvllm

using namespace lvllm;

// Workgroups for `units` n-tiles (or SwiGLU pairs) handed out round-robin when at most `cap` may run: with
// tuning().gemm_balance the fewest that need no more rounds than `cap` would, so that no workgroup idles through the
// last round while others still stream (qkv of an 8B model: 384 tiles, 256 -> 192 workgroups, two full rounds).
static inline int balanced_groups(int units, int cap) {
  if (cap < 1) cap = 1;
  if (units <= cap) return units < 1 ? 1 : units;
  if (!tuning().gemm_balance) return cap;
  const int rounds = (units + cap - 1) / cap;
  return (units + rounds - 1) / rounds;
}

// Bytes of fp32 workspace lvllm_skinny_gemm needs for this shape (0 when K is not split over
// workgroups).
// k-steps one wave may own: its X fragments (MT * steps * 4 VGPRs) must stay in registers
static inline int max_steps_per_wave(int M) { return M <= 32 ? 16 : 8; }

extern "C" int64_t lvllm_skinny_gemm_workspace_bytes(int M, int N, int K) {
  const int total_steps = K / 32;
  const int cap = kGemmWaves * max_steps_per_wave(M);
  const int ksplit = (total_steps + cap - 1) / cap;
  return ksplit > 1 ? (int64_t)ksplit * M * N * 4 : 0;
}

// Y[M,N] = X[M,K] . W[N,K]^T (+ bias[N]).  X rows ldx elements apart; W and Y contiguous.
// Returns 0 on success, 3 when the shape is outside this kernel's envelope (caller falls back
// to a library GEMM): M > 64, K % 32 != 0, N % 16 != 0, W >= 4 GiB, or fp32.
extern "C" int lvllm_pack_weight(void* dst, const void* src, int N, int K, int dtype, void* stream) {
  LV_CHECK(dtype == LVLLM_BF16 || dtype == LVLLM_F16, "16-bit weights only");
  LV_CHECK(N % 16 == 0 && K % 32 == 0, "N must be a multiple of 16 and K of 32");
  LV_CHECK(dst != src, "packing is out of place");
  const int64_t nchunks = (int64_t)N * K / 8;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (uint4*)dst, (const uint4*)src, nchunks, K);
  LV_LAUNCH_CHECK();
  return 0;
}

// Greedy sampling fused into a projection: tokens[m] = argmax_n (X . W^T)[m, n] over the values rounded
// to the element type (what torch.argmax of the projection's output sees; ties: the smaller n), without
// writing the [M, N] result.  workspace: lvllm_skinny_gemm_argmax_workspace_bytes(M) bytes.
extern "C" int64_t lvllm_skinny_gemm_argmax_workspace_bytes(int M) { return (int64_t)1024 * (M > 0 ? M : 1) * 8; }
// for any shape of the envelope: adds the fp32 partials when K is split over workgroups at this M
extern "C" int64_t lvllm_skinny_gemm_argmax_workspace_bytes_ex(int M, int N, int K) {
  return lvllm_skinny_gemm_argmax_workspace_bytes(M) + lvllm_skinny_gemm_workspace_bytes(M, N, K);
}

extern "C" int lvllm_skinny_gemm_ex(void* y, const void* x, const void* w, const void* bias, int M, int N,
                                    int K, int64_t ldx, int dtype, int packed, int act, int partial_out,
                                    int* ksplit_out, void* workspace, int64_t workspace_bytes, void* stream);

extern "C" int lvllm_skinny_gemm_argmax(int64_t* tokens, const void* x, const void* w_packed, int M, int N, int K,
                                        int64_t ldx, int dtype, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
  return lvllm_skinny_gemm_ex(tokens, x, w_packed, nullptr, M, N, K, ldx, dtype, 1, 3, 0, nullptr, workspace,
                              workspace_bytes, stream);
}

// `act` = 1: X rows are [gate (K) | up (K)] and the kernel multiplies W by silu(gate)*up
// (the SwiGLU activation fused into the down projection); `act` = 2: W rows are [gate (N/2) | up (N/2)]
// and y [M, N/2] = silu(X.gate^T) * (X.up^T), the activation applied in the epilogue (the gate_up
// projection and silu_and_mul in one launch, bit-identical to the pair); `partial_out` != 0: leave the fp32
// split-K partials in `workspace` ([ksplit, M, N]) and do not write y (the caller's next kernel
// sums them: lvllm_fused_add_rms_norm_splitk).  *ksplit_out receives the number of partials.
extern "C" int lvllm_skinny_gemm_ex(void* y, const void* x, const void* w, const void* bias, int M, int N,
                                    int K, int64_t ldx, int dtype, int packed, int act, int partial_out,
                                    int* ksplit_out, void* workspace, int64_t workspace_bytes, void* stream);

extern "C" int lvllm_skinny_gemm(void* y, const void* x, const void* w, const void* bias, int M, int N,
                                 int K, int64_t ldx, int dtype, int packed, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  return lvllm_skinny_gemm_ex(y, x, w, bias, M, N, K, ldx, dtype, packed, 0, 0, nullptr, workspace,
                              workspace_bytes, stream);
}

extern "C" int lvllm_skinny_gemm_ex(void* y, const void* x, const void* w, const void* bias, int M, int N,
                                    int K, int64_t ldx, int dtype, int packed, int act, int partial_out,
                                    int* ksplit_out, void* workspace, int64_t workspace_bytes, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (!(dtype == LVLLM_BF16 || dtype == LVLLM_F16) || M > 64 || (K % 32) != 0 || (N % 16) != 0 ||
      (int64_t)N * K * 2 >= ((int64_t)1 << 32) - 16 || (ldx % 8) != 0 ||
      ((int64_t)(M - 1) * ldx + (act == 1 ? 2 : 1) * (int64_t)K) * 2 >= ((int64_t)1 << 31) ||
      ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) != 0)) {
    set_error("lvllm_skinny_gemm: shape outside the kernel's envelope");
    return 3;
  }
  const int total_steps = K / 32;
  const int cap = kGemmWaves * max_steps_per_wave(M);
  int ksplit = (total_steps + cap - 1) / cap;
  // a caller that sums the partials itself may ask for a finer split of K (each workgroup then reads only its
  // K slice of X; the price is ksplit fp32 slabs): tuning "gemm_partials_ksplit"
  if (partial_out && act == 0 && tuning().gemm_partials_ksplit > ksplit &&
      total_steps >= tuning().gemm_partials_ksplit * kGemmWaves)
    ksplit = tuning().gemm_partials_ksplit;
  const int steps_per_wg = (total_steps + ksplit - 1) / ksplit;
  const int steps_per_wave = (steps_per_wg + kGemmWaves - 1) / kGemmWaves;  // <= max_steps_per_wave(M)
  const int ntiles = N / 16;
  // one workgroup (8 waves, ~236 VGPRs: nothing else fits beside it) per CU by default; a host
  // running steps on several streams asks for fewer (common.h Tuning)
  const int gemm_cus = (ntiles >= tuning().gemm_wide_min_tiles && tuning().gemm_workgroups_wide > 0) ? tuning().gemm_workgroups_wide
                                                                            : tuning().gemm_workgroups;
  int groups = gemm_cus / ksplit;
  if (groups < 1) groups = 1;
  const bool glu_reduce = act == 2 && ksplit > 1;  // K split over workgroups: SwiGLU inside the reduce pass
  if (act == 2) {  // SwiGLU: y is [M, N / 2]
    LV_CHECK(N % 32 == 0 && !partial_out, "the SwiGLU epilogue needs N % 32 == 0 and writes y itself");
    if (glu_reduce) act = 0;  // plain tiles into the fp32 partials
  }
  groups = balanced_groups(act == 2 ? ntiles / 2 : ntiles, groups);  // (SwiGLU: (gate, up) pairs are handed out)
  void* const tokens_out = y;
  const bool amax_reduce = act == 3 && ksplit > 1;  // K split over workgroups: arg-max inside a reduce pass
  float* amax_cand = nullptr;
  if (act == 3) {  // arg-max: y is int64 [M]; candidates (and, with split K, the fp32 partials) go through `workspace`
    LV_CHECK(!partial_out && bias == nullptr, "the arg-max epilogue takes no bias and writes the tokens itself");
    if (amax_reduce) {
      LV_CHECK(workspace != nullptr && workspace_bytes >= (int64_t)ksplit * M * N * 4 + (int64_t)64 * M * 8,
               "workspace too small (see lvllm_skinny_gemm_argmax_workspace_bytes_ex)");
      amax_cand = (float*)((char*)workspace + (int64_t)ksplit * M * N * 4);
      act = 0;  // plain tiles into the fp32 partials
    } else {
      LV_CHECK(workspace != nullptr && workspace_bytes >= (int64_t)groups * M * 8,
               "workspace too small (see lvllm_skinny_gemm_argmax_workspace_bytes)");
      y = workspace;
    }
  }
  if (ksplit_out) *ksplit_out = ksplit;
  LV_CHECK(!(partial_out && bias != nullptr), "partial_out leaves the bias to the caller");
  float* partial = nullptr;
  if (ksplit > 1 || partial_out) {
    LV_CHECK(workspace != nullptr && workspace_bytes >= (int64_t)ksplit * M * N * 4,
             "workspace too small (see lvllm_skinny_gemm_workspace_bytes)");
    partial = (float*)workspace;
  }
  hipStream_t s = (hipStream_t)stream;
  const int MT = (M + 15) / 16;
  const bool k8 = steps_per_wave <= 8;
#define LV_SG(T_, MT_)                                                                              \
  do {                                                                                              \
    if (k8)                                                                                         \
      launch_skinny<T_, MT_, 8>(y, partial, x, w, bias, M, N, K, ldx, steps_per_wave, ntiles, groups, \
                                ksplit, packed != 0, act, s);                                       \
    else                                                                                            \
      launch_skinny<T_, MT_, 16>(y, partial, x, w, bias, M, N, K, ldx, steps_per_wave, ntiles, groups, \
                                 ksplit, packed != 0, act, s);                                      \
  } while (0)
#define LV_SG_MT(T_)                           \
  switch (MT) {                                \
    case 1: LV_SG(T_, 1); break;               \
    case 2: LV_SG(T_, 2); break;               \
    default:                                   \
      launch_skinny<T_, 4, 8>(y, partial, x, w, bias, M, N, K, ldx, steps_per_wave, ntiles, groups, ksplit, packed != 0, act, s); \
      break;                                   \
  }
  if (dtype == LVLLM_BF16) { LV_SG_MT(BF16) } else { LV_SG_MT(F16) }
#undef LV_SG_MT
#undef LV_SG
  LV_LAUNCH_CHECK();
  if (amax_reduce) {
    if (dtype == LVLLM_BF16)
      hipLaunchKernelGGL((skinny_gemm_reduce_argmax_kernel<BF16>), dim3(64, M), dim3(256), 0, s, amax_cand, partial, M,
                         N, ksplit);
    else
      hipLaunchKernelGGL((skinny_gemm_reduce_argmax_kernel<F16>), dim3(64, M), dim3(256), 0, s, amax_cand, partial, M,
                         N, ksplit);
    hipLaunchKernelGGL(skinny_argmax_reduce_kernel, dim3(M), dim3(256), 0, s, (int64_t*)tokens_out,
                       (const float*)amax_cand, 64, M);
    LV_LAUNCH_CHECK();
    return 0;
  }
  if (act == 3) {
    hipLaunchKernelGGL(skinny_argmax_reduce_kernel, dim3(M), dim3(256), 0, s, (int64_t*)tokens_out, (const float*)y,
                       groups, M);
    LV_LAUNCH_CHECK();
    return 0;
  }
  if (glu_reduce) {
    const int grid = (int)(((int64_t)M * (N / 2) / 4 + 255) / 256);
    if (dtype == LVLLM_BF16)
      hipLaunchKernelGGL((skinny_gemm_reduce_swiglu_kernel<BF16>), dim3(grid), dim3(256), 0, s, (uint16_t*)y, partial,
                         (const uint16_t*)bias, M, N, ksplit);
    else
      hipLaunchKernelGGL((skinny_gemm_reduce_swiglu_kernel<F16>), dim3(grid), dim3(256), 0, s, (uint16_t*)y, partial,
                         (const uint16_t*)bias, M, N, ksplit);
    LV_LAUNCH_CHECK();
    return 0;
  }
  if (ksplit > 1 && !partial_out) {
    const int64_t MN = (int64_t)M * N;
    const int threads = 256;
    const int grid = (int)((MN / 4 + threads - 1) / threads);
    if (dtype == LVLLM_BF16)
      hipLaunchKernelGGL((skinny_gemm_reduce_kernel<BF16>), dim3(grid), dim3(threads), 0, s, (uint16_t*)y,
                         partial, (const uint16_t*)bias, MN, N, ksplit);
    else
      hipLaunchKernelGGL((skinny_gemm_reduce_kernel<F16>), dim3(grid), dim3(threads), 0, s, (uint16_t*)y,
                         partial, (const uint16_t*)bias, MN, N, ksplit);
    LV_LAUNCH_CHECK();
  }
  return 0;
}

// W8A8: Y[M,N] = T((fp8(X / x_scale) . W8^T) * x_scale * w_scale + bias).  X [M, K] in T (ldx apart),
// W8 = fp8 weights [N, K] packed by lvllm_pack_weight on their [N, K/2] 16-bit view, per-tensor
// scales on the device (w8a8_utils.py:103-156 of the reference with a static activation scale).
// W8A8 waves own at most 8 k-steps (64 k each): the variant whose activations take the row-order +
// LDS-transpose route (its fragments leave room for the landing registers); longer K is split over more
// workgroups instead.  LVLLM_W8_STEPS=16 restores the old split for A/B runs.
static inline int w8_max_steps_per_wave(int M) {
  static const int forced = getenv("LVLLM_W8_STEPS") ? atoi(getenv("LVLLM_W8_STEPS")) : 0;
  if (forced == 16) return max_steps_per_wave(M);
  return 8;
}

extern "C" int64_t lvllm_skinny_gemm_w8a8_workspace_bytes(int M, int N, int K) {
  const int total_steps = (K / 2) / 32;
  const int cap = kGemmWaves * w8_max_steps_per_wave(M);
  const int ksplit = (total_steps + cap - 1) / cap;
  return ksplit > 1 ? (int64_t)ksplit * M * N * 4 : 0;
}

extern "C" int lvllm_skinny_gemm_w8a8_ex(void* y, const void* x, const void* w_packed, const void* bias,
                                         const float* x_scale, const float* w_scale, int M, int N, int K,
                                         int64_t ldx, int dtype, int act, void* workspace,
                                         int64_t workspace_bytes, void* stream);

extern "C" int lvllm_skinny_gemm_w8a8(void* y, const void* x, const void* w_packed, const void* bias,
                                      const float* x_scale, const float* w_scale, int M, int N, int K,
                                      int64_t ldx, int dtype, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
  return lvllm_skinny_gemm_w8a8_ex(y, x, w_packed, bias, x_scale, w_scale, M, N, K, ldx, dtype, 0, workspace,
                                   workspace_bytes, stream);
}

// act = 2: W rows are [gate | up], y [M, N/2] = silu_and_mul of the projection (see lvllm_skinny_gemm_ex)
extern "C" int lvllm_skinny_gemm_w8a8_ex(void* y, const void* x, const void* w_packed, const void* bias,
                                         const float* x_scale, const float* w_scale, int M, int N, int K,
                                         int64_t ldx, int dtype, int act, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  LV_CHECK(act == 0 || act == 2 || act == 3 || act == 4, "lvllm_skinny_gemm_w8a8_ex: act must be 0, 2, 3 or 4");
  const bool leave_partials = act == 4;  // the raw fp32 split-K partials stay in `workspace`, y is not written
  if (leave_partials) act = 0;
  LV_CHECK(x_scale != nullptr && w_scale != nullptr, "scales are device pointers to one float each");
  if (!(dtype == LVLLM_BF16 || dtype == LVLLM_F16) || M > 64 || (K % 64) != 0 || (N % 16) != 0 ||
      (int64_t)N * K >= ((int64_t)1 << 32) - 16 || (ldx % 8) != 0 ||
      ((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y) & 15) != 0)) {
    set_error("lvllm_skinny_gemm_w8a8: shape outside the kernel's envelope");
    return 3;
  }
  const int K2 = K / 2;  // the kernel's K: 16-bit units of the weight rows
  const int total_steps = K2 / 32;
  const int cap = kGemmWaves * w8_max_steps_per_wave(M);
  const int ksplit = (total_steps + cap - 1) / cap;
  const int steps_per_wg = (total_steps + ksplit - 1) / ksplit;
  const int steps_per_wave = (steps_per_wg + kGemmWaves - 1) / kGemmWaves;
  const int ntiles = N / 16;
  int groups = ((ntiles >= tuning().gemm_wide_min_tiles && tuning().gemm_workgroups_wide > 0) ? tuning().gemm_workgroups_wide
                                                                      : tuning().gemm_workgroups) / ksplit;
  if (groups < 1) groups = 1;
  if (act == 2) LV_CHECK(N % 32 == 0 && ksplit == 1, "the SwiGLU epilogue needs N % 32 == 0 and K within one workgroup");
  groups = balanced_groups(act == 2 ? ntiles / 2 : ntiles, groups);
  void* const tokens_out = y;
  if (act == 3) {  // arg-max epilogue (see lvllm_skinny_gemm_argmax): y is int64 [M], candidates via `workspace`
    LV_CHECK(ksplit == 1 && bias == nullptr && M <= 32,
             "the W8A8 arg-max epilogue needs M <= 32, K within one workgroup and no bias");  // its LDS slots: see kernel
    LV_CHECK(workspace != nullptr && workspace_bytes >= (int64_t)groups * M * 8, "workspace too small");
    y = workspace;
  }
  float* partial = nullptr;
  LV_CHECK(!leave_partials || (ksplit > 1 && bias == nullptr),
           "act = 4 needs K split over workgroups (lvllm_skinny_gemm_w8a8_workspace_bytes > 0) and no bias");
  if (ksplit > 1) {
    LV_CHECK(workspace != nullptr && workspace_bytes >= (int64_t)ksplit * M * N * 4,
             "workspace too small (see lvllm_skinny_gemm_w8a8_workspace_bytes)");
    partial = (float*)workspace;
  }
  hipStream_t s = (hipStream_t)stream;
  const int MT = (M + 15) / 16;
  const bool k8 = steps_per_wave <= 8;
#define LV_SG8(T_, MT_)                                                                                   \
  do {                                                                                                    \
    if (k8 || MT_ == 4)                                                                                   \
      launch_skinny<T_, MT_, 8, true>(y, partial, x, w_packed, bias, M, N, K2, ldx, steps_per_wave, ntiles, \
                                      groups, ksplit, true, act, s, x_scale, w_scale);                    \
    else                                                                                                  \
      launch_skinny<T_, MT_, (MT_ == 4 ? 8 : 16), true>(y, partial, x, w_packed, bias, M, N, K2, ldx,       \
                                                       steps_per_wave, ntiles, groups, ksplit, true, act, s, \
                                                       x_scale, w_scale);                                 \
  } while (0)
#define LV_SG8_MT(T_)          \
  switch (MT) {                \
    case 1: LV_SG8(T_, 1); break; \
    case 2: LV_SG8(T_, 2); break; \
    default: LV_SG8(T_, 4); break; \
  }
  if (dtype == LVLLM_BF16) { LV_SG8_MT(BF16) } else { LV_SG8_MT(F16) }
#undef LV_SG8_MT
#undef LV_SG8
  LV_LAUNCH_CHECK();
  if (act == 3) {
    hipLaunchKernelGGL(skinny_argmax_reduce_kernel, dim3(M), dim3(256), 0, s, (int64_t*)tokens_out, (const float*)y,
                       groups, M);
    LV_LAUNCH_CHECK();
    return 0;
  }
  if (ksplit > 1 && !leave_partials) {
    const int64_t MN = (int64_t)M * N;
    const int grid = (int)((MN / 4 + 255) / 256);
    if (dtype == LVLLM_BF16)
      hipLaunchKernelGGL((skinny_gemm_reduce_kernel<BF16>), dim3(grid), dim3(256), 0, s, (uint16_t*)y, partial,
                         (const uint16_t*)bias, MN, N, ksplit, x_scale, w_scale);
    else
      hipLaunchKernelGGL((skinny_gemm_reduce_kernel<F16>), dim3(grid), dim3(256), 0, s, (uint16_t*)y, partial,
                         (const uint16_t*)bias, MN, N, ksplit, x_scale, w_scale);
    LV_LAUNCH_CHECK();
  }
  return 0;
}

// W8A8 with activations that arrive already quantised: x_fp8 [M, K] bytes (rows ldx BYTES apart), written by
// lvllm_rms_norm_quant / lvllm_fused_add_rms_norm_quant / lvllm_fused_add_rms_norm_splitk_quant or by this entry's own
// SwiGLU epilogue -- i.e. static_scaled_fp8_quant(x, *x_scale), which is what lvllm_skinny_gemm_w8a8_ex computes in its
// prologue: results are bit-identical to that entry's on the unquantised x.  act 0 / 2 (SwiGLU, y [M, N/2]) / 4 (raw
// split-K partials left in `workspace`).  act = 2 may also (y_fp8 != null) or only (y == null) write the activation as
// fp8 with *y_fp8_scale, for a following lvllm_skinny_gemm_w8a8_q.  M <= 32, K within 8 k-steps of 64 per wave (K <=
// 4096 per workgroup; longer K is split over workgroups as in lvllm_skinny_gemm_w8a8_ex).  3 = outside the envelope.
extern "C" int lvllm_skinny_gemm_w8a8_q(void* y, void* y_fp8, const float* y_fp8_scale, const void* x_fp8,
                                        const void* w_packed, const void* bias, const float* x_scale,
                                        const float* w_scale, int M, int N, int K, int64_t ldx, int dtype, int act,
                                        void* workspace, int64_t workspace_bytes, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  LV_CHECK(act == 0 || act == 2 || act == 4, "lvllm_skinny_gemm_w8a8_q: act must be 0, 2 or 4");
  const bool leave_partials = act == 4;
  if (leave_partials) act = 0;
  LV_CHECK(x_scale != nullptr && w_scale != nullptr, "scales are device pointers to one float each");
  LV_CHECK((y_fp8 == nullptr) == (y_fp8_scale == nullptr), "y_fp8 and y_fp8_scale: both or none");
  LV_CHECK(y_fp8 == nullptr || act == 2, "the fp8 output belongs to the SwiGLU epilogue (act = 2)");
  LV_CHECK(y != nullptr || y_fp8 != nullptr || leave_partials, "nowhere to write the result");
  if (!(dtype == LVLLM_BF16 || dtype == LVLLM_F16) || M > 32 || (K % 64) != 0 || (N % 16) != 0 ||
      (int64_t)N * K >= ((int64_t)1 << 32) - 16 || (ldx % 16) != 0 || (int64_t)(M - 1) * ldx + K >= ((int64_t)1 << 31) ||
      ((((uintptr_t)x_fp8 | (uintptr_t)w_packed | (uintptr_t)y) & 15) != 0) || (((uintptr_t)y_fp8) & 3) != 0) {
    set_error("lvllm_skinny_gemm_w8a8_q: shape outside the kernel's envelope");
    return 3;
  }
  const int K2 = K / 2;
  const int total_steps = K2 / 32;
  const int cap = kGemmWaves * 8;
  const int ksplit = (total_steps + cap - 1) / cap;
  const int steps_per_wg = (total_steps + ksplit - 1) / ksplit;
  const int steps_per_wave = (steps_per_wg + kGemmWaves - 1) / kGemmWaves;
  const int ntiles = N / 16;
  int groups = ((ntiles >= tuning().gemm_wide_min_tiles && tuning().gemm_workgroups_wide > 0) ? tuning().gemm_workgroups_wide
                                                                      : tuning().gemm_workgroups) / ksplit;
  if (groups < 1) groups = 1;
  if (act == 2) LV_CHECK(N % 32 == 0 && ksplit == 1, "the SwiGLU epilogue needs N % 32 == 0 and K within one workgroup");
  groups = balanced_groups(act == 2 ? ntiles / 2 : ntiles, groups);
  LV_CHECK(!leave_partials || (ksplit > 1 && bias == nullptr),
           "act = 4 needs K split over workgroups (lvllm_skinny_gemm_w8a8_workspace_bytes > 0) and no bias");
  float* partial = nullptr;
  if (ksplit > 1) {
    LV_CHECK(workspace != nullptr && workspace_bytes >= (int64_t)ksplit * M * N * 4,
             "workspace too small (see lvllm_skinny_gemm_w8a8_workspace_bytes)");
    partial = (float*)workspace;
  }
  hipStream_t s = (hipStream_t)stream;
  const int MT = (M + 15) / 16;
#define LV_SGQ(T_)                                                                                                   \
  do {                                                                                                               \
    if (MT == 1)                                                                                                     \
      launch_skinny<T_, 1, 8, true, true>(y, partial, x_fp8, w_packed, bias, M, N, K2, ldx, steps_per_wave, ntiles,  \
                                          groups, ksplit, true, act, s, x_scale, w_scale, (uint8_t*)y_fp8, y_fp8_scale); \
    else                                                                                                             \
      launch_skinny<T_, 2, 8, true, true>(y, partial, x_fp8, w_packed, bias, M, N, K2, ldx, steps_per_wave, ntiles,  \
                                          groups, ksplit, true, act, s, x_scale, w_scale, (uint8_t*)y_fp8, y_fp8_scale); \
  } while (0)
  if (dtype == LVLLM_BF16) LV_SGQ(BF16); else LV_SGQ(F16);
#undef LV_SGQ
  LV_LAUNCH_CHECK();
  if (ksplit > 1 && !leave_partials) {
    const int64_t MN = (int64_t)M * N;
    const int grid = (int)((MN / 4 + 255) / 256);
    if (dtype == LVLLM_BF16)
      hipLaunchKernelGGL((skinny_gemm_reduce_kernel<BF16>), dim3(grid), dim3(256), 0, s, (uint16_t*)y, partial,
                         (const uint16_t*)bias, MN, N, ksplit, x_scale, w_scale);
    else
      hipLaunchKernelGGL((skinny_gemm_reduce_kernel<F16>), dim3(grid), dim3(256), 0, s, (uint16_t*)y, partial,
                         (const uint16_t*)bias, MN, N, ksplit, x_scale, w_scale);
    LV_LAUNCH_CHECK();
  }
  return 0;
}

// ---- 65..256 rows --------------------------------------------------------------------------
static inline int stream_gemm_ksplit(int M, int N, int K) {
  const int KC = 4;
  const int total_steps = K / 32;
  const int groups = (N / 16 + kGemmWaves - 1) / kGemmWaves;
  int ksplit = tuning().gemm_workgroups / groups;  // fill the CUs the host allows
  const int max_split = (total_steps + 2 * KC - 1) / (2 * KC);  // at least two chunks per workgroup
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit > 16) ksplit = 16;
  if (ksplit < 1) ksplit = 1;
  return ksplit;
}

extern "C" int64_t lvllm_stream_gemm_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K < 32) return 0;
  const int ksplit = stream_gemm_ksplit(M, N, K);
  return ksplit > 1 ? (int64_t)ksplit * M * N * 4 : 0;
}

// Y[M,N] = X[M,K] . W[N,K]^T (+ bias[N]) for 1 <= M <= 256 rows (meant for 65..256; the register-resident
// kernel above is faster below).  W packed by lvllm_pack_weight.  Returns 3 outside the envelope.
extern "C" int lvllm_stream_gemm(void* y, const void* x, const void* w_packed, const void* bias, int M, int N,
                                 int K, int64_t ldx, int dtype, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (!(dtype == LVLLM_BF16 || dtype == LVLLM_F16) || M > 256 || (K % 32) != 0 || (N % 16) != 0 ||
      (int64_t)N * K * 2 >= ((int64_t)1 << 32) - 16 || (ldx % 8) != 0 ||
      ((int64_t)(M - 1) * ldx + K) * 2 >= ((int64_t)1 << 31) ||
      ((((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y) & 15) != 0)) {
    set_error("lvllm_stream_gemm: shape outside the kernel's envelope");
    return 3;
  }
  const int ntiles = N / 16;
  const int groups_total = (ntiles + kGemmWaves - 1) / kGemmWaves;
  const int ksplit = stream_gemm_ksplit(M, N, K);
  const int total_steps = K / 32;
  const int KC = 4;
  int steps_per_split = (total_steps + ksplit - 1) / ksplit;
  steps_per_split = (steps_per_split + KC - 1) / KC * KC;  // whole chunks
  float* partial = nullptr;
  if (ksplit > 1) {
    LV_CHECK(workspace != nullptr && workspace_bytes >= (int64_t)ksplit * M * N * 4,
             "workspace too small (see lvllm_stream_gemm_workspace_bytes)");
    partial = (float*)workspace;
  }
  int groups = tuning().gemm_workgroups / ksplit;
  if (groups < 1) groups = 1;
  if (groups > groups_total) groups = groups_total;
  hipStream_t s = (hipStream_t)stream;
  const size_t smem = M > 128 ? (size_t)2 * 256 * 256 : M > 64 ? (size_t)3 * 128 * 256 : (size_t)3 * 64 * 256;  // [NST][ROWS][RB]
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kern, dim3(groups, ksplit), dim3(kGemmWaves * 64), smem, s, (uint16_t*)y, partial,
                       (const uint16_t*)x, (const uint16_t*)w_packed, (const uint16_t*)bias, M, N, K, ldx, ntiles,
                       steps_per_split);
  };
  if (dtype == LVLLM_BF16) {
    if (M <= 64) go(stream_gemm_kernel<BF16, 4, 4, 3>);
    else if (M <= 128) go(stream_gemm_kernel<BF16, 8, 4, 3>);
    else go(stream_gemm_kernel<BF16, 16, 4, 2>);
  } else {
    if (M <= 64) go(stream_gemm_kernel<F16, 4, 4, 3>);
    else if (M <= 128) go(stream_gemm_kernel<F16, 8, 4, 3>);
    else go(stream_gemm_kernel<F16, 16, 4, 2>);
  }
  LV_LAUNCH_CHECK();
  if (ksplit > 1) {
    const int64_t MN = (int64_t)M * N;
    const int grid = (int)((MN / 4 + 255) / 256);
    if (dtype == LVLLM_BF16)
      hipLaunchKernelGGL((skinny_gemm_reduce_kernel<BF16>), dim3(grid), dim3(256), 0, s, (uint16_t*)y, partial,
                         (const uint16_t*)bias, MN, N, ksplit);
    else
      hipLaunchKernelGGL((skinny_gemm_reduce_kernel<F16>), dim3(grid), dim3(256), 0, s, (uint16_t*)y, partial,
                         (const uint16_t*)bias, MN, N, ksplit);
    LV_LAUNCH_CHECK();
  }
  return 0;
}
