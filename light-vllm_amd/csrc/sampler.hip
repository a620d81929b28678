// Device-side sampler: everything between the lm_head's logits and the next token ids of a decode step, in one call
// (one launch, or two: selection, then a draw shared by several workgroups per row), so that a step with sampled
// (non-greedy) requests stays inside the captured graph and the multi-step burst.
//
// WHAT (reference: light_vllm/decoding/backends/sampler.py:90-200 forward; :238-277 _apply_min_tokens_penalty;
// :281-301 _apply_penalties; :304-330 _apply_top_k_top_p; :333-347 _apply_min_p; :434-454 _multinomial):
//   per row: stop tokens banned while fewer than min_tokens outputs exist -> repetition / frequency / presence
//   penalties from the prompt and output token histories -> division by the temperature -> top-k (keep everything
//   >= the k-th largest value) -> top-p (sorted ascending, drop the prefix whose probability mass is <= 1 - p, never
//   the largest) -> min-p (drop what is less likely than min_p x the most likely) -> an exponential-race multinomial
//   draw (arg-max of p_i / q_i, q_i ~ Exp(1)); greedy rows (temperature < 1e-5) take the arg-max of the penalised
//   logits.  The reference runs these as ~25 torch launches (two full sorts among them) on the [rows, vocab] matrix.
//
// HOW: one workgroup of 1024 threads per row.  The row's fp32 working copy lives in a caller-provided scratch (it
// stays in the XCD's L2 between passes).  Nothing is sorted: the k-th largest value and the top-p cut are found by a
// descent over LDS histograms (select_key below: one level of linear bins under the row's maximum, then 11 bits at a
// time of the order-preserving integer image of the float) -- counts for top-k, probability mass for top-p.  Mass is
// accumulated in 2^-40 fixed point with integer atomics, so the result does not depend on the order in which the
// adds arrive: the kernel is deterministic.
// Per-request state lives on the device in `state slots` (SamplerParams + one int32 per vocabulary entry: bit 31 =
// seen in the prompt, bits 0..30 = occurrences in the output); the kernel appends the token it drew to that state, so
// the k model steps of a burst need no host round trip.  Random numbers: Philox4x32-10 keyed by the request's seed,
// counter = (vocabulary index / 4, number of tokens the request has drawn so far): a request's stream depends on
// nothing but its seed and its own step count.
//
// Differences from the reference, stated: arithmetic is fp32 on the model-dtype logits (the reference's live code
// keeps the model dtype through penalties and filters, its golden vectors here were recorded in fp32); exact ties at
// the top-p cut are kept as a group (the reference keeps whichever of them its unstable sort happened to put last);
// the random stream is Philox, not torch's generator -- seeded requests repeat themselves, they do not repeat torch.
#include <float.h>

#include <type_traits>

#include "common.h"

namespace lvllm {

struct SamplerParams {  // one per state slot; 128 bytes; the host writes it, the kernel advances output_len
  float temperature, top_p, min_p, presence, frequency, repetition;
  int32_t top_k;       // <= 0 or >= vocab: off
  int32_t min_tokens;  // stop tokens are banned while output_len < min_tokens
  uint64_t seed;
  int32_t output_len;  // tokens drawn so far (device-updated)
  int32_t num_banned;
  int32_t banned[20];
};
static_assert(sizeof(SamplerParams) == LVLLM_SAMPLER_PARAMS_BYTES, "include/lvllm_hip.h states the size");

constexpr int kSamplerThreads = 1024;
constexpr int kSamplerMaxParts = 8;      // workgroups that share the first pass of a row
// the tail of a scratch row behind the (4-aligned) vocabulary: [0, 25) the meeting of the first pass (maxima | indices |
// counter | XCD ids), [28, 32) what the draw launch needs (maximum, cut, mode, arg-max), [32, 57) the meeting of the draw
constexpr int kSamplerRecord = 28, kSamplerDrawArea = 32, kSamplerRowTail = 64;  // floats of a scratch row behind the (4-aligned) vocabulary
constexpr float kSamplingEps = 1e-5f;  // sampling_params.py:14 (_SAMPLING_EPS)

__device__ __forceinline__ uint32_t order_key(float x) {  // monotone: a < b  <=>  key(a) < key(b)
  const uint32_t b = __builtin_bit_cast(uint32_t, x);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
  uint32_t c2 = 0, c3 = 0;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename LT>
__device__ __forceinline__ float logit_to_float(LT v);
template <>
__device__ __forceinline__ float logit_to_float<float>(float v) { return v; }
struct BF16Bits { uint16_t v; };
struct F16Bits { uint16_t v; };
template <>
__device__ __forceinline__ float logit_to_float<BF16Bits>(BF16Bits v) { return BF16::to_float(v.v); }
template <>
__device__ __forceinline__ float logit_to_float<F16Bits>(F16Bits v) { return F16::to_float(v.v); }

// f(item, index) for `n` items of a row, the block striding over them; U independent loads are issued before the first
// one is used.  One workgroup walks a whole row: a loop that waits for each 2- or 4-byte load in turn runs at the
// memory latency (30 us for a 128 k row of bf16), so the passes below move 16-byte groups and keep several in flight.
template <int U, typename LOAD, typename F>
__device__ __forceinline__ void for_each_in_row(const int n, LOAD&& load, F&& f, const int first = 0) {
  const int end = first + n;
  int i = first + threadIdx.x;
  for (; i + (U - 1) * kSamplerThreads < end; i += U * kSamplerThreads) {
    decltype(load(0)) v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load(i + u * kSamplerThreads);
#pragma unroll
    for (int u = 0; u < U; ++u) f(v[u], i + u * kSamplerThreads);
  }
  for (; i < end; i += kSamplerThreads) f(load(i), i);
}

template <typename V>
struct alignas(16) Group {  // 16 bytes of a row
  static constexpr int N = 16 / sizeof(V);
  V v[N];
};

// group g of a row of n values: one 16-byte load when the row allows it, element by element otherwise (rows whose
// base or pitch is not a multiple of 16 bytes, and the ragged last group, whose missing values read as `pad`)
template <typename V>
__device__ __forceinline__ Group<V> load_group(const V* __restrict__ row, const int g, const int n, const bool aligned,
                                               const V pad) {
  constexpr int N = Group<V>::N;
  Group<V> r;
  if (aligned && (g + 1) * N <= n) {
    r = *reinterpret_cast<const Group<V>*>(row + g * N);
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) r.v[e] = g * N + e < n ? row[g * N + e] : pad;
  }
  return r;
}

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// f(value, index) for every value of a row -- or of its 16-byte groups [g0, g0 + ng) (ng < 0: all).  Only the ragged
// last group is checked element by element (a guard per element was a branch per element: these passes are bound by
// the instructions one CU can issue).
template <typename V, typename F>
__device__ __forceinline__ void for_each_value(const V* __restrict__ row, const int n, F&& f, const int g0 = 0,
                                               const int ng = -1) {
  constexpr int N = Group<V>::N;
  const bool al = aligned16(row);
  const int groups = (n + N - 1) / N, full = n / N;
  const int first = ng < 0 ? 0 : g0, end = ng < 0 ? groups : g0 + ng;
  const int end_full = end < full ? end : full;
  if (end_full > first)
    for_each_in_row<4>(end_full - first, [&](int g) { return load_group<V>(row, g, n, al, V{}); },
                       [&](const Group<V>& v, int g) {
#pragma unroll
                         for (int e = 0; e < N; ++e) f(v.v[e], g * N + e);
                       }, first);
  if (end > full && full >= first && threadIdx.x < n - full * N) f(row[full * N + threadIdx.x], full * N + (int)threadIdx.x);
}

// (value, index) arg-max of the block, ties to the smaller index; result in every thread
__device__ inline void block_argmax(float& v, int& i, float* sv, int* si) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float ov = __shfl_xor(v, m);
    const int oi = __shfl_xor(i, m);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { sv[wave] = v; si[wave] = i; }
  __syncthreads();
  float rv = lane < kSamplerThreads / 64 ? sv[lane] : -INFINITY;
  int ri = lane < kSamplerThreads / 64 ? si[lane] : 0x7fffffff;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    const float ov = __shfl_xor(rv, m);
    const int oi = __shfl_xor(ri, m);
    if (ov > rv || (ov == rv && oi < ri)) { rv = ov; ri = oi; }
  }
  v = __shfl(rv, 0);
  i = __shfl(ri, 0);
}

__device__ inline uint64_t block_sum_u64(uint64_t v, uint64_t* red) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  uint64_t r = lane < kSamplerThreads / 64 ? red[lane] : 0;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) r += __shfl_xor(r, m);
  return __shfl(r, 0);
}

// Histogram search by wave 0: the first bin -- walking down from the last one (DESC) or up from bin 0 -- at which the
// running total reaches `target` (STRICT: exceeds it).  Writes the bin and the total accumulated before it; bin = -1
// when no bin qualifies.
template <int NBINS, bool DESC, bool STRICT, typename H>
__device__ inline void find_bin(const H* hist, uint64_t target, int* bin_out, uint64_t* before_out) {
  if (threadIdx.x < 64) {
    constexpr int PER = NBINS / 64;
    const int lane = threadIdx.x;
    const int base = DESC ? NBINS - 1 - lane * PER : lane * PER;
    uint64_t mine = 0;
#pragma unroll 4
    for (int j = 0; j < PER; ++j) mine += hist[DESC ? base - j : base + j];
    uint64_t incl = mine;  // inclusive scan over lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    const uint64_t excl = incl - mine;
    const bool hit = STRICT ? (excl <= target && incl > target) : (excl < target && incl >= target);
    const uint64_t ballot = __ballot(hit);
    if (ballot == 0) {
      if (lane == 0) { *bin_out = -1; *before_out = 0; }
    } else if (lane == __ffsll((long long)ballot) - 1) {
      uint64_t run = excl;
      int b = -1;
      for (int j = 0; j < PER; ++j) {
        const int idx = DESC ? base - j : base + j;
        const uint64_t h = hist[idx];
        const bool here = STRICT ? (run + h > target) : (run + h >= target);
        if (here) { b = idx; break; }
        run += h;
      }
      *bin_out = b;
      *before_out = run;
    }
  }
  __syncthreads();
}

// Selection without a sort.  MASS = false: the key of the `want`-th largest value of the row.  MASS = true: the
// smallest key whose probability mass from the bottom, counted inside {key >= floor_key}, exceeds drop_frac x the
// mass of that set (*found = false when no key does).
//
// Level 1 bins the row LINEARLY by its distance below the maximum (1/64 per bin, the last bin takes everything 32 or
// more below): the values of a row of logits crowd a few octaves, so bins cut from the float's exponent bits put
// 128 k LDS atomics on ~40 addresses and the pass costs what the serialised atomics cost; cut linearly they spread
// over hundreds.  The last bin is not counted with atomics at all (a register sum per thread).  The levels below
// work on the order-preserving integer keys of the values inside the chosen bin, 11 bits at a time from the span
// the bin can hold, until single keys are told apart -- two more passes for ordinary logits, so the answer is the
// exact k-th value / cut, whatever the binning above it was.
constexpr int kLinBins = 2048;
constexpr float kLinPerUnit = 64.f;

__device__ __forceinline__ int lin_bin(float m, float x) {
  return (int)fminf((m - x) * kLinPerUnit, (float)(kLinBins - 1));  // NaN -> the last bin
}

template <bool MASS>
__device__ __forceinline__ uint32_t select_key(const float* __restrict__ x_row, const int vocab, const float m,
                                      const uint32_t floor_key, uint64_t target, const float drop_frac, uint64_t* hist64,
                                      uint64_t* red, int* s_bin, uint64_t* s_before, bool* found) {
  const int tid = threadIdx.x;
  *found = true;
  // counts fit 32 bits (a wave-wide 32-bit LDS atomic moves half the bank words of a 64-bit one; the passes are bound
  // by the one LDS unit of the CU they run on), mass needs its 2^-40 fixed point
  using H = typename std::conditional<MASS, uint64_t, uint32_t>::type;
  H* hist = reinterpret_cast<H*>(hist64);
  for (int b = tid; b < kLinBins; b += kSamplerThreads) hist[b] = 0;
  __syncthreads();
  // (the last bin -- everything 32 or more below the maximum -- is not counted with atomics: mass is summed in
  // registers, counts are what the other bins leave of the vocabulary)
  uint64_t tail = 0;
  for_each_value<float>(x_row, vocab, [&](float x, int) {
    if (MASS && order_key(x) < floor_key) return;
    const int b = lin_bin(m, x);
    if constexpr (MASS) {
      // 2^-40 fixed point; below e^-27.8 the product is < 1 and truncates to 0: no exponential for those (most of a
      // real vocabulary)
      const uint64_t w = x - m < -27.8f ? 0ull : (uint64_t)(expf(x - m) * 1099511627776.f);
      if (b < kLinBins - 1) atomicAdd(&hist[b], (H)w);
      else tail += w;
    } else {
      if (b < kLinBins - 1) atomicAdd(&hist[b], (H)1);
    }
  });
  if constexpr (MASS) {
    tail = block_sum_u64(tail, red);
  } else {
    __syncthreads();
    const uint64_t counted = block_sum_u64((uint64_t)hist[tid] + (tid + kSamplerThreads < kLinBins - 1 ? (uint64_t)hist[tid + kSamplerThreads] : 0), red);
    tail = (uint64_t)vocab - counted;
  }
  if (tid == 0) hist[kLinBins - 1] = (H)tail;
  __syncthreads();
  if (MASS) {
    const uint64_t z = block_sum_u64((uint64_t)hist[tid] + (uint64_t)hist[tid + kSamplerThreads], red);
    target = (uint64_t)((double)drop_frac * (double)z);
  }
  if (MASS) find_bin<kLinBins, true, true>(hist, target, s_bin, s_before);   // from the smallest values up
  else find_bin<kLinBins, false, false>(hist, target, s_bin, s_before);      // from the largest values down
  const int lb = *s_bin;
  if (lb < 0) {  // MASS: the whole mass is <= the target (p -> 0).  Counts: only with NaNs in the row
    *found = false;
    __syncthreads();
    return 0;
  }
  target -= *s_before;
  // keys the bin can hold: its value bounds widened by more than the rounding of (m - x) and of the bounds themselves
  const float slack = (fabsf(m) + 32.f) * 1e-6f;
  const float flo = lb == kLinBins - 1 ? -INFINITY : (m - (float)(lb + 1) / kLinPerUnit) - slack;
  const float fhi = (m - (float)lb / kLinPerUnit) + slack;
  uint32_t lo = lb == kLinBins - 1 ? 0u : order_key(flo);
  const uint32_t hi = order_key(fhi);
  if (hi < lo) lo = 0;
  int bits = 32 - __clz((int)((hi - lo) | 1u));
  __syncthreads();
#pragma unroll 1
  while (true) {
    const int shift = bits > 11 ? bits - 11 : 0;
    for (int b = tid; b < 2048; b += kSamplerThreads) hist[b] = 0;
    __syncthreads();
    for_each_value<float>(x_row, vocab, [&](float x, int) {
      // (two float compares turn nearly every element away before any integer work: the values of one bin)
      if (!(x >= flo && x <= fhi)) return;
      const uint32_t key = order_key(x);
      if (MASS && key < floor_key) return;
      const uint32_t d = key - lo;
      if (key < lo || ((uint64_t)d >> bits) != 0 || lin_bin(m, x) != lb) return;
      const uint64_t w = MASS ? (uint64_t)(expf(x - m) * 1099511627776.f) : 1ull;
      atomicAdd(&hist[d >> shift], (H)w);
    });
    __syncthreads();
    if (MASS) find_bin<2048, false, true>(hist, target, s_bin, s_before);
    else find_bin<2048, true, false>(hist, target, s_bin, s_before);
    const int b = *s_bin;
    if (b < 0) {  // cannot happen for a bin level 1 chose (same elements, same integer weights); stay safe
      *found = false;
      __syncthreads();
      return 0;
    }
    target -= *s_before;
    lo += (uint32_t)b << shift;
    bits = shift;
    __syncthreads();
    if (shift == 0) break;
  }
  return lo;
}

// Where the workgroups that share a pass of a row meet: `area` = 25 words of the row's scratch tail (8 values, 8
// indices, the arrival counter, 8 XCD ids).  Every workgroup leaves its partial arg-max (larger value wins, ties to
// the smaller index; index 0x7fffffff = nothing) and returns 0 -- except the one that arrives last, which gets the
// merged (value, index) and returns 1, or 2 when the parts did not all run on its XCD (it must then redo the pass
// alone: the others' stores were only pushed as far as THEIR L2).  No fence wider than the workgroup is issued.
__device__ __forceinline__ int meet_parts(float* area, const int W, const int part, float& bv, int& bi, int* s_last) {
  if (W == 1) return 1;
  const int tid = threadIdx.x;
  int* iarea = reinterpret_cast<int*>(area);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have been acknowledged by the L2
  __syncthreads();
  if (tid == 0) {
    // The record itself travels write-through (agent-scope stores = sc1: they leave this XCD's L2) and is read back
    // with agent-scope loads after the reader's own atomic has returned, so the three words -- the XCD id among them
    // -- are fresh wherever the parts ran; only the ROW's working values rely on the shared L2, and the ids say
    // whether they may.  (With plain stores a part that ran on another XCD left its id in THAT L2 and the last
    // workgroup compared against the id of an earlier launch: the fallback could not fire when it was needed.)
    __hip_atomic_store(area + part, bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(iarea + kSamplerMaxParts + part, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(iarea + 2 * kSamplerMaxParts + 1 + part, (int)(xcc & 0xf), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // acknowledged before the arrival is counted
    *s_last = __hip_atomic_fetch_add(iarea + 2 * kSamplerMaxParts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == W - 1;
  }
  __syncthreads();
  if (!*s_last) return 0;
  if (tid == 0)  // for the next launch (an agent-scope store: the counter lives at the memory side)
    __hip_atomic_store(iarea + 2 * kSamplerMaxParts, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bv = -INFINITY;
  bi = 0x7fffffff;
  bool same_l2 = true;
  for (int w = 0; w < W; ++w) {
    const float v = __hip_atomic_load(area + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int i = __hip_atomic_load(iarea + kSamplerMaxParts + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int x = __hip_atomic_load(iarea + 2 * kSamplerMaxParts + 1 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    same_l2 = same_l2 && x == (int)(xcc & 0xf);
    if (i != 0x7fffffff && (v > bv || (v == bv && i < bi) || bi == 0x7fffffff)) { bv = v; bi = i; }
  }
  return same_l2 ? 1 : 2;
}

// ln(q) of the exponential race: q = -ln(u), u uniform in the OPEN interval (0, 1).  23 random bits + 0.5 is exact in
// fp32 (24 significant bits), so u lies in [2^-24, 1 - 2^-24]: with 24 bits, (r >> 8) + 0.5 = 16 777 215.5 rounded to
// 2^24, u was 1.0 once in 2^24 draws, q = 0 and the score +inf -- that token won whatever its probability.
// The clamp keeps the score finite whatever v_log_f32 returns at the ends.
__device__ __forceinline__ float draw_uniform(const uint32_t r) {
  return ((float)(r >> 9) + 0.5f) * 1.1920928955078125e-07f;  // 2^-23
}
__device__ __forceinline__ float draw_neg_log_q(const uint32_t r) {
  const float q = fmaxf(-__logf(draw_uniform(r)), 1.17549435e-38f);
  return __logf(q);
}

// The draw over the 4-value groups [first, first + n) of a working row: min-p, then the arg-max over the kept tokens of
// (x - m) - log(q), q ~ Exp(1) (Philox4x32-10, counter = (group, step)); partial result in (best, besti).
__device__ __forceinline__ void draw_range(const float* __restrict__ x_row, const int vocab, const int first, const int n,
                                           const float m, const uint32_t cut_key, const float min_p, const uint64_t seed,
                                           const uint32_t step, float* __restrict__ prow, float& best, int& besti) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const bool x_al = aligned16(x_row);
  for_each_in_row<4>(n, [&](int i4) { return load_group<float>(x_row, i4, vocab, x_al, -INFINITY); },
                     [&](Group<float> f, int i4) {
    bool any = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i4 * 4 + e;
      float x = f.v[e];
      bool keep = order_key(x) >= cut_key && x > -INFINITY;
      if (keep && min_p > 0.f) keep = !(expf(x - m) < min_p);
      if (!keep) x = -INFINITY;
      if (prow != nullptr && i < vocab) prow[i] = x;
      f.v[e] = x;
      any = any || keep;
    }
    if (!any) return;  // random numbers only where a token is still in the race (top-k 50: 50 of 128 k)
    uint32_t r[4];
    philox4x32_10((uint32_t)i4, step, k0, k1, r);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (f.v[e] > -INFINITY) {
        // arg-max of p_i / q_i, q_i = -ln(u_i) ~ Exp(1), as x_i - ln(q_i); the hardware logarithm (v_log_f32) is
        // this kernel's own definition of its random stream: nothing outside compares these bits
        const float sc = (f.v[e] - m) - draw_neg_log_q(r[e]);
        const int i = i4 * 4 + e;
        if (sc > best || (sc == best && i < besti)) { best = sc; besti = i; }
      }
    }
  }, first);
}

// SPLIT_DRAW: the draw is a launch of its own (sampler_draw_kernel), shared by the workgroups of a row like the first
// pass; this kernel leaves (maximum, cut, arg-max) of the rows that need one in the row's tail.
template <typename LT, bool SPLIT_DRAW>
__global__ __launch_bounds__(kSamplerThreads) void sampler_kernel(
    int64_t* __restrict__ tokens_out, const LT* __restrict__ logits, const int64_t logits_stride, const int vocab,
    const int32_t* __restrict__ state_slot, SamplerParams* __restrict__ params, int32_t* __restrict__ counts,
    const int64_t counts_stride, const int num_slots, float* __restrict__ scratch, const int64_t scratch_stride,
    float* __restrict__ processed_out, const int64_t processed_stride, const int update_state) {
  __shared__ uint64_t hist[2048];
  constexpr int kSurvivors = 2048;  // values >= the k-th largest that top-p walks from LDS
  __shared__ __attribute__((aligned(16))) float survivors[kSurvivors];
  __shared__ int s_count;
  __shared__ float sv[16];
  __shared__ int si[16];
  __shared__ uint64_t red[16];
  __shared__ int s_bin;
  __shared__ uint64_t s_before;
  __shared__ bool s_found;

  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  const LT* lrow = logits + (int64_t)row * logits_stride;
  const int sid = state_slot != nullptr ? state_slot[row] : -1;
  // gridDim.y workgroups share the FIRST pass of a row (its 16-byte groups cut into gridDim.y ranges): one workgroup
  // = one CU does ~12 vector operations per element and pass whatever the memory delivers, 10-30 us for a 128 k row.
  // Each leaves its partial arg-max in the row's tail of `scratch`; the one that arrives last (an atomic counter
  // there, no waiting) merges them and carries on alone with the passes that need the whole row's maximum.
  // The workgroups of a row meet through ONE L2: the launcher asks for the split only when rows % 8 == 0, so that
  // workgroup (row, part) = linear id row + rows * part lands on XCD row % 8 for every part (workgroups go to the
  // XCDs round-robin); stores reach that L2 (the L1 writes through), nothing is flushed or invalidated -- as
  // device-scope fences the meeting cost 60 us (every workgroup wrote back and invalidated a whole L2).  Every part
  // records the XCD it ran on; if they ever differ the last workgroup does the whole pass again by itself.
  const int W = gridDim.y, part = blockIdx.y;
  constexpr int NL = Group<LT>::N;
  const int lgroups = (vocab + NL - 1) / NL;
  const int share = (lgroups + W - 1) / W;
  int g0 = part * share, ng = max(0, min(share, lgroups - g0));
  float* tail = W > 1 ? scratch + (int64_t)row * scratch_stride + ((vocab + 3) & ~3) : nullptr;
  __shared__ int s_last;
  auto merge_parts = [&](float& bv, int& bi) __attribute__((always_inline)) -> int {
    const int r = meet_parts(tail, W, part, bv, bi, &s_last);
    if (r == 2) { g0 = 0; ng = lgroups; }
    return r;
  };

  if (sid < 0 || sid >= num_slots) {  // plain greedy row: arg-max of the logits as they are
    float bv;
    int bi;
    bool redo = false;
    do {
      bv = -INFINITY;
      bi = 0x7fffffff;
      for_each_value<LT>(lrow, vocab, [&](LT v, int i) {
        const float x = logit_to_float<LT>(v);
        if (x > bv || (x == bv && i < bi) || bi == 0x7fffffff) { bv = x; bi = i; }
      }, g0, ng);
      block_argmax(bv, bi, sv, si);
      if (redo) break;  // (the second round covered the whole row)
      const int r = merge_parts(bv, bi);
      if (r == 0) return;
      redo = r == 2;
    } while (redo);
    if (processed_out != nullptr)
      for (int i = tid; i < vocab; i += kSamplerThreads)
        processed_out[(int64_t)row * processed_stride + i] = logit_to_float<LT>(lrow[i]);
    if (tid == 0) tokens_out[row] = bi;
    return;  // (no state slot: sampler_draw_kernel skips the row by that)
  }

  __shared__ SamplerParams P;  // one copy per workgroup (its banned[] list is indexed dynamically)
  if (tid < (int)(sizeof(SamplerParams) / 4)) ((uint32_t*)&P)[tid] = ((const uint32_t*)&params[sid])[tid];
  __syncthreads();
  int32_t* crow = counts + (int64_t)sid * counts_stride;
  float* x_row = scratch + (int64_t)row * scratch_stride;
  const bool greedy = P.temperature < kSamplingEps;
  const bool do_pen = P.presence != 0.f || P.frequency != 0.f || P.repetition != 1.f;
  const int nban = P.output_len < P.min_tokens ? min(P.num_banned, 20) : 0;
  const float temp = greedy ? 1.f : P.temperature;  // sampling_metadata.py: greedy rows divide by 1

  // ---- pass A: ban, penalties, temperature; the row's maximum ----
  float bv;
  int bi;
  bool redo = false;
  do {
    bv = -INFINITY;
    bi = 0x7fffffff;
  {
    constexpr int N = Group<LT>::N;  // 8 logits of 16 bits (4 of fp32) = N / 4 groups of counts and of working values
    struct Item { Group<LT> l; Group<int32_t> c[N / 4]; };
    const bool l_al = aligned16(lrow), c_al = aligned16(crow), x_al = aligned16(x_row);
    for_each_in_row<2>(ng, [&](int g) {
      Item it;
      it.l = load_group<LT>(lrow, g, vocab, l_al, LT{});
#pragma unroll
      for (int h = 0; h < N / 4; ++h)
        it.c[h] = do_pen ? load_group<int32_t>(crow, g * (N / 4) + h, vocab, c_al, 0) : Group<int32_t>{};
      return it;
    }, [&](const Item& it, int g) {
#pragma unroll
      for (int h = 0; h < N / 4; ++h) {
        Group<float> o;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int e = h * 4 + e4, i = g * N + e;
          float x = logit_to_float<LT>(it.l.v[e]);
          for (int b = 0; b < nban; ++b)
            if (P.banned[b] == i) x = -INFINITY;
          if (do_pen) {
            const uint32_t c = (uint32_t)it.c[h].v[e4];
            if (c != 0) x = x > 0.f ? x / P.repetition : x * P.repetition;
            const float oc = (float)(c & 0x7fffffffu);
            x -= P.frequency * oc;
            x -= P.presence * (oc > 0.f ? 1.f : 0.f);
          }
          x = x / temp;
          o.v[e4] = x;
          if (i < vocab && (x > bv || (x == bv && i < bi) || bi == 0x7fffffff)) { bv = x; bi = i; }
        }
        const int i0 = g * N + h * 4;
        if (x_al && i0 + 4 <= vocab) {
          *reinterpret_cast<Group<float>*>(x_row + i0) = o;
        } else {
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4)
            if (i0 + e4 < vocab) x_row[i0 + e4] = o.v[e4];
        }
      }
    }, g0);
  }
    block_argmax(bv, bi, sv, si);
    if (redo) break;  // (the second round covered the whole row)
    const int r = merge_parts(bv, bi);
    if (r == 0) return;
    redo = r == 2;
  } while (redo);
  const float m = bv;
  int token = bi;

  if (!greedy) {
    // ---- top-k: the key of the k-th largest value ----
    uint32_t kth_key = 0;  // everything is >= key 0
    if (P.top_k > 0 && P.top_k < vocab) {
      const uint32_t k = select_key<false>(x_row, vocab, m, 0u, (uint64_t)P.top_k, 0.f, hist, red, &s_bin, &s_before,
                                           &s_found);
      if (s_found) kth_key = k;
      __syncthreads();
    }
    // ---- top-p: the smallest key whose mass from the bottom (inside the top-k set) exceeds (1 - p) Z ----
    uint32_t cut_key = kth_key;
    if (P.top_p < 1.f) {
      // With top-k in front, top-p only concerns the few values >= the k-th: one pass collects them in LDS (any order:
      // the selection sums integers) and the three histogram passes walk that list instead of the vocabulary.  More
      // survivors than the list holds (ties at the k-th value, a huge k): the passes walk the row as before.
      const float* src = x_row;
      int n_src = vocab;
      if (kth_key != 0) {
        if (tid == 0) s_count = 0;
        __syncthreads();
        for_each_value<float>(x_row, vocab, [&](float x, int) {
          if (order_key(x) >= kth_key) {
            const int at = atomicAdd(&s_count, 1);
            if (at < kSurvivors) survivors[at] = x;
          }
        });
        __syncthreads();
        if (s_count <= kSurvivors) {
          src = survivors;
          n_src = s_count;
        }
      }
      const uint32_t c = select_key<true>(src, n_src, m, kth_key, 0, 1.f - P.top_p, hist, red, &s_bin, &s_before,
                                          &s_found);
      const uint32_t max_key = order_key(m);
      cut_key = s_found ? c : max_key;  // nothing exceeds the target: only the largest survives ("at least one")
      if (cut_key > max_key) cut_key = max_key;
      if (cut_key < kth_key) cut_key = kth_key;
      __syncthreads();
    }
    if constexpr (SPLIT_DRAW) {  // the draw is the next launch's: leave it what it needs
      if (tid == 0) {
        tail[kSamplerRecord + 0] = m;
        reinterpret_cast<uint32_t*>(tail)[kSamplerRecord + 1] = cut_key;
        reinterpret_cast<int*>(tail)[kSamplerRecord + 2] = 1;
        reinterpret_cast<int*>(tail)[kSamplerRecord + 3] = bi;
      }
      return;
    }
    // ---- min-p + the draw ----
    float best = -INFINITY;
    int besti = 0x7fffffff;
    draw_range(x_row, vocab, 0, (vocab + 3) / 4, m, cut_key, P.min_p, P.seed, (uint32_t)P.output_len,
               processed_out != nullptr ? processed_out + (int64_t)row * processed_stride : nullptr, best, besti);
    block_argmax(best, besti, sv, si);
    token = besti != 0x7fffffff ? besti : bi;
  } else if (processed_out != nullptr) {
    for (int i = tid; i < vocab; i += kSamplerThreads) processed_out[(int64_t)row * processed_stride + i] = x_row[i];
  }

  if (tid == 0) {
    if constexpr (SPLIT_DRAW) reinterpret_cast<int*>(tail)[kSamplerRecord + 2] = 0;  // (a greedy row: nothing to draw)
    tokens_out[row] = token;
    if (update_state) {  // the drawn token joins the request's output history
      crow[token] += 1;
      params[sid].output_len = P.output_len + 1;
    }
  }
}

// The draw of the rows sampler_kernel<.., true> left one for: gridDim.y workgroups per row, met like the first pass.
__global__ __launch_bounds__(kSamplerThreads) void sampler_draw_kernel(
    int64_t* __restrict__ tokens_out, const int vocab, const int32_t* __restrict__ state_slot,
    SamplerParams* __restrict__ params, int32_t* __restrict__ counts, const int64_t counts_stride, const int num_slots,
    float* __restrict__ scratch, const int64_t scratch_stride, float* __restrict__ processed_out,
    const int64_t processed_stride, const int update_state) {
  __shared__ float sv[16];
  __shared__ int si[16];
  __shared__ int s_last;
  const int row = blockIdx.x, W = gridDim.y, part = blockIdx.y, tid = threadIdx.x;
  const int sid = state_slot[row];
  if (sid < 0 || sid >= num_slots) return;
  float* x_row = scratch + (int64_t)row * scratch_stride;
  float* tail = x_row + ((vocab + 3) & ~3);
  if (reinterpret_cast<const int*>(tail)[kSamplerRecord + 2] != 1) return;
  const float m = tail[kSamplerRecord + 0];
  const uint32_t cut_key = reinterpret_cast<const uint32_t*>(tail)[kSamplerRecord + 1];
  const int bi = reinterpret_cast<const int*>(tail)[kSamplerRecord + 3];
  const float min_p = params[sid].min_p;
  const uint64_t seed = params[sid].seed;
  const int output_len = params[sid].output_len;
  const int groups = (vocab + 3) / 4;
  const int share = (groups + W - 1) / W;
  int g0 = part * share, ng = max(0, min(share, groups - g0));
  float* prow = processed_out != nullptr ? processed_out + (int64_t)row * processed_stride : nullptr;
  float best;
  int besti;
  bool redo = false;
  do {
    best = -INFINITY;
    besti = 0x7fffffff;
    draw_range(x_row, vocab, g0, ng, m, cut_key, min_p, seed, (uint32_t)output_len, prow, best, besti);
    block_argmax(best, besti, sv, si);
    if (redo) break;  // (the second round covered the whole row)
    const int r = meet_parts(tail + kSamplerDrawArea, W, part, best, besti, &s_last);
    if (r == 0) return;
    redo = r == 2;
    if (redo) { g0 = 0; ng = groups; }
  } while (redo);
  if (tid == 0) {
    const int token = besti != 0x7fffffff ? besti : bi;
    tokens_out[row] = token;
    if (update_state) {
      counts[(int64_t)sid * counts_stride + token] += 1;
      params[sid].output_len = output_len + 1;
    }
  }
}

// A state slot's vocabulary row from the request's histories: bit 31 = in the prompt, low bits = output occurrences.
__global__ void sampler_init_row_kernel(int32_t* __restrict__ crow, const int vocab, const int64_t* __restrict__ prompt,
                                        const int n_prompt, const int64_t* __restrict__ output, const int n_output) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_prompt) {
    const int64_t t = prompt[i];
    if (t >= 0 && t < vocab) atomicOr((unsigned int*)&crow[t], 0x80000000u);
  } else if (i < n_prompt + n_output) {
    const int64_t t = output[i - n_prompt];
    if (t >= 0 && t < vocab) atomicAdd((unsigned int*)&crow[t], 1u);
  }
}

}  // namespace lvllm

using namespace lvllm;

extern "C" int lvllm_sample_rows(int64_t* tokens_out, const void* logits, int64_t logits_stride, int logits_dtype,
                                 int num_rows, int vocab, const int32_t* state_slot, void* params, int32_t* counts,
                                 int64_t counts_stride, int num_slots, float* scratch, int64_t scratch_stride,
                                 float* processed_out, int64_t processed_stride, int update_state, void* stream) {
  LV_CHECK(num_rows >= 0 && vocab > 0, "bad shape");
  if (num_rows == 0) return 0;
  LV_CHECK(tokens_out != nullptr && logits != nullptr, "null tokens_out / logits");
  LV_CHECK(state_slot == nullptr || (params != nullptr && counts != nullptr && scratch != nullptr && num_slots > 0),
           "rows with state need params, counts and scratch");
  LV_CHECK(scratch == nullptr || scratch_stride >= vocab, "scratch rows shorter than the vocabulary");
  LV_CHECK(counts == nullptr || counts_stride >= vocab, "count rows shorter than the vocabulary");
  hipStream_t s = (hipStream_t)stream;
  auto* pp = (SamplerParams*)params;
  // workgroups per row for the first pass: as many as leave no CU idle, when the scratch rows have the tail for their
  // partial results (zero-initialised by the caller once; the kernel leaves it zeroed)
  static const int num_cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return n;
  }();
  int W = 1;
  // (rows % 8 == 0: the workgroups of a row then share an XCD and its L2, see the kernel)
  if (scratch != nullptr && scratch_stride >= ((vocab + 3) & ~3) + kSamplerRowTail && vocab >= 8192 &&
      num_rows % 8 == 0) {
    W = num_cus / num_rows;
    W = W < 1 ? 1 : (W > kSamplerMaxParts ? kSamplerMaxParts : W);
  }
  // the draw as a launch of its own, shared like the first pass, when rows may need one
  const bool split_draw = W > 1 && state_slot != nullptr;
#define LV_SAMPLE(LT)                                                                                                 \
  do {                                                                                                                \
    if (split_draw)                                                                                                   \
      hipLaunchKernelGGL((sampler_kernel<LT, true>), dim3(num_rows, W), dim3(kSamplerThreads), 0, s, tokens_out,      \
                         (const LT*)logits, logits_stride, vocab, state_slot, pp, counts, counts_stride, num_slots,  \
                         scratch, scratch_stride, processed_out, processed_stride, update_state);                    \
    else                                                                                                              \
      hipLaunchKernelGGL((sampler_kernel<LT, false>), dim3(num_rows, W), dim3(kSamplerThreads), 0, s, tokens_out,     \
                         (const LT*)logits, logits_stride, vocab, state_slot, pp, counts, counts_stride, num_slots,  \
                         scratch, scratch_stride, processed_out, processed_stride, update_state);                    \
  } while (0)
  switch (logits_dtype) {
    case LVLLM_F32: LV_SAMPLE(float); break;
    case LVLLM_F16: LV_SAMPLE(F16Bits); break;
    case LVLLM_BF16: LV_SAMPLE(BF16Bits); break;
    default: LV_CHECK(false, "unsupported logits dtype");
  }
#undef LV_SAMPLE
  if (split_draw)
    hipLaunchKernelGGL(sampler_draw_kernel, dim3(num_rows, W), dim3(kSamplerThreads), 0, s, tokens_out, vocab, state_slot,
                       pp, counts, counts_stride, num_slots, scratch, scratch_stride, processed_out, processed_stride,
                       update_state);
  LV_LAUNCH_CHECK();
  return 0;
}

// Test hook: out[3 i .. 3 i + 2] = { u, -ln(u) clamped, ln(q) } of the draw arithmetic for the random word r[i]
// (draw_uniform / draw_neg_log_q above, the same device code the draw runs), so the ends of the range are checked on
// the hardware logarithm itself (tests/test_sampler_gpu.py).
__global__ void sampler_draw_probe_kernel(const uint32_t* __restrict__ r, float* __restrict__ out, const int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float u = draw_uniform(r[i]);
  out[3 * i] = u;
  out[3 * i + 1] = fmaxf(-__logf(u), 1.17549435e-38f);
  out[3 * i + 2] = draw_neg_log_q(r[i]);
}

extern "C" int lvllm_sampler_draw_probe(const uint32_t* r, float* out, int n, void* stream) {
  LV_CHECK(r != nullptr && out != nullptr && n >= 0, "bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(sampler_draw_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, r, out, n);
  LV_LAUNCH_CHECK();
  return 0;
}

extern "C" int lvllm_sampler_init_row(int32_t* counts_row, int vocab, const int64_t* prompt_tokens, int n_prompt,
                                      const int64_t* output_tokens, int n_output, void* stream) {
  LV_CHECK(counts_row != nullptr && vocab > 0 && n_prompt >= 0 && n_output >= 0, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(counts_row, 0, (size_t)vocab * 4, s) != hipSuccess) {
    set_error("lvllm_sampler_init_row: memset failed");
    return 2;
  }
  const int n = n_prompt + n_output;
  if (n > 0) {
    hipLaunchKernelGGL(sampler_init_row_kernel, dim3((n + 255) / 256), dim3(256), 0, s, counts_row, vocab, prompt_tokens,
                       n_prompt, output_tokens, n_output);
    LV_LAUNCH_CHECK();
  }
  return 0;
}
