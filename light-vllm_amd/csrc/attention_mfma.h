// Paged-attention decode for MI355X (gfx950 / CDNA4): paged_attention_v1, paged_attention_v2
// and the v2 partition reduce.
//
// WHAT (reference semantics, csrc/attention/attention_kernels.cu:86-496, 564-669):
//   for every (sequence, query head): softmax(scale * q.K^T (+ALiBi)) . V over the
//   sequence's KV blocks, addressed through block_tables; fp32 logits/softmax/
//   accumulation; the softmax normaliser is 1/(sum + 1e-6); v2 does this per
//   512-token partition, stores (max_logit, exp_sum, normalised partial out)
//   per partition and merges them in a second pass.
//
// HOW (MI355X-first, not the reference's launch shape):
//   * one workgroup per (kv head, sequence, partition) serves ALL query heads
//     of the GQA group (up to 16) from a single read of the K/V blocks -- the
//     reference launches one workgroup per *query* head and re-reads every KV
//     block H/KVH times.
//   * the paged K layout [D/8][BS][8] *is* the A-operand fragment layout of
//     v_mfma_f32_16x16x32_{bf16,f16} with rows = tokens: lane (g = lane>>4,
//     c = lane&15) needs K[token c][d = 32j + 8g .. +7], which is the 16-byte
//     chunk (d8 = 4j + g, token c) of the block.  Four fully coalesced
//     dwordx4 wave loads (1 KiB each) fetch a 16-token K tile straight into
//     MFMA operand registers; nothing is staged through LDS.
//   * S = K.Q^T lands with lane (g, c) holding tokens 4g..4g+3 of head c.  The
//     P.V product uses the same token->k-slot permutation on both operands
//     (k-slot 8g + e  <->  tile e>>2, token 4g + (e&3)), so the probabilities
//     never move between lanes: V is fetched as 8-byte pieces
//     V[d][4g..4g+3] (each wave load covers 512 contiguous bytes per tile).
//   * online softmax per wave (running max / sum, fp32), waves of a workgroup
//     own interleaved 32-token pairs and are merged once through LDS.
//   * two register sets per wave: the loads of pair i+1 are in flight while
//     pair i is multiplied (32 KiB per wave, >= 128 KiB per CU outstanding).
//
// Rounding: q.k products are exact bf16 x bf16 in fp32 accumulation; exp is
// __expf; probabilities are rounded to T before P.V as in the reference
// (attention_kernels.cu:398-400) -- here un-normalised exp values, the
// normaliser 1/(sum+1e-6) is applied in fp32 at the end.
#pragma once
#include "trace.h"
#include <float.h>

#include <type_traits>

#include "attention_params.h"
#include "common.h"

namespace lvllm {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// S = K.Q^T : 16 tokens x 16 heads, contraction over 32 head-dim elements
template <typename T>
__device__ __forceinline__ f32x4_t mfma_qk(u32x4_t a, u32x4_t b, f32x4_t c);
template <>
__device__ __forceinline__ f32x4_t mfma_qk<BF16>(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma_qk<F16>(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// O^T += V^T.P^T : 16 head-dim rows x 16 heads, contraction over 16 tokens
template <typename T>
__device__ __forceinline__ f32x4_t mfma_pv(u32x2_t a, u32x2_t b, f32x4_t c);
template <>
__device__ __forceinline__ f32x4_t mfma_pv<BF16>(u32x2_t a, u32x2_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a),
                                                   __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma_pv<F16>(u32x2_t a, u32x2_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, a),
                                               __builtin_bit_cast(f16x4_t, b), c, 0, 0, 0);
}

// two fp32 -> one dword of two T (round to nearest even): a single v_cvt_pk_{bf16,f16}_f32
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <>
__device__ __forceinline__ uint32_t pack2<BF16>(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
template <>
__device__ __forceinline__ uint32_t pack2<F16>(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
}

// max without the canonicalising v_max x,x,x that fmaxf() costs on values the compiler cannot
// prove quiet (MFMA results, selects).  max2() is v_med3_f32(a, b, +inf): a target intrinsic, so
// the compiler still sees a vector instruction reading its operands and inserts the wait
// states an MFMA result needs before a vector read.  vmax()/vmax3() are inline asm, which the
// hazard recogniser does not look into: they may only consume results of ordinary vector
// instructions, NEVER an MFMA accumulator directly.
__device__ __forceinline__ float max2(float a, float b) {
  return __builtin_amdgcn_fmed3f(a, b, __builtin_inff());
}
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// maximum over the four 16-lane rows of a wave (lanes c, c + 16, c + 32, c + 48), in every lane: two VALU swaps
// (v_permlane16_swap, v_permlane32_swap) instead of two trips through the LDS crossbar.  (Used by prefill_chunk.h.
// In the decode kernel below it -- with base-2 exponentials and a rescale skipped when no maximum moved -- changed
// nothing measurable: 22.96 against 22.96 us at the metric's shape, 14.3 against 14.5 with an fp8 cache, A/B of round 3,
// profiles/r03_tuning.md section 8; that kernel waits for memory, not for its softmax.)
__device__ __forceinline__ float rows_max(float m) {
  const uint32_t u = __builtin_bit_cast(uint32_t, m);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  m = fmaxf(__builtin_bit_cast(float, a[0]), __builtin_bit_cast(float, a[1]));
  const uint32_t w = __builtin_bit_cast(uint32_t, m);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__builtin_bit_cast(float, b[0]), __builtin_bit_cast(float, b[1]));
}

// gfx9-family raw buffer descriptor word 3 (32-bit data format, no swizzle)
constexpr int kSrdFlags = 0x00020000;
// cache policy of the K/V stream loads (buffer_load aux bits; 2 = nt: data read once)
#ifndef LVLLM_ATTN_AUX
#define LVLLM_ATTN_AUX 2  // measured: nt loads -8 % (profiles/r01_tuning.md)
#endif

// ---------------------------------------------------------------------------
// The kernel.  Unit of work = one 16-token tile of one kv head:
//   K tile  : NS  x buffer_load_dwordx4 per lane (1 KiB per wave instruction)
//   V tile  : NDT x buffer_load_dwordx2 per lane (512 B per wave instruction)
// Tiles of a partition are dealt round-robin to the NWAVES waves; each wave
// keeps NBUF tiles in flight.  All tile loads go through a per-tile buffer
// descriptor whose size is 0 for tiles past the end of the context: such
// loads fetch nothing and return zeros, so the loop body has no branches and
// the compiler's vmcnt waits stay exact (the next tiles remain in flight
// while one tile is multiplied).
// ---------------------------------------------------------------------------
#ifndef LVLLM_ATTN_MIN_WAVES_PER_SIMD
#define LVLLM_ATTN_MIN_WAVES_PER_SIMD 2
#endif
// ROPE instantiation: 1 = the rotation of Q is shared by the waves (wave j rotates fragment pair j once, everybody
// reads the result from LDS after the prologue's barrier); 0 = every wave rotates all of Q (round 2: ~800 vector
// instructions per wave, eight times over, before the first tile)
#ifndef LVLLM_ATTN_ROPE_SHARED_Q
#define LVLLM_ATTN_ROPE_SHARED_Q 1
#endif
// timing-diagnosis builds of the ROPE instantiation (WRONG results; tools/ab_rope_attn.sh): 1 = no rotation arithmetic,
// 2 = no cache stores, 4 = no new-token work at all (loads, rotation, stores, stash), 8 = no cos / sin loads
// 1: the new token's K / V row goes into the caches AFTER the key walk (from the LDS stash), not in the prologue: vmcnt
// retires in issue order, so every tile load the writing wave issued after those stores (128 scattered 2-byte
// writes of the V row among them) waited for their acknowledgement, and the workgroup's merge for that wave -- 2.5 us
// of a 26 us launch (diagnosis builds, profiles/r03_tuning.md section 10)
#ifndef LVLLM_ATTN_ROPE_LATE_STORES
#define LVLLM_ATTN_ROPE_LATE_STORES 1
#endif
#ifndef LVLLM_ATTN_ROPE_DIAG
#define LVLLM_ATTN_ROPE_DIAG 0
#endif
// 1: the ROPE instantiation runs one wave MORE than it has tile walkers, and that wave does nothing but the step's new
// token: rotate the key row, quantise, leave key and value in the LDS stash (before the prologue's barrier), then write
// both rows into the caches and exit.  It issues no tile load, so nothing queues behind its scattered cache stores
// (vmcnt retires in issue order: a walker's loads would), and the stores are acknowledged while the others walk the
// context instead of holding the workgroup's end.  Measured (round 4, profiles/r04_tuning.md section 9): nothing at
// bs 32 x 1 024 (24.2-24.3 against 24.2 us; fp8 17.1-17.2 against 16.8-17.0) and -9 % at bs 64 x 2 048 over an fp8 cache
// (nine waves put five on a SIMD when two workgroups share a CU: only one fits) -- so 0 (a tile walker writes the rows
// after its walk, round 3) stays the default; the switch remains for the record.
// 1: the new token's cache rows are written by the wave that owns the context's LAST tile, straight from the tile
// registers it has just patched with the new values -- the whole 16-token V tile of the kv head (full 32-byte sectors,
// 256 / 512 contiguous bytes per wave store) and the K chunks of the token and its sector partner -- instead of 16 + 128
// narrow stores (2-byte / 1-byte pieces of 128 different rows: every one a partial-sector write the memory side turns
// into a read-modify-write).  Only when the slot IS the last position of the context as the block table places it
// (what an engine passes); any other slot keeps the narrow stores.  0: always the narrow stores (round 3).
// Measured (round 4, profiles/r04_tuning.md section 11): bit-identical and SLOWER -- 25.1 against 24.2-24.5 us, over an
// fp8 cache 18.5 against 15.9: stores under a condition inside the tile body cost every tile its exact vmcnt waits.
// 2: the owner only stashes the patched tile in LDS inside the walk and the rows leave behind the merge as 16-byte
// stores of whole sectors: also slower (25.5 / 18.6 us) -- the ISA shows why: with either form in the tile body hipcc
// gives up the counted vmcnt waits of the loop (56 -> 21 waits, 14 of them vmcnt(0)).  Off; kept for the record.
#ifndef LVLLM_ATTN_ROPE_TILE_STORES
#define LVLLM_ATTN_ROPE_TILE_STORES 0
#endif
#ifndef LVLLM_ATTN_ROPE_KV_WAVE
#define LVLLM_ATTN_ROPE_KV_WAVE 0
#endif
// 4 fp8 (e4m3fn) of one dword -> 4 T in two dwords, each T(float(fp8) * scale).
// scale == 1 (`scaled` false, the common case): gfx950 converts two fp8 straight to two bf16 / f16 in ONE instruction
// (v_cvt_scalef32_pk_{bf16,f16}_fp8 with a scale of 1.0) -- exact, every e4m3 value is a bf16 and an f16 value -- half
// the vector instructions of the fp8 -> f32 -> T route, which the fp8 attention kernels are bound by.
template <typename T>
__device__ __forceinline__ uint32_t fp8x2_to_T(uint32_t w, bool upper);
template <>
__device__ __forceinline__ uint32_t fp8x2_to_T<BF16>(uint32_t w, bool upper) {
  return upper ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
               : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
}
template <>
__device__ __forceinline__ uint32_t fp8x2_to_T<F16>(uint32_t w, bool upper) {
  return upper ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true))
               : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false));
}
template <typename T>
__device__ __forceinline__ u32x2_t dequant4(uint32_t w, float scale, bool scaled) {
  if (!scaled) return u32x2_t{fp8x2_to_T<T>(w, false), fp8x2_to_T<T>(w, true)};
  f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
  f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
  lo *= scale;
  hi *= scale;
  return u32x2_t{pack2<T>(lo.x, lo.y), pack2<T>(hi.x, hi.y)};
}

// KV8: fp8 KV cache.  Same tiles, half the bytes: a K tile is NS/2 wave loads of 16 bytes per lane
// (chunk = 16 head-dim values of one token), each feeding TWO k-slices after conversion; the
// contraction only needs K and Q to agree on which d sits in which (slice, lane group, element),
// so Q is loaded with the permutation the fp8 chunks impose: slice 2i+h, group g, element e <->
// d = 64i + 16g + 8h + e.  A V piece is 4 bytes per lane.  Conversion is v_cvt_pk_f32_fp8 (+ scale)
// + v_cvt_pk_{bf16,f16}_f32: ~100 vector instructions per tile against ~7 us of HBM time per tile
// and wave at full bandwidth -- free.
// ROPE: see AttnParams (rotation of q and the new k, cache write of the new k and v, inside this launch).
// (head sizes up to 128: NWAVES + 1 waves put three on one SIMD, so the register bound becomes 3 per SIMD = 168
// registers, which the larger head sizes' accumulators do not fit without spilling)
template <bool ROPE, int D>
constexpr int attn_extra_waves() { return ROPE && D <= 128 && LVLLM_ATTN_ROPE_KV_WAVE != 0 ? 1 : 0; }
template <typename T, int D, int BS, int NWAVES, int NBUF, bool KV8, bool ROPE = false, bool SCALED = false>
__global__ __launch_bounds__((NWAVES + attn_extra_waves<ROPE, D>()) * 64,
                             ((attn_extra_waves<ROPE, D>()) && LVLLM_ATTN_MIN_WAVES_PER_SIMD < 3
                                  ? 3 : LVLLM_ATTN_MIN_WAVES_PER_SIMD)) void paged_attn_mfma_kernel(
    const AttnParams p) {
  using S = typename T::store_t;
  LVLLM_TRACE_BEGIN();
  static_assert(sizeof(S) == 2, "MFMA path is for 16-bit element types");
  static_assert(BS == 8 || BS == 16 || BS == 32, "a 16-token tile is one block, half a block, or two 8-token blocks");
  static_assert(BS != 8 || (!KV8 && !ROPE), "8-token blocks: 16-bit caches, separate rope / cache-write launches");
  static_assert(!(ROPE && KV8) || D % 128 == 0, "fused rotation over an fp8 cache: Q fragment j pairs with j + NSQ/2");
  static_assert(NBUF >= 2 && NBUF <= 6, "register sets per wave");
  constexpr int NS = (D + 31) / 32;   // k-slices of the QK product
  constexpr int NDT = (D + 15) / 16;  // 16-row d-tiles of the PV product
  constexpr int DPAD = NDT * 16;
  constexpr int KVB = KV8 ? 1 : 2;             // bytes per cache element
  constexpr int kHeadBytes = D * BS * KVB;     // one kv head's K (or V) bytes inside a block
  constexpr int NKL = KV8 ? (D + 63) / 64 : NS;  // K wave loads (16 bytes per lane) per tile
  constexpr int NSQ = KV8 ? 2 * NKL : NS;        // k-slices actually multiplied (fp8: two per load)
  static_assert(!KV8 || D % 16 == 0, "fp8 cache: head size must be a multiple of x = 16");
  static_assert(!ROPE || D % 64 == 0, "fused rotation: NeoX halves on k-slice boundaries");
  using vraw_t = typename std::conditional<KV8, uint32_t, u32x2_t>::type;  // V piece as loaded

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;

  const int G = p.num_heads / p.num_kv_heads;
  const int HG = (G + 15) >> 4;
  const int kvh = blockIdx.x / HG;
  const int hg = blockIdx.x - kvh * HG;
  const int head0 = kvh * G + hg * 16;
  const int nh = min(16, G - hg * 16);
  const int nh_lds = min(16, G);
  const int seq = blockIdx.y;
  const int part = blockIdx.z;
  const int seq_len = p.seq_lens[seq];

  int t0 = 0, t1 = seq_len;
  // empty share: nothing to do (the reference's early exit, attention_kernels.cu:116-119)
  if (p.partitioned && !split_range(seq_len, p.num_splits, part, &t0, &t1, p.split_tiles)) return;
  const int ntiles = (t1 - t0 + 15) >> 4;
  const int tile0 = t0 >> 4;
  // tiles of this wave: lt = wave + j * NWAVES, j = 0 .. nmy-1 (the new-token wave of the ROPE instantiation walks none)
  constexpr int kXW = attn_extra_waves<ROPE, D>();
  const bool kv_only = kXW != 0 && wave == NWAVES;
  const int nmy = (ntiles > wave && !kv_only) ? (ntiles - wave + NWAVES - 1) / NWAVES : 0;

  const int32_t* block_table = p.block_tables + (int64_t)seq * p.max_num_blocks_per_seq;
  const char* kbytes = (const char*)p.k_cache + (int64_t)kvh * p.kv_head_stride * KVB;
  const char* vbytes = (const char*)p.v_cache + (int64_t)kvh * p.kv_head_stride * KVB;
  const int64_t bsb = p.kv_block_stride * KVB;

  // per-lane byte offsets inside a (block, kv head) region
  // BS == 8: a tile is TWO blocks (tokens 0-7 | 8-15).  Every load is issued once per block with the lanes of the
  // other block pushed out of the descriptor's range (they fetch nothing and return zeros) and the two results
  // are OR-ed: the in-range lanes of each issue still read 512 (K) / 256 (V) contiguous bytes.
  constexpr int kOut = 0x40000000;  // beyond any (block, head) region
  const int koff = BS == 8 ? (g * 8 + (c & 7)) * 16 : (g * BS + c) * 16;  // K chunk (d8 = g (+4j), token c)
  const int voff = BS == 8 ? (c * 8 + ((4 * g) & 7)) * KVB : (c * BS + 4 * g) * KVB;  // V piece (row c (+16t), tokens 4g..4g+3)
  const bool in_a = BS != 8 || c < 8;    // K: this lane's token lies in the tile's first block
  const bool vin_a = BS != 8 || g < 2;   // V: this lane's 4 tokens lie in the tile's first block

  // Physical block number of this wave's j-th tile.  The index is wave-uniform, so
  // this is a scalar load; callers request it one rotation before it is needed.
  // The index is clamped by the table width (a kernel argument), not by the sequence length,
  // so the first block-table loads do not wait for the seq_lens load.
  const int last_block = p.max_num_blocks_per_seq - 1;
  // (BS == 8: two numbers per tile, packed low | high)
  auto block_number = [&](const int j) __attribute__((always_inline)) -> int64_t {
    const int blk = ((tile0 + wave + j * NWAVES) << 4) / BS;
    const uint32_t a = min((uint32_t)block_table[min(blk, last_block)], (uint32_t)p.max_block);
    if constexpr (BS == 8) {
      const uint32_t b = min((uint32_t)block_table[min(blk + 1, last_block)], (uint32_t)p.max_block);
      return (int64_t)(((uint64_t)b << 32) | a);
    }
    return (int64_t)a;
  };

  auto load_tile = [&](u32x4_t (&k)[NKL], vraw_t (&v)[NDT], const int j, const int64_t bnp)
                       __attribute__((always_inline)) {
    const int lt = wave + j * NWAVES;
    const bool valid = j < nmy;
    const int64_t bn = (int64_t)(uint32_t)bnp;
    const int tok_base = (tile0 + lt) << 4;
    const int off = (BS == 32) ? (tok_base & 16) : 0;  // second half of a 32-token block
    __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(kbytes + bn * bsb), 0, valid ? kHeadBytes : 0, kSrdFlags);
    __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(vbytes + bn * bsb), 0, valid ? kHeadBytes : 0, kSrdFlags);
    // chunks past the head size and rows with d >= D fall outside kHeadBytes -> zeros
    // (a chunk is 16 bytes of one token in both cache types: 8 T or 16 fp8)
    if constexpr (BS == 8) {
      const int64_t bn2 = (int64_t)((uint64_t)bnp >> 32);
      __amdgpu_buffer_rsrc_t kr2 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(kbytes + bn2 * bsb), 0, valid ? kHeadBytes : 0, kSrdFlags);
      __amdgpu_buffer_rsrc_t vr2 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(vbytes + bn2 * bsb), 0, valid ? kHeadBytes : 0, kSrdFlags);
      const int ka = in_a ? koff : kOut, kb = in_a ? kOut : koff;
      const int va = vin_a ? voff : kOut, vb = vin_a ? kOut : voff;
#pragma unroll
      for (int jj = 0; jj < NKL; ++jj) {
        const u32x4_t x = __builtin_amdgcn_raw_buffer_load_b128(kr, ka + jj * (4 * BS * 16), 0, LVLLM_ATTN_AUX);
        const u32x4_t y = __builtin_amdgcn_raw_buffer_load_b128(kr2, kb + jj * (4 * BS * 16), 0, LVLLM_ATTN_AUX);
        k[jj] = x | y;
      }
#pragma unroll
      for (int t = 0; t < NDT; ++t) {
        const u32x2_t x = __builtin_amdgcn_raw_buffer_load_b64(vr, va + t * (16 * BS * 2), 0, LVLLM_ATTN_AUX);
        const u32x2_t y = __builtin_amdgcn_raw_buffer_load_b64(vr2, vb + t * (16 * BS * 2), 0, LVLLM_ATTN_AUX);
        v[t] = x | y;
      }
      return;
    }
    // vmcnt retires in issue order and K is needed first, so K is requested first -- pinned: left alone, the
    // scheduler put the V requests in front, and the wait for K in front of K.Q^T became a wait for everything
    // (found in the ISA, round 3: -5 % on the fp8 launch, -3 % on the 16-bit one)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < NKL; ++jj)
      k[jj] = __builtin_amdgcn_raw_buffer_load_b128(kr, koff + jj * (4 * BS * 16), off * 16, LVLLM_ATTN_AUX);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NDT; ++t) {
      if constexpr (KV8)
        v[t] = __builtin_amdgcn_raw_buffer_load_b32(vr, voff + t * (16 * BS), off, LVLLM_ATTN_AUX);
      else
        v[t] = __builtin_amdgcn_raw_buffer_load_b64(vr, voff + t * (16 * BS * 2), off * 2, LVLLM_ATTN_AUX);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- ROPE: the first K/V tiles are requested before anything else: the block-table read and the HBM round trip
  // of tile 0 (and 1) then overlap the position -> cos / sin -> rotation -> cache-write prologue and its barrier
  // instead of following them (the tile that ends with the new token takes that token from the LDS stash, never
  // from the bytes this launch is writing) ----
  // NBUF register sets: set s holds tile s, s + NBUF, ...; bnr[s] the block number of the set's NEXT tile
  u32x4_t kset[NBUF][NKL];
  vraw_t vset[NBUF][NDT];
  int64_t bnr[NBUF];
#pragma unroll
  for (int s = 0; s < NBUF; ++s) bnr[s] = block_number(s);
  auto first_loads = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s + 1 < NBUF; ++s) {
      load_tile(kset[s], vset[s], s, bnr[s]);
      bnr[s] = block_number(s + NBUF);
    }
  };
  if constexpr (ROPE) {
    if (!kv_only) first_loads();  // (without the prologue the Q loads go first, as before: 25.0 against 25.4 us)
  }

  // ---- Q fragments (B operand of the QK product): Q[head c][d = 32j + 8g ..] ----
  constexpr bool kSharedQ = LVLLM_ATTN_ROPE_SHARED_Q != 0 && NSQ / 2 <= NWAVES && NSQ >= 2;
  // (ROPE: the new K / V row is the job of a wave that rotates no Q, when there is one)
  constexpr int kKvWave = kXW ? NWAVES : (kSharedQ && NSQ / 2 < NWAVES ? NSQ / 2 : 0);
  constexpr bool kLateStores = ROPE && LVLLM_ATTN_ROPE_LATE_STORES != 0;
  u32x4_t qf[NSQ];
  {
    const S* qrow = (const S*)p.q + (int64_t)seq * p.q_stride + (int64_t)(head0 + c) * D;
#pragma unroll
    for (int j = 0; j < NSQ; ++j) {
      // bf16/f16 cache: d = 32j + 8g ..; fp8 cache: d = 64(j>>1) + 16g + 8(j&1) .. (see KV8 above)
      const int d0 = KV8 ? 64 * (j >> 1) + 16 * g + 8 * (j & 1) : 32 * j + 8 * g;
      qf[j] = u32x4_t{0, 0, 0, 0};
      // (shared rotation: wave w loads the pair it rotates, fragments w and w + NSQ/2)
      const bool mine = !(ROPE && kSharedQ) || (j % (NSQ / 2 > 0 ? NSQ / 2 : 1)) == wave;
      if (mine && c < nh && d0 < D) qf[j] = *reinterpret_cast<const u32x4_t*>(qrow + d0);
    }
  }
  // ---- ROPE: rotate Q in registers; one wave rotates the new K, stores K and V to the caches and to LDS ----
  // NeoX pairs d with d + D/2: chunk j (d = 32j + 8g ..) pairs with chunk j + NS/2 of the SAME lane, and the
  // cos / sin of both are the 8 values at d mod D/2: the rotation never leaves the lane.
  S* sm_knew = nullptr;
  S* sm_vnew = nullptr;
  bool owns_new_token = false;  // this workgroup's share ends with the step's new token
  constexpr bool kTileStores = ROPE && LVLLM_ATTN_ROPE_LATE_STORES != 0 && LVLLM_ATTN_ROPE_TILE_STORES != 0;
  // mode 2: the last tile's owner only STASHES its patched tile in LDS inside the walk (no store under a condition in
  // the tile body); the rows leave behind the merge as 16-byte stores of whole sectors
  constexpr bool kTileStash = kTileStores && LVLLM_ATTN_ROPE_TILE_STORES == 2;
  constexpr int kTileRowBytes = 16 * KVB;              // one V row of the 16-token tile
  constexpr int kTileKChunks = D * KVB / 16;           // 16-byte K chunks per token
  char* tile_stash_v = nullptr;                        // [D rows][16 tokens]
  char* tile_stash_k = nullptr;                        // [chunk][2 tokens] x 16 bytes
  bool tile_store_ok = false;   // the new rows leave with the last tile's registers (LVLLM_ATTN_ROPE_TILE_STORES)
  if constexpr (ROPE) {
    constexpr int DPAD_ = ((D + 15) / 16) * 16;
    const int nh_l = min(16, G);
    // stash behind the merge area (the merge starts while other waves may still read the stash)
    sm_knew = reinterpret_cast<S*>(smem_raw + (size_t)NWAVES * 16 * 2 * sizeof(float) +
                                   (size_t)NWAVES * nh_l * DPAD_ * sizeof(float));
    sm_vnew = sm_knew + D;
    tile_stash_v = reinterpret_cast<char*>(sm_vnew + D) + (size_t)2 * ((D + 31) / 32) * 1024;
    tile_stash_k = tile_stash_v + D * kTileRowBytes;
    if (seq_len > 0) {
      const int64_t pos = p.positions[seq];
      const S* cosp = (const S*)p.cos_sin_cache + pos * D;
      const S* sinp = cosp + D / 2;
      auto rot8 = [&](u32x4_t& xv, u32x4_t& yv, const int d0) __attribute__((always_inline)) {
        if constexpr (LVLLM_ATTN_ROPE_DIAG & 8) return;
        const u32x4_t cv = *reinterpret_cast<const u32x4_t*>(cosp + d0);
        const u32x4_t sv = *reinterpret_cast<const u32x4_t*>(sinp + d0);
        if constexpr (LVLLM_ATTN_ROPE_DIAG & 1) { xv.x ^= cv.x & sv.x & 1u; return; }
        S* x = reinterpret_cast<S*>(&xv);
        S* y = reinterpret_cast<S*>(&yv);
        const S* cc = reinterpret_cast<const S*>(&cv);
        const S* ss = reinterpret_cast<const S*>(&sv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {  // the arithmetic of pos_encoding.hip rotate<T>, operation for operation
          const float xf = T::to_float(x[e]), yf = T::to_float(y[e]);
          const float cf = T::to_float(cc[e]), sf = T::to_float(ss[e]);
          const float xc = T::to_float(T::from_float(xf * cf));
          const float ys = T::to_float(T::from_float(yf * sf));
          const float yc = T::to_float(T::from_float(yf * cf));
          const float xs = T::to_float(T::from_float(xf * sf));
          x[e] = T::from_float(__fsub_rn(xc, ys));
          y[e] = T::from_float(__fadd_rn(yc, xs));
        }
      };
      // 16-bit cache: fragment j holds d = 32j + 8g ..; fp8 cache: d = 64(j>>1) + 16g + 8(j&1) .. (see KV8)
      if constexpr (kSharedQ) {
        // wave j < NSQ/2 rotates pair (j, j + NSQ/2) and leaves both fragments in LDS, [fragment][lane] x 16 bytes
        u32x4_t* sm_q = reinterpret_cast<u32x4_t*>(sm_vnew + D);
#pragma unroll
        for (int j = 0; j < NSQ / 2; ++j) {
          if (wave == j) {
            if (c < nh) rot8(qf[j], qf[j + NSQ / 2], KV8 ? 64 * (j >> 1) + 16 * g + 8 * (j & 1) : 32 * j + 8 * g);
            sm_q[j * 64 + lane] = qf[j];
            sm_q[(j + NSQ / 2) * 64 + lane] = qf[j + NSQ / 2];
          }
        }
      } else if (c < nh) {
#pragma unroll
        for (int j = 0; j < NSQ / 2; ++j)
          rot8(qf[j], qf[j + NSQ / 2], KV8 ? 64 * (j >> 1) + 16 * g + 8 * (j & 1) : 32 * j + 8 * g);
      }
      owns_new_token = !p.partitioned || t1 == seq_len;
      if constexpr (kTileStores) {
        if (owns_new_token) {  // (wave-uniform: every wave answers alike)
          int64_t slot = p.slot_mapping[seq];
          if (slot >= p.num_slots) slot = -1;
          const int P = seq_len - 1;
          const int blk = ((P >> 4) << 4) / BS;
          const int64_t bn = (int64_t)min((uint32_t)block_table[min(blk, last_block)], (uint32_t)p.max_block);
          tile_store_ok = slot >= 0 && slot / BS == bn && (int)(slot % BS) == (P % BS);
        }
      }
      if (kLateStores && owns_new_token && wave == kKvWave && !(LVLLM_ATTN_ROPE_DIAG & 4)) {
        // The new token's rows, ONE rotation pair (and two value elements) per lane: the whole workgroup waits at the
        // barrier below for this wave.  With a 16-byte chunk pair per lane it ran 16 rotations and -- over an fp8 cache --
        // 24 quantisations with their divisions in 8 .. 16 lanes, ~800 vector instructions that the diagnosis builds priced
        // at 1.2 us of the fused fp8 launch (round 4, profiles/r04_tuning.md section 10); per element the arithmetic is
        // the same (rotate<T> of pos_encoding.hip; fp8_kv_quant4 on one value), so the stash holds the same bytes.
        // (the cache rows themselves are written after the key walk, from the stash)
        const float ksc = SCALED ? p.k_scale : 1.0f, vsc = SCALED ? p.v_scale : 1.0f;  // (scale 1: the division folds away)
        const S* krow = (const S*)p.k_new + (int64_t)seq * p.k_new_stride + (int64_t)kvh * D;
        const S* vrow = (const S*)p.v_new + (int64_t)seq * p.v_new_stride + (int64_t)kvh * D;
#pragma unroll
        for (int i = lane; i < D / 2; i += 64) {
          S x = krow[i], y = krow[D / 2 + i];
          if constexpr (!(LVLLM_ATTN_ROPE_DIAG & 9)) {
            const float xf = T::to_float(x), yf = T::to_float(y);
            const float cf = T::to_float(cosp[i]), sf = T::to_float(sinp[i]);
            const float xc = T::to_float(T::from_float(xf * cf));
            const float ys = T::to_float(T::from_float(yf * sf));
            const float yc = T::to_float(T::from_float(yf * cf));
            const float xs = T::to_float(T::from_float(xf * sf));
            x = T::from_float(__fsub_rn(xc, ys));
            y = T::from_float(__fadd_rn(yc, xs));
          }
          if constexpr (KV8) {
            uint8_t* sk = reinterpret_cast<uint8_t*>(sm_knew);
            sk[i] = (uint8_t)fp8_kv_quant4(T::to_float(x), 0.f, 0.f, 0.f, ksc);
            sk[D / 2 + i] = (uint8_t)fp8_kv_quant4(T::to_float(y), 0.f, 0.f, 0.f, ksc);
          } else {
            sm_knew[i] = x;
            sm_knew[D / 2 + i] = y;
          }
        }
#pragma unroll
        for (int i = lane; i < D; i += 64) {
          const S v = vrow[i];
          if constexpr (KV8) reinterpret_cast<uint8_t*>(sm_vnew)[i] = (uint8_t)fp8_kv_quant4(T::to_float(v), 0.f, 0.f, 0.f, vsc);
          else sm_vnew[i] = v;
        }
      }
      if (!kLateStores && owns_new_token && wave == kKvWave && !(LVLLM_ATTN_ROPE_DIAG & 4)) {
        // lanes 0 .. D/16-1: one (x, y) chunk pair of the key row each; lanes 0 .. D/8-1: one chunk of the value row
        int64_t slot = p.slot_mapping[seq];
        if (slot >= p.num_slots) slot = -1;
        if constexpr (LVLLM_ATTN_ROPE_DIAG & 2) slot = -1;
        const int64_t blk = slot >= 0 ? slot / BS : 0;
        const int boff = slot >= 0 ? (int)(slot - blk * BS) : 0;
        if (lane < D / 16) {
          const S* krow = (const S*)p.k_new + (int64_t)seq * p.k_new_stride + (int64_t)kvh * D;
          u32x4_t kx = *reinterpret_cast<const u32x4_t*>(krow + 8 * lane);
          u32x4_t ky = *reinterpret_cast<const u32x4_t*>(krow + D / 2 + 8 * lane);
          rot8(kx, ky, 8 * lane);
          if constexpr (KV8) {
            // fp8 cache (x = 16): the rotated values -- already rounded to T, as the separate launches see them --
            // are quantised with fp8_kv_quant4; 8 values = 8 bytes of the 16-byte chunk d16 = d / 16
            auto q8 = [&](const u32x4_t& v) __attribute__((always_inline)) -> u32x2_t {
              const S* e = reinterpret_cast<const S*>(&v);
              u32x2_t r;
              r.x = fp8_kv_quant4(T::to_float(e[0]), T::to_float(e[1]), T::to_float(e[2]), T::to_float(e[3]), p.k_scale);
              r.y = fp8_kv_quant4(T::to_float(e[4]), T::to_float(e[5]), T::to_float(e[6]), T::to_float(e[7]), p.k_scale);
              return r;
            };
            const u32x2_t qx = q8(kx), qy = q8(ky);
            uint8_t* sk = reinterpret_cast<uint8_t*>(sm_knew);
            *reinterpret_cast<u32x2_t*>(sk + 8 * lane) = qx;
            *reinterpret_cast<u32x2_t*>(sk + D / 2 + 8 * lane) = qy;
            if (slot >= 0 && !kLateStores) {
              uint8_t* kc8 = (uint8_t*)p.k_cache + (blk * p.kv_block_stride + (int64_t)kvh * p.kv_head_stride) + boff * 16;
              const int dx = 8 * lane, dy = D / 2 + 8 * lane;
              *reinterpret_cast<u32x2_t*>(kc8 + (int64_t)(dx / 16) * BS * 16 + (dx % 16)) = qx;
              *reinterpret_cast<u32x2_t*>(kc8 + (int64_t)(dy / 16) * BS * 16 + (dy % 16)) = qy;
            }
          } else {
            *reinterpret_cast<u32x4_t*>(sm_knew + 8 * lane) = kx;
            *reinterpret_cast<u32x4_t*>(sm_knew + D / 2 + 8 * lane) = ky;
            if (slot >= 0 && !kLateStores) {
              S* kc = (S*)p.k_cache + (blk * p.kv_block_stride + (int64_t)kvh * p.kv_head_stride) + boff * 8;
              *reinterpret_cast<u32x4_t*>(kc + (int64_t)lane * BS * 8) = kx;
              *reinterpret_cast<u32x4_t*>(kc + (int64_t)(D / 16 + lane) * BS * 8) = ky;
            }
          }
        }
        if (lane < D / 8) {
          const S* vrow = (const S*)p.v_new + (int64_t)seq * p.v_new_stride + (int64_t)kvh * D;
          const u32x4_t vv = *reinterpret_cast<const u32x4_t*>(vrow + 8 * lane);
          const S* ve = reinterpret_cast<const S*>(&vv);
          if constexpr (KV8) {
            u32x2_t qv;
            qv.x = fp8_kv_quant4(T::to_float(ve[0]), T::to_float(ve[1]), T::to_float(ve[2]), T::to_float(ve[3]), p.v_scale);
            qv.y = fp8_kv_quant4(T::to_float(ve[4]), T::to_float(ve[5]), T::to_float(ve[6]), T::to_float(ve[7]), p.v_scale);
            uint8_t* sv = reinterpret_cast<uint8_t*>(sm_vnew);
            *reinterpret_cast<u32x2_t*>(sv + 8 * lane) = qv;
            if (slot >= 0 && !kLateStores) {
              uint8_t* vc8 = (uint8_t*)p.v_cache + (blk * p.kv_block_stride + (int64_t)kvh * p.kv_head_stride) +
                             (int64_t)(8 * lane) * BS + boff;
              const uint8_t* qb = reinterpret_cast<const uint8_t*>(&qv);
#pragma unroll
              for (int e = 0; e < 8; ++e) vc8[(int64_t)e * BS] = qb[e];
            }
          } else {
            *reinterpret_cast<u32x4_t*>(sm_vnew + 8 * lane) = vv;
            if (slot >= 0 && !kLateStores) {
              S* vc = (S*)p.v_cache + (blk * p.kv_block_stride + (int64_t)kvh * p.kv_head_stride) +
                      (int64_t)(8 * lane) * BS + boff;
#pragma unroll
              for (int e = 0; e < 8; ++e) vc[(int64_t)e * BS] = ve[e];
            }
          }
        }
      }
    }
    __syncthreads();  // the stash is visible to the wave that owns the last tile
    if constexpr (kSharedQ) {
      if (seq_len > 0) {
        const u32x4_t* sm_q = reinterpret_cast<const u32x4_t*>(sm_vnew + D);
#pragma unroll
        for (int j = 0; j < NSQ; ++j) qf[j] = sm_q[j * 64 + lane];
      }
    }
  }
  // fp8 caches: scales other than 1 take their own instantiation (SCALED).  As a run-time flag the choice became a
  // branch around EVERY conversion of the unrolled loop -- two arms with an s_waitcnt vmcnt(0) in each, 32 branches
  // per tile (found in the ISA, round 3: profiles/r03_tuning.md section 1)
  constexpr bool k_scaled = KV8 && SCALED, v_scaled = KV8 && SCALED;
  const float alibi = (p.alibi_slopes != nullptr && c < nh) ? p.alibi_slopes[head0 + c] : 0.f;

  // running softmax state of this wave: column c of lanes (g, c) is head head0 + c
  float m_run = -FLT_MAX;
  float l_run = 0.f;  // per-lane partial sum over its own tokens
  f32x4_t acc[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // One tile: S = K.Q^T (NS MFMAs), online softmax, O^T += V^T.P^T (NDT MFMAs).
  auto compute_tile = [&](u32x4_t (&kraw)[NKL], vraw_t (&v)[NDT], const int j)
                          __attribute__((always_inline)) {
    if constexpr (ROPE) {
      // the tile that ends with the new token: its K chunks and V elements come from the stash, not from the
      // cache (whose bytes at that slot this launch is only now writing)
      const int P = seq_len - 1;
      if (owns_new_token && (tile0 + wave + j * NWAVES) == (P >> 4)) {
        const int cs = P & 15;
        if constexpr (KV8) {
          // fp8 tile: chunk (d16 = 4jj + g, token c) is 16 bytes; a V piece 4 bytes (tokens 4g..4g+3 of row 16t + c)
          const uint8_t* sk = reinterpret_cast<const uint8_t*>(sm_knew);
          const uint8_t* sv = reinterpret_cast<const uint8_t*>(sm_vnew);
          if (c == cs) {
#pragma unroll
            for (int jj = 0; jj < NKL; ++jj)
              if ((4 * jj + g) * 16 < D) kraw[jj] = *reinterpret_cast<const u32x4_t*>(sk + (4 * jj + g) * 16);
          }
          if (g == (cs >> 2)) {
            const int sh = 8 * (cs & 3);
#pragma unroll
            for (int t = 0; t < NDT; ++t) v[t] = (v[t] & ~(0xffu << sh)) | ((uint32_t)sv[16 * t + c] << sh);
          }
        } else {
          if (c == cs) {
#pragma unroll
            for (int jj = 0; jj < NS; ++jj)
              kraw[jj] = *reinterpret_cast<const u32x4_t*>(sm_knew + (4 * jj + g) * 8);
          }
          if (g == (cs >> 2)) {
            const int e = cs & 3;
#pragma unroll
            for (int t = 0; t < NDT; ++t) {
              const uint32_t nv = sm_vnew[16 * t + c];
              uint32_t w = (e & 2) ? v[t].y : v[t].x;
              w = (e & 1) ? ((w & 0x0000ffffu) | (nv << 16)) : ((w & 0xffff0000u) | nv);
              if (e & 2) v[t].y = w; else v[t].x = w;
            }
          }
        }
        if constexpr (kTileStash) {
          if (tile_store_ok) {
#pragma unroll
            for (int t = 0; t < NDT; ++t)
              *reinterpret_cast<vraw_t*>(tile_stash_v + (16 * t + c) * kTileRowBytes + g * (int)sizeof(vraw_t)) = v[t];
            if ((c | 1) == (cs | 1)) {
#pragma unroll
              for (int jj = 0; jj < NKL; ++jj)
                if (4 * jj + g < kTileKChunks)
                  *reinterpret_cast<u32x4_t*>(tile_stash_k + ((4 * jj + g) * 2 + (c & 1)) * 16) = kraw[jj];
            }
          }
        } else if constexpr (kTileStores) {
          if (tile_store_ok) {
            // the patched tile goes back to the cache: V whole (every lane's piece: full sectors), K the chunks of the
            // token and of its sector partner (tokens 2i, 2i + 1 share 32 bytes); bytes of other tokens are rewritten
            // with what this launch loaded from them
            const int64_t bn = (int64_t)min((uint32_t)block_table[min(((P >> 4) << 4) / BS, last_block)], (uint32_t)p.max_block);
            const int off = (BS == 32) ? (((P >> 4) << 4) & 16) : 0;
            __amdgpu_buffer_rsrc_t kw = __builtin_amdgcn_make_buffer_rsrc((void*)(kbytes + bn * bsb), 0, kHeadBytes, kSrdFlags);
            __amdgpu_buffer_rsrc_t vw = __builtin_amdgcn_make_buffer_rsrc((void*)(vbytes + bn * bsb), 0, kHeadBytes, kSrdFlags);
            const int kst = ((c | 1) == (cs | 1)) ? koff : kOut;
#pragma unroll
            for (int jj = 0; jj < NKL; ++jj)
              __builtin_amdgcn_raw_buffer_store_b128(kraw[jj], kw, kst + jj * (4 * BS * 16), off * 16, 0);
#pragma unroll
            for (int t = 0; t < NDT; ++t) {
              if constexpr (KV8) __builtin_amdgcn_raw_buffer_store_b32(v[t], vw, voff + t * (16 * BS), off, 0);
              else __builtin_amdgcn_raw_buffer_store_b64(v[t], vw, voff + t * (16 * BS * 2), off * 2, 0);
            }
          }
        }
      }
    }
    f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (KV8) {
#pragma unroll
      for (int jj = 0; jj < NSQ; ++jj) {
        const u32x4_t w = kraw[jj >> 1];
        const u32x2_t a = dequant4<T>((jj & 1) ? w.z : w.x, p.k_scale, k_scaled);
        const u32x2_t b = dequant4<T>((jj & 1) ? w.w : w.y, p.k_scale, k_scaled);
        s = mfma_qk<T>(u32x4_t{a.x, a.y, b.x, b.y}, qf[jj], s);
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < NS; ++jj) s = mfma_qk<T>(kraw[jj], qf[jj], s);
    }
    // lane (g, c): logits of tokens tok0 .. tok0+3 for head c
    const int tok0 = ((tile0 + wave + j * NWAVES) << 4) + 4 * g;
    float x[4];
    float m_loc = -FLT_MAX;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y = s[r] * p.scale;
      y += (alibi != 0.f) ? alibi * (float)(tok0 + r - seq_len + 1) : 0.f;
      y = (tok0 + r < t1) ? y : -FLT_MAX;  // masked / out-of-context tokens
      x[r] = y;
      m_loc = fmaxf(m_loc, y);
    }
    m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 16));
    m_loc = fmaxf(m_loc, __shfl_xor(m_loc, 32));
    const float m_new = fmaxf(m_run, m_loc);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      x[r] = (tok0 + r < t1) ? __expf(x[r] - m_new) : 0.f;
      psum += x[r];
    }
    l_run = l_run * alpha + psum;
    u32x2_t pb;  // B operand: P[token 4g + r][head c], rounded to T (attention_kernels.cu:398-400)
    pb.x = pack2<T>(x[0], x[1]);
    pb.y = pack2<T>(x[2], x[3]);
    // V of tokens beyond the context may hold anything (NaN included): zero it,
    // as the reference does for the last block (attention_kernels.cu:420-430).
    uint32_t mx = 0xffffffffu, my = 0xffffffffu;
    if (tok0 + 0 >= t1) mx &= 0xffff0000u;
    if (tok0 + 1 >= t1) mx &= 0x0000ffffu;
    if (tok0 + 2 >= t1) my &= 0xffff0000u;
    if (tok0 + 3 >= t1) my &= 0x0000ffffu;
#pragma unroll
    for (int t = 0; t < NDT; ++t) {
      acc[t] *= alpha;
      u32x2_t va;
      if constexpr (KV8) va = dequant4<T>(v[t], p.v_scale, v_scaled);
      else va = v[t];
      va.x &= mx;
      va.y &= my;
      acc[t] = mfma_pv<T>(va, pb, acc[t]);  // rows: d = 16t + (lane&15); cols: head
    }
  };

  // ---- main loop: NBUF register sets rotate; NBUF-1 tiles stay in flight ----
  // Before tile j + u (set u) is multiplied, the loads of tile j + u + NBUF - 1 go into the set tile j + u - 1 has
  // just left.  Tiles past the wave's last one load nothing (zero-size descriptors) and multiply zeros against
  // masked logits: the running maximum, sum and accumulators are unchanged by them.
  {
    if constexpr (!ROPE) first_loads();
    for (int j = 0; j < nmy; j += NBUF) {
#pragma unroll
      for (int u = 0; u < NBUF; ++u) {
        constexpr int kSets = NBUF;
        const int s = (u + kSets - 1) % kSets;
        load_tile(kset[s], vset[s], j + u + NBUF - 1, bnr[s]);
        bnr[s] = block_number(j + u + 2 * NBUF - 1);
        compute_tile(kset[u], vset[u], j + u);
      }
    }
  }

  // ---- ROPE: the new token's rows go into the caches now, from the stash (see LVLLM_ATTN_ROPE_LATE_STORES) -- behind
  // the merge, not in front of it (round 4): the merge's __syncthreads() waits for every outstanding store of a wave
  // (vmcnt(0) before s_barrier), so with the rows written after the walk the whole workgroup's merge waited ~0.8 us for
  // the acknowledgement of 144 scattered stores; here they travel together with the result's own stores ----
  auto store_new_rows = [&]() __attribute__((always_inline)) {
  if constexpr (kTileStash) {
    if (owns_new_token && tile_store_ok && wave == kKvWave && !(LVLLM_ATTN_ROPE_DIAG & 6)) {
      const int P = seq_len - 1;
      const int64_t slot = p.slot_mapping[seq];  // (tile_store_ok: valid and the context's last position)
      const int64_t bn = slot / BS;
      const int off = (BS == 32) ? (((P >> 4) << 4) & 16) : 0;
      char* vdst = (char*)p.v_cache + (bn * p.kv_block_stride + (int64_t)kvh * p.kv_head_stride) * KVB + off * KVB;
      char* kdst = (char*)p.k_cache + (bn * p.kv_block_stride + (int64_t)kvh * p.kv_head_stride) * KVB;
#pragma unroll
      for (int i = lane; i < D * kTileRowBytes / 16; i += 64) {
        const int r = (i * 16) / kTileRowBytes, h = (i * 16) % kTileRowBytes;
        *reinterpret_cast<u32x4_t*>(vdst + (int64_t)r * BS * KVB + h) = *reinterpret_cast<const u32x4_t*>(tile_stash_v + i * 16);
      }
      const int tok0 = (P % BS) & ~1;
      if (lane < 2 * kTileKChunks)
        *reinterpret_cast<u32x4_t*>(kdst + ((int64_t)(lane >> 1) * BS + tok0 + (lane & 1)) * 16) =
            *reinterpret_cast<const u32x4_t*>(tile_stash_k + lane * 16);
    }
  }
  if constexpr (kLateStores) {
    if (owns_new_token && !tile_store_ok && wave == kKvWave && !(LVLLM_ATTN_ROPE_DIAG & 6)) {
      int64_t slot = p.slot_mapping[seq];
      if (slot >= p.num_slots) slot = -1;
      if (slot >= 0) {
        const int64_t blk = slot / BS;
        const int boff = (int)(slot - blk * BS);
        if constexpr (KV8) {
          const uint8_t* sk = reinterpret_cast<const uint8_t*>(sm_knew);
          const uint8_t* sv = reinterpret_cast<const uint8_t*>(sm_vnew);
          if (lane < D / 16) {
            uint8_t* kc8 = (uint8_t*)p.k_cache + (blk * p.kv_block_stride + (int64_t)kvh * p.kv_head_stride) + boff * 16;
            const int dx = 8 * lane, dy = D / 2 + 8 * lane;
            *reinterpret_cast<u32x2_t*>(kc8 + (int64_t)(dx / 16) * BS * 16 + (dx % 16)) =
                *reinterpret_cast<const u32x2_t*>(sk + 8 * lane);
            *reinterpret_cast<u32x2_t*>(kc8 + (int64_t)(dy / 16) * BS * 16 + (dy % 16)) =
                *reinterpret_cast<const u32x2_t*>(sk + D / 2 + 8 * lane);
          }
          if (lane < D / 8) {
            uint8_t* vc8 = (uint8_t*)p.v_cache + (blk * p.kv_block_stride + (int64_t)kvh * p.kv_head_stride) +
                           (int64_t)(8 * lane) * BS + boff;
            const u32x2_t qv = *reinterpret_cast<const u32x2_t*>(sv + 8 * lane);
            const uint8_t* qb = reinterpret_cast<const uint8_t*>(&qv);
#pragma unroll
            for (int e = 0; e < 8; ++e) vc8[(int64_t)e * BS] = qb[e];
          }
        } else {
          if (lane < D / 16) {
            S* kc = (S*)p.k_cache + (blk * p.kv_block_stride + (int64_t)kvh * p.kv_head_stride) + boff * 8;
            *reinterpret_cast<u32x4_t*>(kc + (int64_t)lane * BS * 8) = *reinterpret_cast<const u32x4_t*>(sm_knew + 8 * lane);
            *reinterpret_cast<u32x4_t*>(kc + (int64_t)(D / 16 + lane) * BS * 8) =
                *reinterpret_cast<const u32x4_t*>(sm_knew + D / 2 + 8 * lane);
          }
          if (lane < D / 8) {
            S* vc = (S*)p.v_cache + (blk * p.kv_block_stride + (int64_t)kvh * p.kv_head_stride) +
                    (int64_t)(8 * lane) * BS + boff;
            const u32x4_t vv = *reinterpret_cast<const u32x4_t*>(sm_vnew + 8 * lane);
            const S* ve = reinterpret_cast<const S*>(&vv);
#pragma unroll
            for (int e = 0; e < 8; ++e) vc[(int64_t)e * BS] = ve[e];
          }
        }
      }
    }
  }
  };
  if constexpr (kXW != 0) {
    if (kv_only) {
      store_new_rows();
      return;
    }
  }
  // ---- merge the waves of the workgroup -------------------------------------
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);

  float* sm_m = reinterpret_cast<float*>(smem_raw);  // [NWAVES][16]
  float* sm_l = sm_m + NWAVES * 16;                  // [NWAVES][16]
  float* sm_acc = sm_l + NWAVES * 16;                // [NWAVES][nh_lds][DPAD]
  if (g == 0) {
    sm_m[wave * 16 + c] = m_run;
    sm_l[wave * 16 + c] = l_run;
  }
  if (c < nh) {
    float* dst = sm_acc + ((int64_t)(wave * nh_lds + c)) * DPAD + 4 * g;
#pragma unroll
    for (int t = 0; t < NDT; ++t)
      *reinterpret_cast<f32x4_t*>(dst + 16 * t) = acc[t];  // acc[t][r]: d = 16t + 4g + r
  }
  __syncthreads();

  const int P = p.partitioned ? p.max_num_partitions : 1;
  for (int idx = threadIdx.x; idx < nh * D; idx += NWAVES * 64) {
    const int h = idx / D, d = idx - h * D;
    float M = -FLT_MAX;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) M = fmaxf(M, sm_m[w * 16 + h]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
      const float f = __expf(sm_m[w * 16 + h] - M);
      L += sm_l[w * 16 + h] * f;
      o += sm_acc[(w * nh_lds + h) * DPAD + d] * f;
    }
    o *= __fdividef(1.f, L + 1e-6f);
    const int64_t row = ((int64_t)seq * p.num_heads + head0 + h) * P + (p.partitioned ? part : 0);
    const S ot = T::from_float(o);
    reinterpret_cast<S*>(p.out)[row * D + d] = ot;
    if (p.out_fp8 != nullptr)  // (the host allows the twin only for single-pass launches: `out` is the final result)
      p.out_fp8[row * D + d] = (uint8_t)fp8_act_quant4(T::to_float(ot), 0.f, 0.f, 0.f, 1.0f / p.out_fp8_scale[0]);
    if (p.partitioned && d == 0) {
      p.max_logits[row] = M;
      p.exp_sums[row] = L;
    }
  }
  store_new_rows();
  LVLLM_TRACE_END(2);
}

// ---- host side: instantiation ladder (head size x block size x waves) ----
#ifndef LVLLM_ATTN_NBUF
#define LVLLM_ATTN_NBUF 2  // measured: 2 sets beat 3 by 13 % at bs32/seq1024 (profiles/r01_tuning.md)
#endif
// fp8 caches take their own count.  Measured, round 3 (profiles/r03_tuning.md section 1): EVERY added set made the fp8
// launch slower (2 / 3 / 4 sets: 15.3 / 17.4 / 17.8 us), so did 16 waves per workgroup and two-tile softmax steps with
// two or three pairs in flight (16.7 / 22.4 us): about 64 KiB in flight per CU cover the HBM latency, more backs up
// the memory pipeline, and a wave that cannot issue its next request cannot issue the arithmetic behind it either.
#ifndef LVLLM_ATTN_NBUF_KV8
#define LVLLM_ATTN_NBUF_KV8 2
#endif

template <typename T, int D, int BS, int NWAVES>
static void launch_mfma(const AttnParams& p, int num_seqs, int num_parts, hipStream_t stream) {
  const int G = p.num_heads / p.num_kv_heads;
  const int HG = (G + 15) / 16;
  const int nh_lds = G < 16 ? G : 16;
  constexpr int DPAD = ((D + 15) / 16) * 16;
  const size_t smem = (size_t)NWAVES * 16 * 2 * sizeof(float) +
                      (size_t)NWAVES * nh_lds * DPAD * sizeof(float) +
                      (p.positions != nullptr ? (size_t)2 * D * 2 + (size_t)2 * ((D + 31) / 32) * 1024 + (size_t)D * 32 + (size_t)D * 4 : 0);  // ROPE: the new
                                                                  // token's k and v, the rotated Q fragments (<= 2 NS KiB)
  auto launch = [&](auto kern) {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    // (the ROPE instantiation brings its new-token wave: attn_extra_waves)
    const int threads = (NWAVES + (p.positions != nullptr ? attn_extra_waves<true, D>() : 0)) * 64;
    hipLaunchKernelGGL(kern, dim3(p.num_kv_heads * HG, num_seqs, num_parts), dim3(threads), smem, stream, p);
  };
  if constexpr (D % 128 == 0 && BS != 8) {
    if (p.positions != nullptr && p.kv_fp8) {  // fused rotation + quantised cache write (host checked the envelope)
      if (p.k_scale != 1.f || p.v_scale != 1.f)
        launch(paged_attn_mfma_kernel<T, D, BS, NWAVES, LVLLM_ATTN_NBUF_KV8, true, true, true>);
      else
        launch(paged_attn_mfma_kernel<T, D, BS, NWAVES, LVLLM_ATTN_NBUF_KV8, true, true>);
      return;
    }
  }
  if constexpr (D % 64 == 0 && BS != 8) {
    if (p.positions != nullptr) {  // fused rotation + cache write (host checked: G <= 16, NeoX)
      launch(paged_attn_mfma_kernel<T, D, BS, NWAVES, LVLLM_ATTN_NBUF, false, true>);
      return;
    }
  }
  if constexpr (D % 16 == 0 && BS != 8) {
    if (p.kv_fp8) {
      if (p.k_scale != 1.f || p.v_scale != 1.f)
        launch(paged_attn_mfma_kernel<T, D, BS, NWAVES, LVLLM_ATTN_NBUF_KV8, true, false, true>);
      else
        launch(paged_attn_mfma_kernel<T, D, BS, NWAVES, LVLLM_ATTN_NBUF_KV8, true>);
      return;
    }
  }
  launch(paged_attn_mfma_kernel<T, D, BS, NWAVES, LVLLM_ATTN_NBUF, false>);
}

template <typename T, int D, int BS>
static void launch_mfma_waves(const AttnParams& p, int num_seqs, int num_parts,
                              int max_tokens_per_wg, hipStream_t stream) {
  // 16-token tiles per workgroup decide how many waves can be kept busy
  const int tiles = (max_tokens_per_wg + 15) / 16;
  const int G = p.num_heads / p.num_kv_heads;
  const bool lds8_ok =
      (size_t)8 * (G < 16 ? G : 16) * (((D + 15) / 16) * 16) * 4 + 1024 <= 160 * 1024;
#ifndef LVLLM_ATTN_NWAVES_LONG
#define LVLLM_ATTN_NWAVES_LONG 8
#endif
  if (tiles >= 16 && lds8_ok && tuning().attn_waves == 8)  // >= 2 tiles per wave
    launch_mfma<T, D, BS, LVLLM_ATTN_NWAVES_LONG>(p, num_seqs, num_parts, stream);
  else
    launch_mfma<T, D, BS, 4>(p, num_seqs, num_parts, stream);
}

template <typename T, int D>
static int launch_mfma_bs(const AttnParams& p, int block_size, int num_seqs, int num_parts,
                          int max_tokens_per_wg, hipStream_t stream) {
  switch (block_size) {
    case 16: launch_mfma_waves<T, D, 16>(p, num_seqs, num_parts, max_tokens_per_wg, stream); break;
#ifndef LVLLM_ATTN_TUNE_ONLY
    case 32: launch_mfma_waves<T, D, 32>(p, num_seqs, num_parts, max_tokens_per_wg, stream); break;
    case 8:
      LV_CHECK(!p.kv_fp8 && p.positions == nullptr, "8-token blocks: 16-bit caches, no fused rotation");
      launch_mfma_waves<T, D, 8>(p, num_seqs, num_parts, max_tokens_per_wg, stream);
      break;
#endif
    default: LV_CHECK(false, "Unsupported block size: " + std::to_string(block_size));
  }
  return 0;
}

// block sizes the MFMA kernel covers (8-token blocks go to the generic kernel)
inline bool mfma_block_size(int bs) { return bs == 8 || bs == 16 || bs == 32; }

template <typename T>
int launch_mfma_hs(const AttnParams& p, int head_size, int block_size, int num_seqs,
                   int num_parts, int max_tokens_per_wg, hipStream_t stream) {
  switch (head_size) {
#define LV_HS(D_) \
  case D_: return launch_mfma_bs<T, D_>(p, block_size, num_seqs, num_parts, max_tokens_per_wg, stream);
#ifdef LVLLM_ATTN_TUNE_ONLY  // tuning builds: one shape, seconds to compile
    LV_HS(128)
#else
    LV_HS(64) LV_HS(80) LV_HS(96) LV_HS(112) LV_HS(120) LV_HS(128) LV_HS(192) LV_HS(256)
#endif
#undef LV_HS
    default: LV_CHECK(false, "Unsupported head size: " + std::to_string(head_size));
  }
  return 0;
}

}  // namespace lvllm
