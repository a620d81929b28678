// Causal varlen attention of prompt chunks of 64+ query tokens over the paged cache: the 32x32x16-MFMA body
// (prefill_mfma.h is the 16x16x32 body and the owner of every other case: fp8 caches, ALiBi,
// soft cap, sliding windows, head sizes other than 64 / 128, short chunks).
//
// Same call, same semantics, same parameter block as prefill_mfma.h (flash_attn.py:538-555); what
// changes is the shape of the work:
//   * a workgroup = 8 waves = 256 columns = (query token, head of the GQA group) pairs, 32 per wave; the
//     waves share ONE 64-key tile stream through LDS (2 or 3 stages, one barrier per tile): half the K/V bytes
//     per FLOP of the 128-column workgroup.  Each wave copies 1/8 of every tile through registers
//     (buffer_load_dwordx4 when the tile before is published, ds_write_b128 a phase later): direct-to-LDS
//     copies cost their wave 60-185 issue cycles per KiB among MFMAs and were 22 % of the time;
//   * S^T = K.Q^T with v_mfma_f32_32x32x16: a lane holds 32 keys of ONE column, so the column maximum is
//     in-lane v_max's plus one half-swap, and an MFMA holds the vector issue port for 8 of its 32 cycles
//     (16x16x32: 8 of 16), which leaves the other wave of the SIMD room for its vector work;
//   * the paged layouts stay MFMA operands: a K fragment (32 keys x 16 d) is one 16-byte chunk per lane of
//     the [D/8][BS][8] block, a V^T fragment (32 d x 16 keys) 8 consecutive tokens of a [D][BS] row.  The
//     LDS images are the blocks' head slices, K verbatim, V with the lanes of each copy permuted so that a
//     fragment read (ds_read_b128) is 32 consecutive chunks: no bank conflicts, no padding;
//   * P never moves between lanes: the K fragment of MFMA row i holds a PERMUTED key (groups 4..7 and 8..11 of
//     every 16 trade places -- only a lane's LDS read address changes), so that the S^T accumulator registers of
//     a lane are the consecutive keys 8hi..8hi+7 of every 16, the B-operand layout of P.V;
//   * accumulators are rescaled only when some column's maximum grew by more than 2^kGrow over the value
//     the exponentials use (probabilities stay below 2^kGrow; fp32 sums and the bf16/f16 rounding of P are
//     relative, so nothing is lost).
//
// DENSE (the twin for encode-only models, lvllm_varlen_attention): K and V are the caller's row-major [token][KVH][D]
// rows (slices of a fused qkv projection), copied as whole rows into row-major LDS images -- no pack pass in front of
// the launch.  K fragments are 16-byte row reads as before; a V^T fragment is two ds_read_b64_tr_b16 (the hardware
// transposes 4 keys x 16 columns per 16 lanes).  Chunk c of row r sits at r * 2D + 16 * (c ^ f(r)); f makes the row
// reads, the transposed reads and the copy's writes conflict-free (tools/lds_image_check.py).  Instantiated for head
// size 64 (the encoders' size: 32 x 512 tokens, 16 heads 91 -> 72 us per call, the pack pass included); at 128 the
// image addresses are 16 more live registers than the 245 of the paged body and the spills cost more than the pack
// pass did (8 x 1 024, 32 heads: 127 -> 174 us) -- those launches keep the pack pass (profiles/r04_tuning.md, 12).
#pragma once
#include "prefill_mfma.h"
#include "prefill_partitions.h"

#ifndef LVLLM_PREFILL32_PINGPONG
#define LVLLM_PREFILL32_PINGPONG 0  // 1: the two waves of a SIMD run half a tile out of phase (3 stages instead of 2)
#endif
#ifndef LVLLM_PREFILL32_STAGES
#define LVLLM_PREFILL32_STAGES (LVLLM_PREFILL32_PINGPONG ? 3 : 2)
#endif
#ifndef LVLLM_PREFILL32_KWIN
#define LVLLM_PREFILL32_KWIN 8
#endif
#ifndef LVLLM_PREFILL32_PRIO
#define LVLLM_PREFILL32_PRIO 0  // 1: s_setprio 1 for waves 4..7; 2: s_setprio 1 around the MFMA phases of every wave
#endif
#ifndef LVLLM_PREFILL32_DIAG
// timing-diagnosis builds (WRONG results; tools/ab_prefill32.sh): 1 = no copies after the prologue, 2 = no
// exponentials, 4 = no barriers, 8 = no maximum / rescale, 16 = no LDS reads (fragments = registers)
#define LVLLM_PREFILL32_DIAG 0
#endif
#ifndef LVLLM_PREFILL32_LAZYMAX
#define LVLLM_PREFILL32_LAZYMAX 1  // 0: the column maximum is computed for every tile
#endif
#ifndef LVLLM_PREFILL32_GROW
#define LVLLM_PREFILL32_GROW 6  // log2 of the growth of a column maximum that forces a rescale
#endif

#ifndef LVLLM_PREFILL32_PAIRS
#define LVLLM_PREFILL32_PAIRS 1  // the main loop runs two tiles per trip (stage = compile-time constant); 0: one
#endif
#ifndef LVLLM_PREFILL32_STAMPS
#define LVLLM_PREFILL32_STAMPS 0  // diagnosis builds: 1 = s_memtime at the phase boundaries of one workgroup (tools/stamps_prefill32.py);
                                  // 2 = also start / first barrier / loop end / exit of EVERY workgroup (tools/wg_timeline_prefill32.py)
#endif

namespace lvllm {

typedef float f32x16_t __attribute__((ext_vector_type(16)));
#if LVLLM_PREFILL32_STAMPS
constexpr int kStampTiles = 24, kStampPerTile = 8;
__device__ unsigned long long g_prefill32_stamps[8 * kStampTiles * kStampPerTile];
// per workgroup (LVLLM_PREFILL32_STAMPS == 2, tools/wg_timeline_prefill32.py): 100 MHz wall clock at kernel entry,
// first barrier passed, loop end, exit (wave 0); tiles walked; hardware id (XCC, SE, CU)
constexpr int kWgRecords = 4096, kWgFields = 6;
__device__ unsigned long long g_prefill32_wg[kWgRecords * kWgFields];
#endif

template <typename T>
__device__ __forceinline__ f32x16_t mfma32(u32x4_t a, u32x4_t b, f32x16_t c);
template <>
__device__ __forceinline__ f32x16_t mfma32<BF16>(u32x4_t a, u32x4_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16_t mfma32<F16>(u32x4_t a, u32x4_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0,
                                                0, 0);
}

// lanes 32..63 of a <-> lanes 0..31 of b
__device__ __forceinline__ void half_swap(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// Accumulator layout of the 32x32 MFMA: lane (col = lane & 31, hi = lane >> 5), register r holds row
// (r & 3) + 8 * (r >> 2) + 4 * hi.
typedef short i16x4_t __attribute__((ext_vector_type(4)));

template <typename T, int D, int BS, bool DENSE = false, int NW = 8>
__global__ __launch_bounds__(NW * 64, 2) void paged_prefill_mfma32_kernel(const PrefillParams p, const ChunkScratch sc) {
  // NW = waves per workgroup.  8: 256 columns share a tile stream.  4 (DENSE only): 128 columns, two workgroups per CU at
  // the same two waves per SIMD -- the prologue and epilogue of one run under the tile loop of the other, and the two
  // waves of a SIMD are no longer tied to one barrier; each wave copies twice as much of a tile.
  static_assert(NW == 8 || (NW == 4 && DENSE), "4-wave workgroups: the dense twin only");
  using S = typename T::store_t;
  static_assert(sizeof(S) == 2, "16-bit element types");
  static_assert(BS == 16 || BS == 32, "block size 16 or 32");
  static_assert(D % 32 == 0 && D <= 128, "head size 32..128, a multiple of 32");
  constexpr int KT = 64;                         // keys per tile
  constexpr int NKS = D / 16;                    // k-steps of S^T = K.Q^T
  constexpr int NDB = D / 32;                    // 32-row blocks of O^T
  constexpr int kSlice = D * BS * 2;             // bytes of a block's head slice (K or V)
  constexpr int kBlocksPerTile = KT / BS;
  constexpr int kImage = kBlocksPerTile * kSlice;  // bytes of the K (or V) image of a tile
  constexpr int kStage = 2 * kImage;
  constexpr int kStages = LVLLM_PREFILL32_STAGES;
  constexpr bool kPingPong = LVLLM_PREFILL32_PINGPONG != 0;
  static_assert(kStages >= (kPingPong ? 3 : 2), "stage count");
  constexpr int kPiecesPerWave = D / (4 * NW);   // 1-KiB copies per wave and tile: 2 * kImage / 1024 / NW
  constexpr int kPiecesPerSlice = kSlice / 1024;
  constexpr float kLog2e = 1.4426950408889634f;
  constexpr float kMasked = -FLT_MAX;
  constexpr float kMInit = -1e30f;
  constexpr float kGrow = (float)LVLLM_PREFILL32_GROW;
  // DENSE: the row-major images
  constexpr int kRowB = D * 2;            // bytes of a row
  constexpr int kCPR = D / 8;             // 16-byte chunks of a row
  constexpr int kRowsPerPiece = 64 / kCPR;
  static_assert(!DENSE || D == 64, "dense twin: head size 64 (128 spills, see the header)");
  auto img = [](const int row, const int c) __attribute__((always_inline)) -> int {
    const int f = D == 64 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : (((row & 3) << 2) | ((row >> 2) & 3));
    return row * kRowB + ((c ^ f) << 4);
  };

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, hi = lane >> 5;
#if LVLLM_PREFILL32_STAMPS == 2
  const unsigned long long wg_t0 = wall_clock64();
  const int wg_id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#endif

  const int GP = 1 << p.gp_shift;    // heads per query token among the 32 columns (group size rounded up)
  const int TQW = 32 >> p.gp_shift;  // query tokens per wave
  const int TQWG = NW * TQW;
  const int G = p.num_heads / p.num_kv_heads;
  const int HG = (G + 31) >> 5;
  const int kvh = blockIdx.x / HG;
  const int hg = blockIdx.x - kvh * HG;
  const int head0 = kvh * G + hg * 32;
  const int nh = min(GP, G - hg * 32);
  const int seq = blockIdx.y;
  // blockIdx.z = partition of the key walk (prefill_partitions.h; one when sc.tmp_out is null) x query tile
  const int qtiles = gridDim.z / sc.num_parts;
  const int part = blockIdx.z / qtiles;
  const int qtile = qtiles - 1 - (blockIdx.z - part * qtiles);  // heaviest first

  const int qbeg = p.query_start_loc[seq];
  const int qlen = p.query_start_loc[seq + 1] - qbeg;
  const int seq_len = p.seq_lens != nullptr ? p.seq_lens[seq] : qlen;
  const int ctx = seq_len - qlen;
  if (qtile * TQWG >= qlen || ctx < 0) return;
  const int t_first = qtile * TQWG + wave * TQW;
  const int nq = max(0, min(TQW, qlen - t_first));  // live query tokens of this wave (0: copies only)

  // keys: the workgroup walks [0, khi_walk), this wave computes the tiles below khi
  const int khi = nq == 0 ? 0 : (p.causal ? ctx + t_first + nq : seq_len);
  const int khi_walk = p.causal ? ctx + min(qlen, (qtile + 1) * TQWG) : seq_len;
  // tiles j0 .. ntiles-1 are walked (a partition: its share of them; none: the workgroup has nothing to add, and no
  // row of it reaches the partition, so the merge never reads what it would have written)
  int ntiles = (khi_walk + KT - 1) / KT;
  int my_ntiles = (khi + KT - 1) / KT;
  int j0 = 0;
  if (sc.tmp_out != nullptr) {
    j0 = part * (sc.part_tokens / KT);
    // (the last partition is open-ended: a context longer than the caller's max_seq_len is walked, not cut off)
    if (part + 1 < sc.num_parts) ntiles = min(ntiles, j0 + sc.part_tokens / KT);
    my_ntiles = min(my_ntiles, ntiles);
    if (j0 >= ntiles) return;
  }

  const int32_t* block_table = p.block_tables + (int64_t)seq * p.max_num_blocks_per_seq;
  const char* kbytes = (const char*)p.k_cache + (int64_t)kvh * p.kv_head_stride * 2;
  const char* vbytes = (const char*)p.v_cache + (int64_t)kvh * p.kv_head_stride * 2;
  const int64_t bsb = p.kv_block_stride * 2;

  // ---- this lane's column: query token t_first + col / GP, head head0 + col % GP ----
  const int cq = col >> p.gp_shift, ch = col & (GP - 1);
  const bool live = ch < nh && cq < nq;
  const int vlast = live ? (p.causal ? ctx + t_first + cq : seq_len - 1) : -1;  // last visible key
  u32x4_t qf[NKS];
  {
    const S* qrow = (const S*)p.q + (int64_t)(qbeg + t_first + cq) * p.q_stride + (int64_t)(head0 + ch) * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      qf[ks] = u32x4_t{0, 0, 0, 0};
      if (live) qf[ks] = *reinterpret_cast<const u32x4_t*>(qrow + 16 * ks + 8 * hi);
    }
  }
  const float kf = p.scale * kLog2e;

  const int last_block = p.max_num_blocks_per_seq - 1;
  const bool arithmetic_blocks = p.block_tables == nullptr;  // the dense twin's scratch tiles (prefill_attention.hip)
  const int first_block = qbeg / BS + seq;
  // Block numbers: lane l of `bt_chunk` holds the block this wave copies in tile 64 c + l of the current chunk c of
  // 64 tiles -- one vector load per 64 tiles, then a v_readlane per tile.  (A scalar load per tile sat in the
  // loop with its wait right behind it: a few hundred cycles of every wave's tile.)
  int bt_chunk = 0;
  auto load_block_chunk = [&](const int c, const int ld_blk_) __attribute__((always_inline)) {
    const int blk = min(((c << 6) + lane) * kBlocksPerTile + ld_blk_, last_block);
    bt_chunk = arithmetic_blocks ? first_block + blk : (int)min((uint32_t)block_table[blk], (uint32_t)p.max_block);
  };
  auto block_of_tile = [&](const int j) __attribute__((always_inline)) -> int {
    return __builtin_amdgcn_readlane(bt_chunk, j & 63);
  };

  // ---- copies: wave w moves kPiecesPerWave KiB of every tile: K (w < 4) or V, all inside one block ----
  extern __shared__ __attribute__((aligned(16))) char kv_lds[];
  const int ld_kind = wave / (NW / 2);
  const int ld_piece_first = (wave % (NW / 2)) * kPiecesPerWave;
  const int ld_blk = ld_piece_first / kPiecesPerSlice;     // block of the tile
  const int ld_piece0 = ld_piece_first % kPiecesPerSlice;  // first piece inside the block's slice
  // lane -> source offset inside a piece.  K: verbatim.  V, BS 16: piece = 32 rows of 32 bytes, lane i reads
  // chunk (row i & 31, half i >> 5); BS 32: piece = 16 rows of 64 bytes, lane i reads (row i & 15, chunk i >> 4).
  const int ld_voffset = ld_kind == 0 ? lane * 16
                         : BS == 16   ? (lane & 31) * 32 + (lane >> 5) * 16
                                      : (lane & 15) * 64 + (lane >> 4) * 16;
  const char* ld_base = ld_kind ? vbytes : kbytes;
  // through registers: fetch(j) issues the loads of tile j, stash(j) writes them to tile j's stage (a block past
  // the walk reads as zeros through its zero-size descriptor)
  u32x4_t staged[kPiecesPerWave];
  // DENSE: wave w copies rows 16 (w & 3) .. + 15 (8 waves; 32 (w & 1) .. + 31 with 4) of the tile's K (the first
  // half of the waves) or V, a piece = kRowsPerPiece whole rows
  const int dn_row = (128 / NW) * (wave % (NW / 2)) + lane / kCPR, dn_c = lane % kCPR;
  static_assert(!DENSE || kPiecesPerWave * kRowsPerPiece == 128 / NW, "a wave's pieces are its share of the 64 rows");
  const int64_t dn_stride_b = (ld_kind ? p.dense_v_stride : p.dense_k_stride) * 2;
  const uint32_t dn_voff = (uint32_t)((int64_t)(qbeg + dn_row) * dn_stride_b + (int64_t)kvh * kRowB + dn_c * 16);
  auto fetch = [&](const int j, const int bn32) __attribute__((always_inline)) {
    if constexpr (DENSE) {
      const bool valid = j * KT < khi_walk;
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(ld_kind ? p.v_cache : p.k_cache), 0,
          valid ? (int)(ld_kind ? p.dense_v_bytes : p.dense_k_bytes) : 0, kSrdFlags);
      const uint32_t tile_off = (uint32_t)(j * KT) * (uint32_t)dn_stride_b;
#pragma unroll
      for (int i = 0; i < kPiecesPerWave; ++i)  // (a row past the caller's last one reads as zeros)
        staged[i] = __builtin_amdgcn_raw_buffer_load_b128(
            r, dn_voff + tile_off + (uint32_t)(i * kRowsPerPiece) * (uint32_t)dn_stride_b, 0, 0);
      return;
    }
    const bool valid = (j * kBlocksPerTile + ld_blk) * BS < khi_walk;
    __amdgpu_buffer_rsrc_t r =
        __builtin_amdgcn_make_buffer_rsrc((void*)(ld_base + (int64_t)bn32 * bsb), 0, valid ? kSlice : 0, kSrdFlags);
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i)
      staged[i] = __builtin_amdgcn_raw_buffer_load_b128(r, ld_voffset, (ld_piece0 + i) * 1024, 0);
  };
  // V past the sequence may hold anything (NaN included) and 0 * NaN is NaN: the copying wave zeroes those tokens
  // on their way to LDS (only in the tile that holds the end of the sequence).  K needs nothing: a logit of a key
  // past the sequence is replaced, not used.
  // This lane's 16 bytes of a V piece are 8 consecutive tokens of one row, starting at ld_tok0 of the tile.
  const int ld_tok0 = ld_blk * BS + (BS == 16 ? (lane >> 5) : (lane >> 4)) * 8;
  // The stage of tile j: with two stages, (j - j0) & 1 -- the main loop runs two tiles per trip, and a tile's stage is
  // a compile-time constant there: every fragment address is a loop-invariant register plus an immediate (the dense
  // images have ten distinct swizzled bases; adding the stage to each cost 43 vector instructions per tile).
  auto stage_off = [&](const int j) __attribute__((always_inline)) -> int {
    return (kStages == 2 ? ((j - j0) & 1) : (j % kStages)) * kStage;
  };
  auto stash = [&](const int j, const int so) __attribute__((always_inline)) {
    if constexpr (DENSE) {
      char* st = kv_lds + so + ld_kind * kImage;
      const bool tail = ld_kind == 1 && j * KT + KT > seq_len;  // (wave-uniform) a lane's 16 bytes are one token's
#pragma unroll
      for (int i = 0; i < kPiecesPerWave; ++i) {
        const int row = dn_row + i * kRowsPerPiece;
        u32x4_t v = staged[i];
        if (tail && j * KT + row >= seq_len) v = u32x4_t{0, 0, 0, 0};
        *reinterpret_cast<u32x4_t*>(st + img(row, dn_c)) = v;
      }
      return;
    }
    char* dst = kv_lds + so + ld_kind * kImage + ld_blk * kSlice + ld_piece0 * 1024 + lane * 16;
    if (ld_kind == 1 && j * KT + KT > seq_len) {  // (wave-uniform; the copy registers themselves are never modified)
      const int nvalid = seq_len - (j * KT + ld_tok0);  // <= 0: none, >= 8: all
      uint32_t m[4];
#pragma unroll
      for (int q2 = 0; q2 < 4; ++q2) m[q2] = nvalid >= 2 * q2 + 2 ? 0xffffffffu : nvalid == 2 * q2 + 1 ? 0x0000ffffu : 0u;
#pragma unroll
      for (int i = 0; i < kPiecesPerWave; ++i)
        *reinterpret_cast<u32x4_t*>(dst + i * 1024) =
            u32x4_t{staged[i].x & m[0], staged[i].y & m[1], staged[i].z & m[2], staged[i].w & m[3]};
    } else {
#pragma unroll
      for (int i = 0; i < kPiecesPerWave; ++i) *reinterpret_cast<u32x4_t*>(dst + i * 1024) = staged[i];
    }
  };

  // ---- fragment addresses inside a stage ----
  // K fragment (half kh, k-step ks): key 32 kh + col, chunk d8 = 2 ks + hi
  // MFMA row i of a half does NOT hold key i: within every 16 keys the groups 4..7 and 8..11 trade places, so that
  // the accumulator registers of a lane -- rows {0..3, 8..11} + 4 hi of every 16 -- are the CONSECUTIVE keys
  // 8 hi .. 8 hi + 7: exactly the B-operand layout P.V wants.  The probabilities are packed where they are; no
  // cross-lane move.  (Register r of half kh of lane (col, hi) is key 32 kh + 16 (r >> 3) + 8 hi + (r & 7).)
  const int krow = (col & ~0xc) | ((col & 4) << 1) | ((col & 8) >> 1);  // key (inside the half) held by MFMA row col
  int koff[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const int key = 32 * kh + krow;
    koff[kh] = (key / BS) * kSlice + hi * (BS * 16) + (key % BS) * 16;
  }
  constexpr int kKStep = 2 * BS * 16;
  // V^T fragment (row block db, k-step ks = keys 16 ks + 8 hi ..): row 32 db + col
  auto voff = [&](const int db, const int ks) __attribute__((always_inline)) -> int {
    if constexpr (BS == 16) return kImage + ks * kSlice + db * 1024 + hi * 512 + col * 16;
    else return kImage + (ks >> 1) * kSlice + (2 * db + (col >> 4)) * 1024 + (2 * (ks & 1) + hi) * 256 + (col & 15) * 16;
  };

  float m_run = kMInit, l_run = 0.f;
  f32x16_t acc[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;

  auto half_max = [&](float x) __attribute__((always_inline)) -> float {
    uint32_t a = __builtin_bit_cast(uint32_t, x), b = a;
    half_swap(a, b);
    return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
  };

  // One tile = three phases.  qk and pv are MFMA + LDS reads only, softmax is vector ALU only.  The LDS reads
  // run a window of kWin fragments ahead of the MFMAs that consume them (a ds_read_b128 takes a few hundred
  // cycles to return with 8 waves reading; two in flight per wave, as the compiler arranges by itself, left the
  // MFMA pipe waiting for operands a quarter of the time), and the first window of V is read BEFORE the softmax.
  constexpr int kWin = 8;
  constexpr int kWinK = LVLLM_PREFILL32_KWIN;  // window of the K fragments
  constexpr int NV = 4 * NDB;  // V^T fragments of a tile, in the order (row block db, k-step ks)
  // DENSE, transposed reads: lane 4 q + pp of every 16 addresses row q of a block of 4 keys, columns 4 pp .. 4 pp + 3
  // of the block's 16; lane i of the 16 receives column i of the 4 keys.  The 16-lane groups of a wave are
  // (d half col >> 4, key half hi), so a lane ends up with the 8 keys 8 hi .. 8 hi + 7 of row 32 db + col.
  const int tr_q = (lane & 15) >> 2, tr_c = 2 * (col >> 4) + ((lane & 3) >> 1), tr_b = 8 * (lane & 1);
  auto kread = [&](const char* st, const int kh, const int ks) __attribute__((always_inline)) -> u32x4_t {
    if constexpr (LVLLM_PREFILL32_DIAG & 16) return qf[(ks + 1) % NKS];
    if constexpr (DENSE) return *reinterpret_cast<const u32x4_t*>(st + img(32 * kh + krow, 2 * ks + hi));
    return *reinterpret_cast<const u32x4_t*>(st + koff[kh] + ks * kKStep);
  };
  auto vread = [&](const char* st, const int i) __attribute__((always_inline)) -> u32x4_t {
    if constexpr (LVLLM_PREFILL32_DIAG & 16) return qf[i % NKS];
    if constexpr (DENSE) {
      typedef __attribute__((address_space(3))) i16x4_t* lds_tr_t;
      const int db = i >> 2, ks = i & 3;
      const int key = 16 * ks + 8 * hi + tr_q;
      const i16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(st + kImage + img(key, 4 * db + tr_c) + tr_b));
      const i16x4_t up =
          __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(st + kImage + img(key + 4, 4 * db + tr_c) + tr_b));
      const u32x2_t a = __builtin_bit_cast(u32x2_t, lo), b = __builtin_bit_cast(u32x2_t, up);
      return u32x4_t{a.x, a.y, b.x, b.y};
    }
    return *reinterpret_cast<const u32x4_t*>(st + voff(i >> 2, i & 3));
  };
  auto qk = [&](const int so, f32x16_t (&s)[2]) __attribute__((always_inline)) {
    const char* st = kv_lds + so;
    if constexpr (LVLLM_PREFILL32_PRIO == 2) __builtin_amdgcn_s_setprio(1);  // MFMA phases win the issue arbitration
    u32x4_t kw[kWinK];
#pragma unroll
    for (int i = 0; i < kWinK; ++i) kw[i] = kread(st, i / NKS, i % NKS);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kh][r] = 0.f;
    // (a scheduling fence after every step: the order written is the order wanted -- left alone, the scheduler
    // sinks every read to just before its use, and sched_group_barrier cannot say WHICH reads go first)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2 * NKS; ++i) {
      s[i / NKS] = mfma32<T>(kw[i % kWinK], qf[i % NKS], s[i / NKS]);
      if (i + kWinK < 2 * NKS) kw[i % kWinK] = kread(st, (i + kWinK) / NKS, (i + kWinK) % NKS);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (LVLLM_PREFILL32_PRIO == 2) __builtin_amdgcn_s_setprio(0);
  };
  auto vpre = [&](const int so, u32x4_t (&vw)[kWin]) __attribute__((always_inline)) {
    const char* st = kv_lds + so;
#pragma unroll
    for (int i = 0; i < kWin; ++i) vw[i] = vread(st, i);
  };
  // (masked: a wave-uniform run-time flag.  The two cases share ONE body and differ by a branch around the
  // selects only: with a copy of softmax + P.V per case the accumulators were defined on both arms of a branch,
  // and the register allocator kept two sets of them -- 256 VGPRs and spills, against 151 for one body.)
  auto softmax = [&](const int j, const f32x16_t (&s)[2], u32x4_t (&pb)[4], const bool masked) __attribute__((always_inline)) {
    const int base = j * KT;
    // ---- column maximum (the logits leave the accumulator tuples as scalars and never go back) ----
    float y[2][16];
    const int rel = vlast - base - 8 * hi;  // register r of half kh is key base + 8 hi + (32 kh + 16 (r >> 3) + (r & 7))
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int r = 0; r < 16; ++r) y[kh][r] = s[kh][r];
    if (masked) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[kh][r] = 32 * kh + 16 * (r >> 3) + (r & 7) <= rel ? y[kh][r] : kMasked;
    }
    // ---- probabilities against the maximum in use, issued before the new maximum is known: the maximum is a
    // dependent chain (4 short ones here) ending in a branch, and the exponentials need not wait for it ----
    // (fused multiply-adds and sums as 2-wide packed instructions: half the issue slots of this phase's
    // full-rate arithmetic; the MFMA pipe of this wave is idle here, so the packed forms cost nothing beside it)
    float e[2][16];
    auto exps = [&](const float mk) __attribute__((always_inline)) {
      const f32x2_t kf2 = {kf, kf}, nmk2 = {-mk, -mk};
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_t t = __builtin_elementwise_fma(f32x2_t{y[kh][r], y[kh][r + 1]}, kf2, nmk2);
          if constexpr (LVLLM_PREFILL32_DIAG & 2) {
            e[kh][r] = t.x;
            e[kh][r + 1] = t.y;
          } else {
            e[kh][r] = __builtin_amdgcn_exp2f(t.x);
            e[kh][r + 1] = __builtin_amdgcn_exp2f(t.y);
          }
        }
    };
    exps(m_run * kf);
    auto sum_all = [&]() __attribute__((always_inline)) -> f32x2_t {  // (four chains: a packed add waits for the one before)
      f32x2_t part[4];
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int c4 = (r >> 1) & 3;
          const f32x2_t v = {e[kh][r], e[kh][r + 1]};
          part[c4] = (kh == 0 && r < 8) ? v : part[c4] + v;
        }
      return (part[0] + part[1]) + (part[2] + part[3]);
    };
    f32x2_t psum2 = sum_all();
    if constexpr (!(LVLLM_PREFILL32_DIAG & 8)) {
      // The new maximum is not computed on the way: a probability above 2^kGrow shows in the lane's sum (all
      // terms are >= 0), and only then (wave-uniform, rare after the first tiles; NaN-proof: the test is "not <=")
      // the exact maximum is taken and the usual criterion decides per column -- both lanes of a column see the
      // same maximum, so they decide alike.  Below the limit every probability is <= 32 * 2^kGrow: fine for fp32 sums
      // and for the bf16 / f16 rounding of P.
      constexpr float kSumLimit = 32.f * (float)(1 << LVLLM_PREFILL32_GROW);
      const bool suspicious = LVLLM_PREFILL32_LAZYMAX ? !(psum2.x + psum2.y <= kSumLimit) : true;
      if (__builtin_amdgcn_ballot_w64(suspicious) != 0) {
        float m4[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          m4[c4] = fmaxf(y[0][4 * c4], y[1][4 * c4]);
#pragma unroll
          for (int r = 1; r < 4; ++r) m4[c4] = fmaxf(m4[c4], fmaxf(y[0][4 * c4 + r], y[1][4 * c4 + r]));
        }
        const float m_loc = half_max(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])));
        // ---- some column outgrew the maximum in use by 2^kGrow: rescale, and redo the exponentials.  The new
        // maximum is taken per column, so a column's arithmetic never depends on its neighbours. ----
        const bool grew = (m_loc - m_run) * kf > kGrow;
        if (__builtin_amdgcn_ballot_w64(grew) != 0) {
          const float m_new = grew ? fmaxf(m_run, m_loc) : m_run;
          const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kf);
          l_run *= alpha;
#pragma unroll
          for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[db][r] *= alpha;
          m_run = m_new;
          exps(m_run * kf);
          psum2 = sum_all();
        }
      }
    }
    // ---- packing, and the move to the B-operand layout ----
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        const int r0 = 8 * g2;
        pb[2 * kh + g2] = u32x4_t{pack2<T>(e[kh][r0 + 0], e[kh][r0 + 1]), pack2<T>(e[kh][r0 + 2], e[kh][r0 + 3]),
                                  pack2<T>(e[kh][r0 + 4], e[kh][r0 + 5]), pack2<T>(e[kh][r0 + 6], e[kh][r0 + 7])};
      }
    }
    l_run += psum2.x + psum2.y;
    __builtin_amdgcn_sched_barrier(0);
  };
  // O^T += V^T.P^T
  auto pv = [&](const int so, const u32x4_t (&pb)[4], u32x4_t (&vw)[kWin]) __attribute__((always_inline)) {
    const char* st = kv_lds + so;
    static_assert(NV >= kWin, "the window is at most one tile of V");
    if constexpr (LVLLM_PREFILL32_PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      acc[i >> 2] = mfma32<T>(vw[i % kWin], pb[i & 3], acc[i >> 2]);
      if (i + kWin < NV) {
        vw[i % kWin] = vread(st, i + kWin);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LVLLM_PREFILL32_PRIO == 2) __builtin_amdgcn_s_setprio(0);
  };

  {
    // tiles wholly at or before the wave's first query need no mask
    const int q_first_pos = p.causal ? ctx + t_first : seq_len - 1;
    const int n_plain = min(my_ntiles, min((q_first_pos + 1) / KT, seq_len / KT));
    // Copies: fetch(j+1) is issued when tile j is published, stash(j+1) writes it to LDS one phase before the
    // barrier that publishes it -- after the softmax of tile j, so that the loads have had K.Q^T and the softmax
    // to arrive and the writes drain under P.V.  The stage of tile j+1 held tile j+1-kStages, which every wave
    // has left behind before the barrier of tile j.
    if constexpr (!DENSE) load_block_chunk(j0 >> 6, ld_blk);
    fetch(j0, DENSE ? 0 : block_of_tile(j0));
    stash(j0, stage_off(j0));
#if LVLLM_PREFILL32_STAMPS
    // workgroup (0, 0, the 9th heaviest): every wave stamps the boundaries of its phases in tiles 8 .. 8 + kStampTiles
    const bool stamping = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 8 && lane == 0;
    auto stamp = [&](const int j, const int k) __attribute__((always_inline)) {
      if (stamping && j >= 8 && j < 8 + kStampTiles)
        g_prefill32_stamps[(wave * kStampTiles + (j - 8)) * kStampPerTile + k] = __builtin_amdgcn_s_memtime();
    };
#else
    auto stamp = [&](const int, const int) __attribute__((always_inline)) {};
#endif
    // "publish(j)": this wave's part of tile j is in LDS; meet everybody (all of tile j is); start loading tile j+1.
    // Every wave publishes tiles 0 .. ntiles-1, in order.
    auto publish = [&](const int j) __attribute__((always_inline)) {
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's writes have reached LDS
      stamp(j, 6);
      if constexpr (!(LVLLM_PREFILL32_DIAG & 4)) __builtin_amdgcn_s_barrier();
      stamp(j, 7);
#if LVLLM_PREFILL32_STAMPS == 2
      if (j == 0 && threadIdx.x == 0 && wg_id < kWgRecords) g_prefill32_wg[wg_id * kWgFields + 1] = wall_clock64();
#endif
      if constexpr (!(LVLLM_PREFILL32_DIAG & 1)) {
        if constexpr (!DENSE) {
          if (((j + 1) & 63) == 0) load_block_chunk((j + 1) >> 6, ld_blk);
        }
        fetch(j + 1, DENSE ? 0 : block_of_tile(j + 1));
      }
    };
    // The second wave of every SIMD (waves w and w + 4 share one) runs half a tile out of phase: it publishes
    // tile j+1 between K.Q^T and the softmax of tile j, the first wave before K.Q^T of tile j+1.  Between two
    // barriers the first wave runs {K.Q^T, softmax, P.V} of one tile, the second {softmax, P.V} of the tile
    // before and K.Q^T of this one: the vector-ALU phase of each falls under MFMAs of the other.
    // (One straight-line loop per role, and the tiles a wave only copies in a loop of their own: a tile body
    // under a condition leaves the 64 accumulator registers defined on two paths, and the compiler then moves
    // all of them at every loop end.)
    auto run = [&](auto late_tag) __attribute__((always_inline)) {
      constexpr bool late = decltype(late_tag)::value;
      if constexpr (late) publish(j0);
      int j = j0;
      if constexpr (!late && kStages == 2 && LVLLM_PREFILL32_PAIRS) {
        auto tile = [&](const int jt, auto so_tag) __attribute__((always_inline)) {
          constexpr int so = decltype(so_tag)::value;
          stamp(jt, 0);
          publish(jt);
          stamp(jt, 1);
          f32x16_t s[2];
          u32x4_t pb[4], vw[kWin];
          qk(so, s);
          stamp(jt, 2);
          stamp(jt, 3);
          vpre(so, vw);  // the first window of V: in flight during the softmax
          __builtin_amdgcn_sched_barrier(0);
          softmax(jt, s, pb, jt >= n_plain);
          stamp(jt, 4);
          stash(jt + 1, kStage - so);
          pv(so, pb, vw);
          stamp(jt, 5);
        };
#pragma nounroll
        for (; j + 1 < my_ntiles; j += 2) {
          tile(j, std::integral_constant<int, 0>{});
          tile(j + 1, std::integral_constant<int, kStage>{});
        }
        if (j < my_ntiles) {
          tile(j, std::integral_constant<int, 0>{});
          ++j;
        }
      } else {
#pragma nounroll
        for (; j < my_ntiles; ++j) {
          const int so = stage_off(j);
          stamp(j, 0);
          if constexpr (!late) publish(j);
          stamp(j, 1);
          f32x16_t s[2];
          u32x4_t pb[4], vw[kWin];
          qk(so, s);
          stamp(j, 2);
          if constexpr (late) {
            stash(j + 1, stage_off(j + 1));
            if (j + 1 < ntiles) publish(j + 1);
          }
          stamp(j, 3);
          vpre(so, vw);  // the first window of V: in flight during the softmax
          __builtin_amdgcn_sched_barrier(0);
          softmax(j, s, pb, j >= n_plain);
          stamp(j, 4);
          if constexpr (!late) stash(j + 1, stage_off(j + 1));
          pv(so, pb, vw);
          stamp(j, 5);
        }
      }
#pragma nounroll
      for (j = max(j, j0); j < ntiles; ++j) {  // tiles this wave copies for the others
        if constexpr (!late) publish(j);
        stash(j + 1, stage_off(j + 1));
        if constexpr (late)
          if (j + 1 < ntiles) publish(j + 1);
      }
    };
#if LVLLM_PREFILL32_PRIO == 1
    if (wave / (NW / 2)) __builtin_amdgcn_s_setprio(1);  // the younger half loses every arbitration otherwise
#endif
    if (LVLLM_PREFILL32_PINGPONG == 2 || (kPingPong && (wave / (NW / 2)) != 0)) run(std::true_type{});  // 2: diagnosis
    else run(std::false_type{});
    __builtin_amdgcn_s_waitcnt(0);  // no copy may still be landing when the workgroup's LDS is released
  }
#if LVLLM_PREFILL32_STAMPS == 2
  const unsigned long long wg_t2 = wall_clock64();
#endif

  // ---- normalise and store: register r of block db is d = 32 db + 8 (r >> 2) + 4 hi + (r & 3) ----
  {
    uint32_t a = __builtin_bit_cast(uint32_t, l_run), b = a;
    half_swap(a, b);
    const float l = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    const float inv = l > 0.f ? __fdividef(1.f, l) : 0.f;
    if (live) {
      S* orow = (S*)p.out + (int64_t)(qbeg + t_first + cq) * p.out_stride + (int64_t)(head0 + ch) * D;
      if (sc.tmp_out != nullptr) {  // the partition's share: (maximum in base 2, sum, normalised partial result)
        const int64_t row = ((int64_t)(qbeg + t_first + cq) * p.num_heads + head0 + ch) * sc.num_parts + part;
        orow = (S*)sc.tmp_out + row * D;
        if (hi == 0) {
          sc.max_logits[row] = m_run * kf;
          sc.exp_sums[row] = l;
        }
      }
#pragma unroll
      for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          u32x2_t o;
          o.x = pack2<T>(acc[db][4 * rq + 0] * inv, acc[db][4 * rq + 1] * inv);
          o.y = pack2<T>(acc[db][4 * rq + 2] * inv, acc[db][4 * rq + 3] * inv);
          *reinterpret_cast<u32x2_t*>(orow + 32 * db + 8 * rq + 4 * hi) = o;
        }
    }
  }
#if LVLLM_PREFILL32_STAMPS == 2
  __builtin_amdgcn_s_waitcnt(0);
  if (threadIdx.x == 0 && wg_id < kWgRecords) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_prefill32_wg[wg_id * kWgFields + 0] = wg_t0;
    g_prefill32_wg[wg_id * kWgFields + 2] = wg_t2;
    g_prefill32_wg[wg_id * kWgFields + 3] = wall_clock64();
    g_prefill32_wg[wg_id * kWgFields + 4] = (unsigned long long)ntiles;
    g_prefill32_wg[wg_id * kWgFields + 5] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
}

// The grid of a launch and, for launches that would leave most CUs idle (short chunks of a few sequences over long
// contexts: 8 x (32 tokens over 2 048) is 64 workgroups walking 33 tiles each), the partitions of its key walk:
// as many as fill the CUs, at least 4 tiles (256 keys) each, when the caller states a bound on seq_lens and
// brings the scratch (prefill_partitions.h).
struct Prefill32Plan {
  int gp_shift, qtiles, parts, part_tokens;
  int64_t rows, ws_bytes;
};
inline Prefill32Plan prefill32_plan(int num_seqs, int max_query_len, int num_heads, int num_kv_heads, int head_size,
                                    int max_seq_len, int waves = 8) {
  static const int num_cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return n;
  }();
  Prefill32Plan pl{};
  const int G = num_heads / num_kv_heads;
  const int HG = (G + 31) / 32;
  pl.gp_shift = G == 1 ? 0 : G == 2 ? 1 : G <= 4 ? 2 : G <= 8 ? 3 : G <= 16 ? 4 : 5;
  const int tqwg = waves * (32 >> pl.gp_shift);
  pl.qtiles = (max_query_len + tqwg - 1) / tqwg;
  pl.parts = 1;
  pl.rows = (int64_t)num_seqs * max_query_len * num_heads;
  const int64_t wgs = (int64_t)num_kv_heads * HG * num_seqs * pl.qtiles;
  if (max_seq_len > 0 && num_cus > 0 && wgs * 2 <= num_cus) {
    int parts = (int)(num_cus / wgs);
    const int tiles = (max_seq_len + 63) / 64;
    if (parts > tiles / 4) parts = tiles / 4;
    if (parts > kMaxPartitions) parts = kMaxPartitions;
    if (parts >= 2) {
      pl.parts = parts;
      pl.part_tokens = ((tiles + parts - 1) / parts) * 64;
      pl.ws_bytes = partition_scratch_bytes(pl.rows, parts, head_size);
    }
  }
  return pl;
}

template <typename T, int D, int BS, bool DENSE, int NW = 8>
static int launch_prefill_mfma32_impl(const PrefillParams& p0, int num_seqs, int max_query_len, hipStream_t stream) {
  PrefillParams p = p0;
  const int G = p.num_heads / p.num_kv_heads;
  const int HG = (G + 31) / 32;
  const Prefill32Plan pl = prefill32_plan(num_seqs, max_query_len, p.num_heads, p.num_kv_heads, D, p.max_seq_len, NW);
  p.gp_shift = pl.gp_shift;
  ChunkScratch sc{};
  sc.num_parts = 1;
  if (pl.parts >= 2 && p.causal && p.workspace != nullptr && p.workspace_bytes >= pl.ws_bytes)
    sc = partition_scratch(p.workspace, pl.rows, pl.parts, pl.part_tokens, D);
  const dim3 grid(p.num_kv_heads * HG, num_seqs, pl.qtiles * sc.num_parts);
  const size_t smem = (size_t)LVLLM_PREFILL32_STAGES * 2 * D * 64 * 2;
  auto kern = paged_prefill_mfma32_kernel<T, D, BS, DENSE, NW>;
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), smem, stream, p, sc);
  if (sc.tmp_out != nullptr)
    launch_partition_reduce<T, D>(p, sc, num_seqs, max_query_len, stream);
#if LVLLM_PREFILL32_STAMPS == 2
  if (getenv("LVLLM_PREFILL32_WG_FILE")) {
    static unsigned long long hostw[kWgRecords * kWgFields];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(hostw, HIP_SYMBOL(g_prefill32_wg), sizeof(hostw));
    FILE* f = fopen(getenv("LVLLM_PREFILL32_WG_FILE"), "w");
    if (f) {
      const int n = (int)(grid.x * grid.y * grid.z);
      for (int w = 0; w < n && w < kWgRecords; ++w) {
        fprintf(f, "%d", w);
        for (int k = 0; k < kWgFields; ++k) fprintf(f, " %llu", hostw[w * kWgFields + k]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#endif
#if LVLLM_PREFILL32_STAMPS
  if (getenv("LVLLM_PREFILL32_STAMP_FILE")) {
    static unsigned long long host[8 * kStampTiles * kStampPerTile];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prefill32_stamps), sizeof(host));
    FILE* f = fopen(getenv("LVLLM_PREFILL32_STAMP_FILE"), "w");
    if (f) {
      for (int w = 0; w < 8; ++w)
        for (int t = 0; t < kStampTiles; ++t) {
          fprintf(f, "%d %d", w, t + 8);
          for (int k = 0; k < kStampPerTile; ++k) fprintf(f, " %llu", host[(w * kStampTiles + t) * kStampPerTile + k]);
          fprintf(f, "\n");
        }
      fclose(f);
    }
  }
#endif
  return 0;
}

template <typename T, int D, int BS>
static int launch_prefill_mfma32(const PrefillParams& p0, int num_seqs, int max_query_len, hipStream_t stream) {
  return launch_prefill_mfma32_impl<T, D, BS, false>(p0, num_seqs, max_query_len, stream);
}

// The dense twin (DENSE above): whether a launch of lvllm_varlen_attention can read the caller's rows directly ...
inline bool dense_mfma32_takes(const PrefillParams& p, int head_size, int num_seqs, int max_query_len) {
  return head_size == 64 && !p.kv_fp8 && p.alibi_slopes == nullptr && p.softcap <= 0.f &&
         p.sliding_window <= 0 && p.dense_k_bytes > 0 && p.dense_v_bytes > 0 &&
         p.dense_k_bytes < ((int64_t)1 << 31) && p.dense_v_bytes < ((int64_t)1 << 31) &&
         takes_mfma32(p, num_seqs, max_query_len);
}
// ... and the launch (block_tables = seq_lens = null: context == chunk)
template <typename T>
static int launch_prefill_mfma32_dense(const PrefillParams& p, int head_size, int num_seqs, int max_query_len,
                                       hipStream_t stream) {
  LV_CHECK(head_size == 64, "dense twin: head size 64");
  // 4-wave workgroups (two per CU) where a workgroup walks few tiles and its prologue / epilogue weigh most:
  // 128 x 128 tokens 45.5 -> 35.9 us, 32 x 512 causal 57 -> 52; level at 32 x 512 bidirectional, slower at 8 x 2 048
  // (225 against 204: twice the K / V copies); profiles/r04_tuning.md, 12
  int waves = tuning().varlen_dense_waves;
  if (waves == 0) waves = (max_query_len <= 256 || (p.causal && max_query_len <= 512)) ? 4 : 8;
  if (waves == 4)
    return launch_prefill_mfma32_impl<T, 64, 16, true, 4>(p, num_seqs, max_query_len, stream);
  return launch_prefill_mfma32_impl<T, 64, 16, true>(p, num_seqs, max_query_len, stream);
}

}  // namespace lvllm
