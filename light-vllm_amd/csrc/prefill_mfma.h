// Causal varlen attention of prompt chunks over the PAGED cache (prefill, chunked prefill,
// prefix-cache hits) for MI355X (gfx950 / CDNA4).
//
// WHAT (the call it replaces: light_vllm/decoding/backends/attention/backends/flash_attn.py:538-555,
// flash_attn_varlen_func(q, key_cache, value_cache, cu_seqlens_q, cu_seqlens_k, causal=True,
// block_table=...); the orphaned Triton twin is ops/prefix_prefill.py context_attention_fwd):
//   sequence i owns query tokens query_start_loc[i] .. query_start_loc[i+1]; its context is
//   seq_lens[i] tokens long and ENDS with those query tokens (their K/V were written to the
//   cache by reshape_and_cache before this call).  Query token t (0-based inside the chunk)
//   sits at position ctx + t, ctx = seq_len - query_len, and attends to keys 0 .. ctx + t
//   (bottom-right aligned causal mask), optionally only the last `sliding_window` of them.
//   fp32 logits / softmax / accumulation, probabilities rounded to T before P.V.
//
// HOW: the decode kernel's operand trick carries over unchanged -- the paged K layout is the
// MFMA A-operand layout, V pieces feed the 16x16x16 MFMA, P never moves between lanes
// (attention_mfma.h) -- but the 16 MFMA columns are now (query token, head of the GQA group)
// pairs, NB column blocks per wave, and the waves of a workgroup split the QUERIES (no LDS, no
// barriers, no merge): each wave walks the key tiles its own queries can see.  K/V tiles are
// re-read by the 4 waves of a workgroup and by neighbouring workgroups out of L1/L2; HBM sees
// each block about once per kv head.  Per wave and key tile: NB*(NS+NDT) MFMAs against
// NS+NDT wave loads, so the kernel is MFMA/VALU-bound, not HBM-bound, for chunks >= 64 tokens.
//   * accumulators are rescaled lazily: only when some column's running max grew in this tile
//     (a wave-uniform branch; after the first tiles it is rarely taken);
//   * exp2 with log2(e) folded into the scale;
//   * the heaviest query tiles (end of the chunk) are launched first.
#pragma once
#include <float.h>

#include <type_traits>

#include "attention_mfma.h"

#ifndef LVLLM_PREFILL_EXP
#define LVLLM_PREFILL_EXP 0
#endif
#ifndef LVLLM_PREFILL_WAVES_PER_SIMD
#define LVLLM_PREFILL_WAVES_PER_SIMD 2  // 2: <= 256 VGPRs; 1: the 512-register body for large NB
#endif
#ifndef LVLLM_PREFILL_STAGES
#define LVLLM_PREFILL_STAGES 3  // LDS stages of the LDSKV variant: copies run STAGES-1 pairs ahead
#endif
#ifndef LVLLM_PREFILL_SCHED
#define LVLLM_PREFILL_SCHED 1
#endif

namespace lvllm {

struct PrefillParams {
  void* out;                       // [num_tokens, num_heads, D]
  const void* q;                   // [num_tokens, num_heads, D] (token stride q_stride)
  const void* k_cache;             // [num_blocks, KVH, D/x, BS, x]
  const void* v_cache;             // [num_blocks, KVH, D, BS]
  const int32_t* block_tables;     // [num_seqs, max_num_blocks_per_seq]
  const int32_t* seq_lens;         // [num_seqs] context length INCLUDING the chunk
  const int32_t* query_start_loc;  // [num_seqs + 1]
  const float* alibi_slopes;       // [num_heads] or null
  int num_heads, num_kv_heads, max_num_blocks_per_seq;
  int max_block;       // block numbers from the table are clamped to [0, max_block] (attention_params.h)
  int gp_shift;        // log2 of the GQA group size rounded up to a power of two (<= 16)
  int causal;          // 1: bottom-right aligned causal mask; 0: every query sees the whole context
  int sliding_window;  // <= 0: none; else a query at position p sees keys p-w+1 .. p (causal only)
  float scale;
  float softcap;  // <= 0: none; else logits = cap * tanh(logits / cap)
  int64_t q_stride, out_stride, kv_block_stride, kv_head_stride;  // kv strides in cache elements
  int kv_fp8;                // caches hold OCP e4m3fn bytes (x = 16 layouts); see attention_mfma.h KV8
  float k_scale, v_scale;    // a dequantised element is T(float(fp8) * scale)
  // host side only (prefill_chunk.h): the caller's bound on seq_lens (0: unknown) and scratch for partitioned walks
  int max_seq_len;
  int num_tokens;  // rows of `query` the caller states (0: unknown)
  void* workspace;
  int64_t workspace_bytes;
  // the dense twin of the 32x32 body (prefill_mfma32.h, DENSE): k_cache / v_cache are row-major [token][KVH][D] rows
  // of the caller (token strides in elements), readable for dense_k_bytes / dense_v_bytes from their bases
  int64_t dense_k_stride, dense_v_stride, dense_k_bytes, dense_v_bytes;
};

// Column layout of one 16-column MFMA block: column c = (query token c / GP, head c % GP) with
// GP = 1 << gp_shift = the GQA group size rounded up to a power of two (<= 16).
// EXTRAS = ALiBi / soft cap / sliding window present (every tile is masked and biased per element);
// the plain causal instantiation masks only the tiles that straddle the diagonal.
// LDSKV = the K/V tiles of a pair are staged ONCE per workgroup in LDS (direct-to-LDS buffer loads,
// double buffered, one barrier per pair) and every wave reads its MFMA operands from there: a
// quarter of the global-load instructions and of the L1 traffic of the register path, whose
// per-wave K/V loads cost 45 % of its time.  Needs D % 32 == 0.
// KV8 (with LDSKV only, D % 64 == 0): fp8 caches.  The tile images in LDS are the fp8 bytes (half the
// copy and half the LDS); a K fragment read is 16 fp8 = two k-slices after conversion, a V piece 4
// bytes; Q is loaded with the matching head-dim permutation, exactly as in the decode kernel's KV8.
template <typename T, int D, int BS, int NB, bool EXTRAS, bool LDSKV, bool KV8 = false>
__global__ __launch_bounds__(256, LVLLM_PREFILL_WAVES_PER_SIMD) void paged_prefill_mfma_kernel(const PrefillParams p) {
  static_assert(!KV8 || (LDSKV && D % 64 == 0), "fp8 caches: LDS path, head size a multiple of 64");
  constexpr int KVB = KV8 ? 1 : 2;  // bytes per cache element
  using S = typename T::store_t;
  static_assert(sizeof(S) == 2, "MFMA path is for 16-bit element types");
  static_assert(BS == 16 || BS == 32, "one tile must lie inside one block");
  constexpr int NS = (D + 31) / 32;
  constexpr int NDT = (D + 15) / 16;
  constexpr int kHeadBytes = D * BS * KVB;
  constexpr float kLog2e = 1.4426950408889634f;
  constexpr float kMasked = -FLT_MAX;
  constexpr float kMInit = -1e30f;  // > kMasked: exp2(kMasked - m) == 0 even before any key is seen

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;

  const int GP = 1 << p.gp_shift;
  const int TQB = 16 >> p.gp_shift;  // query tokens per column block
  const int TQW = NB * TQB;          // per wave
  const int TQWG = 4 * TQW;          // per workgroup
  const int G = p.num_heads / p.num_kv_heads;
  const int HG = (G + 15) >> 4;
  const int kvh = blockIdx.x / HG;
  const int hg = blockIdx.x - kvh * HG;
  const int head0 = kvh * G + hg * 16;
  const int nh = min(GP, G - hg * 16);
  const int seq = blockIdx.y;
  const int qtile = gridDim.z - 1 - blockIdx.z;  // heaviest first

  const int qbeg = p.query_start_loc[seq];
  const int qlen = p.query_start_loc[seq + 1] - qbeg;
  const int seq_len = p.seq_lens != nullptr ? p.seq_lens[seq] : qlen;  // no cache: the chunk is the context
  const int ctx = seq_len - qlen;
  const int t_first = qtile * TQWG + wave * TQW;  // first query token (inside the chunk) of this wave
  if (qtile * TQWG >= qlen || ctx < 0) return;    // the whole workgroup is past the chunk
  // (with LDSKV a wave past the end of the chunk stays: it still loads its share of every pair)
  if (!LDSKV && t_first >= qlen) return;
  const int nq = max(0, min(TQW, qlen - t_first));  // live query tokens of this wave

  // keys this wave needs: [klo, khi); LDSKV: the tile walk is the workgroup's (first wave's
  // window start .. last live query), a wave computes only the pairs its own queries can see
  const int khi = nq == 0 ? 0 : (p.causal ? ctx + t_first + nq : seq_len);
  const int wg_first = LDSKV ? qtile * TQWG : t_first;
  const int klo = p.sliding_window > 0 ? max(0, ctx + wg_first - p.sliding_window + 1) : 0;
  const int tile0 = klo >> 4;
  const int khi_walk = !LDSKV ? khi : (p.causal ? ctx + min(qlen, (qtile + 1) * TQWG) : seq_len);
  const int ntiles = ((khi_walk + 15) >> 4) - tile0;  // tiles loaded
  const int my_npairs = khi > (tile0 << 4) ? (khi - (tile0 << 4) + 31) >> 5 : 0;  // pairs computed

  const int32_t* block_table = p.block_tables + (int64_t)seq * p.max_num_blocks_per_seq;
  const char* kbytes = (const char*)p.k_cache + (int64_t)kvh * p.kv_head_stride * KVB;
  const char* vbytes = (const char*)p.v_cache + (int64_t)kvh * p.kv_head_stride * KVB;
  const int64_t bsb = p.kv_block_stride * KVB;
  const int koff = (g * BS + c) * 16;
  const int voff = (c * BS + 4 * g) * 2;

  // column c of block b: query token t_first + b*TQB + c/GP, head head0 + c%GP
  const int cq = c >> p.gp_shift, ch = c & (GP - 1);
  const bool head_ok = ch < nh;

  // ---- Q fragments ----
  u32x4_t qf[NB][NS];
  int qpos[NB];  // absolute position of this lane's column in block b (dead columns: -1)
  int vlast[NB];  // last key this column may see: its own position (causal) or the last key
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int t = b * TQB + cq;
    const bool ok = head_ok && t < nq;
    qpos[b] = ok ? ctx + t_first + t : -1;
    vlast[b] = ok ? (p.causal ? ctx + t_first + t : seq_len - 1) : -1;
    const S* qrow = (const S*)p.q + (int64_t)(qbeg + t_first + t) * p.q_stride + (int64_t)(head0 + ch) * D;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      // 16-bit caches: d = 32j + 8g ..; fp8 caches: d = 64(j>>1) + 16g + 8(j&1) .. (K chunks of 16 d)
      const int d0 = KV8 ? 64 * (j >> 1) + 16 * g + 8 * (j & 1) : 32 * j + 8 * g;
      qf[b][j] = u32x4_t{0, 0, 0, 0};
      if (ok && d0 < D) qf[b][j] = *reinterpret_cast<const u32x4_t*>(qrow + d0);
    }
  }
  const float alibi = (p.alibi_slopes != nullptr && head_ok) ? p.alibi_slopes[head0 + ch] * kLog2e : 0.f;
  const float qk_scale = p.scale * kLog2e;
  const bool use_alibi = p.alibi_slopes != nullptr;
  const bool use_cap = p.softcap > 0.f;
  const int window = p.sliding_window > 0 ? p.sliding_window : 0x3fffffff;

  const int last_block = p.max_num_blocks_per_seq - 1;
  // block_tables == nullptr: the dense path's scratch tiles, placed arithmetically -- sequence s
  // owns blocks cu_seqlens[s] / BS + s ..., which never overlap (prefill_attention.hip)
  const bool arithmetic_blocks = p.block_tables == nullptr;
  const int first_block = qbeg / BS + seq;
  // The table is read through the CONSTANT address space: that keeps the loads scalar (s_load,
  // lgkmcnt) also after the LDS copies and barriers of the LDSKV loop, which the compiler takes
  // for stores that might alias -- as vector loads they came with an s_waitcnt vmcnt(0) that
  // drained the copy pipeline every pair.
  typedef const int32_t __attribute__((address_space(4))) * const_i32_ptr;
  const const_i32_ptr block_table_c = (const_i32_ptr)(uintptr_t)block_table;
  auto block_number = [&](const int j) __attribute__((always_inline)) -> int {
    const int blk = min(((tile0 + j) << 4) / BS, last_block);
    return arithmetic_blocks ? first_block + blk : (int)min((uint32_t)block_table_c[blk], (uint32_t)p.max_block);
  };

  float m_run[NB], l_run[NB];
  f32x4_t acc[NB][NDT];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    m_run[b] = kMInit;
    l_run[b] = 0.f;
#pragma unroll
    for (int t = 0; t < NDT; ++t) acc[b][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // ---- unit of work: a PAIR of 16-token tiles (A, B) = 32 keys --------------------------------
  // S_A, S_B = K.Q^T per tile; one softmax step over the 32 keys; O^T += V^T.P^T with the
  // K=32 MFMA, whose k-slot 8g+e of lane (g, c) is (tile e>>2, token 4g + (e&3)) on BOTH operands:
  // A operand = {V_A piece, V_B piece} as loaded, B operand = {P_A, P_B} as computed.
  const int npairs = (ntiles + 1) >> 1;
  auto tile_rsrc = [&](const char* base, const int j, const int bn32) __attribute__((always_inline)) {
    const int64_t bn = bn32;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + bn * bsb), 0, j < ntiles ? kHeadBytes : 0,
                                             kSrdFlags);
  };
  auto tile_off = [&](const int j) __attribute__((always_inline)) -> int {
    return (BS == 32) ? (((tile0 + j) << 4) & 16) : 0;  // second half of a 32-token block
  };
  auto load_k = [&](u32x4_t (&k)[NS], const int j, const int bn32) __attribute__((always_inline)) {
    __amdgpu_buffer_rsrc_t kr = tile_rsrc(kbytes, j, bn32);
    const int off = tile_off(j);
#pragma unroll
    for (int jj = 0; jj < NS; ++jj)
      k[jj] = __builtin_amdgcn_raw_buffer_load_b128(kr, koff + jj * (4 * BS * 16), off * 16, 0);
  };
  // V pieces of the pair land directly in the 4-register MFMA operand: .xy = tile A, .zw = tile B
  auto load_v = [&](u32x4_t (&v)[NDT], const int half, const int j, const int bn32) __attribute__((always_inline)) {
    __amdgpu_buffer_rsrc_t vr = tile_rsrc(vbytes, j, bn32);
    const int off = tile_off(j);
#pragma unroll
    for (int t = 0; t < NDT; ++t) {
      const u32x2_t x = __builtin_amdgcn_raw_buffer_load_b64(vr, voff + t * (16 * BS * 2), off * 2, 0);
      if (half == 0) {
        v[t].x = x.x;
        v[t].y = x.y;
      } else {
        v[t].z = x.x;
        v[t].w = x.y;
      }
    }
  };
  // ---- LDSKV: stage layout [K tile A | K tile B | V tile A | V tile B], each tile compact:
  // K [D/8][16 tok][16 B] (= the MFMA A-operand order: slice j of lane l at j*1024 + l*16),
  // V [D][16 tok][2 B] (piece of lane (g, c), d-tile t at t*512 + c*32 + g*8).
  constexpr int kTile = D * 16 * KVB;      // bytes of one K (or V) tile
  constexpr int kStage = 4 * kTile;
  constexpr int kStages = LVLLM_PREFILL_STAGES;
  constexpr int kLoadsPerTile = kTile / 1024;  // wave loads of 16 bytes per lane
  extern __shared__ __attribute__((aligned(16))) char kv_lds[];
  // wave w copies one of the four tiles of every pair: K/V = w >> 1, tile A/B = w & 1
  const int ld_kind = wave >> 1, ld_tsel = wave & 1;
  auto issue_pair_loads = [&](const int jp, const int bn32) __attribute__((always_inline)) {
    const int j = 2 * jp + ld_tsel;
    const int tok_off = (BS == 32) ? (((tile0 + j) << 4) & 16) : 0;
    __amdgpu_buffer_rsrc_t r = tile_rsrc(ld_kind ? vbytes : kbytes, j, bn32);
    // lane's 16-byte chunk q = i*64 + lane of the compact tile, in the paged block:
    //   K chunk (d8 = q / 16, tok = q % 16) at (d8 * BS + tok_off + tok) * 16
    //   V chunk (d = q / 2, half = q % 2)    at (d * BS + tok_off + 8 * half) * 2
    int voffset, step, soff0;
    if (BS == 16) {
      voffset = lane * 16;
      step = 1024;
      soff0 = 0;
    } else if (KV8) {
      // fp8: K chunk (d16 = q / 16, tok = q % 16) at (d16 * BS + tok_off + tok) * 16; V chunk = row
      // d = q, its 16 tokens at d * BS + tok_off
      voffset = ld_kind == 0 ? (lane >> 4) * (BS * 16) + (lane & 15) * 16 : lane * BS;
      step = ld_kind == 0 ? 4 * BS * 16 : 64 * BS;
      soff0 = ld_kind == 0 ? tok_off * 16 : tok_off;
    } else if (ld_kind == 0) {
      voffset = (lane >> 4) * (BS * 16) + (lane & 15) * 16;
      step = 4 * BS * 16;
      soff0 = tok_off * 16;
    } else {
      voffset = (lane >> 1) * (BS * 2) + (lane & 1) * 16;
      step = 32 * BS * 2;
      soff0 = tok_off * 2;
    }
    char* dst = kv_lds + (jp % kStages) * kStage + (2 * ld_kind + ld_tsel) * kTile;
#pragma unroll
    for (int i = 0; i < kLoadsPerTile; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                               voffset, soff0 + i * step, 0, 0);
  };
  const bool k_scaled = KV8 && p.k_scale != 1.f, v_scaled = KV8 && p.v_scale != 1.f;
  auto read_pair_k = [&](u32x4_t (&ka)[NS], u32x4_t (&kb)[NS], const int jp) __attribute__((always_inline)) {
    const char* st = kv_lds + (jp % kStages) * kStage + lane * 16;
    if constexpr (KV8) {
#pragma unroll
      for (int i = 0; i < NS / 2; ++i) {
        const u32x4_t wa = *reinterpret_cast<const u32x4_t*>(st + i * 1024);
        const u32x4_t wb = *reinterpret_cast<const u32x4_t*>(st + kTile + i * 1024);
        const u32x2_t a0 = dequant4<T>(wa.x, p.k_scale, k_scaled), a1 = dequant4<T>(wa.y, p.k_scale, k_scaled);
        const u32x2_t a2 = dequant4<T>(wa.z, p.k_scale, k_scaled), a3 = dequant4<T>(wa.w, p.k_scale, k_scaled);
        ka[2 * i] = u32x4_t{a0.x, a0.y, a1.x, a1.y};
        ka[2 * i + 1] = u32x4_t{a2.x, a2.y, a3.x, a3.y};
        const u32x2_t b0 = dequant4<T>(wb.x, p.k_scale, k_scaled), b1 = dequant4<T>(wb.y, p.k_scale, k_scaled);
        const u32x2_t b2 = dequant4<T>(wb.z, p.k_scale, k_scaled), b3 = dequant4<T>(wb.w, p.k_scale, k_scaled);
        kb[2 * i] = u32x4_t{b0.x, b0.y, b1.x, b1.y};
        kb[2 * i + 1] = u32x4_t{b2.x, b2.y, b3.x, b3.y};
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < NS; ++jj) {
        ka[jj] = *reinterpret_cast<const u32x4_t*>(st + jj * 1024);
        kb[jj] = *reinterpret_cast<const u32x4_t*>(st + kTile + jj * 1024);
      }
    }
  };
  auto read_pair_v = [&](u32x4_t (&v)[NDT], const int jp) __attribute__((always_inline)) {
    if constexpr (KV8) {
      const char* st = kv_lds + (jp % kStages) * kStage + 2 * kTile + c * 16 + g * 4;  // row 16t + c, tokens 4g..
#pragma unroll
      for (int t = 0; t < NDT; ++t) {
        const u32x2_t a = dequant4<T>(*reinterpret_cast<const uint32_t*>(st + t * 256), p.v_scale, v_scaled);
        const u32x2_t b = dequant4<T>(*reinterpret_cast<const uint32_t*>(st + kTile + t * 256), p.v_scale, v_scaled);
        v[t] = u32x4_t{a.x, a.y, b.x, b.y};
      }
    } else {
      const char* st = kv_lds + (jp % kStages) * kStage + 2 * kTile + c * 32 + g * 8;
#pragma unroll
      for (int t = 0; t < NDT; ++t) {
        const u32x2_t a = *reinterpret_cast<const u32x2_t*>(st + t * 512);
        const u32x2_t b = *reinterpret_cast<const u32x2_t*>(st + kTile + t * 512);
        v[t] = u32x4_t{a.x, a.y, b.x, b.y};
      }
    }
  };

  // butterfly max over the 4 lane groups g (rows of 16 lanes) without LDS: the gfx950 row swaps.
  // (inline asm: hipcc folds the builtin form of swap(x, x) + max away.)
  auto group_max = [&](float x) __attribute__((always_inline)) -> float {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    b = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    return a;
  };
  auto zero_tail = [&](u32x4_t (&v)[NDT], const int half, const int tile_base) __attribute__((always_inline)) {
    // V beyond the sequence may hold anything (NaN included): zero it (only the last tile can)
    if (tile_base + 16 > seq_len) {
      const int tok0 = tile_base + 4 * g;
      uint32_t mx = 0xffffffffu, my = 0xffffffffu;
      if (tok0 + 0 >= seq_len) mx &= 0xffff0000u;
      if (tok0 + 1 >= seq_len) mx &= 0x0000ffffu;
      if (tok0 + 2 >= seq_len) my &= 0xffff0000u;
      if (tok0 + 3 >= seq_len) my &= 0x0000ffffu;
#pragma unroll
      for (int t = 0; t < NDT; ++t) {
        if (half == 0) {
          v[t].x &= mx;
          v[t].y &= my;
        } else {
          v[t].z &= mx;
          v[t].w &= my;
        }
      }
    }
  };

  // every live column of the wave sees all keys up to this one
  const int q_first_pos = p.causal ? ctx + t_first : seq_len - 1;
  // exponent = y * kf - m * kf: the plain path keeps y = raw q.k and folds scale*log2(e) into
  // the FMA; with extras y is already the biased logit in log2 units
  const float kf = EXTRAS ? 1.f : qk_scale;

  u32x4_t kA[NS], kB[NS];
  u32x4_t vv[NDT];
  int bnA = 0, bnB = 0;
  if constexpr (!LDSKV) {
    bnA = block_number(0);
    bnB = block_number(1);
    load_k(kA, 0, bnA);
    load_k(kB, 1, bnB);
    load_v(vv, 0, 0, bnA);
    load_v(vv, 1, 1, bnB);
    bnA = block_number(2);
    bnB = block_number(3);
  }
  // One pair.  MASKED is a compile-time tag: the pairs wholly at or before the wave's first query
  // (all but the last one or two) run a copy of the body with no mask and no branch but the rescale.
  auto pair_step = [&](const int jp, auto masked_tag) __attribute__((always_inline)) {
    constexpr bool need_mask = decltype(masked_tag)::value;
    const int j = 2 * jp;
    const int base = (tile0 + j) << 4;
    if constexpr (LDSKV) {
      read_pair_k(kA, kB, jp);
      read_pair_v(vv, jp);
    }
    if constexpr (need_mask) {
      zero_tail(vv, 0, base);
      zero_tail(vv, 1, base + 16);
    }
    // ---- S = K.Q^T for both tiles and all column blocks, then the K registers are free ----
    f32x4_t sA[NB], sB[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      sA[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      sB[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jj = 0; jj < NS; ++jj) sA[b] = mfma_qk<T>(kA[jj], qf[b][jj], sA[b]);
#pragma unroll
      for (int jj = 0; jj < NS; ++jj) sB[b] = mfma_qk<T>(kB[jj], qf[b][jj], sB[b]);
    }
#if LVLLM_PREFILL_EXP < 2  // diagnosis builds: 1 = no V stream, 2 = no K stream either (wrong results)
    if constexpr (!LDSKV) {
      load_k(kA, j + 2, bnA);
      load_k(kB, j + 3, bnB);
    }
#endif
    // (1) logits, masks and the new running max of every column block
    float y[NB][8];
    float m_new[NB];
    bool grew = false;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y[b][r] = sA[b][r];
        y[b][4 + r] = sB[b][r];
      }
      if constexpr (EXTRAS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int tok = base + ((e >> 2) << 4) + 4 * g + (e & 3);
          float z = y[b][e] * qk_scale;
          if (use_cap) z = p.softcap * kLog2e * tanhf(y[b][e] * p.scale / p.softcap);
          if (use_alibi) z += alibi * (float)(tok - qpos[b]);
          y[b][e] = z;
        }
      }
      if constexpr (need_mask) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int tok = base + ((e >> 2) << 4) + 4 * g + (e & 3);
          const int dist = vlast[b] - tok;  // >= 0: visible
          const bool vis = EXTRAS ? (dist >= 0 && dist < window) : dist >= 0;
          y[b][e] = vis ? y[b][e] : kMasked;
        }
      }
      // plain fmaxf: this file is built with -fno-honor-nans (build.py), so no canonicalising
      // v_max x,x,x; the compiler sees MFMA results being read and inserts the wait states
      float m_loc = fmaxf(fmaxf(fmaxf(y[b][0], y[b][1]), fmaxf(y[b][2], y[b][3])),
                          fmaxf(fmaxf(y[b][4], y[b][5]), fmaxf(y[b][6], y[b][7])));
      m_loc = group_max(m_loc);
      m_new[b] = fmaxf(m_run[b], m_loc);
      grew |= m_new[b] > m_run[b];
    }
    // (2) one wave-uniform branch: rescale the accumulators only when some running max grew
    if (__builtin_amdgcn_ballot_w64(grew) != 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float alpha = __builtin_amdgcn_exp2f((m_run[b] - m_new[b]) * kf);
        l_run[b] *= alpha;
#pragma unroll
        for (int t = 0; t < NDT; ++t) acc[b][t] *= alpha;
        m_run[b] = m_new[b];
      }
    }
    // (3) one straight-line region.  Issue order (an MFMA leaves half of its 16 cycles to the
    // vector pipe, and a wave issues in order, so the interleaving has to be in the program):
    //   exponentials of block 0 | P.V MFMAs of block b-1 interleaved with the exponentials of
    //   block b | P.V MFMAs of the last block
    auto exponentials = [&](const int b) __attribute__((always_inline)) -> u32x4_t {
      const float mk = m_new[b] * kf;
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        y[b][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(y[b][e], kf, -mk));
        psum += y[b][e];
      }
      l_run[b] += psum;
      return u32x4_t{pack2<T>(y[b][0], y[b][1]), pack2<T>(y[b][2], y[b][3]), pack2<T>(y[b][4], y[b][5]),
                     pack2<T>(y[b][6], y[b][7])};
    };
    u32x4_t pb = exponentials(0);
#pragma unroll
    for (int b = 1; b < NB; ++b) {
#if LVLLM_PREFILL_SCHED
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int t = 0; t < NDT; ++t) acc[b - 1][t] = mfma_qk<T>(vv[t], pb, acc[b - 1][t]);
      pb = exponentials(b);
#if LVLLM_PREFILL_SCHED
#pragma unroll
      for (int t = 0; t < NDT; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, (28 + NDT - 1) / NDT, 0);
      }
#endif
    }
#if LVLLM_PREFILL_SCHED
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int t = 0; t < NDT; ++t) acc[NB - 1][t] = mfma_qk<T>(vv[t], pb, acc[NB - 1][t]);
    if constexpr (!LDSKV) {
#if LVLLM_PREFILL_EXP < 1
      load_v(vv, 0, j + 2, bnA);
      load_v(vv, 1, j + 3, bnB);
#endif
      bnA = block_number(j + 4);
      bnB = block_number(j + 5);
    }
  };
  {
    // pairs whose last key (base + 31) is at or before the wave's first query need no mask
    const int n_plain = EXTRAS ? 0 : min(my_npairs, max(0, (q_first_pos + 1 - (tile0 << 4)) >> 5));
    if constexpr (!LDSKV) {
      int jp = 0;
      for (; jp < n_plain; ++jp) pair_step(jp, std::false_type{});
      for (; jp < npairs; ++jp) pair_step(jp, std::true_type{});
    } else {
      // the stages start as zeros: a tile past the walk is never loaded (zero-size descriptor)
      // and must still read as finite numbers
      for (int i = threadIdx.x; i < kStages * kStage / 16; i += 256)
        reinterpret_cast<u32x4_t*>(kv_lds)[i] = u32x4_t{0, 0, 0, 0};
      __syncthreads();
      // copies run kStages-1 pairs ahead of the multiplies (every pair is issued, valid or not,
      // so that the number of loads in flight is a compile-time constant)
#pragma unroll
      for (int a = 0; a < kStages - 1; ++a) issue_pair_loads(a, block_number(2 * a + ld_tsel));
      int bn = block_number(2 * (kStages - 1) + ld_tsel);
      // s_waitcnt vmcnt(N) lgkmcnt(0): all but this wave's N youngest copies have landed
      constexpr int kInFlight = (kStages - 2) * kLoadsPerTile;
      constexpr int kWait = (kInFlight & 15) | (7 << 4) | (0 << 8) | ((kInFlight >> 4) << 14);
      for (int jp = 0; jp < npairs; ++jp) {
        __builtin_amdgcn_s_waitcnt(kWait);  // this wave's copies of pair jp are in LDS ...
        __builtin_amdgcn_s_barrier();       // ... and everybody's; everybody is done reading pair jp-1
        issue_pair_loads(jp + kStages - 1, bn);  // into the stage pair jp-1 used
        bn = block_number(2 * (jp + kStages) + ld_tsel);
        if (jp < n_plain) pair_step(jp, std::false_type{});
        else if (jp < my_npairs) pair_step(jp, std::true_type{});
      }
      __builtin_amdgcn_s_waitcnt(0);  // no copy may still be landing when the workgroup's LDS is released
    }
  }

  // ---- normalise and store: lane (g, c) holds d = 16t + 4g .. +3 of its column ----
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float l = l_run[b];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? __fdividef(1.f, l) : 0.f;
    if (qpos[b] >= 0) {
      const int t = b * TQB + cq;
      S* orow = (S*)p.out + (int64_t)(qbeg + t_first + t) * p.out_stride + (int64_t)(head0 + ch) * D;
#pragma unroll
      for (int tt = 0; tt < NDT; ++tt) {
        const int d = 16 * tt + 4 * g;
        if (d < D) {
          u32x2_t o;
          o.x = pack2<T>(acc[b][tt][0] * inv, acc[b][tt][1] * inv);
          o.y = pack2<T>(acc[b][tt][2] * inv, acc[b][tt][3] * inv);
          *reinterpret_cast<u32x2_t*>(orow + d) = o;
        }
      }
    }
  }
}

#ifndef LVLLM_PREFILL_NB
#define LVLLM_PREFILL_NB 2
#endif

// the 32x32-MFMA body for long plain chunks (prefill_mfma32.h)
template <typename T, int D, int BS>
static int launch_prefill_mfma32(const PrefillParams& p0, int num_seqs, int max_query_len, hipStream_t stream);

// Whether a plain launch of head size 64 / 128 takes the 32x32-MFMA body: chunks of prefill_mfma32_min_query tokens
// up always.  Shorter chunks (16 tokens up) take it too when its whole grid fits the CUs at once: the launch then
// lasts as long as its slowest workgroup, and the 64-key tile walk is the faster one whatever the number of live
// columns (1 x (32 over 4 096): 74 against 139 us; profiles/r02_prefill_threshold_ab.txt).  Larger launches
// with short chunks are typically mixed steps full of one-token sequences.
inline bool takes_mfma32(const PrefillParams& p0, int num_seqs, int max_query_len) {
  const int min_query = tuning().prefill_mfma32_min_query;
  bool take32 = min_query > 0 && max_query_len >= min_query;
  if (!take32 && min_query > 0 && max_query_len >= 16) {
    static const int num_cus = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
      return n;
    }();
    const int G_ = p0.num_heads / p0.num_kv_heads;
    const int cols_per_token = G_ == 1 ? 1 : G_ == 2 ? 2 : G_ <= 4 ? 4 : G_ <= 8 ? 8 : G_ <= 16 ? 16 : 32;
    const int tqwg = 256 / cols_per_token;
    const int64_t wgs = (int64_t)p0.num_kv_heads * ((G_ + 31) / 32) * num_seqs * ((max_query_len + tqwg - 1) / tqwg);
    take32 = wgs <= num_cus;
  }
  return take32;
}

// launches that are mostly one-token sequences: the decode-style walk (prefill_chunk.h)
template <typename T, int D, int BS>
static int launch_prefill_chunk(const PrefillParams& p0, int num_seqs, int max_query_len, hipStream_t stream);
inline bool chunk_kernel_takes(const PrefillParams& p, int head_size, int num_seqs, int max_query_len);

template <typename T, int D, int BS>
static int launch_prefill_gp(const PrefillParams& p0, int num_seqs, int max_query_len, hipStream_t stream) {
  constexpr int NB = D > 128 ? 1 : LVLLM_PREFILL_NB;  // accumulators: NB * D/4 VGPRs per lane
  if constexpr (D == 64 || D == 128) {
    if (p0.causal && !p0.kv_fp8 && p0.alibi_slopes == nullptr && p0.softcap <= 0.f && p0.sliding_window <= 0 &&
        chunk_kernel_takes(p0, D, num_seqs, max_query_len))
      return launch_prefill_chunk<T, D, BS>(p0, num_seqs, max_query_len, stream);
    const bool take32 = takes_mfma32(p0, num_seqs, max_query_len);
    if (take32 && !p0.kv_fp8 && p0.alibi_slopes == nullptr && p0.softcap <= 0.f && p0.sliding_window <= 0)
      return launch_prefill_mfma32<T, D, BS>(p0, num_seqs, max_query_len, stream);
  }
  PrefillParams p = p0;
  const int G = p.num_heads / p.num_kv_heads;
  const int HG = (G + 15) / 16;
  p.gp_shift = G == 1 ? 0 : G == 2 ? 1 : G <= 4 ? 2 : G <= 8 ? 3 : 4;
  const int tqwg = 4 * NB * (16 >> p.gp_shift);
  const int qtiles = (max_query_len + tqwg - 1) / tqwg;
  const bool extras = p.alibi_slopes != nullptr || p.softcap > 0.f || p.sliding_window > 0;
  const dim3 grid(p.num_kv_heads * HG, num_seqs, qtiles);
  if constexpr (D % 32 == 0) {
    if (tuning().prefill_lds) {
      const size_t smem = (size_t)LVLLM_PREFILL_STAGES * 4 * D * 16 * (p.kv_fp8 ? 1 : 2);  // [K A | K B | V A | V B]
      auto launch = [&](auto kern) {
        if (smem > 64 * 1024)
          (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, p);
      };
      if constexpr (D % 64 == 0) {
        if (p.kv_fp8) {
          if (extras) launch(paged_prefill_mfma_kernel<T, D, BS, NB, true, true, true>);
          else launch(paged_prefill_mfma_kernel<T, D, BS, NB, false, true, true>);
          return 0;
        }
      }
      LV_CHECK(!p.kv_fp8, "fp8 kv cache: the prefill kernel needs a head size that is a multiple of 64");
      if (extras) launch(paged_prefill_mfma_kernel<T, D, BS, NB, true, true>);
      else launch(paged_prefill_mfma_kernel<T, D, BS, NB, false, true>);
      return 0;
    }
  }
  LV_CHECK(!p.kv_fp8, "fp8 kv cache: the prefill kernel needs the LDS path and a head size multiple of 64");
  if (extras)
    hipLaunchKernelGGL((paged_prefill_mfma_kernel<T, D, BS, NB, true, false>), grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((paged_prefill_mfma_kernel<T, D, BS, NB, false, false>), grid, dim3(256), 0, stream, p);
  return 0;
}

template <typename T, int D>
static int launch_prefill_bs(const PrefillParams& p, int block_size, int num_seqs, int max_query_len,
                             hipStream_t stream) {
  switch (block_size) {
    case 16: return launch_prefill_gp<T, D, 16>(p, num_seqs, max_query_len, stream);
    case 32: return launch_prefill_gp<T, D, 32>(p, num_seqs, max_query_len, stream);
    default: LV_CHECK(false, "unsupported block size " + std::to_string(block_size));
  }
  return 0;
}

template <typename T>
int launch_prefill_hs(const PrefillParams& p, int head_size, int block_size, int num_seqs,
                      int max_query_len, hipStream_t stream) {
  switch (head_size) {
#define LV_HS(D_) \
  case D_: return launch_prefill_bs<T, D_>(p, block_size, num_seqs, max_query_len, stream);
    LV_HS(64) LV_HS(80) LV_HS(96) LV_HS(112) LV_HS(120) LV_HS(128) LV_HS(192) LV_HS(256)
#undef LV_HS
    default: LV_CHECK(false, "unsupported head size " + std::to_string(head_size));
  }
  return 0;
}

}  // namespace lvllm
