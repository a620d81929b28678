// Mixed steps of chunked prefill: launches that are mostly ONE-token sequences with a short prompt chunk beside them
// -- the shape a step of the reference's scheduler has most of the time under chunked prefill
// (benchmark_chunked_prefill_throughput.py: a budget of 64 tokens = ~32 decode rows + one chunk of the rest; the live
// backend hands both kinds to flash-attn over the paged cache, flash_attn.py:538-555).
//
// Same call, semantics and parameter block as prefill_mfma.h.  What changes is the shape of the work.  The prefill
// bodies walk the keys of a sequence serially inside one workgroup, 64 keys per barrier, with K/V staged in LDS for
// 128 or 256 query columns: a one-token sequence lights 4 of those columns (GQA group of 4) and nothing splits its
// keys -- such a step ran at the speed of its longest context.  Here the walk is the DECODE kernel's
// (attention_mfma.h): the paged K layout is the MFMA A operand straight from global memory, the 8 waves of a
// workgroup own interleaved 16-key tiles and keep the next one in flight with zero-size descriptors past the end,
// V pieces are masked 8-byte loads, the waves are merged once through LDS.  Added to it:
//   * the 16 MFMA columns are (query token, head of the GQA group) pairs, 16 / GP tokens per workgroup, each column
//     with its own causal horizon: a one-token sequence is one workgroup per kv head (the decode kernel's
//     arithmetic), a chunk of n tokens n GP / 16 of them;
//   * the grid is the list of query groups of the launch, counted over its sequences by each workgroup from
//     query_start_loc (one vector load + a wave scan): no empty workgroups for ragged chunk lengths;
//   * the key range can be cut into partitions across workgroups (blockIdx.z), merged by
//     prefill_chunk_reduce_kernel -- paged_attention_v2's scheme with per-row horizons -- when the launch would not
//     fill the CUs otherwise;
//   * logits in base 2 (v_exp_f32 is 2^x, log2(e) folded into the scale), the tile maximum over the four 16-lane rows
//     by v_permlane16/32_swap instead of LDS round trips: 32 one-token sequences at 1 024 run in 21.8 us
//     (tools/bench_chunk_attn.py; 31.5 through the prefill body).
// The kernel is written for NCG column groups per workgroup (NCG x 16 / GP tokens against one read of K/V, Q
// fragments in LDS, groups taken two at a time, K and V of the next tile requested apart so that NCG x D/4
// accumulator registers fit) and ships with NCG = 1: with 4 groups a tile costs a wave ~4 000 issue cycles (48 MFMAs
// + ~300 vector instructions), the launch is instruction-bound, and launches of short chunks ONLY stay with the
// 32x32 body, which is faster there (profiles/r03_tuning.md section 7).
// Plain causal attention over 16-bit caches, head size 64 or 128, GQA groups of at most 16: everything else stays
// with prefill_mfma.h.
#pragma once
#include "attention_mfma.h"
#include "prefill_partitions.h"

#ifndef LVLLM_CHUNK_NBUF
#define LVLLM_CHUNK_NBUF 2
#endif

namespace lvllm {

// The lane number, recomputed where it is used: the asm "depends" on the tile counter, so it is not hoisted out of
// the tile loop (and it is not volatile: a volatile asm counts as a store to anything, which turns the scalar
// block-table loads into vector loads behind an s_waitcnt vmcnt(0)).  Everything a lane derives from its number --
// its K / V offsets, its Q address in LDS, its columns' horizons -- then lives for a few instructions instead of
// across the loop: the walk is 3-6 registers over 256 otherwise, and each of those comes back from scratch, again
// behind an s_waitcnt vmcnt(0): the prefetch is gone.
__device__ __forceinline__ int lane_now(const int tile_counter) {
  int l;
  asm("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l) : "s"(tile_counter));
  return l;
}

template <typename T, int D, int BS, int NWAVES, int NCG, int NBUF>
__global__ __launch_bounds__(NWAVES * 64, 2) void paged_prefill_chunk_kernel(const PrefillParams p,
                                                                             const ChunkScratch sc,
                                                                             const int qgroups) {
  using S = typename T::store_t;
  static_assert(sizeof(S) == 2, "16-bit element types");
  static_assert(BS == 16 || BS == 32, "a 16-key tile lies inside one block");
  static_assert(D % 32 == 0 && D <= 128, "head size 64 / 128");
  constexpr int NS = D / 32;   // k-slices of S = K.Q^T
  constexpr int NDT = D / 16;  // 16-row d-tiles of O^T += V^T.P^T
  constexpr int kHeadBytes = D * BS * 2;
  constexpr float kLog2e = 1.4426950408889634f;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;

  const int GP = 1 << p.gp_shift;    // heads per query token among the 16 columns of a group
  const int TPG = 16 >> p.gp_shift;  // query tokens per column group
  const int TQ = NCG * TPG;          // query tokens per workgroup
  const int G = p.num_heads / p.num_kv_heads;
  const int kvh = blockIdx.x;
  const int head0 = kvh * G;
  // blockIdx.y: `qgroups` > 0 -- (sequence, query group) on a rectangle, most of it empty when the chunks are
  // ragged; `qgroups` <= 0 -- item number -qgroups = sequences of the launch: the y-th query group of the launch,
  // counted over the sequences in order (a launch then has ceil(T / TQ) + num_seqs workgroup rows at most: in a
  // mixed step of one 32-token chunk beside 32 one-token sequences the rectangle was 7/8 empty workgroups).
  int seq, qg, qbeg, qlen, seq_len;
  if (qgroups > 0) {
    seq = blockIdx.y / qgroups;
    qg = blockIdx.y - seq * qgroups;
    qbeg = p.query_start_loc[seq];
    qlen = p.query_start_loc[seq + 1] - qbeg;
    seq_len = p.seq_lens[seq];
  } else {
    const int num_seqs = -qgroups;
    const int item = blockIdx.y;
    int run = 0;
    seq = -1;
    qg = qbeg = qlen = seq_len = 0;
    for (int base = 0; base < num_seqs; base += 64) {
      // lane s: one sequence's extents and length, all in one round trip; the lane that owns the item hands them on
      const int s = min(base + lane, num_seqs - 1);
      const int q0 = p.query_start_loc[s], q1 = p.query_start_loc[s + 1], sl = p.seq_lens[s];
      const int ng = base + lane < num_seqs ? (q1 - q0 + TQ - 1) / TQ : 0;
      int incl = ng;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
      }
      const uint64_t hit = __builtin_amdgcn_ballot_w64(run + incl > item);
      if (hit != 0) {
        const int l = __ffsll((long long)hit) - 1;
        seq = base + l;
        qg = item - (run + __builtin_amdgcn_readlane(incl, l) - __builtin_amdgcn_readlane(ng, l));
        qbeg = __builtin_amdgcn_readlane(q0, l);
        qlen = __builtin_amdgcn_readlane(q1, l) - qbeg;
        seq_len = __builtin_amdgcn_readlane(sl, l);
        break;
      }
      run += __builtin_amdgcn_readlane(incl, 63);
    }
    if (seq < 0) return;  // past the last query group
    seq = __builtin_amdgcn_readfirstlane(seq);
    qg = __builtin_amdgcn_readfirstlane(qg);
  }
  const int part = blockIdx.z;
  const int ctx = seq_len - qlen;
  const int t_first = qg * TQ;
  if (t_first >= qlen || ctx < 0) return;
  const int nq = min(TQ, qlen - t_first);
  const int ncg_live = (nq + TPG - 1) / TPG;  // column groups with a live token (workgroup-uniform)
  const int khi = ctx + t_first + nq;         // keys the last token of this workgroup sees

  int t0 = 0, t1 = khi;
  if (sc.tmp_out != nullptr) {
    // (the last partition is open-ended: a context longer than the caller's max_seq_len is walked, not cut off)
    t0 = part * sc.part_tokens;
    t1 = part + 1 == sc.num_parts ? khi : min(khi, t0 + sc.part_tokens);
    if (t0 >= khi) return;  // no row of this workgroup reaches the partition: the reduce never reads it
  }
  const int ntiles = (t1 - t0 + 15) >> 4;
  const int tile0 = t0 >> 4;
  const int nmy = ntiles > wave ? (ntiles - wave + NWAVES - 1) / NWAVES : 0;

  const int32_t* block_table = p.block_tables + (int64_t)seq * p.max_num_blocks_per_seq;
  const char* kbytes = (const char*)p.k_cache + (int64_t)kvh * p.kv_head_stride * 2;
  const char* vbytes = (const char*)p.v_cache + (int64_t)kvh * p.kv_head_stride * 2;
  const int64_t bsb = p.kv_block_stride * 2;
  const int last_block = p.max_num_blocks_per_seq - 1;

  auto block_number = [&](const int j) __attribute__((always_inline)) -> int {
    const int blk = ((tile0 + wave + j * NWAVES) << 4) / BS;
    return (int)min((uint32_t)block_table[min(blk, last_block)], (uint32_t)p.max_block);
  };
  // K and V of a tile are requested apart: K of tile j+1 before tile j is multiplied, V of tile j+1 once K.Q^T of
  // tile j has consumed K of tile j -- at most three of the four register quarters (K, V) x (this tile, next tile)
  // are live at any time, which is what lets NCG x D/4 accumulator registers fit beside them without spills (a spill
  // inside the loop is a scratch access, and scratch accesses wait on vmcnt(0): the prefetch is gone).
  auto load_k = [&](u32x4_t (&k)[NS], const int j, const int bn) __attribute__((always_inline)) {
    const int lt = wave + j * NWAVES;
    const int off = (BS == 32) ? (((tile0 + lt) << 4) & 16) : 0;  // second half of a 32-token block
    __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc((void*)(kbytes + (int64_t)bn * bsb), 0,
                                                                  j < nmy ? kHeadBytes : 0, kSrdFlags);
    const int ln = lane_now(j);
    const int koff = ((ln >> 4) * BS + (ln & 15)) * 16;  // K chunk (d8 = g (+4j), token c)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < NS; ++jj)
      k[jj] = __builtin_amdgcn_raw_buffer_load_b128(kr, koff + jj * (4 * BS * 16), off * 16, LVLLM_ATTN_AUX);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_v = [&](u32x2_t (&v)[NDT], const int j, const int bn) __attribute__((always_inline)) {
    const int lt = wave + j * NWAVES;
    const int off = (BS == 32) ? (((tile0 + lt) << 4) & 16) : 0;
    __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void*)(vbytes + (int64_t)bn * bsb), 0,
                                                                  j < nmy ? kHeadBytes : 0, kSrdFlags);
    const int ln = lane_now(j);
    const int voff = ((ln & 15) * BS + 4 * (ln >> 4)) * 2;  // V piece (row c (+16t), tokens 4g..4g+3)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NDT; ++t)
      v[t] = __builtin_amdgcn_raw_buffer_load_b64(vr, voff + t * (16 * BS * 2), off * 2, LVLLM_ATTN_AUX);
    __builtin_amdgcn_sched_barrier(0);
  };

  static_assert(NBUF == 2, "two register sets: this tile and the next");
  u32x4_t kset[2][NS];
  u32x2_t vset[2][NDT];
  int bn_next = block_number(0);
  load_k(kset[0], 0, bn_next);
  load_v(vset[0], 0, bn_next);
  bn_next = block_number(1);

  // ---- this lane's columns: group cg, column c = (query token t_first + cg TPG + c / GP, head head0 + c % GP) ----
  // The Q fragments (B operands of K.Q^T) live in LDS, [cg][k-slice][lane] x 16 bytes, read back per MFMA: NCG x D/8
  // registers of Q beside NCG x D/4 of accumulators and two K/V sets do not fit 256 registers, and with the
  // accumulators pushed into the AGPR half the compiler moved them in and out around every MFMA (57 us instead of 22).
  const int ch = c & (GP - 1), cq = c >> p.gp_shift;
  u32x4_t* sm_q = reinterpret_cast<u32x4_t*>(smem_raw + (size_t)NWAVES * 16 * 2 * sizeof(float) +
                                             (size_t)NWAVES * 16 * D * sizeof(float));
  static_assert(NCG <= NWAVES, "wave w < NCG stages the Q fragments of column group w");
  if (wave < NCG) {
    const int tq = t_first + wave * TPG + cq;
    const bool live = ch < G && tq < qlen;
    const S* qrow = (const S*)p.q + (int64_t)(qbeg + tq) * p.q_stride + (int64_t)(head0 + ch) * D;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      u32x4_t qv = u32x4_t{0, 0, 0, 0};
      if (live) qv = *reinterpret_cast<const u32x4_t*>(qrow + 32 * j + 8 * g);
      sm_q[(wave * NS + j) * 64 + lane] = qv;
    }
  }
  __syncthreads();
  const float kf = p.scale * kLog2e;  // logits are kept in base 2 (v_exp_f32 is 2^x)

  float m_run[NCG], l_run[NCG];
  f32x4_t acc[NCG][NDT];
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
    m_run[cg] = -FLT_MAX;
    l_run[cg] = 0.f;
#pragma unroll
    for (int t = 0; t < NDT; ++t) acc[cg][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // One tile against the first NL column groups (NL = the workgroup's live groups, a compile-time constant of the
  // loop it runs in: as run-time guards these were 40 branches per tile, each a scheduling fence).
  auto compute_tile = [&](auto nl_tag, u32x4_t (&kraw)[NS], u32x2_t (&v)[NDT], u32x2_t (&v_next)[NDT], const int j,
                          const int bn_of_next) __attribute__((always_inline)) {
    constexpr int NL = decltype(nl_tag)::value;
    const int ln = lane_now(j);
    const int tok0 = ((tile0 + wave + j * NWAVES) << 4) + 4 * (ln >> 4);
    const int cq_ = (ln & 15) >> p.gp_shift;
    const int vbase = ctx + t_first + cq_;  // last key column (cg, c) sees: vbase + cg TPG, if the column is live
    const bool head_live = (ln & (GP - 1)) < G;
    // Column groups are taken two at a time (K.Q^T of both, then both softmax chains side by side): a wave has one
    // partner on its SIMD at most, and group after group every step waited for the one before; all four at once
    // needs 16 logit registers more than there are.
    u32x2_t pb[NL];
    float alpha[NL];
    bool moved = false;
#pragma unroll
    for (int c0 = 0; c0 < NL; c0 += 2) {
      constexpr int kPairMax = 2;
      const int np = NL - c0 < kPairMax ? NL - c0 : kPairMax;
      f32x4_t sc4[kPairMax];
#pragma unroll
      for (int i = 0; i < kPairMax; ++i) sc4[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jj = 0; jj < NS; ++jj)
#pragma unroll
        for (int i = 0; i < kPairMax; ++i)
          if (i < np) sc4[i] = mfma_qk<T>(kraw[jj], sm_q[((c0 + i) * NS + jj) * 64 + ln], sc4[i]);
      if (c0 + 2 >= NL) load_v(v_next, j + 1, bn_of_next);  // (K of this tile is consumed: its registers are free)
#pragma unroll
      for (int i = 0; i < kPairMax; ++i) {
        if (i >= np) continue;
        const int cg = c0 + i;
        const bool col_live = head_live && t_first + cg * TPG + cq_ < qlen;
        const int lim = min(col_live ? vbase + cg * TPG : -1, t1 - 1);  // causal horizon, partition end
        float x[4];
        float m_loc = -FLT_MAX;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x[r] = (tok0 + r <= lim) ? sc4[i][r] * kf : -FLT_MAX;
          m_loc = fmaxf(m_loc, x[r]);
        }
        m_loc = rows_max(m_loc);  // over the four 16-lane rows = the tile's 16 keys
        const float m_new = fmaxf(m_run[cg], m_loc);
        alpha[cg] = __builtin_amdgcn_exp2f(m_run[cg] - m_new);
        moved = moved || alpha[cg] != 1.f;
        m_run[cg] = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x[r] = (tok0 + r <= lim) ? __builtin_amdgcn_exp2f(x[r] - m_new) : 0.f;
          psum += x[r];
        }
        l_run[cg] = l_run[cg] * alpha[cg] + psum;
        pb[cg].x = pack2<T>(x[0], x[1]);
        pb[cg].y = pack2<T>(x[2], x[3]);
      }
    }
    // the accumulators are rescaled only when some column's maximum moved (rare after the first tiles)
    if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
      for (int cg = 0; cg < NL; ++cg)
#pragma unroll
        for (int t = 0; t < NDT; ++t) acc[cg][t] *= alpha[cg];
    }
    // V of keys past the end of the partition / context may hold anything (NaN included): zero it
    uint32_t mx = 0xffffffffu, my = 0xffffffffu;
    if (tok0 + 0 >= t1) mx &= 0xffff0000u;
    if (tok0 + 1 >= t1) mx &= 0x0000ffffu;
    if (tok0 + 2 >= t1) my &= 0xffff0000u;
    if (tok0 + 3 >= t1) my &= 0x0000ffffu;
#pragma unroll
    for (int t = 0; t < NDT; ++t) {
      u32x2_t va = v[t];
      va.x &= mx;
      va.y &= my;
#pragma unroll
      for (int cg = 0; cg < NL; ++cg) acc[cg][t] = mfma_pv<T>(va, pb[cg], acc[cg][t]);
    }
  };

  auto walk = [&](auto nl_tag) __attribute__((always_inline)) {
    for (int j = 0; j < nmy; j += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int bn = bn_next;
        load_k(kset[1 - u], j + u + 1, bn);
        bn_next = block_number(j + u + 2);
        compute_tile(nl_tag, kset[u], vset[u], vset[1 - u], j + u, bn);
      }
    }
  };
  static_assert(NCG == 1 || NCG == 2 || NCG == 4, "the switch below lists the live-group counts");
  if constexpr (NCG == 1) {
    walk(std::integral_constant<int, 1>{});
  } else if constexpr (NCG == 2) {
    if (ncg_live == 1) walk(std::integral_constant<int, 1>{});
    else walk(std::integral_constant<int, 2>{});
  } else {
    switch (ncg_live) {
      case 1: walk(std::integral_constant<int, 1>{}); break;
      case 2: walk(std::integral_constant<int, 2>{}); break;
      case 3: walk(std::integral_constant<int, 3>{}); break;
      default: walk(std::integral_constant<int, 4>{}); break;
    }
  }

  // ---- merge the waves, one column group at a time through the same 16-column LDS area ----
  float* sm_m = reinterpret_cast<float*>(smem_raw);  // [NWAVES][16]
  float* sm_l = sm_m + NWAVES * 16;                  // [NWAVES][16]
  float* sm_acc = sm_l + NWAVES * 16;                // [NWAVES][16][D]
  const bool partitioned = sc.tmp_out != nullptr;
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
    if (cg >= ncg_live) break;
    float l = l_run[cg];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (cg > 0) __syncthreads();  // the previous group has been read
    if (g == 0) {
      sm_m[wave * 16 + c] = m_run[cg];
      sm_l[wave * 16 + c] = l;
    }
    {
      float* dst = sm_acc + ((int64_t)(wave * 16 + c)) * D + 4 * g;
#pragma unroll
      for (int t = 0; t < NDT; ++t) *reinterpret_cast<f32x4_t*>(dst + 16 * t) = acc[cg][t];  // d = 16t + 4g + r
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 16 * D; idx += NWAVES * 64) {
      const int col = idx / D, d = idx - col * D;
      const int h = col & (GP - 1), tq = t_first + cg * TPG + (col >> p.gp_shift);
      if (h >= G || tq >= qlen) continue;
      float M = -FLT_MAX;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) M = fmaxf(M, sm_m[w * 16 + col]);
      float L = 0.f, o = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) {
        const float f = __builtin_amdgcn_exp2f(sm_m[w * 16 + col] - M);
        L += sm_l[w * 16 + col] * f;
        o += sm_acc[(w * 16 + col) * D + d] * f;
      }
      o *= L > 0.f ? __fdividef(1.f, L) : 0.f;
      if (!partitioned) {
        reinterpret_cast<S*>(p.out)[(int64_t)(qbeg + tq) * p.out_stride + (int64_t)(head0 + h) * D + d] =
            T::from_float(o);
      } else {
        const int64_t row = ((int64_t)(qbeg + tq) * p.num_heads + head0 + h) * sc.num_parts + part;
        reinterpret_cast<S*>(sc.tmp_out)[row * D + d] = T::from_float(o);
        if (d == 0) {
          sc.max_logits[row] = M;
          sc.exp_sums[row] = L;
        }
      }
    }
  }
}

// The shape of a launch: query groups per sequence, partitions of the key walk, scratch.
struct ChunkPlan {
  int gp_shift, qgroups, parts, part_tokens;
  int64_t rows, ws_bytes;  // rows = scratch rows per partition (an upper bound of tokens x heads)
};
constexpr int kChunkNCG = 1, kChunkWaves = 8, kChunkMaxParts = kMaxPartitions;  // (NCG 2 and 4 build and pass the tests: slower)

// Partitions: when the workgroups of a single pass would leave CUs idle and the contexts are long enough to cut
// (>= 8 tiles per partition: two per wave).  max_seq_len is the caller's bound on seq_lens (0: unknown, no cut).
inline ChunkPlan chunk_plan(int num_seqs, int max_query_len, int num_heads, int num_kv_heads, int head_size,
                            int max_seq_len, int num_tokens = 0) {
  static const int num_cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    return n;
  }();
  ChunkPlan pl{};
  const int G = num_heads / num_kv_heads;
  pl.gp_shift = G == 1 ? 0 : G == 2 ? 1 : G <= 4 ? 2 : G <= 8 ? 3 : 4;
  const int TQ = kChunkNCG * (16 >> pl.gp_shift);
  pl.qgroups = (max_query_len + TQ - 1) / TQ;
  pl.parts = 1;
  pl.rows = (int64_t)num_seqs * max_query_len * num_heads;
  const int64_t wgs = (int64_t)num_kv_heads * (num_tokens > 0 ? (num_tokens + TQ - 1) / TQ + num_seqs / 2
                                                              : (int64_t)num_seqs * pl.qgroups);
  if (max_seq_len > 0 && num_cus > 0 && wgs * 2 <= num_cus + num_cus / 4) {
    int parts = (int)((num_cus + wgs - 1) / wgs);
    if (parts > max_seq_len / 128) parts = max_seq_len / 128;
    if (parts > kChunkMaxParts) parts = kChunkMaxParts;
    if (parts >= 2) {
      pl.parts = parts;
      pl.part_tokens = (((max_seq_len + parts - 1) / parts) + 15) & ~15;
      pl.ws_bytes = partition_scratch_bytes(pl.rows, parts, head_size);
    }
  }
  return pl;
}

// Whether the launch belongs here (the caller has checked: plain causal attention, 16-bit cache): launches that are
// mostly one-token sequences -- at most tuning().prefill_chunk_max_avg_x8 / 8 query tokens per sequence on average
// (2 as shipped), as the caller's num_tokens says -- with no chunk longer than tuning().prefill_chunk_max_query.
// Launches of short chunks ONLY stay with the 32x32 body (16 x (16 over 1 024): 22 us there, 27 here).
inline bool chunk_kernel_takes(const PrefillParams& p, int head_size, int num_seqs, int max_query_len) {
  const int G = p.num_heads / p.num_kv_heads;
  const Tuning& t = tuning();
  return t.prefill_chunk_max_query > 0 && max_query_len <= t.prefill_chunk_max_query && G <= 16 &&
         (head_size == 64 || head_size == 128) && p.num_tokens > 0 &&
         (int64_t)p.num_tokens * 8 <= (int64_t)num_seqs * t.prefill_chunk_max_avg_x8;
}

template <typename T, int D, int BS>
static int launch_prefill_chunk(const PrefillParams& p0, int num_seqs, int max_query_len, hipStream_t stream) {
  PrefillParams p = p0;
  ChunkPlan pl = chunk_plan(num_seqs, max_query_len, p.num_heads, p.num_kv_heads, D, p.max_seq_len, p.num_tokens);
  p.gp_shift = pl.gp_shift;
  ChunkScratch sc{};
  sc.num_parts = 1;
  if (pl.parts >= 2 && p.workspace != nullptr && p.workspace_bytes >= pl.ws_bytes) {
    sc = partition_scratch(p.workspace, pl.rows, pl.parts, pl.part_tokens, D);
  }
  const size_t smem = (size_t)kChunkWaves * 16 * 2 * sizeof(float) + (size_t)kChunkWaves * 16 * D * sizeof(float) +
                      (size_t)kChunkNCG * (D / 32) * 64 * 16;  // merge area | Q fragments
  auto kern = paged_prefill_chunk_kernel<T, D, BS, kChunkWaves, kChunkNCG, LVLLM_CHUNK_NBUF>;
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  // rows of workgroups: the query groups of the launch counted over its sequences (the caller said how many tokens
  // there are), or the (sequence, group) rectangle
  const int TQ = kChunkNCG * (16 >> pl.gp_shift);
  const bool flat = p.num_tokens > 0;
  const int rows = flat ? (p.num_tokens + TQ - 1) / TQ + num_seqs : num_seqs * pl.qgroups;
  hipLaunchKernelGGL(kern, dim3(p.num_kv_heads, rows, sc.num_parts), dim3(kChunkWaves * 64), smem, stream, p, sc,
                     flat ? -num_seqs : pl.qgroups);
  if (sc.tmp_out != nullptr)
    launch_partition_reduce<T, D>(p, sc, num_seqs, max_query_len, stream);
  return 0;
}

}  // namespace lvllm
