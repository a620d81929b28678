/*
 * paged_ops_oracle.c -- CPU restatement of the reference's paged-attention decode operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under light-vllm_amd/ may import, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the reported CPU baseline, never as the product path.
 *
 * Parity pin: this restatement is checked against the reference's own CPU backend
 * (csrc/cpu/ *.cpp built unmodified into oracle/_ref/ by oracle/build_oracle.py) on seeded
 * inputs, and against the golden vectors that build produced (tests/golden/ *.npz).  The
 * reference ships no kernel-level golden vectors of its own.
 *
 * Each function restates the arithmetic of the reference's GPU kernel -- the forward a
 * light-vllm user gets on a GPU -- including its rounding points; the reference's CPU
 * kernels (csrc/cpu) differ from its GPU kernels only in where they round to the element
 * type (noted per function), which is inside the tolerance the tests state.
 *
 * Element types: 0 = float32, 1 = float16, 2 = bfloat16 (enum lvllm_dtype of
 * include/lvllm_hip.h).  16-bit values travel as uint16_t bit patterns.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define PARTITION_SIZE 512 /* csrc/attention/attention_kernels.cu:850 */

/* ---- element conversions (round to nearest even, NaN preserved) ---------------- */
static inline float bf16_to_f(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* quiet NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float f16_to_f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, u;
  if (exp == 0) {
    if (man == 0) {
      u = sign;
    } else { /* subnormal */
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while (!(man & 0x400u));
      u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    u = sign | 0x7f800000u | (man << 13);
  } else {
    u = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f_to_f16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  uint32_t sign = (u >> 16) & 0x8000u;
  uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);          /* NaN */
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);         /* overflow -> inf (>= 65520) */
  if (a < 0x33000001u) return (uint16_t)sign;                      /* underflow -> 0 (<= 2^-25) */
  int e = (int)(a >> 23) - 127;
  uint32_t man = (a & 0x7fffffu) | 0x800000u;
  int shift = (e < -14) ? (13 + (-14 - e)) : 13; /* subnormal halves lose extra bits */
  uint32_t half_man = man >> shift;
  uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (half_man & 1u))) ++half_man;
  uint32_t h;
  if (e < -14)
    h = half_man; /* subnormal (may carry into the smallest normal, which is right) */
  else
    h = ((uint32_t)(e + 15) << 10) + (half_man - 0x400u); /* carry bumps the exponent */
  return (uint16_t)(sign | h);
}

static inline float ld(const void* p, int dt, int64_t i) {
  if (dt == 0) return ((const float*)p)[i];
  if (dt == 1) return f16_to_f(((const uint16_t*)p)[i]);
  return bf16_to_f(((const uint16_t*)p)[i]);
}
static inline void st(void* p, int dt, int64_t i, float f) {
  if (dt == 0)
    ((float*)p)[i] = f;
  else if (dt == 1)
    ((uint16_t*)p)[i] = f_to_f16(f);
  else
    ((uint16_t*)p)[i] = f_to_bf16(f);
}
/* round f to the element type and back (the "(scalar_t)(...)" casts of the kernels) */
static inline float rnd(float f, int dt) {
  if (dt == 0) return f;
  if (dt == 1) return f16_to_f(f_to_f16(f));
  return bf16_to_f(f_to_bf16(f));
}
static inline int esize(int dt) { return dt == 0 ? 4 : 2; }

/* ---- fp8 KV cache: OCP e4m3fn, the format of the reference's NVIDIA path (__NV_E4M3 with
 * __NV_SATFINITE, csrc/quantization/fp8/nvidia/quant_utils.cuh:458-489) and of the gfx950
 * conversion instructions.  (The reference's ROCm path, fp8/amd/hip_float8.h, is the MI300 fnuz
 * format; the cache content is private to the kernels, so the hardware format of the target wins.)
 *   1 sign, 4 exponent (bias 7), 3 mantissa bits; no infinities; 0x7f / 0xff = NaN; max 448. ---- */
static float e4m3_to_f(uint8_t v) {
  const int sign = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 15 && m == 7) r = NAN;
  else if (e == 0) r = ldexpf((float)m, -9);            /* subnormal: m/8 * 2^-6 */
  else r = ldexpf(1.f + (float)m / 8.f, e - 7);
  return sign ? -r : r;
}
static uint8_t f_to_e4m3(float f) { /* round to nearest even, saturate to +-448, NaN -> 0x7f */
  uint8_t sign = signbit(f) ? 0x80 : 0;
  if (isnan(f)) return 0x7f;
  float a = fabsf(f);
  if (a >= 448.f) return sign | 0x7e; /* 448 = 1.75 * 2^8 */
  if (a < ldexpf(1.f, -10)) return sign; /* below half the smallest subnormal (2^-9): rounds to 0; ties at 2^-10 -> even (0) */
  int e;
  float fr = frexpf(a, &e); /* a = fr * 2^e, fr in [0.5, 1) */
  int exp = e - 1;          /* a = (2 fr) * 2^exp, 2 fr in [1, 2) */
  if (exp < -6) {           /* subnormal: quantum 2^-9 */
    float q = a * 512.f;    /* in units of 2^-9 */
    float r = nearbyintf(q);
    if (r >= 8.f) return sign | 0x08; /* rounds up to the smallest normal 2^-6 */
    return sign | (uint8_t)r;
  }
  float q = ldexpf(a, 3 - exp); /* mantissa with 3 fraction bits: in [8, 16) */
  float r = nearbyintf(q);
  if (r >= 16.f) { r = 8.f; exp += 1; }
  if (exp > 8 || (exp == 8 && r > 14.f)) return sign | 0x7e;
  return sign | (uint8_t)(((exp + 7) << 3) | ((int)r - 8));
}
uint8_t oracle_f32_to_e4m3(float f) { return f_to_e4m3(f); }
float oracle_e4m3_to_f32(uint8_t v) { return e4m3_to_f(v); }

/* kv_cache_dtype of the attention functions below: 0 = "auto" (cache holds T), 1 = "fp8"/"fp8_e4m3".
 * A dequantised element is T(float(fp8) * scale) (nvidia/quant_utils.cuh:295-300), then fp32 math. */
static int g_kv_fp8 = 0;
static float g_k_scale = 1.f, g_v_scale = 1.f;
void oracle_set_kv_cache_fp8(int on, float k_scale, float v_scale) {
  g_kv_fp8 = on;
  g_k_scale = k_scale;
  g_v_scale = v_scale;
}
/* block-sparse attention (attention_kernels.cu:209-247, 385-393): a cache block is attended by a
 * head iff (k + offset) % vert_stride == 0 or k > q - local_blocks; skipped tokens get the logit
 * -FLT_MAX and add nothing.  Off when vert_stride <= 1. */
static int g_bs_vert = 0, g_bs_local = 0, g_bs_block = 64, g_bs_step = 0, g_tp_rank = 0;
void oracle_set_blocksparse(int vert_stride, int local_blocks, int block_size, int head_sliding_step,
                            int tp_rank) {
  g_bs_vert = vert_stride;
  g_bs_local = local_blocks;
  g_bs_block = block_size;
  g_bs_step = head_sliding_step;
  g_tp_rank = tp_rank;
}
static int bs_attended(int token, int seq_len, int head, int kv_head, int num_heads, int num_kv_heads,
                       int cache_block_size) {
  if (g_bs_vert <= 1) return 1;
  const int q_bs_block_id = (seq_len - 1) / g_bs_block;
  const int bs_block_offset = g_bs_step >= 0 ? (g_tp_rank * num_heads + head) * g_bs_step + 1
                                             : (g_tp_rank * num_kv_heads + kv_head) * (-g_bs_step) + 1;
  const int k_bs_block_id = (token / cache_block_size) * cache_block_size / g_bs_block;
  const int is_remote = ((k_bs_block_id + bs_block_offset) % g_bs_vert == 0);
  const int is_local = (k_bs_block_id > q_bs_block_id - g_bs_local);
  return is_remote || is_local;
}

static inline float ldkv(const void* cache, int dt, int64_t i, float scale) {
  if (!g_kv_fp8) return ld(cache, dt, i);
  return rnd(e4m3_to_f(((const uint8_t*)cache)[i]) * scale, dt);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* exposed for the tests of the conversions themselves */
uint16_t oracle_f32_to_f16(float f) { return f_to_f16(f); }
float oracle_f16_to_f32(uint16_t h) { return f16_to_f(h); }
uint16_t oracle_f32_to_bf16(float f) { return f_to_bf16(f); }

/* ---------------------------------------------------------------------------------
 * Attention over one token range [t0, t1) of one (sequence, head): the body of
 * paged_attention_kernel, csrc/attention/attention_kernels.cu:111-496.
 *   logits   :201-308  qk = scale * dot(q, k) (+ alibi * (token - seq_len + 1)), fp32
 *   softmax  :310-346  max over valid tokens, __expf, sum, inv = 1 / (sum + 1e-6)
 *   P.V      :359-434  probabilities rounded to T (from_float, :398-400), fp32 accumulate
 *   output   :483-495  rounded to T
 * The reference's CPU kernel (csrc/cpu/attention.cpp:209-214,48-71) keeps fp32
 * probabilities and divides by sum (no 1e-6): a difference below 2^-8 relative.
 * ------------------------------------------------------------------------------- */
/* the (head, head counts) of the attend_range call in progress on this thread: block-sparse masks
 * depend on the head */
static __thread int t_head = 0, t_num_heads = 1, t_num_kv_heads = 1;

static void attend_range(const void* q, int64_t q_off, const void* k_cache, const void* v_cache,
                         const int32_t* block_table, int kv_head, int64_t kv_block_stride,
                         int64_t kv_head_stride, int head_size, int block_size, int x, float scale,
                         float alibi_slope, int seq_len, int t0, int t1, int dt, float* logits,
                         float* out_f32, float* out_max, float* out_sum) {
  const int n = t1 - t0;
  float qk_max = -FLT_MAX;
  if (g_kv_fp8) x = 16; /* 16 / sizeof(cache element), cache_kernels.cu:184-188 */
  for (int i = 0; i < n; ++i) {
    const int tok = t0 + i;
    const int64_t bn = block_table[tok / block_size];
    const int off = tok % block_size;
    const int64_t kb = bn * kv_block_stride + (int64_t)kv_head * kv_head_stride;
    float dot = 0.f;
    for (int d = 0; d < head_size; ++d) {
      /* key_cache[block][head][d / x][off][d % x], cache_kernels.cu:184-188 */
      const int64_t ki = kb + (int64_t)(d / x) * block_size * x + (int64_t)off * x + (d % x);
      dot += ld(q, dt, q_off + d) * ldkv(k_cache, dt, ki, g_k_scale);
    }
    float qk = scale * dot;
    qk += (alibi_slope != 0.f) ? alibi_slope * (float)(tok - seq_len + 1) : 0.f;
    if (!bs_attended(tok, seq_len, t_head, kv_head, t_num_heads, t_num_kv_heads, block_size))
      qk = -FLT_MAX; /* :241-247 */
    logits[i] = qk;
    qk_max = fmaxf(qk_max, qk);
  }
  float exp_sum = 0.f;
  for (int i = 0; i < n; ++i) {
    logits[i] = logits[i] == -FLT_MAX ? 0.f : expf(logits[i] - qk_max);
    exp_sum += logits[i];
  }
  const float inv_sum = 1.f / (exp_sum + 1e-6f);
  for (int i = 0; i < n; ++i) logits[i] = rnd(logits[i] * inv_sum, dt);
  for (int d = 0; d < head_size; ++d) out_f32[d] = 0.f;
  for (int i = 0; i < n; ++i) {
    const int tok = t0 + i;
    const int64_t bn = block_table[tok / block_size];
    const int off = tok % block_size;
    /* value_cache[block][head][d][off], cache_kernels.cu:189-192 */
    const int64_t vb = bn * kv_block_stride + (int64_t)kv_head * kv_head_stride + off;
    const float p = logits[i];
    for (int d = 0; d < head_size; ++d)
      out_f32[d] += p * ldkv(v_cache, dt, vb + (int64_t)d * block_size, g_v_scale);
  }
  *out_max = qk_max;
  *out_sum = exp_sum;
}

/* paged_attention_v1: csrc/attention/attention_kernels.cu:498-527, 690-829 */
void oracle_paged_attention_v1(void* out, const void* query, const void* key_cache,
                               const void* value_cache, int num_seqs, int num_heads, int head_size,
                               int num_kv_heads, float scale, const int32_t* block_tables,
                               const int32_t* seq_lens, int block_size,
                               int max_num_blocks_per_seq, const float* alibi_slopes,
                               int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
                               int dt) {
  const int x = 16 / esize(dt);
  const int G = num_heads / num_kv_heads;
  int max_len = 1;
  for (int s = 0; s < num_seqs; ++s)
    if (seq_lens[s] > max_len) max_len = seq_lens[s];
#pragma omp parallel
  {
    float* logits = (float*)malloc(sizeof(float) * (size_t)max_len);
    float* acc = (float*)malloc(sizeof(float) * (size_t)head_size);
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (int s = 0; s < num_seqs; ++s)
      for (int h = 0; h < num_heads; ++h) {
        float mx, sm;
        t_head = h; t_num_heads = num_heads; t_num_kv_heads = num_kv_heads;
        attend_range(query, (int64_t)s * q_stride + (int64_t)h * head_size, key_cache, value_cache,
                     block_tables + (int64_t)s * max_num_blocks_per_seq, h / G, kv_block_stride,
                     kv_head_stride, head_size, block_size, x, scale,
                     alibi_slopes ? alibi_slopes[h] : 0.f, seq_lens[s], 0, seq_lens[s], dt, logits,
                     acc, &mx, &sm);
        for (int d = 0; d < head_size; ++d)
          st(out, dt, ((int64_t)s * num_heads + h) * head_size + d, acc[d]);
      }
    free(logits);
    free(acc);
  }
}

/* ---------------------------------------------------------------------------------
 * Causal varlen attention of prompt chunks over the paged cache (prefill / chunked prefill /
 * prefix hits).  The reference delegates this to the third-party vllm-flash-attn 2.6.1
 * (requirements.txt:32, not under /root/reference) through
 *   flash_attn_varlen_func(q, key_cache, value_cache, cu_seqlens_q, cu_seqlens_k, causal=True,
 *                          window_size, alibi_slopes, block_table, softcap)
 * light_vllm/decoding/backends/attention/backends/flash_attn.py:538-555.  Restated from the
 * published FlashAttention-2 algorithm with the in-tree definition of the same function as
 * anchor (scaled_dot_product_attention, prefill_only/backends/attention/backends/torch_naive.py:125-149):
 *   logits = scale * q.k (fp32)  [softcap: cap * tanh(logits / cap)]  + alibi * (key - query position)
 *   causal mask aligned to the bottom-right corner: query t of a chunk of L tokens in a context of S
 *   sits at position S - L + t and sees keys 0 .. position (the last `window` of them if window > 0)
 *   P = exp(logits - max) rounded to T for the P.V product, fp32 accumulation, divided by the
 *   fp32 sum of the unrounded exponentials, result rounded to T.
 * ------------------------------------------------------------------------------- */
void oracle_paged_prefill_attention(void* out, const void* query, const void* key_cache,
                                    const void* value_cache, int num_seqs, int num_heads,
                                    int head_size, int num_kv_heads, float scale,
                                    const int32_t* block_tables, const int32_t* seq_lens,
                                    const int32_t* query_start_loc, int block_size,
                                    int max_num_blocks_per_seq, const float* alibi_slopes,
                                    int sliding_window, float softcap, int64_t q_stride,
                                    int64_t out_stride, int64_t kv_block_stride,
                                    int64_t kv_head_stride, int dt) {
  const int x = g_kv_fp8 ? 16 : 16 / esize(dt);
  const int G = num_heads / num_kv_heads;
  int max_len = 1;
  for (int s = 0; s < num_seqs; ++s)
    if (seq_lens[s] > max_len) max_len = seq_lens[s];
  const int num_tokens = query_start_loc[num_seqs];
  /* token -> sequence */
  int* seq_of = (int*)malloc(sizeof(int) * (size_t)(num_tokens > 0 ? num_tokens : 1));
  for (int s = 0; s < num_seqs; ++s)
    for (int t = query_start_loc[s]; t < query_start_loc[s + 1]; ++t) seq_of[t] = s;
#pragma omp parallel
  {
    float* logits = (float*)malloc(sizeof(float) * (size_t)max_len);
    float* acc = (float*)malloc(sizeof(float) * (size_t)head_size);
#pragma omp for collapse(2) schedule(dynamic, 4)
    for (int tok = 0; tok < num_tokens; ++tok)
      for (int h = 0; h < num_heads; ++h) {
        const int s = seq_of[tok];
        const int qlen = query_start_loc[s + 1] - query_start_loc[s];
        const int pos = seq_lens[s] - qlen + (tok - query_start_loc[s]);
        const int32_t* bt = block_tables + (int64_t)s * max_num_blocks_per_seq;
        const int kvh = h / G;
        const int64_t q_off = (int64_t)tok * q_stride + (int64_t)h * head_size;
        int k0 = 0;
        if (sliding_window > 0 && pos - sliding_window + 1 > 0) k0 = pos - sliding_window + 1;
        const int n = pos + 1 - k0;
        float mx = -FLT_MAX;
        for (int i = 0; i < n; ++i) {
          const int key = k0 + i;
          const int64_t kb = (int64_t)bt[key / block_size] * kv_block_stride + (int64_t)kvh * kv_head_stride;
          const int off = key % block_size;
          float dot = 0.f;
          for (int d = 0; d < head_size; ++d)
            dot += ld(query, dt, q_off + d) *
                   ldkv(key_cache, dt, kb + (int64_t)(d / x) * block_size * x + (int64_t)off * x + (d % x), g_k_scale);
          float qk = scale * dot;
          if (softcap > 0.f) qk = softcap * tanhf(qk / softcap);
          if (alibi_slopes) qk += alibi_slopes[h] * (float)(key - pos);
          logits[i] = qk;
          mx = fmaxf(mx, qk);
        }
        float sum = 0.f;
        for (int i = 0; i < n; ++i) {
          const float e = expf(logits[i] - mx);
          sum += e;
          logits[i] = rnd(e, dt);
        }
        for (int d = 0; d < head_size; ++d) acc[d] = 0.f;
        for (int i = 0; i < n; ++i) {
          const int key = k0 + i;
          const int64_t vb = (int64_t)bt[key / block_size] * kv_block_stride + (int64_t)kvh * kv_head_stride +
                             key % block_size;
          const float pr = logits[i];
          for (int d = 0; d < head_size; ++d)
            acc[d] += pr * ldkv(value_cache, dt, vb + (int64_t)d * block_size, g_v_scale);
        }
        const float inv = sum > 0.f ? 1.f / sum : 0.f;
        for (int d = 0; d < head_size; ++d)
          st(out, dt, (int64_t)tok * out_stride + (int64_t)h * head_size + d, acc[d] * inv);
      }
    free(logits);
    free(acc);
  }
  free(seq_of);
}

/* ---------------------------------------------------------------------------------
 * Dense varlen attention without a KV cache: the prefill-only / encode-only backends
 * (light_vllm/prefill_only/backends/attention/backends/torch_naive.py:65-149 is the in-tree
 * definition; flash_attn.py there calls the third-party flash_attn_varlen_func).
 *   query [T, H, D], key/value [T, KVH, D], sequences cut by cu_seqlens;
 *   key/value heads repeated H/KVH times (:98-100); causal (DECODER) masks keys after the query
 *   (:135-140, both inside the same sequence), ENCODER sees the whole sequence;
 *   attn = softmax(q.k * scale + bias) @ v (:147-149).
 * Rounding as the GPU kernel: fp32 logits, P = exp(x - max) rounded to T for P.V, fp32
 * accumulation, divided by the fp32 sum, rounded to T.
 * ------------------------------------------------------------------------------- */
void oracle_varlen_attention(void* out, const void* query, const void* key, const void* value,
                             const int32_t* cu_seqlens, int num_seqs, int num_heads,
                             int num_kv_heads, int head_size, float scale, int causal,
                             int64_t q_stride, int64_t k_stride, int64_t v_stride,
                             int64_t out_stride, int dt) {
  const int G = num_heads / num_kv_heads;
  int max_len = 1;
  for (int s = 0; s < num_seqs; ++s)
    if (cu_seqlens[s + 1] - cu_seqlens[s] > max_len) max_len = cu_seqlens[s + 1] - cu_seqlens[s];
#pragma omp parallel
  {
    float* logits = (float*)malloc(sizeof(float) * (size_t)max_len);
    float* acc = (float*)malloc(sizeof(float) * (size_t)head_size);
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (int s = 0; s < num_seqs; ++s)
      for (int h = 0; h < num_heads; ++h) {
        const int beg = cu_seqlens[s], len = cu_seqlens[s + 1] - cu_seqlens[s];
        const int kvh = h / G;
        for (int i = 0; i < len; ++i) {
          const int n = causal ? i + 1 : len;
          const int64_t q_off = (int64_t)(beg + i) * q_stride + (int64_t)h * head_size;
          float mx = -FLT_MAX;
          for (int j = 0; j < n; ++j) {
            const int64_t k_off = (int64_t)(beg + j) * k_stride + (int64_t)kvh * head_size;
            float dot = 0.f;
            for (int d = 0; d < head_size; ++d) dot += ld(query, dt, q_off + d) * ld(key, dt, k_off + d);
            logits[j] = scale * dot;
            mx = fmaxf(mx, logits[j]);
          }
          float sum = 0.f;
          for (int j = 0; j < n; ++j) {
            const float e = expf(logits[j] - mx);
            sum += e;
            logits[j] = rnd(e, dt);
          }
          for (int d = 0; d < head_size; ++d) acc[d] = 0.f;
          for (int j = 0; j < n; ++j) {
            const int64_t v_off = (int64_t)(beg + j) * v_stride + (int64_t)kvh * head_size;
            for (int d = 0; d < head_size; ++d) acc[d] += logits[j] * ld(value, dt, v_off + d);
          }
          const float inv = sum > 0.f ? 1.f / sum : 0.f;
          for (int d = 0; d < head_size; ++d)
            st(out, dt, (int64_t)(beg + i) * out_stride + (int64_t)h * head_size + d, acc[d] * inv);
        }
      }
    free(logits);
    free(acc);
  }
}

/* paged_attention_v2 + reduce: csrc/attention/attention_kernels.cu:529-669, 848-997.
 * tmp_out/exp_sums/max_logits are filled exactly where the GPU kernel fills them
 * (every partition that holds tokens, also when there is only one). */
void oracle_paged_attention_v2(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                               const void* query, const void* key_cache, const void* value_cache,
                               int num_seqs, int num_heads, int head_size, int num_kv_heads,
                               float scale, const int32_t* block_tables, const int32_t* seq_lens,
                               int block_size, int max_num_blocks_per_seq, int max_num_partitions,
                               const float* alibi_slopes, int64_t q_stride,
                               int64_t kv_block_stride, int64_t kv_head_stride, int dt) {
  const int x = 16 / esize(dt);
  const int G = num_heads / num_kv_heads;
#pragma omp parallel
  {
    float logits[PARTITION_SIZE];
    float* acc = (float*)malloc(sizeof(float) * (size_t)head_size);
#pragma omp for collapse(3) schedule(dynamic, 1)
    for (int s = 0; s < num_seqs; ++s)
      for (int h = 0; h < num_heads; ++h)
        for (int pi = 0; pi < max_num_partitions; ++pi) {
          const int seq_len = seq_lens[s];
          const int t0 = pi * PARTITION_SIZE;
          if (t0 >= seq_len) continue; /* :116-119 */
          const int t1 = t0 + PARTITION_SIZE < seq_len ? t0 + PARTITION_SIZE : seq_len;
          float mx, sm;
          t_head = h; t_num_heads = num_heads; t_num_kv_heads = num_kv_heads;
          attend_range(query, (int64_t)s * q_stride + (int64_t)h * head_size, key_cache,
                       value_cache, block_tables + (int64_t)s * max_num_blocks_per_seq, h / G,
                       kv_block_stride, kv_head_stride, head_size, block_size, x, scale,
                       alibi_slopes ? alibi_slopes[h] : 0.f, seq_len, t0, t1, dt, logits, acc, &mx,
                       &sm);
          const int64_t row = ((int64_t)s * num_heads + h) * max_num_partitions + pi;
          max_logits[row] = mx;
          exp_sums[row] = sm;
          for (int d = 0; d < head_size; ++d) st(tmp_out, dt, row * head_size + d, acc[d]);
        }
    free(acc);
  }
  /* paged_attention_v2_reduce_kernel :577-668 */
#pragma omp parallel for collapse(2)
  for (int s = 0; s < num_seqs; ++s)
    for (int h = 0; h < num_heads; ++h) {
      const int seq_len = seq_lens[s];
      const int np = (seq_len + PARTITION_SIZE - 1) / PARTITION_SIZE;
      const int64_t row = ((int64_t)s * num_heads + h);
      const int64_t prow = row * max_num_partitions;
      if (np == 1) {
        for (int d = 0; d < head_size; ++d)
          st(out, dt, row * head_size + d, ld(tmp_out, dt, prow * head_size + d));
        continue;
      }
      float mx = -FLT_MAX;
      for (int j = 0; j < np; ++j) mx = fmaxf(mx, max_logits[prow + j]);
      float gsum = 0.f;
      for (int j = 0; j < np; ++j) gsum += exp_sums[prow + j] * expf(max_logits[prow + j] - mx);
      const float inv = 1.f / (gsum + 1e-6f);
      for (int d = 0; d < head_size; ++d) {
        float acc = 0.f;
        for (int j = 0; j < np; ++j)
          acc += ld(tmp_out, dt, (prow + j) * head_size + d) *
                 (exp_sums[prow + j] * expf(max_logits[prow + j] - mx)) * inv;
        st(out, dt, row * head_size + d, acc);
      }
    }
}

/* reshape_and_cache: csrc/cache_kernels.cu:164-203 (byte movement, bit-exact) */
void oracle_reshape_and_cache(const void* key, const void* value, void* key_cache,
                              void* value_cache, const int64_t* slot_mapping, int num_tokens,
                              int num_heads, int head_size, int block_size, int x,
                              int64_t key_stride, int64_t value_stride, int dt) {
  const int es = esize(dt);
  for (int64_t t = 0; t < num_tokens; ++t) {
    const int64_t slot = slot_mapping[t];
    if (slot < 0) continue;
    const int64_t block_idx = slot / block_size, block_off = slot % block_size;
    for (int i = 0; i < num_heads * head_size; ++i) {
      const int head = i / head_size, ho = i % head_size;
      const int x_idx = ho / x, x_off = ho % x;
      const int64_t kdst = block_idx * num_heads * (head_size / x) * block_size * x +
                           (int64_t)head * (head_size / x) * block_size * x +
                           (int64_t)x_idx * block_size * x + block_off * x + x_off;
      const int64_t vdst = block_idx * num_heads * head_size * block_size +
                           (int64_t)head * head_size * block_size + (int64_t)ho * block_size +
                           block_off;
      memcpy((char*)key_cache + kdst * es, (const char*)key + (t * key_stride + i) * es, es);
      memcpy((char*)value_cache + vdst * es, (const char*)value + (t * value_stride + i) * es, es);
    }
  }
}

/* reshape_and_cache with kv_cache_dtype "fp8": csrc/cache_kernels.cu:194-202 -- every element
 * becomes fp8(float(x) / scale) (nvidia/quant_utils.cuh:458-489); layouts as above with x = 16. */
void oracle_reshape_and_cache_fp8(const void* key, const void* value, uint8_t* key_cache,
                                  uint8_t* value_cache, const int64_t* slot_mapping, int num_tokens,
                                  int num_heads, int head_size, int block_size, int64_t key_stride,
                                  int64_t value_stride, int dt, float k_scale, float v_scale) {
  const int x = 16;
  for (int64_t t = 0; t < num_tokens; ++t) {
    const int64_t slot = slot_mapping[t];
    if (slot < 0) continue;
    const int64_t block_idx = slot / block_size, block_off = slot % block_size;
    for (int i = 0; i < num_heads * head_size; ++i) {
      const int head = i / head_size, ho = i % head_size;
      const int x_idx = ho / x, x_off = ho % x;
      const int64_t kdst = block_idx * num_heads * (head_size / x) * block_size * x +
                           (int64_t)head * (head_size / x) * block_size * x +
                           (int64_t)x_idx * block_size * x + block_off * x + x_off;
      const int64_t vdst = block_idx * num_heads * head_size * block_size +
                           (int64_t)head * head_size * block_size + (int64_t)ho * block_size +
                           block_off;
      key_cache[kdst] = f_to_e4m3(ld(key, dt, t * key_stride + i) / k_scale);
      value_cache[vdst] = f_to_e4m3(ld(value, dt, t * value_stride + i) / v_scale);
    }
  }
}

/* advance_step: csrc/prepare_inputs/advance_step.cu:14-57 (rows 0 .. num_queries-1) */
void oracle_advance_step(int num_queries, int block_size, int64_t* input_tokens,
                         const int64_t* sampled_token_ids, int64_t* input_positions,
                         int32_t* seq_lens, int64_t* slot_mapping, const int32_t* block_tables,
                         int64_t block_tables_stride) {
  for (int i = 0; i < num_queries; ++i) {
    input_tokens[i] = sampled_token_ids[i];
    const int next_seq_len = seq_lens[i] + 1;
    const int next_input_pos = next_seq_len - 1;
    seq_lens[i] = next_seq_len;
    input_positions[i] = next_input_pos;
    const int32_t* row = block_tables + block_tables_stride * i;
    slot_mapping[i] = (int64_t)row[next_input_pos / block_size] * block_size + next_input_pos % block_size;
  }
}

/* fp8 activation quantisation: csrc/quantization/fp8/common.cu:24-38 (conversion: multiply by the
 * inverted scale or divide, clamp with fmax(-448, fmin(x, 448)) -- so NaN -> +448 -- then round to
 * nearest even), :46-83 (absmax / 448), :164-224 (per token). */
static uint8_t fp8_conv(float v, float scale, int inverted) {
  const float x = inverted ? v * scale : v / scale;
  const float r = fmaxf(-448.f, fminf(x, 448.f));
  return f_to_e4m3(r);
}
void oracle_static_scaled_fp8_quant(uint8_t* out, const void* input, const float* scale, int64_t n, int dt) {
  const float inv = 1.0f / scale[0];
  for (int64_t i = 0; i < n; ++i) out[i] = fp8_conv(ld(input, dt, i), inv, 1);
}
void oracle_dynamic_scaled_fp8_quant(uint8_t* out, const void* input, float* scale, int64_t n, int dt) {
  float m = 0.f;
  for (int64_t i = 0; i < n; ++i) m = fmaxf(m, fabsf(ld(input, dt, i)));
  if (m / 448.f > scale[0]) scale[0] = m / 448.f; /* atomic max into a value that starts <= 0 */
  oracle_static_scaled_fp8_quant(out, input, scale, n, dt);
}
void oracle_dynamic_per_token_scaled_fp8_quant(uint8_t* out, float* scales, const void* input,
                                               const float* scale_ub, int num_tokens, int hidden, int dt) {
  const float min_scaling_factor = 1.0f / (448.f * 512.f);
  for (int t = 0; t < num_tokens; ++t) {
    float m = 0.f;
    for (int i = 0; i < hidden; ++i) m = fmaxf(m, fabsf(ld(input, dt, (int64_t)t * hidden + i)));
    if (scale_ub) m = fminf(m, scale_ub[0]);
    const float s = fmaxf(m / 448.f, min_scaling_factor);
    scales[t] = s;
    for (int i = 0; i < hidden; ++i)
      out[(int64_t)t * hidden + i] = fp8_conv(ld(input, dt, (int64_t)t * hidden + i), s, 0);
  }
}

/* reshape_and_cache_flash: csrc/cache_kernels.cu:218-246 */
void oracle_reshape_and_cache_flash(const void* key, const void* value, void* key_cache,
                                    void* value_cache, const int64_t* slot_mapping,
                                    int num_tokens, int num_heads, int head_size, int block_size,
                                    int64_t block_stride, int64_t key_stride, int64_t value_stride,
                                    int dt) {
  const int es = esize(dt);
  const int n = num_heads * head_size;
  for (int64_t t = 0; t < num_tokens; ++t) {
    const int64_t slot = slot_mapping[t];
    if (slot < 0) continue;
    const int64_t block_idx = slot / block_size, block_off = slot % block_size;
    const int64_t dst = block_idx * block_stride + block_off * n;
    memcpy((char*)key_cache + dst * es, (const char*)key + t * key_stride * es, (size_t)n * es);
    memcpy((char*)value_cache + dst * es, (const char*)value + t * value_stride * es, (size_t)n * es);
  }
}

/* copy_blocks for one layer: csrc/cache_kernels.cu:67-98 (pairs applied in order) */
void oracle_copy_blocks(void* key_cache, void* value_cache, const int64_t* block_mapping,
                        int num_pairs, int64_t block_bytes) {
  for (int p = 0; p < num_pairs; ++p) {
    const int64_t src = block_mapping[2 * p], dst = block_mapping[2 * p + 1];
    memmove((char*)key_cache + dst * block_bytes, (char*)key_cache + src * block_bytes, (size_t)block_bytes);
    memmove((char*)value_cache + dst * block_bytes, (char*)value_cache + src * block_bytes, (size_t)block_bytes);
  }
}

/* swap_blocks: csrc/cache_kernels.cu:24-63 (one block-sized copy per pair) */
void oracle_swap_blocks(const void* src, void* dst, const int64_t* block_mapping, int num_pairs,
                        int64_t block_bytes) {
  for (int p = 0; p < num_pairs; ++p)
    memcpy((char*)dst + block_mapping[2 * p + 1] * block_bytes,
           (const char*)src + block_mapping[2 * p] * block_bytes, (size_t)block_bytes);
}

/* rms_norm: csrc/layernorm_kernels.cu:21-45.  out = ((T)(x * s)) * w, a T x T multiply.
 * (csrc/cpu/layernorm.cpp multiplies in fp32 and rounds once.) */
void oracle_rms_norm(void* out, const void* input, const void* weight, float eps, int num_tokens,
                     int hidden, int dt) {
#pragma omp parallel for
  for (int t = 0; t < num_tokens; ++t) {
    const int64_t row = (int64_t)t * hidden;
    float var = 0.f;
    for (int i = 0; i < hidden; ++i) {
      const float x = ld(input, dt, row + i);
      var += x * x;
    }
    const float s = 1.0f / sqrtf(var / hidden + eps);
    for (int i = 0; i < hidden; ++i) {
      const float tn = rnd(ld(input, dt, row + i) * s, dt);
      st(out, dt, row + i, tn * ld(weight, dt, i));
    }
  }
}

/* fused_add_rms_norm: csrc/layernorm_kernels.cu:254-287 (generic form; the packed
 * fp16 form :200-248 has the same rounding points). */
void oracle_fused_add_rms_norm(void* input, void* residual, const void* weight, float eps,
                               int num_tokens, int hidden, int dt) {
#pragma omp parallel for
  for (int t = 0; t < num_tokens; ++t) {
    const int64_t row = (int64_t)t * hidden;
    float var = 0.f;
    for (int i = 0; i < hidden; ++i) {
      const float z = rnd(ld(input, dt, row + i) + ld(residual, dt, row + i), dt);
      st(residual, dt, row + i, z);
      var += z * z;
    }
    const float s = 1.0f / sqrtf(var / hidden + eps);
    for (int i = 0; i < hidden; ++i) {
      const float tn = rnd(ld(residual, dt, row + i) * s, dt);
      st(input, dt, row + i, tn * ld(weight, dt, i));
    }
  }
}

/* rotary_embedding: csrc/pos_encoding_kernels.cu:10-92.  `x * cos - y * sin` on scalar_t
 * values: every product and the sum/difference are rounded to T.
 * (csrc/cpu/pos_encoding.cpp does the arithmetic in fp32 and rounds once.) */
static void rope_pair(void* arr, int64_t xi, int64_t yi, float c, float s, int dt) {
  const float x = ld(arr, dt, xi), y = ld(arr, dt, yi);
  const float xc = rnd(x * c, dt), ys = rnd(y * s, dt);
  const float yc = rnd(y * c, dt), xs = rnd(x * s, dt);
  st(arr, dt, xi, xc - ys);
  st(arr, dt, yi, yc + xs);
}
void oracle_rotary_embedding(const int64_t* positions, void* query, void* key, int num_tokens,
                             int num_heads, int num_kv_heads, int head_size, int rot_dim,
                             int64_t query_stride, int64_t key_stride, const void* cos_sin_cache,
                             int is_neox, int dt) {
  const int embed = rot_dim / 2;
#pragma omp parallel for
  for (int t = 0; t < num_tokens; ++t) {
    const int64_t cache = positions[t] * rot_dim;
    for (int part = 0; part < 2; ++part) {
      void* arr = part == 0 ? query : key;
      const int nh = part == 0 ? num_heads : num_kv_heads;
      const int64_t stride = part == 0 ? query_stride : key_stride;
      for (int h = 0; h < nh; ++h)
        for (int r = 0; r < embed; ++r) {
          const int64_t base = (int64_t)t * stride + (int64_t)h * head_size;
          const float c = ld(cos_sin_cache, dt, cache + r);
          const float s = ld(cos_sin_cache, dt, cache + embed + r);
          if (is_neox)
            rope_pair(arr, base + r, base + embed + r, c, s, dt);
          else
            rope_pair(arr, base + 2 * r, base + 2 * r + 1, c, s, dt);
        }
    }
  }
}

/* silu_and_mul: csrc/activation_kernels.cu:9-30.  out = ((T)(x / (1 + expf(-x)))) * y */
void oracle_silu_and_mul(void* out, const void* input, int64_t num_tokens, int d, int dt) {
#pragma omp parallel for
  for (int64_t t = 0; t < num_tokens; ++t)
    for (int i = 0; i < d; ++i) {
      const float x = ld(input, dt, t * 2 * d + i), y = ld(input, dt, t * 2 * d + d + i);
      const float a = rnd(x / (1.0f + expf(-x)), dt);
      st(out, dt, t * d + i, a * y);
    }
}
