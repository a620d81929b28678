/*
 * lvllm_hip.h — C-ABI of the MI355X (gfx950) paged-attention decode path.
 *
 * This is the drop-in boundary: every entry point replaces one operator the
 * reference registers in csrc/torch_bindings.cpp (namespaces `_C`,
 * `_C_cache_ops`, `_C_cuda_utils`) and declares in csrc/ops.h / csrc/cache.h.
 * Signatures are plain pointers + sizes + a HIP stream; no torch types.  The
 * torch-side binding (light-vllm_amd/csrc/torch_bindings.cpp) unpacks tensors
 * and forwards here, so `torch.ops._C.*` keeps the reference schemas verbatim.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless a comment says "host";
 *   - strides are in ELEMENTS, as the reference passes them to its kernels;
 *   - `stream` is a hipStream_t (void* here so that C callers need no HIP
 *     headers); kernels are launched asynchronously on it and never
 *     synchronise (graph-capture safe);
 *   - kv_cache_bytes (attention, reshape_and_cache, the fused rope + cache write, prefill): the bytes the
 *     caller owns behind key_cache (== behind value_cache).  The library cannot see allocations: with
 *     kv_cache_bytes = 0 EXTENTS ARE UNCHECKED -- a block table entry, a slot or an element size that does
 *     not belong to the caches makes the kernel read or write outside them (a GPU memory fault, not an
 *     error code).  With the extent stated, block numbers are clamped to the blocks that fit
 *     (kv_cache_bytes / (kv_block_stride * element bytes)) and slots beyond the last one are skipped like
 *     padding slots: a mismatch then gives wrong numbers, never a fault.  The torch bindings always state it.
 *   - return value: 0 on success, non-zero on a rejected argument or HIP
 *     error; lvllm_last_error() returns the message (thread-local).  The torch
 *     binding turns non-zero into RuntimeError, which is what TORCH_CHECK does
 *     in the reference (csrc/attention/attention_kernels.cu:767,804).
 */
#ifndef LVLLM_HIP_H_
#define LVLLM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of activations (scalar_t in the reference) */
enum lvllm_dtype {
  LVLLM_F32 = 0,
  LVLLM_F16 = 1,
  LVLLM_BF16 = 2,
};

/* kv_cache_dtype strings of the reference ("auto" | "fp8" | "fp8_e4m3"),
 * csrc/quantization/fp8/amd/quant_utils.cuh:547-573.  FP8_E4M3 = OCP e4m3fn bytes (the format of
 * the reference's NVIDIA path and of gfx950's conversion instructions), caches with x = 16, strides
 * in bytes; accepted by reshape_and_cache and paged_attention_v1/v2 (16-bit queries, block size
 * 16 | 32, head size % 16 == 0), a stored element is fp8(float(x) / scale), a read one
 * T(float(fp8) * scale) (fp8/nvidia/quant_utils.cuh:295-300,458-489). */
enum lvllm_kv_dtype {
  LVLLM_KV_AUTO = 0,
  LVLLM_KV_FP8_E4M3 = 1,
};

const char* lvllm_last_error(void);
/* build identification: "lvllm_hip gfx950 <abi version>" */
const char* lvllm_version(void);
/* Launch-shape knobs for the host's concurrency level (process-wide): "gemm_workgroups" (default
 * 256; 128 when two steps run on two streams), "gemm_balance" (1 | 0: a decode GEMM takes the fewest workgroups
 * <= gemm_workgroups that need no more rounds of n-tiles than gemm_workgroups would), "attn_waves" (8 | 4), "attn_splits" (paged_attention_v2:
 * 0 = shares chosen per call, n >= 1 = n shares, -1 = the reference's 512-token partitions),
 * "cache_tile_min_tokens" (reshape_and_cache: token count from which the LDS-tiled kernel is used),
 * "prefill_mfma32_min_query" (paged_prefill_attention: plain launches -- head size 64 or 128, 16-bit cache -- whose
 * longest chunk has at least this many query tokens take the 32x32-MFMA body, and so do launches with chunks of 16+
 * tokens whose grid fits the CUs at once; 0 = never), "varlen_dense" (1 | 0: lvllm_varlen_attention's long plain
 * launches read key/value in place instead of packing them first), "varlen_dense_waves" (waves per workgroup of that
 * launch: 8, 4, or 0 = by the longest sequence).  lvllm_get_tuning reads a knob back. */
int lvllm_set_tuning(const char* key, int value);
int lvllm_get_tuning(const char* key, int* value);

/* ---- attention (replaces csrc/ops.h:8-27, attention_kernels.cu:808-997) --- */

/* paged_attention_v1: out[num_seqs,num_heads,head_size] (contiguous).
 * query rows may be strided (q_stride = query.stride(0)).
 * key_cache  [num_blocks, num_kv_heads, head_size/x, block_size, x], x=16/sizeof(cache elt)
 * value_cache[num_blocks, num_kv_heads, head_size, block_size]
 * block_tables int32 [num_seqs, max_num_blocks_per_seq]; seq_lens int32 [num_seqs].
 * alibi_slopes: float32 [num_heads] or NULL.
 * blocksparse_*: as csrc/ops.h:13-16; vert_stride <= 1 means dense. */
int lvllm_paged_attention_v1(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int block_size,
    int max_seq_len, int max_num_blocks_per_seq, const float* alibi_slopes,
    int64_t q_stride, int64_t kv_block_stride, int64_t kv_head_stride,
    int dtype, int kv_dtype, float k_scale, float v_scale, int tp_rank,
    int blocksparse_local_blocks, int blocksparse_vert_stride,
    int blocksparse_block_size, int blocksparse_head_sliding_step, int64_t kv_cache_bytes, void* stream);

/* paged_attention_v2: as v1, split in partitions of 512 tokens
 * (attention_kernels.cu:850) and merged by a reduce pass.  Scratch, caller
 * allocated exactly as light_vllm/decoding/backends/attention/ops/paged_attn.py:156-166:
 *   tmp_out   [num_seqs,num_heads,max_num_partitions,head_size] (dtype)
 *   exp_sums, max_logits float32 [num_seqs,num_heads,max_num_partitions]
 * They are SCRATCH, and what they hold after the call DIFFERS from the reference by default:
 *   - the reference cuts every context at 512 tokens: slot j of a (seq, head) row holds partition j's
 *     (max logit, exp sum, normalised partial output) (attention_kernels.cu:349-357,483-495);
 *   - this library cuts a context into n EQUAL SHARES of whole 16-token tiles, n chosen per call to fill the
 *     GPU and n <= max_num_partitions: slot j holds share j's quantities (same kind, other token ranges).
 *     n = 1 when the batch alone fills the GPU (e.g. 32 sequences x 8 kv heads): ONE PASS writes `out`
 *     directly, no reduce launch, and the scratch is NOT TOUCHED.
 * `out` is the same attention result either way.  A caller that reads the scratch (none in the reference:
 * paged_attn.py allocates and drops it) selects the reference's partitioning with
 * lvllm_set_tuning("attn_splits", -1): 512-token partitions whatever the batch, every non-empty slot
 * written, then the reduce pass -- tests/test_ops_gpu.py compares the three tensors with the oracle's. */
int lvllm_paged_attention_v2(
    void* out, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int block_size,
    int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions,
    const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
    float v_scale, int tp_rank, int blocksparse_local_blocks,
    int blocksparse_vert_stride, int blocksparse_block_size,
    int blocksparse_head_sliding_step, int64_t kv_cache_bytes, void* stream);

/* The two passes of paged_attention_v2 individually, for profiling and for the benchmark's
 * per-kernel timing: phases = 1 runs only the partition pass (writes tmp_out / exp_sums /
 * max_logits), 2 only the reduce pass (attention_kernels.cu:564-669), 3 both (== v2). */
int lvllm_paged_attention_v2_phases(
    void* out, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, int block_size,
    int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions,
    const float* alibi_slopes, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale,
    float v_scale, int tp_rank, int blocksparse_local_blocks,
    int blocksparse_vert_stride, int blocksparse_block_size,
    int blocksparse_head_sliding_step, int64_t kv_cache_bytes, int phases, void* stream);

/* Extension (not a reference operator): the three launches a decode step makes per layer --
 *   rotary_embedding(positions, query, key, head_size, cos_sin_cache, is_neox)      csrc/ops.h:35-37
 *   reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, "auto")     csrc/cache.h:18-22
 *   paged_attention_v2(out, ..., query, key_cache, value_cache, ...)                csrc/ops.h:18-27
 * -- as ONE (qwen2.py:151-154 -> layer.py:83-105 of the reference run them back to back).  One token per
 * sequence: query [num_seqs, num_heads, D], key / value [num_seqs, num_kv_heads, D] (row strides in elements),
 * positions / slot_mapping int64 [num_seqs].  The attention workgroup of a (sequence, kv head) rotates its query
 * heads and the new key in registers (arithmetic and roundings of lvllm_rotary_embedding), writes the rotated key
 * and the value into the paged caches at the slot, and attends to them from registers.  `out`, the caches and the
 * scratch hold bit for bit what the three calls leave; query and key are NOT rotated in place (nothing downstream
 * of attention reads them).  NeoX pairing, rot_dim == head_size in {64, 128, 256}, 16-bit types, block_size
 * 16 | 32, GQA group <= 16; kv_dtype AUTO (scales 1.0) or FP8_E4M3 (head size 128 | 256; the rotated key and the
 * value are quantised with their scales on the way into the cache, as lvllm_reshape_and_cache does).  Returns 3 --
 * and does nothing -- outside that envelope. */
int lvllm_rope_cache_paged_attention(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query, const void* key,
    const void* value, void* key_cache, void* value_cache, int num_seqs, int num_heads, int head_size,
    int num_kv_heads, float scale, const int32_t* block_tables, const int32_t* seq_lens,
    const int64_t* positions, const int64_t* slot_mapping, const void* cos_sin_cache, int rot_dim, int is_neox,
    int block_size, int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions, int64_t q_stride,
    int64_t key_stride, int64_t value_stride, int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
    int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes, void* stream);

/* ---- cache ops (replaces csrc/cache.h:9-33, cache_kernels.cu) ------------- */

/* reshape_and_cache: scatter key/value [num_tokens,num_heads,head_size]
 * (row strides key_stride/value_stride) into the paged caches via
 * slot_mapping int64 [num_tokens]; slot < 0 is skipped (cache_kernels.cu:164-203). */
int lvllm_reshape_and_cache(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int x, int64_t key_stride, int64_t value_stride, int dtype,
    int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes, void* stream);

/* The same over caches whose blocks are kv_block_stride cache elements apart (>= num_heads * head_size * block_size,
 * a multiple of 16 bytes): what key_cache.stride(0) says.  The attention entries have always taken that stride; with
 * it the cache ops follow a padded allocation too -- blocks exactly 32 KiB apart put the same tile of every
 * sequence of a freshly filled cache on the same HBM channels (sequence i's blocks start i x 2 MiB in), and 1 KiB of
 * padding per block makes the decode attention launch 12 % faster there (profiles/r03_tuning.md section 9).
 * lvllm_reshape_and_cache is this with the dense stride. */
int lvllm_reshape_and_cache_strided(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int x, int64_t key_stride, int64_t value_stride, int dtype,
    int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes, int64_t kv_block_stride, void* stream);

/* reshape_and_cache_flash: cache layout [num_blocks, block_size, num_heads,
 * head_size], block_stride = key_cache.stride(0) (cache_kernels.cu:206-247). */
int lvllm_reshape_and_cache_flash(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int64_t block_stride, int64_t key_stride,
    int64_t value_stride, int dtype, int kv_dtype, float k_scale, float v_scale,
    void* stream);

/* copy_blocks: for every layer and every (src,dst) pair copy one block in the
 * key cache and in the value cache (cache_kernels.cu:67-148).
 * key_cache_ptrs/value_cache_ptrs: DEVICE arrays of num_layers device
 * pointers; block_mapping: DEVICE int64 [num_pairs,2]; block_bytes = bytes of
 * one block of one layer's K (== V) cache. */
int lvllm_copy_blocks(const void* const* key_cache_ptrs,
                      const void* const* value_cache_ptrs,
                      const int64_t* block_mapping, int num_layers,
                      int num_pairs, int64_t block_bytes, void* stream);

/* swap_blocks: block_mapping is a HOST int64 [num_pairs,2] array
 * (cache_kernels.cu:40-43).  src/dst may each be device or (pinned) host
 * memory: src_is_device / dst_is_device say which (cache_kernels.cu:26-39).
 * Runs of consecutive (src,dst) block numbers are merged into one copy. */
int lvllm_swap_blocks(const void* src, void* dst, const int64_t* block_mapping,
                      int num_pairs, int64_t block_bytes, int src_is_device,
                      int dst_is_device, void* stream);

/* ---- norm / rope / activation (csrc/ops.h:29-45) -------------------------- */

/* rms_norm: out = T(x * rsqrt(mean(x^2)+eps)) * w   (layernorm_kernels.cu:21-45) */
int lvllm_rms_norm(void* out, const void* input, const void* weight,
                   float epsilon, int num_tokens, int hidden_size, int dtype,
                   void* stream);

/* fused_add_rms_norm: residual = T(input+residual); input = norm(residual)*w
 * in place (layernorm_kernels.cu:200-287). */
int lvllm_fused_add_rms_norm(void* input, void* residual, const void* weight,
                             float epsilon, int num_tokens, int hidden_size,
                             int dtype, void* stream);

/* rotary_embedding: in-place on query [num_tokens, num_heads*head_size] and
 * key [num_tokens, num_kv_heads*head_size] with row strides query_stride /
 * key_stride; cos_sin_cache [max_position, rot_dim] = [cos | sin] in dtype;
 * positions int64 [num_tokens] (pos_encoding_kernels.cu:10-92). */
int lvllm_rotary_embedding(const int64_t* positions, void* query, void* key,
                           int num_tokens, int num_heads, int num_kv_heads,
                           int head_size, int rot_dim, int64_t query_stride,
                           int64_t key_stride, const void* cos_sin_cache,
                           int is_neox, int dtype, void* stream);

/* silu_and_mul: out[t,i] = T(silu(x[t,i])) * x[t,d+i]  (activation_kernels.cu:9-30) */
int lvllm_silu_and_mul(void* out, const void* input, int64_t num_tokens, int d,
                       int dtype, void* stream);

/* Exact (erf) GELU, elementwise: out = T(0.5 x (1 + erf(x / sqrt 2))) in fp32 (torch.nn.functional.gelu of the
 * encoder models' MLP, a torch op in the reference).  16-bit types, numel % 8 == 0, out may alias x. */
int lvllm_gelu(void* out, const void* x, int64_t numel, int dtype, void* stream);

/* ---- extension (no counterpart operator in the reference): weight-streaming GEMM --------
 * Y[M,N] = X[M,K] . W[N,K]^T (+ bias[N]) for decode batches (M <= 64), bf16/f16, fp32
 * accumulate: the dense projections of light_vllm/backends/linear.py:134-139 (F.linear) at
 * decode sizes, where they are HBM-bound.  Returns 3 when the shape is outside the kernel's
 * envelope (caller uses a library GEMM).  `workspace`: lvllm_skinny_gemm_workspace_bytes()
 * bytes of device memory (fp32 partials when K is split over workgroups), may be NULL if 0.
 * `packed` != 0: W is in the MFMA-fragment order written by lvllm_pack_weight
 * ([N/16][K/32][4][16][8]); every wave load is then one contiguous KiB. */
int64_t lvllm_skinny_gemm_workspace_bytes(int M, int N, int K);
int lvllm_skinny_gemm(void* y, const void* x, const void* w, const void* bias, int M, int N,
                      int K, int64_t ldx, int dtype, int packed, void* workspace,
                      int64_t workspace_bytes, void* stream);
/* Reorders row-major W[N,K] into the packed order (out of place; N % 16 == 0, K % 32 == 0). */
int lvllm_pack_weight(void* dst, const void* src, int N, int K, int dtype, void* stream);
/* Greedy sampling fused into a projection (the lm_head of a decode step): tokens[m] = argmax over n of
 * (X . W^T)[m, n], compared after rounding to the element type (what torch.argmax of the projection's
 * output sees; ties go to the smaller n), without writing the [M, N] result.  M <= 64, packed weights, no
 * bias; when K is split over workgroups (K > 4096 at M <= 32, > 2048 above) the fp32 partials pass through
 * the workspace and the arg-max runs inside the reduce pass.  tokens: int64 [M] on the device. */
int64_t lvllm_skinny_gemm_argmax_workspace_bytes(int M);                    /* K within one workgroup */
int64_t lvllm_skinny_gemm_argmax_workspace_bytes_ex(int M, int N, int K);   /* any shape of the envelope */
int lvllm_skinny_gemm_argmax(int64_t* tokens, const void* x, const void* w_packed, int M, int N, int K,
                             int64_t ldx, int dtype, void* workspace, int64_t workspace_bytes, void* stream);
/* The same product for 1 <= M <= 256 rows (meant for 65..256: a large decode batch, a prefill chunk):
 * X goes through LDS, the waves of a workgroup split N, K is split over workgroups where N alone
 * would not fill the GPU (fp32 partials in `workspace`, lvllm_stream_gemm_workspace_bytes).
 * W packed by lvllm_pack_weight.  Returns 3 outside the envelope (M > 256, K % 32, N % 16, W >= 4 GiB). */
int64_t lvllm_stream_gemm_workspace_bytes(int M, int N, int K);
int lvllm_stream_gemm(void* y, const void* x, const void* w_packed, const void* bias, int M, int N,
                      int K, int64_t ldx, int dtype, void* workspace, int64_t workspace_bytes,
                      void* stream);
/* As lvllm_skinny_gemm, plus: act = 1 -> X rows are [gate | up] (2K wide) and the kernel
 * multiplies by T(T(silu(gate)) * up) (silu_and_mul fused into the down projection);
 * act = 2 -> W rows are [gate (N/2) | up (N/2)] and y is [M, N/2] = silu_and_mul of the projection,
 * applied in the epilogue (N % 32 == 0, K not split over workgroups, no partial_out);
 * partial_out != 0 -> the fp32 partial sums [ksplit, M, N] stay in `workspace` (at least
 * max(ksplit,1)*M*N*4 bytes), y is not written, *ksplit_out = number of partials. */
int lvllm_skinny_gemm_ex(void* y, const void* x, const void* w, const void* bias, int M, int N,
                         int K, int64_t ldx, int dtype, int packed, int act, int partial_out,
                         int* ksplit_out, void* workspace, int64_t workspace_bytes, void* stream);
/* LayerNorm(x + y) of the encoder models (xlm_roberta.py; torch ops in the reference): z = T(x + y),
 * out = T((z - mean) * rsqrt(var + eps) * weight + bias), statistics in fp32.  y may be NULL (plain
 * LayerNorm).  16-bit element types, hidden_size % 8 == 0, contiguous rows, out may alias x. */
int lvllm_add_layer_norm(void* out, const void* x, const void* y, const void* weight, const void* bias,
                         float epsilon, int num_tokens, int hidden_size, int dtype, void* stream);
/* fused_add_rms_norm whose input is the fp32 split-K partials of the preceding projection:
 * x = T(sum_s partials[s]); residual = T(x + residual); out = norm(residual) * weight. */
int lvllm_fused_add_rms_norm_splitk(void* out, void* residual, const float* partials,
                                    int num_partials, const void* weight, float epsilon,
                                    int num_tokens, int hidden_size, int dtype, void* stream);
/* The same behind a W8A8 projection that left its raw partials (lvllm_skinny_gemm_w8a8_ex, act = 4):
 * x = T(sum_s partials[s] * (*x_scale * *w_scale)), the value that projection's own reduce pass writes. */
int lvllm_fused_add_rms_norm_splitk_scaled(void* out, void* residual, const float* partials,
                                           int num_partials, const void* weight, float epsilon,
                                           int num_tokens, int hidden_size, int dtype,
                                           const float* x_scale, const float* w_scale, void* stream);
/* The norm launches of a W8A8 decode step with an fp8 twin of their result: out_fp8 [num_tokens, hidden] bytes =
 * static_scaled_fp8_quant(normalised row, *q_scale) (csrc/quantization/fp8/common.cu:24-38,171-176) on the value as
 * rounded to T, bit for bit -- the next projection (lvllm_skinny_gemm_w8a8_q) then takes its activations already
 * quantised instead of re-quantising them in every workgroup.  16-bit element types, hidden_size % 8 == 0, 16-byte
 * aligned rows.  lvllm_rms_norm_quant: `out` (the T result) may be NULL.  lvllm_fused_add_rms_norm_quant: residual <-
 * T(input + residual) as lvllm_fused_add_rms_norm; the normalised rows go to out_fp8 and, when write_normed != 0, to
 * `input` (otherwise `input` is only read).  lvllm_fused_add_rms_norm_splitk_quant: `out` may be NULL; x_scale /
 * w_scale as in lvllm_fused_add_rms_norm_splitk_scaled (both NULL: plain partials); out_fp8 / q_scale both NULL = that
 * entry. */
int lvllm_rms_norm_quant(void* out, void* out_fp8, const float* q_scale, const void* input, const void* weight,
                         float epsilon, int num_tokens, int hidden_size, int dtype, void* stream);
int lvllm_fused_add_rms_norm_quant(void* input, void* residual, const void* weight, float epsilon, int num_tokens,
                                   int hidden_size, int dtype, void* out_fp8, const float* q_scale, int write_normed,
                                   void* stream);
int lvllm_fused_add_rms_norm_splitk_quant(void* out, void* residual, const float* partials, int num_partials,
                                          const void* weight, float epsilon, int num_tokens, int hidden_size,
                                          int dtype, const float* x_scale, const float* w_scale, void* out_fp8,
                                          const float* q_scale, void* stream);
/* rotary_embedding (rot_dim == head_size) and reshape_and_cache of the rotated key and the
 * value in one launch; returns 3 outside its envelope (use the two separate entry points). */
int lvllm_rotary_embedding_and_cache(
    const int64_t* positions, void* query, void* key, const void* value, int num_tokens, int num_heads,
    int num_kv_heads, int head_size, int rot_dim, int64_t query_stride, int64_t key_stride,
    int64_t value_stride, const void* cos_sin_cache, int is_neox, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int block_size, int dtype, void* stream);

/* ... and into an fp8 (e4m3fn, x = 16) cache: the rotated key and the value are quantised with their
 * scales on the way in (as lvllm_reshape_and_cache with kv_dtype FP8_E4M3 would after the rotation). */
int lvllm_rotary_embedding_and_cache_ex(
    const int64_t* positions, void* query, void* key, const void* value, int num_tokens, int num_heads,
    int num_kv_heads, int head_size, int rot_dim, int64_t query_stride, int64_t key_stride,
    int64_t value_stride, const void* cos_sin_cache, int is_neox, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int block_size, int dtype, int kv_dtype, float k_scale, float v_scale,
    int64_t kv_cache_bytes, void* stream);

/* ... and over caches whose blocks are kv_block_stride cache elements apart (see lvllm_reshape_and_cache_strided). */
int lvllm_rotary_embedding_and_cache_strided(
    const int64_t* positions, void* query, void* key, const void* value, int num_tokens, int num_heads,
    int num_kv_heads, int head_size, int rot_dim, int64_t query_stride, int64_t key_stride,
    int64_t value_stride, const void* cos_sin_cache, int is_neox, void* key_cache, void* value_cache,
    const int64_t* slot_mapping, int block_size, int dtype, int kv_dtype, float k_scale, float v_scale,
    int64_t kv_cache_bytes, int64_t kv_block_stride, void* stream);

/* The reduce pass of a QKV projection whose K was split over workgroups (lvllm_skinny_gemm_ex with partial_out: fp32
 * slabs partials[num_partials][num_tokens][(num_heads + 2 num_kv_heads) * head_size], columns [q | k | v]) +
 * rotary_embedding + reshape_and_cache in ONE launch: qkv [num_tokens, (num_heads + 2 num_kv_heads) * head_size]
 * (contiguous, 16-bit) receives T(sum of the slabs + bias) with q and k rotated, the caches the rotated key and the value
 * -- bit for bit what lvllm_skinny_gemm's own reduce pass followed by lvllm_rotary_embedding_and_cache_strided leaves
 * (steps of 33..64 rows: one launch less per layer).  bias: [row] of T or NULL.  3 = outside the envelope. */
int lvllm_rotary_embedding_and_cache_splitk(
    const int64_t* positions, void* qkv, const float* partials, int num_partials, const void* bias, int num_tokens,
    int num_heads, int num_kv_heads, int head_size, int rot_dim, const void* cos_sin_cache, int is_neox,
    void* key_cache, void* value_cache, const int64_t* slot_mapping, int block_size, int dtype, int kv_dtype,
    float k_scale, float v_scale, int64_t kv_cache_bytes, int64_t kv_block_stride, void* stream);

/* Causal varlen attention of prompt chunks over the paged cache: prefill, chunked prefill and
 * prefix-cache hits.  Replaces the reference's third-party call
 *   flash_attn_varlen_func(q, key_cache, value_cache, cu_seqlens_q=query_start_loc,
 *                          cu_seqlens_k=seq_start_loc, causal=True, block_table=...)
 * (light_vllm/decoding/backends/attention/backends/flash_attn.py:538-555; same job as the Triton
 * context_attention_fwd, ops/prefix_prefill.py).  Sequence i owns query tokens
 * query_start_loc[i] .. query_start_loc[i+1]; seq_lens[i] counts the whole context INCLUDING those
 * tokens, whose K/V must already be in the cache (reshape_and_cache runs first, flash_attn.py:488-500).
 * Query t of the chunk sits at position seq_len - query_len + t and sees keys 0 .. that position
 * (the last sliding_window of them when sliding_window > 0) when causal != 0, every key of its
 * sequence when causal == 0 (encoder attention; no ALiBi / window there); softcap > 0 applies
 * cap * tanh(logit / cap).  16-bit dtypes, head sizes of paged_attention, block_size 16 or 32;
 * key_cache / value_cache in the paged_attention layouts.  query [T, num_heads, head_size] with token
 * stride q_stride, out likewise with out_stride (elements). */
int lvllm_paged_prefill_attention(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, const int32_t* query_start_loc,
    int max_query_len, int block_size, int max_num_blocks_per_seq, const float* alibi_slopes,
    int causal, int sliding_window, float softcap, int64_t q_stride, int64_t out_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int dtype, int kv_dtype, void* stream);

/* The same over an fp8 (e4m3fn) cache: kv_dtype = LVLLM_KV_FP8_E4M3 with its scales (head size a
 * multiple of 64); lvllm_paged_prefill_attention is this with scales 1.0. */
int lvllm_paged_prefill_attention_ex(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, const int32_t* query_start_loc,
    int max_query_len, int block_size, int max_num_blocks_per_seq, const float* alibi_slopes,
    int causal, int sliding_window, float softcap, int64_t q_stride, int64_t out_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale, float v_scale,
    int64_t kv_cache_bytes, void* stream);

/* The same with what lets a launch that would leave CUs idle cut its key walk into partitions across workgroups
 * (paged_attention_v2's scheme; csrc/prefill_partitions.h): max_seq_len, the caller's bound on seq_lens (the
 * reference passes max_seqlen_k to flash_attn_varlen_func, flash_attn.py:547; 0: unknown, no partitions), and
 * `workspace` of at least lvllm_paged_prefill_workspace_bytes(...) bytes, 256-byte aligned (null / too small: no
 * partitions).  Launches of a few sequences with short chunks over long contexts are the ones cut (8 x (32 tokens over
 * 2 048): 38 -> see profiles/r03_bench_chunk_attn.txt).  Results do not depend on whether the walk was cut beyond the
 * rounding of the partial results to the model dtype.  num_tokens: the rows of `query` (query_start_loc[num_seqs] or
 * more; 0: unknown) -- launches that are mostly one-token sequences (the mixed steps of chunked prefill) are
 * recognised by it and walk K/V the way paged_attention does (csrc/prefill_chunk.h).
 * lvllm_paged_prefill_attention_ex is this with num_tokens 0 and no workspace. */
int64_t lvllm_paged_prefill_workspace_bytes(int num_seqs, int num_tokens, int max_query_len, int num_heads,
                                            int num_kv_heads, int head_size, int max_seq_len);
int lvllm_paged_prefill_attention_ws(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    int num_seqs, int num_heads, int head_size, int num_kv_heads, float scale,
    const int32_t* block_tables, const int32_t* seq_lens, const int32_t* query_start_loc,
    int max_query_len, int block_size, int max_num_blocks_per_seq, const float* alibi_slopes,
    int causal, int sliding_window, float softcap, int64_t q_stride, int64_t out_stride,
    int64_t kv_block_stride, int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale, float v_scale,
    int64_t kv_cache_bytes, int num_tokens, int max_seq_len, void* workspace, int64_t workspace_bytes, void* stream);

/* Dense varlen attention without a KV cache: the encode-only / prefill-only path
 * (light_vllm/prefill_only/backends/attention/backends/flash_attn.py: flash_attn_varlen_func(q, k, v,
 * cu_seqlens, causal=...); in-tree definition torch_naive.py:65-149).  query [T, num_heads, D],
 * key/value [T, num_kv_heads, D] (token strides in elements), cu_seqlens int32 [num_seqs + 1] on the
 * device, causal = 1 for AttentionType.DECODER, 0 for ENCODER.  No host synchronisation.  Plain launches (head size
 * 64, no ALiBi / soft cap / window) whose longest sequence has 64+ tokens are ONE launch: the 32x32-MFMA body
 * reads the key/value rows where they lie (tuning key "varlen_dense", default 1).  Everything else is two: key/value
 * are packed into paged tiles in `workspace` (sequence s owns the blocks cu_seqlens[s] / 16 + s ...), then
 * lvllm_paged_prefill_attention runs over them with block_tables = seq_lens = NULL, which selects that arithmetic
 * placement.  Both give the same bits.
 * workspace: at least lvllm_varlen_attention_workspace_bytes(...) bytes, 256-byte aligned (required either way). */
int64_t lvllm_varlen_attention_workspace_bytes(int num_tokens, int num_seqs, int max_seq_len,
                                               int num_kv_heads, int head_size);
int lvllm_varlen_attention(
    void* out, const void* query, const void* key, const void* value, const int32_t* cu_seqlens,
    int num_tokens, int num_seqs, int max_seq_len, int num_heads, int num_kv_heads, int head_size,
    float scale, int causal, const float* alibi_slopes, int sliding_window, float softcap,
    int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t out_stride, int dtype,
    void* workspace, int64_t workspace_bytes, void* stream);

/* advance_step: csrc/prepare_inputs/advance_step.cu:14-57, torch_bindings.cpp:75-77.  Moves the
 * first num_queries rows of a decode batch one token forward on the device (tokens <- sampled
 * ids, seq_lens += 1, positions, slot_mapping through block_tables [num_seqs, stride] int32). */
int lvllm_advance_step(int num_seqs, int num_queries, int block_size, int64_t* input_tokens,
                       const int64_t* sampled_token_ids, int64_t* input_positions, int32_t* seq_lens,
                       int64_t* slot_mapping, const int32_t* block_tables, int64_t block_tables_stride,
                       void* stream);

/* The same inside a multi-step decode (num_scheduler_steps > 1: k model steps replayed back to back, this
 * call between them): token_log (nullable) also receives the sampled ids, row for row -- step j's tokens
 * are kept for the host in row j of a [k, num_seqs] buffer -- and with skip_empty_rows != 0 a row whose
 * seq_lens[i] <= 0 (padding of a captured batch) is left untouched. */
int lvllm_advance_step_ex(int num_seqs, int num_queries, int block_size, int64_t* input_tokens,
                          const int64_t* sampled_token_ids, int64_t* input_positions,
                          int32_t* seq_lens, int64_t* slot_mapping, const int32_t* block_tables,
                          int64_t block_tables_stride, int64_t* token_log, int skip_empty_rows,
                          void* stream);

/* ---- device-side sampler (replaces the torch sequence of light_vllm/decoding/backends/sampler.py:90-200:
 * _apply_min_tokens_penalty :238-277, _apply_penalties :281-301, _apply_top_k_top_p :304-330, _apply_min_p :333-347,
 * _greedy_sample / _multinomial :350-454) ----
 * One call per step (one launch, or two -- see `scratch`): tokens_out[row] int64 <- the token drawn for row `row` of
 * `logits` [num_rows, vocab]
 * (row stride logits_stride elements, element type `logits_dtype` = enum lvllm_dtype).
 * state_slot (nullable) int32 [num_rows]: the request's slot in the device-resident sampler state, or < 0 for a
 * plain greedy row (arg-max of the logits, ties to the smaller index -- torch.argmax).  A slot is
 *   params + slot * LVLLM_SAMPLER_PARAMS_BYTES : { float temperature, top_p, min_p, presence_penalty,
 *       frequency_penalty, repetition_penalty; int32 top_k (<= 0 or >= vocab: off); int32 min_tokens; uint64 seed;
 *       int32 output_len; int32 num_banned; int32 banned[20] }   (stop tokens banned while output_len < min_tokens)
 *   counts + slot * counts_stride : int32 [vocab], bit 31 = token occurs in the prompt, bits 0..30 = occurrences in
 *       the output so far (lvllm_sampler_init_row fills it from the histories).
 * update_state != 0: the drawn token is added to the slot's counts and output_len += 1 on the device, so the model
 * steps of a multi-step burst need no host round trip.  scratch: float [num_rows, scratch_stride >= vocab] working
 * copy of the rows (needed when state_slot is given).  With scratch_stride >= ((vocab + 3) & ~3) + 64 and the floats
 * behind the vocabulary ZERO when the buffer is first used, up to 8 workgroups share the first pass of a row
 * (penalties, temperature, the row's maximum) and -- as a second launch -- its draw (launches of a multiple of 8
 * rows), and meet in that tail -- the kernel leaves its arrival counter at zero; such a buffer belongs
 * to one launch at a time (launches in flight on different streams bring their own).  processed_out (nullable) float [num_rows, processed_stride]:
 * the logits as they stand before the draw (-inf = filtered out), for tests.  temperature < 1e-5 = greedy on the
 * penalised logits.  Random numbers: Philox4x32-10, key = seed, counter = (vocabulary index / 4, output_len).
 * Deterministic: integer histograms and 2^-40 fixed-point mass sums, no floating-point atomics. */
#define LVLLM_SAMPLER_PARAMS_BYTES 128
int lvllm_sample_rows(int64_t* tokens_out, const void* logits, int64_t logits_stride, int logits_dtype,
                      int num_rows, int vocab, const int32_t* state_slot, void* params, int32_t* counts,
                      int64_t counts_stride, int num_slots, float* scratch, int64_t scratch_stride,
                      float* processed_out, int64_t processed_stride, int update_state, void* stream);
/* counts_row int32 [vocab] <- 0, then bit 31 for every prompt token and +1 for every output token (ids outside
 * [0, vocab) -- padding -- are ignored).  Token lists are device (or device-visible pinned) int64 arrays. */
int lvllm_sampler_init_row(int32_t* counts_row, int vocab, const int64_t* prompt_tokens, int n_prompt,
                           const int64_t* output_tokens, int n_output, void* stream);

/* Test hook for the draw's arithmetic: out float [n, 3] <- { u, q = -ln(u), ln(q) } for the 32-bit random words
 * r [n] (device pointers), computed by the device code the draw itself runs: u = ((r >> 9) + 0.5) * 2^-23 lies
 * strictly inside (0, 1), so q > 0 and the race score x - ln(q) is finite for every r. */
int lvllm_sampler_draw_probe(const uint32_t* r, float* out, int n, void* stream);

/* convert_fp8: csrc/cache_kernels.cu:334-410, torch_bindings.cpp:261-264 ("only for testing" there).
 * to_fp8 != 0: dst (bytes) = fp8(float(src) / scale); else dst = T(float(fp8 src) * scale).  `dtype` is the
 * element type of the non-fp8 side; contiguous buffers of num_elems elements. */
int lvllm_convert_fp8(void* dst, const void* src, float scale, int64_t num_elems, int dtype, int to_fp8,
                      int kv_dtype, void* stream);

/* ---- fp8 activation quantisation (csrc/quantization/fp8/common.cu, torch_bindings.cpp:185-202) ----
 * out: OCP e4m3fn bytes (c10::Float8_e4m3fn).  static: out = fp8(clamp(x * (1/scale), +-448)), scale[1].
 * dynamic: scale[0] (<= 0 on entry) = max|x| / 448 first, then as static.  per token:
 * scales[token] = max(min(absmax, *scale_ub) / 448, 1/(448*512)), out = fp8(clamp(x / scale)). */
int lvllm_static_scaled_fp8_quant(void* out, const void* input, const float* scale, int64_t num_elems,
                                  int dtype, void* stream);
int lvllm_dynamic_scaled_fp8_quant(void* out, const void* input, float* scale, int64_t num_elems, int dtype,
                                   void* stream);
int lvllm_dynamic_per_token_scaled_fp8_quant(void* out, float* scales, const void* input,
                                             const float* scale_ub, int num_tokens, int hidden_size,
                                             int dtype, void* stream);

/* W8A8 decode projection on the weight-streaming kernel: y = T((fp8(x / *x_scale) . w8^T) * *x_scale *
 * *w_scale + bias) -- torch._scaled_mm(fp8(x), fp8(w), scale_a, scale_b, bias) of w8a8_utils.py:147-156
 * with a static activation scale, the activation quantised while the MFMA operands are built
 * (arithmetic of static_scaled_fp8_quant).  x [M <= 64, K] in `dtype` (rows ldx apart), w_packed =
 * fp8 e4m3fn weights [N, K] packed by lvllm_pack_weight on their [N, K/2] 16-bit view; K % 64 == 0,
 * N % 16 == 0.  Returns 3 outside the envelope. */
int64_t lvllm_skinny_gemm_w8a8_workspace_bytes(int M, int N, int K);
int lvllm_skinny_gemm_w8a8(void* y, const void* x, const void* w_packed, const void* bias,
                           const float* x_scale, const float* w_scale, int M, int N, int K, int64_t ldx,
                           int dtype, void* workspace, int64_t workspace_bytes, void* stream);
/* As lvllm_skinny_gemm_w8a8 with act = 0; act = 2: W rows are [gate (N/2) | up (N/2)] and y is [M, N/2],
 * silu_and_mul applied in the epilogue (N % 32 == 0, K not split over workgroups); act = 3: y is int64 [M],
 * the arg-max over n of the rounded result (as lvllm_skinny_gemm_argmax; workspace of
 * lvllm_skinny_gemm_argmax_workspace_bytes(M), no bias, K not split over workgroups); act = 4: K IS split over
 * workgroups (lvllm_skinny_gemm_w8a8_workspace_bytes > 0) and the raw fp32 partial sums [ksplit, M, N] stay in
 * `workspace` -- unscaled, y not written, no bias -- for lvllm_fused_add_rms_norm_splitk_scaled. */
int lvllm_skinny_gemm_w8a8_ex(void* y, const void* x, const void* w_packed, const void* bias,
                              const float* x_scale, const float* w_scale, int M, int N, int K,
                              int64_t ldx, int dtype, int act, void* workspace,
                              int64_t workspace_bytes, void* stream);

/* W8A8 with activations that arrive ALREADY quantised: x_fp8 [M <= 32, K] bytes, rows ldx BYTES apart, =
 * static_scaled_fp8_quant(x, *x_scale) as written by the *_quant norm entries above or by this entry's SwiGLU epilogue.
 * Results are bit-identical to lvllm_skinny_gemm_w8a8_ex on the unquantised x (same quantisation arithmetic, same
 * MFMA order).  act 0 | 2 (SwiGLU: y [M, N/2]; y_fp8 != NULL: the activation also -- or, with y == NULL, only --
 * leaves as fp8 [M, N/2] quantised with *y_fp8_scale, for a following call of this entry) | 4 (raw split-K partials
 * in `workspace`, as lvllm_skinny_gemm_w8a8_ex).  `dtype` = element type of y and bias.  3 = outside the envelope. */
int lvllm_skinny_gemm_w8a8_q(void* y, void* y_fp8, const float* y_fp8_scale, const void* x_fp8,
                             const void* w_packed, const void* bias, const float* x_scale, const float* w_scale,
                             int M, int N, int K, int64_t ldx, int dtype, int act, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* paged_attention_v2 / lvllm_rope_cache_paged_attention with an fp8 twin of the result: out_fp8 [num_seqs, num_heads,
 * head_size] bytes = static_scaled_fp8_quant(out, *out_fp8_scale), for a W8A8 output projection that takes its
 * activations already quantised (lvllm_skinny_gemm_w8a8_q).  Same arguments otherwise (no ALiBi, no block-sparse
 * arguments).  Only launches that are NOT cut into shares write the twin (bs 32 x 8 kv heads: one pass); otherwise 3
 * is returned and nothing was launched. */
int lvllm_paged_attention_v2_q(
    void* out, void* out_fp8, const float* out_fp8_scale, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key_cache, const void* value_cache, int num_seqs, int num_heads, int head_size,
    int num_kv_heads, float scale, const int32_t* block_tables, const int32_t* seq_lens, int block_size,
    int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions, int64_t q_stride, int64_t kv_block_stride,
    int64_t kv_head_stride, int dtype, int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes,
    void* stream);
int lvllm_rope_cache_paged_attention_q(
    void* out, void* out_fp8, const float* out_fp8_scale, float* exp_sums, float* max_logits, void* tmp_out,
    const void* query, const void* key, const void* value, void* key_cache, void* value_cache, int num_seqs,
    int num_heads, int head_size, int num_kv_heads, float scale, const int32_t* block_tables, const int32_t* seq_lens,
    const int64_t* positions, const int64_t* slot_mapping, const void* cos_sin_cache, int rot_dim, int is_neox,
    int block_size, int max_seq_len, int max_num_blocks_per_seq, int max_num_partitions, int64_t q_stride,
    int64_t key_stride, int64_t value_stride, int64_t kv_block_stride, int64_t kv_head_stride, int dtype,
    int kv_dtype, float k_scale, float v_scale, int64_t kv_cache_bytes, void* stream);

/* ---- device queries (csrc/cuda_utils.h, torch_bindings.cpp:271-279) ------- */
int64_t lvllm_get_device_attribute(int64_t attribute, int64_t device_id);
int64_t lvllm_get_max_shared_memory_per_block_device_attribute(int64_t device_id);

#ifdef __cplusplus
}
#endif
#endif /* LVLLM_HIP_H_ */
