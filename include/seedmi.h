/*
 * seedmi — C ABI of the MI355X-native SEED tokenize-and-generate hot path (libseedmi.so, gfx950 only).
 *
 * This header is the drop-in boundary.  The reference (AILab-CVC/SEED @ 2024_10_08) is pure Python on
 * PyTorch/xformers and has no FFI of its own; the entry points below are what a ctypes binding placed behind
 *   models/seed_llama_tokenizer.py:75-90,185-202        (ImageTokenizer.encode / SeedLlamaTokenizer.encode_image)
 *   models/seed_qformer/qformer_quantizer.py:288-307    (Blip2QformerQuantizer.get_codebook_indices)
 *   models/llama_xformer.py:661-743                     (LlamaForCausalLM.forward)
 * binds (see INTEGRATION.md for the stub).  Each kernel-level function cites the reference lines it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / HIP types in signatures (`stream` is a hipStream_t passed
 *     as void*; NULL = the null stream).
 *   - every pointer is a DEVICE pointer owned by the caller; bf16 tensors are raw uint16 storage, row-major with an
 *     explicit leading dimension in elements.  Pointers must be 16-byte aligned, leading dimensions multiples of 8.
 *   - every call is asynchronous and stream-ordered; nothing is allocated or freed inside the library; scratch
 *     memory is sized with *_workspace_bytes() and passed in.
 *   - return value: 0 (SEEDMI_OK) or a negative SEEDMI_E_* code; seedmi_last_error() returns a thread-local
 *     description of the last failure.  Errors correspond to the reference's Python asserts / ValueErrors
 *     (seed_llama_tokenizer.py:192, eva_vit.py:227-228, llama_xformer.py:515-522).
 *   - thread safety: calls on distinct streams with distinct workspaces may run concurrently; weight structs are
 *     read-only.
 */
#ifndef SEEDMI_H
#define SEEDMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEEDMI_OK 0
#define SEEDMI_E_SHAPE (-1)
#define SEEDMI_E_DTYPE (-2)
#define SEEDMI_E_ALIGN (-3)
#define SEEDMI_E_ARCH (-4)
#define SEEDMI_E_HIP (-5)

/* Bumped whenever an existing entry point changes its argument list or a struct in this header changes layout (new entry points alone
 * do not bump it).  1: round 1.  2: seedmi_sample_token_bf16 / seedmi_rope_kv_append / seedmi_llama_decode_attention_bf16 gained an
 * argument before `stream`, seedmi_vit_layer_t grew by six pointers (LayerNorm fold).  3: seedmi_gemm_ext_t grew (tile-span statistics).
 * 4 (round 5): the calibration kernels seedmi_bench_stream_read / seedmi_bench_mfma_bf16 left the library (tools/calib/libseedcal.so), llama /
 * split-K workspaces carry a tag written by seedmi_llama_workspace_init / seedmi_gemm_skinny_workspace_init which the status calls require,
 * the llama workspace lost its stream-K region (smaller seedmi_llama_workspace_bytes), seedmi_set_option lost its history / A-B keys.
 * A caller compiled against this header must check  seedmi_version() == SEEDMI_ABI_VERSION  before its first call: a mismatch
 * means arguments and struct strides no longer line up (silent corruption, not an error).  seed_amd/lib.py does. */
#define SEEDMI_ABI_VERSION 4

int seedmi_version(void);
/* The 16-bit element type this build of the library computes in: 0 = bf16 (libseedmi.so: what BASELINE.json's configs name), 1 = IEEE
 * fp16 (libseedmi_f16.so: the reference's shipped setting, configs/tokenizer/seed_llama_tokenizer_hf.yaml:3 `fp16: True`,
 * models/seed_llama_tokenizer.py:58-59,86-87, configs/llm/seed_llama_8b.yaml:4).  Both are built from the same sources (-DSEEDMI_F16
 * switches the conversions, the rounding points' target type and v_mfma_f32_16x16x32_{bf16,f16}); every "bf16" in an entry point's name
 * or comment below reads "the library's 16-bit element".  A caller picks the library by the dtype of the tensors it passes. */
int seedmi_compute_dtype(void);
const char* seedmi_last_error(void);
/* 0 if the current device is a gfx950 (MI355X); SEEDMI_E_ARCH otherwise. */
int seedmi_check_device(void);
/* Tuning overrides: PROCESS-WIDE selections between kernels / schedules (relaxed atomics: reads are race-free, but they are not part of
 * the per-stream thread-safety contract - set them before concurrent use; production callers leave the defaults).  The product library
 * keeps only the keys below; every option marked "=" computes bit-identical results under every value, "~" marks the three that do not.
 * History values of the schedules, timing-only ablations, rejected kernel variants and A/B knobs (gemm_sched 81 / 31 / ..., attn_vit 4 / 6 / 7,
 * gemm_residual_nt, gemm_prefetch_residual, tokenize_tile_stats, skinny_waves / skinny_rows / skinny_nt, decode_persistent, *_ablate ...)
 * exist only in the -DSEEDMI_DEVTOOLS build (libseedmi_dev.so, used by tools/).  Unknown keys or values return SEEDMI_E_SHAPE.
 *
 *   key                    values (default first)        what it selects
 * = gemm                   0 | 64 | 65 | 66..70 | 128 | 256   tile kernel: by shape | 64x64 deep-ring small-M kernel (65: its four-wave form without
 *                                                        producer waves) | shaped small-M kernel with dedicated producer waves: 64x64 / 128x64 /
 *                                                        64x128 tiles / the shape by cost / 32x64 | 128x128 | persistent 256x256
 * = gemm_small             1 | 2 | 0                     small M (128x128 tiles cannot give every CU a workgroup; one image: M = 257): the shaped
 *                                                        kernel up to M = 1100, the 64x64 kernel beyond | always the 64x64 kernel | neither
 * = gemm64_xcd             1 | 0                         64x64 kernel: XCD-contiguous tile order (the m-tiles sharing a W panel on one XCD) |
 *                                                        workgroup b takes tile b
 * = gemm_sched             -1 = 8273 | 24657 | 57425 | 0 schedule of the 256x256 kernel: two-phase K-tile, position-free body | + seam (the next
 *                                                        tile's operands requested by the K loop's last K-tiles) | + peeled first K-tiles whose
 *                                                        waits leave the output stores in flight | the plain four-phase schedule of round 2
 * = gemm_persist           1 | 0                         one workgroup per CU walking tiles | one workgroup per tile
 * = gemm_streamk           1 | 0                         stream-K tail when the caller passes a workspace (seedmi_gemm_bf16_ws)
 * = gemm_group_m           0 | 1..64                     m-tiles per L2 tile group (0: by shape)
 * = gemm_min_tiles         160 | 1..4096                 fewer 256x256 tiles than this -> 128x128 kernel
 * = tokenize_streams       2 | 1..4                      concurrent sub-batches inside seedmi_tokenize
 * = tokenize_streamk       0 | 1                         stream-K tail for the tokenizer's big GEMMs
 * = tokenize_split_rounds  0 | 1                         whole rounds of 256x256 tiles + a 128x128-tiled remainder call
 * = tokenize_vq_head       1 | 0                         encode_task_layer's last Linear fused into the VQ argmin kernel
 * ~ tokenize_lnfold        1 | 0                         LayerNorm folded into qkv / fc1 (one rounding) | explicit LayerNorm launches (the
 *                                                        reference's two roundings): the documented fidelity switch of the ViT
 * ~ skinny_splitk          1 | 0 | 2 | 3                 decode GEMM: balanced split-K where the shape asks | one tile per workgroup | always cut |
 *                                                        cut only badly filled shapes - equal up to the order of the fp32 K summation
 * ~ prefill_tiled          1 | 0                         LDS-tiled causal prefill attention | the row-at-a-time kernel
 * = decode_fused           1 | 0                         RoPE + cache append inside the decode attention launch
 * = decode_attn_early      1 | 0 | 2                     which cached rows the fused decode attention requests ahead of the rotation
 * = attn_vit               5 | 0 | 1 | 2 | 3             ViT attention (257 tokens, head dim 88): staggered sixteen-wave | generic full-row |
 *                                                        twelve-wave | sixteen-wave | sixteen-wave, 16-byte stores
 * = attn_xcd               1 | 0                         all heads of an image on one XCD (attn_vit 5, full launches)
 * = attn_small             1 | 2..16 | 0                 ViT attention launches with fewer (image, head) items than half the CUs (one image: 16):
 *                                                        every item's query tiles split over several workgroups (automatic | that many) | never
 * = attn_store_wait        1 | 0                         attn_vit 5: the wait for the next item's K / Q leaves the output stores in flight
 * = attn_trv               1 | 0                         generic kernel: V through ds_read_b64_tr_b16 | a transposed LDS image */
int seedmi_set_option(const char* key, int value);

/* ---- GEMM epilogues ------------------------------------------------------------------------------------------ */
#define SEEDMI_EPI_NONE 0          /* C = A W^T                                                                   */
#define SEEDMI_EPI_BIAS 1          /* nn.Linear                                                                   */
#define SEEDMI_EPI_BIAS_GELU 2     /* eva_vit.py:60-61, qformer_causual.py:321-322 (exact-erf GELU of the half fc1, by table) */
#define SEEDMI_EPI_BIAS_RESIDUAL 3 /* eva_vit.py:201-202, qformer_causual.py:252-254, llama_xformer.py:316,322     */
#define SEEDMI_EPI_BIAS_TANH 4     /* qformer_quantizer.py:219-221                                                */
#define SEEDMI_EPI_SWIGLU 5        /* llama_xformer.py:186 on row-interleaved gate/up weights; C is [M, N/2]        */
#define SEEDMI_EPI_PATCH_EMBED 6   /* eva_vit.py:229 + 373-377: conv bias + pos_embed, output rows skip one cls row */
#define SEEDMI_EPI_RELU 7          /* qformer_quantizer.py:279-285 (image_down: bias-free Linear + ReLU; bias may be NULL) */

/* C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N]);  bf16 in/out, fp32 MFMA accumulation.  K % 64 == 0.
 * residual/ldr: used by BIAS_RESIDUAL (same row as C) and PATCH_EMBED (pos_embed, row = m % row_group + row_extra;
 * C row = m + (m / row_group + 1) * row_extra).  row_group/row_extra are ignored by the other epilogues. */
int seedmi_gemm_bf16(int M, int N, int K, const void* A, int lda, const void* W, int ldw, const void* bias,
                     const void* residual, int ldr, int epilogue, void* C, int ldc, int row_group, int row_extra,
                     void* stream);
/* Same with a caller-owned workspace of seedmi_gemm_workspace_bytes() (256-byte aligned; its first 4 KiB must have been zeroed once,
 * e.g. by hipMemsetAsync ahead of the first call; may be reused by later calls on the SAME stream).  With it the persistent 256x256
 * kernel cuts its last, partial round of tiles along K into equal ranges (stream-K): every workgroup ends at the same time, and a
 * tile shared by two workgroups continues the fp32 accumulator image its partner published (the same k-ordered chain: results are
 * bit-identical to the plain call).  The first 4 KiB are one flag word per workgroup; every flag is cleared again by the workgroup
 * that consumed it, so the area is all-zero after each launch and the call may be captured in a hipGraph and replayed.  The LAST
 * word (offset 4092) is a sticky error word: non-zero (1 + workgroup id) after a launch in which a workgroup gave up waiting for its
 * partner's partial tile (bounded spin) - that tile of C is then wrong; clear the word to re-arm.  workspace == NULL is exactly
 * seedmi_gemm_bf16. */
size_t seedmi_gemm_workspace_bytes(void);
/* LayerNorm folded into the two GEMMs around it (eva_vit.py:199-202: norm1 -> attn.qkv, norm2 -> mlp.fc1).
 * Consumer (BIAS / BIAS_GELU, N % 64 == 0, ln_stats != NULL): A holds the UN-normalised rows x, W holds half(weight * gamma); ln_stats
 * [M + (M & 1)][2] fp32 = (mean, rstd) of every row of x (an even number of rows must be readable: they are fetched in pairs), ln_colsum [N]
 * fp32 = sum_k W[n][k], bias_f32 [N] fp32 = bias_n + sum_k beta_k weight[n][k], both 16-byte aligned; the epilogue forms
 * rstd * (acc - mean * colsum) + bias_f32 = LayerNorm(x) weight^T + bias with one rounding to half (`bias` is ignored).
 * Producer (BIAS_RESIDUAL, N % 64 == 0, stats_out != NULL): stats_out [N / 64][stats_ld >= M][2] fp32 (span-major planes) receives, per
 * 64-column span and row, (sum, sum of squares) of the half outputs; seedmi_layernorm_stats_finalize reduces them to (mean, rstd).
 * Both may be NULL (= seedmi_gemm_bf16_ws). */
typedef struct {
    const float* ln_stats;
    const float* ln_colsum;
    const float* bias_f32;
    float* stats_out;
    int stats_ld;
    /* ABI 3 - statistics by 256-column TILE, finalized inside the consumer (no seedmi_layernorm_stats_finalize launch in between).
     * Producer: stats_by_tile = 1 writes ceil(N / 256) planes [plane][stats_ld][2] (one (sum, sum of squares) pair per row and n-tile, the
     * four 64-column spans of a tile summed in span order: deterministic) instead of N / 64 span planes; stats_ld must be even.
     * Consumer: ln_planes > 0 says ln_stats points at such planes (ln_planes of them, ln_ld rows each, over ln_cols columns in all) and not
     * at finished (mean, rstd) pairs: every tile sums its rows' ln_planes pairs in plane order and forms
     * mean = sum / ln_cols, rstd = rsqrt(max(sumsq / ln_cols - mean^2, 0) + ln_eps) itself (one expression in every kernel: csrc/common.h
     * seedmi_ln_finish, so an image gives the same ids whichever kernel its batch size selects).
     * Where seedmi_gemm_tile_stats_supported(M, N) is 1 (the 256x256 kernel, large M): up to 6 TILE planes, ln_ld even.
     * Where it is 0 (the 64x64 / 128x128 small-M kernels, e.g. one image): the PRODUCER side (stats_by_tile) is not available - those
     * kernels write the N / 64 span planes - but the CONSUMER side is: ln_planes <= 64 SPAN planes in span order (what a small-M
     * BIAS_RESIDUAL producer wrote: plane s = columns 64 s .. 64 s + 63), any ln_ld >= M, summed in plane order and finished in the
     * consumer's epilogue.  seedmi_tokenize uses this for one-image batches (no seedmi_layernorm_stats_finalize launch per ViT GEMM). */
    int stats_by_tile;
    int ln_planes;
    int ln_ld;
    int ln_cols;
    float ln_eps;
} seedmi_gemm_ext_t;
/* 1 if seedmi_gemm_bf16_ext(M, N, ...) runs on the 256x256 persistent kernel under the current options (the precondition of
 * stats_by_tile / ln_planes), else 0. */
int seedmi_gemm_tile_stats_supported(int M, int N);
int seedmi_gemm_bf16_ext(int M, int N, int K, const void* A, int lda, const void* W, int ldw, const void* bias,
                         const void* residual, int ldr, int epilogue, void* C, int ldc, int row_group, int row_extra,
                         const seedmi_gemm_ext_t* ext, void* workspace, size_t workspace_bytes, void* stream);
/* (mean, rstd = rsqrt(var + eps)) of every row of x [rows, cols] bf16, fp32 statistics like nn.LayerNorm: directly from x ... */
int seedmi_layernorm_stats_bf16(const void* x, int ldx, int rows, int cols, float eps, void* stats_f32x2, void* stream);
/* ... or from the per-span (sum, sum of squares) partials [spans][stats_ld >= rows][2] a BIAS_RESIDUAL GEMM wrote (summed in span
 * order: deterministic). */
int seedmi_layernorm_stats_finalize(const void* partial_f32x2, int spans, int stats_ld, int rows, int cols, float eps,
                                    void* stats_f32x2, void* stream);
int seedmi_gemm_bf16_ws(int M, int N, int K, const void* A, int lda, const void* W, int ldw, const void* bias,
                        const void* residual, int ldr, int epilogue, void* C, int ldc, int row_group, int row_extra,
                        void* workspace, size_t workspace_bytes, void* stream);

/* nn.LayerNorm with fp32 statistics (eva_vit.py:199-202, blip2.py:179-184, qformer_causual.py:96,254,336). */
int seedmi_layernorm_bf16(const void* x, int ldx, const void* gamma, const void* beta, float eps, void* out, int ldo,
                          int rows, int cols, void* stream);
/* LlamaRMSNorm.forward (llama_xformer.py:105-113). */
int seedmi_rmsnorm_bf16(const void* x, int ldx, const void* gamma, float eps, void* out, int ldo, int rows, int cols,
                        void* stream);
/* Same, writing the fragment-major activation layout consumed by seedmi_gemm_skinny_packed_bf16(a_packed = 1):
 * element (m, k) at (((m>>4)*(cols>>5) + (k>>5))*64 + ((k>>3)&3)*16 + (m&15))*8 + (k&7); out holds ceil(rows/16)*16 rows. */
int seedmi_rmsnorm_packed_bf16(const void* x, int ldx, const void* gamma, float eps, void* out_packed, int rows, int cols,
                               void* stream);
/* PatchEmbed unfold (eva_vit.py:222-229): img [B,chans,hw,hw] (fp32 or bf16) -> col [B*(hw/patch)^2, kpad] bf16,
 * k = (c, kh, kw), zero padded to kpad. */
int seedmi_im2col_patch(const void* img, int img_is_fp32, void* col, int batch, int chans, int hw, int patch, int kpad,
                        void* stream);
/* dst[(g*group_rows + r0 + i), :cols] = src[i, :cols]  (cls rows: eva_vit.py:373-377; query expand: qformer_quantizer.py:293) */
int seedmi_fill_rows(void* dst, int ld, int group_rows, int r0, int ngroups, const void* src, int lds, int nsrc,
                     int cols, void* stream);
/* softmax(Q K^T * scale [causal]) V for <= 288 keys; Q/K/V/O are [batch*n, ld] with head h in columns [h*hd,(h+1)*hd).
 * eva_vit.py:139-156, qformer_causual.py:189-236.  round_scores=1 rounds S to bf16 before the softmax like the
 * reference's half matmul output.  head_dim in {64, 88}. */
int seedmi_attention_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                          int batch, int heads, int head_dim, int nq, int nk, float scale, int causal, int round_scores,
                          void* stream);
/* ||e||^2 of every codebook row with the reference's bf16 rounding points (qformer_quantizer.py:95). */
int seedmi_vq_code_sqnorm(const void* codebook, void* ee_f32, int n_embed, int dim, void* stream);
/* VectorQuantizer2 nearest neighbour (qformer_quantizer.py:94-98): ids[r] = argmin_n d(z_r, e_n), first index on ties,
 * int64 output.  dim must be 32. */
int seedmi_vq_argmin_bf16(const void* z, int ldz, const void* codebook, const void* ee_f32, void* ids_i64, int rows,
                          int n_embed, int dim, void* stream);
/* encode_task_layer's last Linear (qformer_quantizer.py:219-223, Linear(hidden, 32) after the Tanh) fused in front of the argmin
 * (SURVEY 8a a13 -> a14): t [rows, hidden] bf16 = the Tanh output, w [32, hidden], bias [32] (may be NULL); z = half(t w^T + bias) goes
 * into the sweep from LDS and, when z_out != NULL, to z_out [rows, 32] (row stride ldz).  hidden % 8 == 0, hidden <= 1024. */
int seedmi_vq_head_argmin_bf16(const void* t, int ldt, int hidden, const void* w, int ldw, const void* bias, const void* codebook,
                               const void* ee_f32, void* ids_i64, void* z_out, int ldz, int rows, int n_embed, int dim, void* stream);

/* ---- LLaMA pieces ---------------------------------------------------------------------------------------------- */
/* nn.Embedding gather (llama_xformer.py:544): out[i,:] = table[ids[i],:]. */
int seedmi_embed_rows(const void* ids_i64, const void* table, int ldt, void* out, int ldo, int n, int cols, int vocab,
                      void* stream);
/* apply_rotary_pos_emb + KV append (llama_xformer.py:160-168,234-239): qkv [B*T, 3*H*hd] -> q_out [B*T, H*hd],
 * caches [B][H][tmax][hd] written at positions past_len..past_len+T-1.  cos/sin: [max_pos, hd] bf16 tables; positions are
 * clamped to [0, max_pos) and a device-resident length to the cache capacity (the reference device-asserts / reallocates). */
int seedmi_rope_kv_append(const void* qkv, int ldqkv, const void* pos_ids_i64, const void* cos_t, const void* sin_t,
                          void* q_out, int ldq, void* k_cache, void* v_cache, int B, int T, int H, int hd, int tmax,
                          int past_len, const void* past_len_dev, int max_pos, void* stream);
/* *counter += delta on the stream (the decode graph advances its device-resident cache length with it). */
int seedmi_add_i32(void* counter_dev, int delta, void* stream);
/* dst[i] += inc[i], i < n (the continuous-batching step advances the cache length of every ACTIVE slot with it). */
int seedmi_add_i32_vec(void* dst_i32, const void* inc_i32, int n, void* stream);
/* Decode step (T == 1) of LlamaAttention.forward with apply_rotary_pos_emb, the cache append and the attention in one launch
 * (llama_xformer.py:147-168, 228-256): qkv [B, 3*H*hd] (q|k|v) un-rotated; the new key/value row is written to the caches at
 * past_len (or *past_len_dev) and attended to together with the cached rows.  pos_ids_i64 [B] may be NULL (position = past). */
int seedmi_llama_decode_attention_bf16(const void* qkv, int ldqkv, const void* pos_ids_i64, const void* cos_t, const void* sin_t,
                                       void* k_cache, void* v_cache, void* out, int ldo, int B, int H, int hd, int tmax,
                                       int past_len, float scale, int out_packed, const void* past_len_dev, int max_pos,
                                       void* stream);
/* xformers.ops.memory_efficient_attention semantics (llama_xformer.py:244-256), head_dim 128:
 * q [B*T, H*hd]; caches [B][H][tmax][hd] holding kv_len = past_len + T keys; causal (top-left aligned on the
 * last T positions) when T > 1. */
int seedmi_llama_attention_bf16(const void* q, int ldq, const void* k_cache, const void* v_cache, void* out, int ldo,
                                int B, int T, int H, int hd, int tmax, int past_len, float scale, int out_packed,
                                const void* past_len_dev, void* stream);
/* Skinny GEMM for decode (M <= 64): same contract as seedmi_gemm_bf16 restricted to NONE/BIAS_RESIDUAL/SWIGLU. */
int seedmi_gemm_skinny_bf16(int M, int N, int K, const void* A, int lda, const void* W, int ldw, const void* residual,
                            int ldr, int epilogue, void* C, int ldc, void* stream);
/* Decode GEMM with the RMSNorm in front of it folded in (llama_xformer.py:105-113 + the nn.Linear that follows): A holds the
 * UN-normalised rows (fragment-major when a_packed), W_packed holds weight * gamma; the kernel accumulates the rows' sums of squares
 * from the activation fragments it streams and scales its fp32 accumulators by rsqrt(mean(x^2) + rms_eps) before the single bf16
 * rounding.  rms_eps == 0 disables the scaling (plain packed GEMM).  x_packed_out (optional, BIAS_RESIDUAL only) receives a second,
 * fragment-major copy of the result: the residual stream as the next folded GEMM wants it. */
int seedmi_gemm_skinny_norm_bf16(int M, int N, int K, const void* A, int a_packed, const void* W_packed, float rms_eps,
                                 const void* residual, int ldr, int epilogue, void* C, int ldc, int c_packed, void* x_packed_out,
                                 void* stream);
/* The same GEMM in its balanced split-K form (M <= 32, a_packed = 1): 64-row weight tiles, the (tile, k-step) space cut into one equal
 * contiguous range per resident workgroup, cut tiles summed in a fixed workgroup order through `workspace`
 * (seedmi_gemm_skinny_workspace_bytes() bytes, 256-byte aligned).  The first 4 KiB of the workspace are flag words: ZERO them once
 * after allocation (hipMemset); every launch leaves them zero again, and word 1023 is a sticky error word (non-zero = a launch gave
 * up waiting for a partner workgroup: its result is wrong - only possible when the launch could not be fully resident).  One workspace
 * serves any number of launches ON ONE STREAM (launches that may overlap need one each).  A shape that divides into whole tiles per
 * workgroup (16-row tiles a multiple of the CU count, or of three times it) runs uncut: no hand-off, the workspace untouched.
 * workspace == NULL, M > 32, a_packed == 0 or seedmi_set_option("skinny_splitk", 0) select the kernel of seedmi_gemm_skinny_norm_bf16;
 * all forms give the same values up to the order of the fp32 K summation. */
size_t seedmi_gemm_skinny_workspace_bytes(void);
/* Once after allocation (instead of the hipMemset; stream-ordered, no synchronisation): zeroes the flag words and the error word and writes
 * a tag into word 1022, by which seedmi_gemm_skinny_ws_status tells an initialised workspace from uninitialised memory. */
int seedmi_gemm_skinny_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
/* Reads (and, once reported, clears) the sticky error word of such a workspace: SEEDMI_OK, SEEDMI_E_HIP with the workgroup that gave up
 * in seedmi_last_error(), or SEEDMI_E_SHAPE for a workspace that never saw seedmi_gemm_skinny_workspace_init (no tag).  SYNCHRONISES `stream` (one 4-byte copy to the host): call it after a decode loop, not inside one. */
int seedmi_gemm_skinny_ws_status(void* workspace, size_t workspace_bytes, void* stream);
int seedmi_gemm_skinny_norm_ws_bf16(int M, int N, int K, const void* A, int a_packed, const void* W_packed, float rms_eps,
                                    const void* residual, int ldr, int epilogue, void* C, int ldc, int c_packed, void* x_packed_out,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* rows [rows, cols] (row stride ldx) -> the fragment-major activation layout, no arithmetic. */
int seedmi_pack_activations_bf16(const void* x, int ldx, void* out_packed, int rows, int cols, void* stream);
/* The same GEMM on fragment-major weights: tile (16 rows) x k-step (32) blocks of 1 KiB laid out in the MFMA operand's
 * lane order, so every wave streams one contiguous region of HBM (row-major weights put the 16 rows of a load on the
 * same channels: 2.3 vs > 4 TB/s).  Pack once per weight with seedmi_pack_skinny_weights. */
size_t seedmi_pack_skinny_weights_bytes(int N, int K);
int seedmi_pack_skinny_weights(const void* W, int ldw, int N, int K, void* out, void* stream);
int seedmi_gemm_skinny_packed_bf16(int M, int N, int K, const void* A, int lda, const void* W_packed, const void* residual,
                                   int ldr, int epilogue, void* C, int ldc, int a_packed, int c_packed, void* stream);

/* ---- path level: SEED-2 tokenizer ------------------------------------------------------------------------------ */
typedef struct {
    const void *ln1_w, *ln1_b;      /* blocks.N.norm1                                   */
    const void *qkv_w, *qkv_b;      /* attn.qkv.weight [3D,D]; bias = cat(q_bias,0,v_bias) */
    const void *proj_w, *proj_b;    /* attn.proj                                        */
    const void *ln2_w, *ln2_b;      /* norm2                                            */
    const void *fc1_w, *fc1_b;      /* mlp.fc1 [F,D]                                    */
    const void *fc2_w, *fc2_b;      /* mlp.fc2 [D,F]                                    */
    /* optional LayerNorm fold (all six or none): half(qkv.weight * norm1.weight), its fp32 row sums, fp32 qkv bias + qkv.weight norm1.bias;
     * the same for fc1 with norm2.  NULL -> explicit LayerNorm launches. */
    const void *qkv_wg, *qkv_cs, *qkv_bf;
    const void *fc1_wg, *fc1_cs, *fc1_bf;
} seedmi_vit_layer_t;

typedef struct {
    const void *qkv_w, *qkv_b;      /* attention.self.{query,key,value} stacked [3Q,Q]  */
    const void *ao_w, *ao_b, *ao_ln_w, *ao_ln_b;          /* attention.output            */
    int has_cross;
    const void *cq_w, *cq_b;        /* crossattention.self.query [Q,Q]                  */
    const void *ckv_w, *ckv_b;      /* crossattention.self.{key,value} stacked [2Q,D]   */
    const void *co_w, *co_b, *co_ln_w, *co_ln_b;          /* crossattention.output       */
    const void *ffn_w1, *ffn_b1;    /* intermediate_query.dense [FF,Q]                  */
    const void *ffn_w2, *ffn_b2, *ffn_ln_w, *ffn_ln_b;    /* output_query                */
} seedmi_qf_layer_t;

typedef struct {
    int img_size, patch, vit_dim, vit_depth, vit_heads, vit_ffn;
    int qf_dim, qf_layers, qf_heads, qf_ffn, n_query, n_embed, code_dim;
    int kpad;                        /* patch-embed K (3*patch*patch) zero padded to a multiple of 64 */
    const void *patch_w, *patch_b;   /* [D, kpad], [D]                                    */
    const void* pos_embed;           /* [n_tokens, D]                                     */
    const void* cls_pos0;            /* [D] = half(cls_token + pos_embed[0])              */
    const seedmi_vit_layer_t* vit;   /* host array [vit_depth]                            */
    const void *ln_vision_w, *ln_vision_b;
    const void* query_ln;            /* [n_query, Q] = embeddings.LayerNorm(query_tokens), input independent */
    const seedmi_qf_layer_t* qf;     /* host array [qf_layers]                            */
    const void *head_w0, *head_b0;   /* encode_task_layer.0 [Q,Q]                         */
    const void *head_w1, *head_b1;   /* encode_task_layer.2 [code_dim,Q]                  */
    const void* codebook;            /* quantize.embedding.weight [n_embed, code_dim]     */
    const void* code_sqnorm;         /* fp32 [n_embed] from seedmi_vq_code_sqnorm         */
} seedmi_tokenizer_weights_t;

typedef struct {                     /* optional device outputs for parity checks; any may be NULL */
    void* image_embeds;              /* [B*n_tokens, D] bf16  (ln_vision output)          */
    void* qformer_out;               /* [B*n_query, Q] bf16                               */
    void* z;                         /* [B*n_query, code_dim] bf16                        */
} seedmi_tokenizer_taps_t;

/* Sized for any number of sub-batches (see below), so one allocation serves every setting. */
size_t seedmi_tokenize_workspace_bytes(const seedmi_tokenizer_weights_t* w, int batch);
/* Blip2QformerQuantizer.get_codebook_indices + ImageTokenizer.encode (qformer_quantizer.py:288-307,
 * seed_llama_tokenizer.py:75-90): images [B,3,S,S] (fp32 or bf16) -> ids int64 [B, n_query] in [0, n_embed). */
int seedmi_tokenize(const seedmi_tokenizer_weights_t* w, const void* images, int images_fp32, int batch, void* ids_i64,
                    const seedmi_tokenizer_taps_t* taps, void* workspace, size_t workspace_bytes, void* stream);
/* Sub-batch overlap and the ONE piece of state this library owns.  A batch >= 32 is split into seedmi_set_option("tokenize_streams")
 * sub-batches (default 2) that run on side streams forked from / joined into `stream` by events, so that one sub-batch's kernels fill the
 * partial last round of the other's GEMMs (+5 %); to the caller the call is still ordered on `stream` alone.  seedmi_tokenize creates those
 * objects itself at first use - per calling thread and device up to 3 non-blocking streams and 4 events, kept until the thread exits -
 * which is the exception to "nothing is created inside the library".  A caller that wants none of it either sets "tokenize_streams" to 1
 * (everything on `stream`) or passes its OWN objects through seedmi_tokenize_fj: hipStream_t side_stream[n_side] (any streams of the
 * device other than `stream`), hipEvent_t fork_event and join_event[n_side] (hipEventDisableTiming is enough), n_side = 0..3 (the
 * number of sub-batches is n_side + 1 whatever the option says; a batch below 32 is never split).  The objects must not be in use by a
 * concurrent call.  fj == NULL is seedmi_tokenize. */
typedef struct {
    void* side_stream[3];
    void* fork_event;
    void* join_event[3];
    int n_side;
} seedmi_fork_join_t;
int seedmi_tokenize_fj(const seedmi_tokenizer_weights_t* w, const void* images, int images_fp32, int batch, void* ids_i64,
                       const seedmi_tokenizer_taps_t* taps, void* workspace, size_t workspace_bytes, const seedmi_fork_join_t* fj,
                       void* stream);

/* ---- the step after the path: next-token selection ------------------------------------------------------------------- */
/* Greedy argmax (uniforms_f32 == NULL or top_p == 0; first index on ties) or temperature + top-p sampling of one token per row
 * of logits bf16 [batch, ldl] (columns >= vocab ignored), the logits-processor + multinomial part of GenerationMixin.sample for
 * the scripts' generation_config (scripts/seed_llama_inference_8B.py:81-87): weights exp((l - max) / temperature); a token is
 * kept iff the weight mass ranked before it (descending weight, ties by ascending id) is < top_p * total; the draw is the
 * inverse CDF over the kept tokens in that order at uniforms_f32[step * batch + row] (in [0,1)), step = *step_dev (device int32,
 * may be NULL = 0) + step_offset.  Writes tok_out_i64[row] and, when history_i64 != NULL, history_i64[row * history_ld + step].
 * n_steps > 0 bounds the step: beyond it the last uniforms row is reused and nothing is recorded (a graph replayed too often
 * must not leave its buffers).  No host sync: a captured decode graph replays it with the step read from device memory. */
int seedmi_sample_token_bf16(const void* logits, int ldl, int batch, int vocab, float temperature, float top_p,
                             const void* uniforms_f32, const void* step_dev, int step_offset, void* tok_out_i64,
                             void* history_i64, int history_ld, int n_steps, void* stream);

/* ---- the step before the path: image pre-processing -------------------------------------------------------------- */
#define SEEDMI_RESIZE_BILINEAR 2   /* PIL.Image.BILINEAR: transforms.Resize default, models/transforms.py:13,16      */
#define SEEDMI_RESIZE_BICUBIC 3    /* PIL.Image.BICUBIC: interpolation=3, models/seed_llama_tokenizer.py:51           */
size_t seedmi_preprocess_workspace_bytes(int in_h, int in_w, int resize_h, int resize_w, int filter);
/* transforms.Resize -> [CenterCrop] -> ToTensor -> Normalize on one uint8 RGB image (models/seed_llama_tokenizer.py:50-56,
 * models/transforms.py:8-21).  rgb_hwc: device uint8 [in_h][row_stride bytes] with 3-byte pixels; the image is resized to
 * resize_h x resize_w with Pillow's antialiased 8-bit resampler (bit-exact: same fixed-point coefficients and the uint8
 * rounding after each of the two passes), the window [crop_top, +out_h) x [crop_left, +out_w) of the result is taken,
 * divided by 255 and normalised with mean3/std3 (host pointers) in fp32, and written as [3][out_h][out_w] fp32 or bf16.
 * out_u8_hwc (optional) receives the resized+cropped uint8 image [out_h][out_w][3] for parity checks. */
int seedmi_preprocess_image_u8(const void* rgb_hwc, int in_h, int in_w, int row_stride, int resize_h, int resize_w, int filter,
                               int crop_top, int crop_left, int out_h, int out_w, const float* mean3, const float* std3,
                               void* out_chw, int out_is_fp32, void* out_u8_hwc, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ---- path level: de-tokenizer front half ------------------------------------------------------------------------- */
typedef struct {
    int n_embed, code_dim, code_pad;  /* code_pad = code_dim zero padded to a multiple of 64 (GEMM K granularity)   */
    int dim, heads, ffn, depth;       /* blocks_image: vit.Block(768, 12 heads, mlp 4.0) x decode_depth              */
    int n_query, down1, down2, down3, out_dim;   /* image_down widths (256, 128, 32); distill output (1024)          */
    const void* codebook_pad;         /* quantize.embedding.weight [n_embed, code_pad] (columns >= code_dim zero)    */
    const void *dec_w0, *dec_b0;      /* decode_task_layer.0 padded to [code_pad, code_pad], [code_pad]              */
    const void *dec_w1, *dec_b1;      /* decode_task_layer.2 [dim, code_pad] (K zero padded), [dim]                  */
    const void* pos_embed_image;      /* [n_query, dim]                                                              */
    const seedmi_vit_layer_t* blocks; /* host array [depth]: blocks_image.N (qkv bias is the module's own [3*dim])    */
    const void *down_w0, *down_w1, *down_w2;     /* image_down.{0,2,4}.weight [down1,dim] [down2,down1] [down3,down2] */
    const void *distill_w, *distill_b;           /* distill_image_proj [out_dim, n_query*down3], [out_dim]           */
} seedmi_detok_weights_t;

size_t seedmi_detokenize_workspace_bytes(const seedmi_detok_weights_t* w, int batch);
/* Blip2QformerQuantizer.get_codebook_entry (qformer_quantizer.py:309-338, use_qformer_image = False branch) as called by
 * ImageTokenizer.decode (seed_llama_tokenizer.py:92-100): ids int64 [B, n_query] -> unCLIP image embeds bf16 [B, out_dim]
 * (the conditioning handed to the diffusers pipeline, which stays outside this library).  hidden (optional) receives the
 * blocks_image output [B*n_query, dim] for parity checks. */
int seedmi_detokenize(const seedmi_detok_weights_t* w, const void* ids_i64, int batch, void* embeds, void* hidden,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- path level: LLaMA forward --------------------------------------------------------------------------------- */
typedef struct {
    const void *ln1_w;               /* input_layernorm                                   */
    const void *qkv_w;               /* {q,k,v}_proj stacked [3h,h]                       */
    const void *o_w;                 /* o_proj [h,h]                                      */
    const void *ln2_w;               /* post_attention_layernorm                          */
    const void *gate_up_w;           /* gate_proj/up_proj row-interleaved [2F,h]          */
    const void *down_w;              /* down_proj [h,F]                                   */
    void *k_cache, *v_cache;         /* [B][H][tmax][hd] bf16, caller owned               */
    /* optional fragment-major copies for the decode (M <= 64) path; NULL = stream the row-major weights */
    const void *qkv_wp, *o_wp, *gate_up_wp, *down_wp;
} seedmi_llama_layer_t;

typedef struct {
    int hidden, layers, heads, ffn, vocab, vocab_pad, max_pos, tmax, batch_cap;
    float rms_eps;
    const void* embed;               /* embed_tokens [vocab, h]                           */
    const seedmi_llama_layer_t* layer; /* host array [layers]                             */
    const void* norm_w;              /* model.norm                                        */
    const void* lm_head;             /* [vocab_pad, h] (rows >= vocab zero)               */
    const void *cos_t, *sin_t;       /* [max_pos, hd] bf16                                */
    const void* lm_head_p;           /* optional fragment-major lm_head (vocab_pad rows)  */
    int norm_folded;                 /* 1: qkv_wp / gate_up_wp / lm_head_p hold weight * gamma of the RMSNorm in front of them (input_layernorm,
                                      * post_attention_layernorm, model.norm): decode steps then run without norm launches */
} seedmi_llama_weights_t;

/* The first seedmi_gemm_skinny_workspace_bytes() bytes of a workspace (whatever batch and T it was sized for) are the split-K area of
 * seedmi_gemm_skinny_norm_ws_bf16, used by decode steps (T == 1, batch <= 32).  What must be initialised is ONLY that area's 1022 flag
 * words, the tag and the sticky error word - seedmi_llama_workspace_init does it (or a hipMemset of the whole workspace followed by it);
 * every other byte may hold anything, NaN / Inf bit patterns included: each region (activation images, their padded fragment rows beyond
 * the batch, partial-tile images, logits staging) is written by the step before it is read, and padded rows are never reduced into real
 * ones (tests/test_gpu_llama.py::test_llama_workspace_needs_only_its_flag_area_initialised poisons the workspace with 0xFF first).
 * Every step clears its hand-off flags itself, the sticky error word (32-bit word 1023) is only ever cleared by
 * seedmi_llama_decode_status; prefills never touch the area, so one workspace may serve prefills and decode steps alike. */
size_t seedmi_llama_workspace_bytes(const seedmi_llama_weights_t* w, int batch, int T);
/* Once after allocating a llama workspace of any (batch, T) (stream-ordered): seedmi_gemm_skinny_workspace_init on its split-K area - the
 * only initialisation a workspace needs (see above).  A workspace that skips it computes correctly only from all-zero memory, and
 * seedmi_llama_decode_status refuses it (SEEDMI_E_SHAPE). */
int seedmi_llama_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
/* LlamaForCausalLM.forward, eval, use_cache (llama_xformer.py:661-743): ids/pos int64 [B,T]; appends to the static
 * KV cache at past_len; logits bf16 [B*T_out, ldl] where T_out = T (all positions, reference behaviour) or 1
 * (last position only, decode fast path) depending on last_only. */
int seedmi_llama_forward(const seedmi_llama_weights_t* w, const void* ids_i64, const void* pos_i64, int batch, int T,
                         int past_len, int last_only, void* logits, int ldl, void* workspace, size_t workspace_bytes,
                         void* stream);
/* Same with the cache length read from device memory (int32*) by the kernels that need it (T must be 1; positions are
 * taken as *past_len_dev when pos_i64 is NULL): no launch argument changes from step to step, so one captured
 * hipGraph of this call replays the whole decode loop (the reference pays a host round trip per layer per step,
 * llama_xformer.py:255). past_len is still needed on the host for capacity checks (pass the current upper bound). */
int seedmi_llama_forward_ex(const seedmi_llama_weights_t* w, const void* ids_i64, const void* pos_i64, int batch, int T,
                            int past_len, const void* past_len_dev, int last_only, void* logits, int ldl, void* workspace,
                            size_t workspace_bytes, void* stream);
/* LlamaModel.forward's other two surfaces (llama_xformer.py:502-541, 569-570, 613-617): exactly one of ids_i64 and
 * inputs_embeds (bf16 [batch*T, hidden]) is given; hidden_states (optional, bf16 [(layers+1)][batch*T][hidden], last_only must be
 * 0) receives the input of every decoder layer and, last, the final-norm output - the tuple output_hidden_states=True returns. */
int seedmi_llama_forward_io(const seedmi_llama_weights_t* w, const void* ids_i64, const void* inputs_embeds, const void* pos_i64,
                            int batch, int T, int past_len, const void* past_len_dev, int last_only, void* logits, int ldl,
                            void* hidden_states, void* workspace, size_t workspace_bytes, void* stream);
/* One decode step over `batch` independent slots of the static KV cache, slot b at its own cache length lens_i32[b] (device int32):
 * the step a continuous-batching serving loop replays (gradio_demo/seed_llama_flask.py:166-174 generates one request at a time; the
 * survey's f4 row asks for batching on top).  tok_i64 [batch] are the slots' current tokens; logits bf16 [batch, ldl] (last
 * position of every slot).  Rows are independent: an idle slot (length 0, any token) costs its share of the weight stream only. */
int seedmi_llama_decode_slots(const seedmi_llama_weights_t* w, const void* tok_i64, const void* lens_i32, int batch, void* logits,
                              int ldl, void* workspace, size_t workspace_bytes, void* stream);
/* seedmi_gemm_skinny_ws_status for the split-K area inside a decode workspace (the one passed to the T = 1 forward / decode_slots calls
 * of this batch): SEEDMI_OK when every decode step since the last check completed its cut tiles, SEEDMI_E_HIP otherwise (the logits of
 * those steps are then invalid).  SYNCHRONISES `stream`; a decode loop calls it once at its end (seed_amd/llama_engine.py does). */
int seedmi_llama_decode_status(const seedmi_llama_weights_t* w, int batch, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEEDMI_H */
