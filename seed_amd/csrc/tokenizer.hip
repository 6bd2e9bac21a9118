// Path-level entry: SEED-2 image tokenizer forward (EVA-ViT-g -> ln_vision -> causal Q-Former -> task MLP -> VQ).
//
// Host-side orchestration of the kernel-level C ABI; mirrors, step for step,
//   Blip2QformerQuantizer.get_codebook_indices   models/seed_qformer/qformer_quantizer.py:288-307
//   VisionTransformer.forward_features            models/seed_qformer/eva_vit.py:369-385
//   BertModel.forward / BertLayer.forward         models/seed_qformer/qformer_causual.py:769-931, 359-444
//
// The batch is a pure map over images, so large batches are split into two sub-batches that run the same kernel
// sequence on two HIP streams (fork/join with events on the caller's stream).  With one 128 KiB-LDS GEMM workgroup
// per CU a single stream leaves the chip partly idle in every kernel's last round of tiles and during the
// HBM-bound LayerNorm / attention kernels; the second stream's workgroups fill those holes.  Nothing is allocated
// per call (streams/events are created once), there is no host sync, and fork/join by events is graph-capturable.
#include <atomic>
#include "common.h"
#include "seedmi_internal.h"
#include "../../include/seedmi.h"

namespace {

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base((char*)p) {}
    void* take(size_t bytes) {
        void* r = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return r;
    }
};

struct TokWs {
    bf16_t *col, *x, *xn, *qkv, *h, *kv, *qx, *qa, *qt, *qqkv, *qh, *z;
    float *stats, *spart;     // LayerNorm fold: (mean, rstd) per row; per-span (sum, sumsq) partials of the last residual GEMM
    int spans;
    void* sk;                 // stream-K workspace of the big GEMMs (flags in its first 4 KiB, cleared at the start of every call)
    size_t sk_bytes;
    size_t bytes;
};

std::atomic<int> g_tok_streamk{0};                    // seedmi_set_option("tokenize_streamk", 0|1)
std::atomic<int> g_tok_split{0};                      // seedmi_set_option("tokenize_split_rounds", 0|1): whole rounds of 256x256 tiles + a 128x128 remainder
std::atomic<int> g_tok_tilestats{0};                  // seedmi_set_option("tokenize_tile_stats", 0|1): LayerNorm statistics by 256-column tile, finalized inside
                                          // the consuming GEMM (no seedmi_layernorm_stats_finalize launches between the ViT GEMMs)
std::atomic<int> g_tok_vqhead{1};                     // seedmi_set_option("tokenize_vq_head", 0|1): last head Linear fused into the VQ argmin kernel
std::atomic<int> g_tok_lnfold{1};                     // seedmi_set_option("tokenize_lnfold", 0|1): LayerNorm folded into qkv / fc1 when the weights carry it

TokWs carve(const seedmi_tokenizer_weights_t* w, int B, void* ws) {
    const int grid = w->img_size / w->patch;
    const size_t P = (size_t)grid * grid, NT = P + 1;
    const size_t M = (size_t)B * NT, Mq = (size_t)B * w->n_query;
    const size_t D = w->vit_dim, F = w->vit_ffn, Q = w->qf_dim, FF = w->qf_ffn;
    Carver c(ws);
    TokWs t;
    t.col = (bf16_t*)c.take((size_t)B * P * w->kpad * 2);
    t.x = (bf16_t*)c.take(M * D * 2);
    t.xn = (bf16_t*)c.take(M * D * 2);
    t.qkv = (bf16_t*)c.take(M * 3 * D * 2);
    t.h = (bf16_t*)c.take(M * F * 2);
    t.kv = (bf16_t*)c.take(M * 2 * Q * 2);
    t.qx = (bf16_t*)c.take(Mq * Q * 2);
    t.qa = (bf16_t*)c.take(Mq * Q * 2);
    t.qt = (bf16_t*)c.take(Mq * Q * 2);
    t.qqkv = (bf16_t*)c.take(Mq * 3 * Q * 2);
    t.qh = (bf16_t*)c.take(Mq * FF * 2);
    t.z = (bf16_t*)c.take(Mq * 64 * 2);
    t.spans = (int)((D + 63) / 64);
    t.stats = (float*)c.take((M + 1) * 2 * sizeof(float));      // (the fold's LDS-DMA reads row pairs: one row of slack)
    t.spart = (float*)c.take((M + 1) * t.spans * 2 * sizeof(float));          // (+ 1: the by-tile planes keep an even row stride)
    t.sk_bytes = seedmi_gemm_workspace_bytes();
    t.sk = c.take(t.sk_bytes);
    t.bytes = c.off;
    return t;
}

#define CK(call)                          \
    do {                                  \
        const int rc_ = (call);           \
        if (rc_ != SEEDMI_OK) return rc_; \
    } while (0)
#define HIPCK(call)                                                                  \
    do {                                                                             \
        const hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                      \
            seedmi_set_error("seedmi_tokenize: %s: %s", #call, hipGetErrorString(e_)); \
            return SEEDMI_E_HIP;                                                     \
        }                                                                            \
    } while (0)

// One sub-batch: its slice of the inputs/outputs, its own workspace, its stream.
struct Part {
    const seedmi_tokenizer_weights_t* w;
    const char* images;
    int images_fp32;
    int B;
    long long* ids;
    seedmi_tokenizer_taps_t taps;      // already offset to this part's rows (null pointers stay null)
    TokWs t;
    hipStream_t s;
};

// phases: 0 = patch embed; 1..depth = ViT blocks; depth+1 = ln_vision + query expand; then Q-Former layers; last = head + VQ
int n_phases(const seedmi_tokenizer_weights_t* w) { return 1 + w->vit_depth + 1 + w->qf_layers + 1; }

// big GEMMs (M = B * 257 rows): with seedmi_set_option("tokenize_streamk", 1) they take the stream-K workspace.  Off by default:
// measured at B = 256 (tools/tok_ab.py, profiles/r02_tok_ab.json) the two sub-batch streams already fill every kernel's partial last
// round with the other stream's workgroups (134.9 ms per pass), and stream-K's balanced endings take that overlap away (141.5 ms);
// on one stream it is neutral (140.6 vs 140.8 ms).  It pays for isolated GEMM calls (seedmi_gemm_bf16_ws: +2..4 %).
// A big GEMM as whole rounds plus a remainder.  The persistent 256x256 kernel runs ceil(tiles / CUs) rounds; B = 256 images are
// 256 + 1 m-tiles (257 tokens each), so the N = 1408 GEMMs (6 n-tiles) were 1542 tiles = 6.02 rounds: a seventh round on six CUs, 14 %
// of proj / fc2.  When dropping the last few m-tiles saves a whole round, those rows go to a second call, which the library sends to the
// 128x128 kernel (M < 1024): a quarter of a round on 2 x 11 workgroups.  Pointers of the row-indexed operands are offset; results are
// those of the single call (each output element's K chain does not depend on the tile shape; tools/tok_ab.py checks the ids).
// Measured at B = 256 (profiles/r02_tok_ab.json): one stream 131.0 vs 132.6 ms, two streams 129.0 vs 126.6 ms - with two sub-batch
// streams the other stream's kernels already run in the empty part of a last round and the extra launches only cost.  Off by default
// (the default is two streams); seedmi_set_option("tokenize_split_rounds", 1) for single-stream callers.
int gemm_rounds(int M, int N, int K, const bf16_t* A, int lda, const void* W, int ldw, const void* bias, const bf16_t* R, int ldr, int epi,
                bf16_t* C, int ldc, const seedmi_gemm_ext_t* ext, void* sk, size_t skb, void* s) {
    int m_main = M;
    if (g_tok_split && M >= 2048) {
        const int cus = seedmi_device_cus(seedmi_current_device());
        const int tn = (N + 255) / 256, tm = (M + 255) / 256;
        const int rounds = (tm * tn + cus - 1) / cus;
        for (int drop = 1; drop <= 3 && drop < tm; ++drop) {         // rows of the last `drop` m-tiles -> remainder
            const int r2 = ((tm - drop) * tn + cus - 1) / cus;
            if (r2 < rounds && M - (tm - drop) * 256 < 1024) { m_main = (tm - drop) * 256; break; }
        }
    }
    CK(seedmi_gemm_bf16_ext(m_main, N, K, A, lda, W, ldw, bias, R, ldr, epi, C, ldc, 0, 0, ext, sk, skb, s));
    if (m_main < M) {
        seedmi_gemm_ext_t e2 = {nullptr, nullptr, nullptr, nullptr, 0};
        if (ext) {
            e2 = *ext;
            if (e2.ln_stats) e2.ln_stats += 2 * (size_t)m_main;
            if (e2.stats_out) e2.stats_out += 2 * (size_t)m_main;           // (span-major planes: the plane stride stays stats_ld)
        }
        CK(seedmi_gemm_bf16_ext(M - m_main, N, K, A + (size_t)m_main * lda, lda, W, ldw, bias, R ? R + (size_t)m_main * ldr : nullptr, ldr, epi,
                                C + (size_t)m_main * ldc, ldc, 0, 0, ext ? &e2 : nullptr, nullptr, 0, s));
    }
    return SEEDMI_OK;
}

// big GEMMs (M = B * 257 rows): with seedmi_set_option("tokenize_streamk", 1) they take the stream-K workspace.  Off by default:
// measured at B = 256 (tools/tok_ab.py, profiles/r02_tok_ab.json) the two sub-batch streams already fill every kernel's partial last
// round with the other stream's workgroups (134.9 ms per pass), and stream-K's balanced endings take that overlap away (141.5 ms);
// on one stream it is neutral (140.6 vs 140.8 ms).  It pays for isolated GEMM calls (seedmi_gemm_bf16_ws: +2..4 %).
#define GEMM_WS(M_, N_, K_, A_, lda_, W_, ldw_, b_, R_, ldr_, epi_, C_, ldc_, rg_, re_) \
    seedmi_gemm_bf16_ws(M_, N_, K_, A_, lda_, W_, ldw_, b_, R_, ldr_, epi_, C_, ldc_, rg_, re_, g_tok_streamk ? t.sk : nullptr, \
                        g_tok_streamk ? t.sk_bytes : 0, s)
#define GEMM_R(M_, N_, K_, A_, lda_, W_, ldw_, b_, R_, ldr_, epi_, C_, ldc_, ext_) \
    gemm_rounds(M_, N_, K_, A_, lda_, W_, ldw_, b_, R_, ldr_, epi_, C_, ldc_, ext_, g_tok_streamk ? t.sk : nullptr, \
                g_tok_streamk ? t.sk_bytes : 0, s)

int run_phase(const Part& p, int phase) {
    const seedmi_tokenizer_weights_t* w = p.w;
    const TokWs& t = p.t;
    void* s = (void*)p.s;
    const int B = p.B;
    const int grid = w->img_size / w->patch;
    const int P = grid * grid, NT = P + 1;
    const int M = B * NT, Mq = B * w->n_query;
    const int D = w->vit_dim, F = w->vit_ffn, H = w->vit_heads, hd = D / H;
    const int Q = w->qf_dim, FF = w->qf_ffn, QH = w->qf_heads, qhd = Q / QH;
    const int nq = w->n_query;

    if (phase == 0) {
        // patch embed: unfold -> GEMM(+conv bias, +pos_embed, rows shifted past the cls slot); cls rows
        CK(seedmi_im2col_patch(p.images, p.images_fp32, t.col, B, 3, w->img_size, w->patch, w->kpad, s));
        CK(GEMM_WS(B * P, D, w->kpad, t.col, w->kpad, w->patch_w, w->kpad, w->patch_b, w->pos_embed, D,
                   SEEDMI_EPI_PATCH_EMBED, t.x, D, P, 1));
        CK(seedmi_fill_rows(t.x, D, NT, 0, B, w->cls_pos0, D, 1, D, s));
        if (g_tok_lnfold && w->vit_depth > 0 && w->vit[0].qkv_wg)      // statistics of block 0's norm1 (later ones come from the GEMMs)
            CK(seedmi_layernorm_stats_bf16(t.x, D, M, D, 1e-6f, t.stats, s));
        return SEEDMI_OK;
    }
    phase -= 1;
    if (phase < w->vit_depth) {                                             // Block.forward (eva_vit.py:199-202)
        const seedmi_vit_layer_t& L = w->vit[phase];
        const float vit_scale = 1.0f / sqrtf((float)hd);
        if (g_tok_lnfold && L.qkv_wg) {
            // LayerNorm folded into the GEMMs around it: qkv / fc1 read the residual stream itself (weights carry gamma, the epilogue
            // applies mean / rstd / beta), proj / fc2 emit the row statistics of what they write.  78 LayerNorm passes over the token
            // stream (read + write 0.37 GB each at B = 256) become 78 reductions of 22 partials per row.
            seedmi_gemm_ext_t e_qkv = {t.stats, (const float*)L.qkv_cs, (const float*)L.qkv_bf, nullptr, 0};
            seedmi_gemm_ext_t e_fc1 = {t.stats, (const float*)L.fc1_cs, (const float*)L.fc1_bf, nullptr, 0};
            seedmi_gemm_ext_t e_res = {nullptr, nullptr, nullptr, t.spart, (int)M};
            // Statistics by tile: where all four GEMMs of the block run on the 256x256 kernel (large sub-batches), proj / fc2 write one
            // (sum, sum of squares) pair per row and n-tile (6 planes for D = 1408 instead of 22 span planes) and qkv / fc1 finalize the rows
            // of each of their tiles themselves: the two seedmi_layernorm_stats_finalize launches per block - a 65 792-thread kernel queued
            // behind the other stream's GEMM workgroups, on the critical path between two GEMMs of this stream - disappear.
            // (Block 0's norm1 statistics come finished from the patch-embed phase.)
            const bool by_tile = g_tok_tilestats && !g_tok_split && (D + 255) / 256 <= 6 && seedmi_gemm_tile_stats_supported((int)M, D) &&
                                 seedmi_gemm_tile_stats_supported((int)M, 3 * D) && seedmi_gemm_tile_stats_supported((int)M, F);
            if (by_tile) {
                const int ld = ((int)M + 1) & ~1, planes = (D + 255) / 256;
                e_res.stats_ld = ld;
                e_res.stats_by_tile = 1;
                seedmi_gemm_ext_t e_par = {t.spart, nullptr, nullptr, nullptr, 0, 0, planes, ld, D, 1e-6f};
                e_fc1 = e_par; e_fc1.ln_colsum = (const float*)L.fc1_cs; e_fc1.bias_f32 = (const float*)L.fc1_bf;
                if (phase > 0) { e_qkv = e_par; e_qkv.ln_colsum = (const float*)L.qkv_cs; e_qkv.bias_f32 = (const float*)L.qkv_bf; }
            }
            // Small sub-batches (one image: M = 257): none of the block's GEMMs runs on the 256x256 kernel, and the consumers (64x64 / 128x128
            // kernels) finish the row statistics in their own epilogue from the producers' span planes - same sums, same order, same finishing
            // expression as the finalize kernel, so the same bits - and the two finalize launches per block disappear.
            const bool in_consumer = !by_tile && !seedmi_gemm_tile_stats_supported((int)M, 3 * D) && !seedmi_gemm_tile_stats_supported((int)M, F) &&
                                     !seedmi_gemm_tile_stats_supported((int)M, D) && t.spans <= 64;
            if (in_consumer) {
                seedmi_gemm_ext_t e_par = {t.spart, nullptr, nullptr, nullptr, 0, 0, t.spans, (int)M, D, 1e-6f};
                e_fc1 = e_par; e_fc1.ln_colsum = (const float*)L.fc1_cs; e_fc1.bias_f32 = (const float*)L.fc1_bf;
                if (phase > 0) { e_qkv = e_par; e_qkv.ln_colsum = (const float*)L.qkv_cs; e_qkv.bias_f32 = (const float*)L.qkv_bf; }
            }
            CK(GEMM_R(M, 3 * D, D, t.x, D, L.qkv_wg, D, nullptr, nullptr, 0, SEEDMI_EPI_BIAS, t.qkv, 3 * D, &e_qkv));
            CK(seedmi_attention_bf16(t.qkv, 3 * D, t.qkv + D, 3 * D, t.qkv + 2 * D, 3 * D, t.xn, D, B, H, hd, NT, NT,
                                     vit_scale, 0, 1, s));
            CK(GEMM_R(M, D, D, t.xn, D, L.proj_w, D, L.proj_b, t.x, D, SEEDMI_EPI_BIAS_RESIDUAL, t.x, D, &e_res));
            if (!by_tile && !in_consumer) CK(seedmi_layernorm_stats_finalize(t.spart, t.spans, (int)M, (int)M, D, 1e-6f, t.stats, s));
            CK(GEMM_R(M, F, D, t.x, D, L.fc1_wg, D, nullptr, nullptr, 0, SEEDMI_EPI_BIAS_GELU, t.h, F, &e_fc1));
            CK(GEMM_R(M, D, F, t.h, F, L.fc2_w, F, L.fc2_b, t.x, D, SEEDMI_EPI_BIAS_RESIDUAL, t.x, D, &e_res));
            if (!by_tile && !in_consumer && phase + 1 < w->vit_depth) CK(seedmi_layernorm_stats_finalize(t.spart, t.spans, (int)M, (int)M, D, 1e-6f, t.stats, s));
            return SEEDMI_OK;
        }
        CK(seedmi_layernorm_bf16(t.x, D, L.ln1_w, L.ln1_b, 1e-6f, t.xn, D, M, D, s));
        CK(GEMM_R(M, 3 * D, D, t.xn, D, L.qkv_w, D, L.qkv_b, nullptr, 0, SEEDMI_EPI_BIAS, t.qkv, 3 * D, nullptr));
        CK(seedmi_attention_bf16(t.qkv, 3 * D, t.qkv + D, 3 * D, t.qkv + 2 * D, 3 * D, t.xn, D, B, H, hd, NT, NT,
                                 vit_scale, 0, 1, s));
        CK(GEMM_R(M, D, D, t.xn, D, L.proj_w, D, L.proj_b, t.x, D, SEEDMI_EPI_BIAS_RESIDUAL, t.x, D, nullptr));
        CK(seedmi_layernorm_bf16(t.x, D, L.ln2_w, L.ln2_b, 1e-6f, t.xn, D, M, D, s));
        CK(GEMM_R(M, F, D, t.xn, D, L.fc1_w, D, L.fc1_b, nullptr, 0, SEEDMI_EPI_BIAS_GELU, t.h, F, nullptr));
        CK(GEMM_R(M, D, F, t.h, F, L.fc2_w, F, L.fc2_b, t.x, D, SEEDMI_EPI_BIAS_RESIDUAL, t.x, D, nullptr));
        return SEEDMI_OK;
    }
    phase -= w->vit_depth;
    if (phase == 0) {
        // ln_vision (blip2.py:179-184) -> image_embeds in xn; queries = LayerNorm(query_tokens) expanded over the batch
        CK(seedmi_layernorm_bf16(t.x, D, w->ln_vision_w, w->ln_vision_b, 1e-5f, t.xn, D, M, D, s));
        if (p.taps.image_embeds) HIPCK(hipMemcpyAsync(p.taps.image_embeds, t.xn, (size_t)M * D * 2, hipMemcpyDeviceToDevice, p.s));
        CK(seedmi_fill_rows(t.qx, Q, nq, 0, B, w->query_ln, Q, nq, Q, s));
        return SEEDMI_OK;
    }
    phase -= 1;
    if (phase < w->qf_layers) {                                             // BertLayer.forward (qformer_causual.py:359-434)
        const seedmi_qf_layer_t& L = w->qf[phase];
        const float q_scale = 1.0f / sqrtf((float)qhd);
        // causal self-attention on the 32 queries + BertSelfOutput
        CK(seedmi_gemm_bf16(Mq, 3 * Q, Q, t.qx, Q, L.qkv_w, Q, L.qkv_b, nullptr, 0, SEEDMI_EPI_BIAS, t.qqkv, 3 * Q, 0, 0, s));
        CK(seedmi_attention_bf16(t.qqkv, 3 * Q, t.qqkv + Q, 3 * Q, t.qqkv + 2 * Q, 3 * Q, t.qa, Q, B, QH, qhd, nq, nq,
                                 q_scale, 1, 1, s));
        CK(seedmi_gemm_bf16(Mq, Q, Q, t.qa, Q, L.ao_w, Q, L.ao_b, t.qx, Q, SEEDMI_EPI_BIAS_RESIDUAL, t.qt, Q, 0, 0, s));
        CK(seedmi_layernorm_bf16(t.qt, Q, L.ao_ln_w, L.ao_ln_b, 1e-12f, t.qx, Q, Mq, Q, s));
        if (L.has_cross) {
            CK(seedmi_gemm_bf16(Mq, Q, Q, t.qx, Q, L.cq_w, Q, L.cq_b, nullptr, 0, SEEDMI_EPI_BIAS, t.qa, Q, 0, 0, s));
            CK(GEMM_WS(M, 2 * Q, D, t.xn, D, L.ckv_w, D, L.ckv_b, nullptr, 0, SEEDMI_EPI_BIAS, t.kv, 2 * Q, 0, 0));
            CK(seedmi_attention_bf16(t.qa, Q, t.kv, 2 * Q, t.kv + Q, 2 * Q, t.qt, Q, B, QH, qhd, nq, NT, q_scale, 0, 1, s));
            CK(seedmi_gemm_bf16(Mq, Q, Q, t.qt, Q, L.co_w, Q, L.co_b, t.qx, Q, SEEDMI_EPI_BIAS_RESIDUAL, t.qa, Q, 0, 0, s));
            CK(seedmi_layernorm_bf16(t.qa, Q, L.co_ln_w, L.co_ln_b, 1e-12f, t.qx, Q, Mq, Q, s));
        }
        CK(seedmi_gemm_bf16(Mq, FF, Q, t.qx, Q, L.ffn_w1, Q, L.ffn_b1, nullptr, 0, SEEDMI_EPI_BIAS_GELU, t.qh, FF, 0, 0, s));
        CK(seedmi_gemm_bf16(Mq, Q, FF, t.qh, FF, L.ffn_w2, FF, L.ffn_b2, t.qx, Q, SEEDMI_EPI_BIAS_RESIDUAL, t.qt, Q, 0, 0, s));
        CK(seedmi_layernorm_bf16(t.qt, Q, L.ffn_ln_w, L.ffn_ln_b, 1e-12f, t.qx, Q, Mq, Q, s));
        return SEEDMI_OK;
    }
    // encode_task_layer: Linear -> Tanh -> Linear (qformer_quantizer.py:219-223), then the VQ argmin
    const int cd = w->code_dim;
    if (p.taps.qformer_out) HIPCK(hipMemcpyAsync(p.taps.qformer_out, t.qx, (size_t)Mq * Q * 2, hipMemcpyDeviceToDevice, p.s));
    CK(seedmi_gemm_bf16(Mq, Q, Q, t.qx, Q, w->head_w0, Q, w->head_b0, nullptr, 0, SEEDMI_EPI_BIAS_TANH, t.qa, Q, 0, 0, s));
    if (g_tok_vqhead && Q <= 1024 && (Q % 8) == 0) {
        // Linear(768, 32) fused in front of the argmin: z reaches the sweep from LDS (and the tap, if asked for, from the same kernel)
        CK(seedmi_vq_head_argmin_bf16(t.qa, Q, Q, w->head_w1, Q, w->head_b1, w->codebook, w->code_sqnorm, p.ids, p.taps.z, cd, Mq, w->n_embed, cd,
                                      s));
        return SEEDMI_OK;
    }
    CK(seedmi_gemm_bf16(Mq, cd, Q, t.qa, Q, w->head_w1, Q, w->head_b1, nullptr, 0, SEEDMI_EPI_BIAS, t.z, cd, 0, 0, s));
    if (p.taps.z) HIPCK(hipMemcpyAsync(p.taps.z, t.z, (size_t)Mq * cd * 2, hipMemcpyDeviceToDevice, p.s));
    CK(seedmi_vq_argmin_bf16(t.z, cd, w->codebook, w->code_sqnorm, p.ids, Mq, w->n_embed, cd, s));
    return SEEDMI_OK;
}

constexpr int SPLIT_MIN_BATCH = 32;      // below this the kernels are too small for a second stream to help
std::atomic<int> g_tok_streams{2};                    // seedmi_set_option("tokenize_streams", 1|2)

constexpr int MAX_PARTS = 4;
// side streams and fork/join events of the sub-batch overlap: one set per (thread, device), created on first use and kept for the
// life of the thread (a device change no longer leaks the previous device's objects: each device keeps its own slot)
struct ForkJoin {
    hipStream_t side[MAX_PARTS - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[MAX_PARTS - 1] = {nullptr, nullptr, nullptr};
};
struct ForkJoinSet {
    ForkJoin dev[SEEDMI_MAX_DEVICES];
    ~ForkJoinSet() {
        for (ForkJoin& f : dev) {
            if (!f.fork) continue;
            (void)hipEventDestroy(f.fork);
            for (int i = 0; i < MAX_PARTS - 1; ++i) {
                (void)hipEventDestroy(f.join[i]);
                (void)hipStreamDestroy(f.side[i]);
            }
        }
    }
};
thread_local ForkJoinSet g_fj_set;

int ensure_forkjoin(ForkJoin** out) {
    ForkJoin& f = g_fj_set.dev[seedmi_current_device()];
    *out = &f;
    if (f.fork) return SEEDMI_OK;
    HIPCK(hipEventCreateWithFlags(&f.fork, hipEventDisableTiming));
    for (int i = 0; i < MAX_PARTS - 1; ++i) {
        HIPCK(hipStreamCreateWithFlags(&f.side[i], hipStreamNonBlocking));
        HIPCK(hipEventCreateWithFlags(&f.join[i], hipEventDisableTiming));
    }
    return SEEDMI_OK;
}

int n_parts(int batch) {
    const int n = g_tok_streams.load(std::memory_order_relaxed);
    return (n >= 2 && batch >= SPLIT_MIN_BATCH) ? n : 1;
}
int part_size(int batch, int nparts, int i) { return batch / nparts + (i < batch % nparts ? 1 : 0); }

size_t ws_for(const seedmi_tokenizer_weights_t* w, int batch, int nparts) {
    size_t total = 0;
    for (int i = 0; i < nparts; ++i) total += carve(w, part_size(batch, nparts, i), nullptr).bytes;
    return total;
}
size_t total_ws(const seedmi_tokenizer_weights_t* w, int batch) { return ws_for(w, batch, n_parts(batch)); }

}  // namespace

int seedmi_tokenizer_set_streams(int n) {
    if (n < 1 || n > MAX_PARTS) return SEEDMI_E_SHAPE;
    g_tok_streams = n;
    return SEEDMI_OK;
}
int seedmi_tokenizer_set_lnfold(int v) {
    if (v != 0 && v != 1) return SEEDMI_E_SHAPE;
    g_tok_lnfold = v;
    return SEEDMI_OK;
}
int seedmi_tokenizer_set_split(int v) {
    if (v != 0 && v != 1) return SEEDMI_E_SHAPE;
    g_tok_split = v;
    return SEEDMI_OK;
}
int seedmi_tokenizer_set_tilestats(int v) {
    if (v != 0 && v != 1) return SEEDMI_E_SHAPE;
    g_tok_tilestats = v;
    return SEEDMI_OK;
}
int seedmi_tokenizer_set_vqhead(int v) {
    if (v != 0 && v != 1) return SEEDMI_E_SHAPE;
    g_tok_vqhead = v;
    return SEEDMI_OK;
}
int seedmi_tokenizer_set_streamk(int v) {
    if (v != 0 && v != 1) return SEEDMI_E_SHAPE;
    g_tok_streamk = v;
    return SEEDMI_OK;
}

extern "C" size_t seedmi_tokenize_workspace_bytes(const seedmi_tokenizer_weights_t* w, int batch) {
    if (!w || batch <= 0) return 0;
    // sized for either mode so a later seedmi_set_option("tokenize_streams", ...) cannot under-allocate
    size_t best = 0;
    for (int np = 1; np <= MAX_PARTS && np <= batch; ++np) {
        const size_t b = ws_for(w, batch, np);
        if (b > best) best = b;
    }
    return best;
}

extern "C" int seedmi_tokenize(const seedmi_tokenizer_weights_t* w, const void* images, int images_fp32, int batch,
                               void* ids_i64, const seedmi_tokenizer_taps_t* taps, void* workspace,
                               size_t workspace_bytes, void* stream) {
    return seedmi_tokenize_fj(w, images, images_fp32, batch, ids_i64, taps, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int seedmi_tokenize_fj(const seedmi_tokenizer_weights_t* w, const void* images, int images_fp32, int batch,
                                  void* ids_i64, const seedmi_tokenizer_taps_t* taps, void* workspace,
                                  size_t workspace_bytes, const seedmi_fork_join_t* caller_fj, void* stream) {
    if (!w || !images || !ids_i64 || batch <= 0) {
        seedmi_set_error("seedmi_tokenize: null argument or batch=%d", batch);
        return SEEDMI_E_SHAPE;
    }
    if (w->img_size % w->patch || w->vit_dim % w->vit_heads || w->qf_dim % w->qf_heads) {
        seedmi_set_error("seedmi_tokenize: inconsistent dims (img %d / patch %d, D %d / heads %d, Q %d / heads %d)",
                         w->img_size, w->patch, w->vit_dim, w->vit_heads, w->qf_dim, w->qf_heads);
        return SEEDMI_E_SHAPE;
    }
    // caller-owned fork/join objects decide the number of sub-batches themselves (n_side + 1); the library's own set follows the option
    int nparts = n_parts(batch);
    if (caller_fj) {
        if (caller_fj->n_side < 0 || caller_fj->n_side > MAX_PARTS - 1) {
            seedmi_set_error("seedmi_tokenize_fj: n_side=%d (0..%d)", caller_fj->n_side, MAX_PARTS - 1);
            return SEEDMI_E_SHAPE;
        }
        nparts = (batch >= SPLIT_MIN_BATCH) ? caller_fj->n_side + 1 : 1;
        if (nparts > 1 && !caller_fj->fork_event) {
            seedmi_set_error("seedmi_tokenize_fj: fork_event is NULL");
            return SEEDMI_E_SHAPE;
        }
        for (int i = 0; i + 1 < nparts; ++i)
            if (!caller_fj->side_stream[i] || !caller_fj->join_event[i]) {
                seedmi_set_error("seedmi_tokenize_fj: side_stream[%d] / join_event[%d] is NULL", i, i);
                return SEEDMI_E_SHAPE;
            }
    }
    const size_t need = ws_for(w, batch, nparts);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255)) {
        seedmi_set_error("seedmi_tokenize: workspace %zu bytes (need %zu, 256-byte aligned)", workspace_bytes, need);
        return SEEDMI_E_ALIGN;
    }
    const int grid = w->img_size / w->patch;
    const size_t NT = (size_t)grid * grid + 1;
    const size_t img_bytes = (size_t)3 * w->img_size * w->img_size * (images_fp32 ? 4 : 2);
    const bool split = nparts > 1;
    Part parts[MAX_PARTS];
    int b_begin = 0;
    char* ws = (char*)workspace;
    for (int i = 0; i < nparts; ++i) {
        Part& p = parts[i];
        p.w = w;
        p.B = part_size(batch, nparts, i);
        p.images = (const char*)images + (size_t)b_begin * img_bytes;
        p.images_fp32 = images_fp32;
        p.ids = (long long*)ids_i64 + (size_t)b_begin * w->n_query;
        p.taps.image_embeds = (taps && taps->image_embeds) ? (char*)taps->image_embeds + (size_t)b_begin * NT * w->vit_dim * 2 : nullptr;
        p.taps.qformer_out = (taps && taps->qformer_out) ? (char*)taps->qformer_out + (size_t)b_begin * w->n_query * w->qf_dim * 2 : nullptr;
        p.taps.z = (taps && taps->z) ? (char*)taps->z + (size_t)b_begin * w->n_query * w->code_dim * 2 : nullptr;
        p.t = carve(w, p.B, ws);
        ws += p.t.bytes;
        p.s = (hipStream_t)stream;
        b_begin += p.B;
    }
    ForkJoin* fj = nullptr;
    ForkJoin given;
    if (split) {
        if (caller_fj) {
            given.fork = (hipEvent_t)caller_fj->fork_event;
            for (int i = 0; i + 1 < nparts; ++i) {
                given.side[i] = (hipStream_t)caller_fj->side_stream[i];
                given.join[i] = (hipEvent_t)caller_fj->join_event[i];
            }
            fj = &given;
        } else {
            CK(ensure_forkjoin(&fj));
        }
        HIPCK(hipEventRecord(fj->fork, (hipStream_t)stream));
        for (int i = 1; i < nparts; ++i) {
            parts[i].s = fj->side[i - 1];
            HIPCK(hipStreamWaitEvent(fj->side[i - 1], fj->fork, 0));
        }
    }
    if (g_tok_streamk)
        for (int i = 0; i < nparts; ++i)     // stream-K flags: a cleared word never equals a launch's epoch
            HIPCK(hipMemsetAsync(parts[i].t.sk, 0, 4096, parts[i].s));
    const int np = n_phases(w);
    for (int ph = 0; ph < np; ++ph)
        for (int i = 0; i < nparts; ++i) CK(run_phase(parts[i], ph));
    for (int i = 1; i < nparts; ++i) {
        HIPCK(hipEventRecord(fj->join[i - 1], fj->side[i - 1]));
        HIPCK(hipStreamWaitEvent((hipStream_t)stream, fj->join[i - 1], 0));
    }
    return SEEDMI_OK;
}
