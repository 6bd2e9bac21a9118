// Path-level entry: SEED-2 de-tokenizer front half (ids -> unCLIP image embeds), SURVEY.md section 8(f)3.
//
// Host-side orchestration of the kernel-level C ABI; mirrors, step for step,
//   Blip2QformerQuantizer.get_codebook_entry     models/seed_qformer/qformer_quantizer.py:309-338 (use_qformer_image=False)
//   VectorQuantizer2.get_codebook_entry          models/seed_qformer/qformer_quantizer.py:125-140 (nn.Embedding gather)
//   vit.Block / Attention / Mlp                  models/seed_qformer/vit.py:41-47, 85-105, 147-150
// The model is .half()'ed by ImageTokenizer (seed_llama_tokenizer.py:62-63) and this function runs outside autocast, so every
// Linear / LayerNorm / softmax output is a half tensor: the same rounding points are kept here in bf16.
//
// The two 32-wide Linears of decode_task_layer run on the ordinary MFMA GEMM with K zero padded to 64 (the packed codebook
// and weights carry the padding, so padded lanes contribute exact zeros); "+ pos_embed_image" is the PATCH_EMBED epilogue
// with a 32-row period and no row shift.
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

struct DetokWs {
    bf16_t *zq, *t0, *x, *xn, *qkv, *h, *d1, *d2, *d3;
    size_t bytes;
};

DetokWs carve(const seedmi_detok_weights_t* w, int B, void* ws) {
    const size_t R = (size_t)B * w->n_query;
    Carver c(ws);
    DetokWs t;
    t.zq = (bf16_t*)c.take(R * w->code_pad * 2);
    t.t0 = (bf16_t*)c.take(R * w->code_pad * 2);
    t.x = (bf16_t*)c.take(R * w->dim * 2);
    t.xn = (bf16_t*)c.take(R * w->dim * 2);
    t.qkv = (bf16_t*)c.take(R * 3 * w->dim * 2);
    t.h = (bf16_t*)c.take(R * w->ffn * 2);
    t.d1 = (bf16_t*)c.take(R * w->down1 * 2);
    t.d2 = (bf16_t*)c.take(R * w->down2 * 2);
    t.d3 = (bf16_t*)c.take(R * w->down3 * 2);
    t.bytes = c.off;
    return t;
}

#define CK(call)                          \
    do {                                  \
        const int rc_ = (call);           \
        if (rc_ != SEEDMI_OK) return rc_; \
    } while (0)

}  // namespace

extern "C" size_t seedmi_detokenize_workspace_bytes(const seedmi_detok_weights_t* w, int batch) {
    if (!w || batch <= 0) return 0;
    return carve(w, batch, nullptr).bytes;
}

extern "C" int seedmi_detokenize(const seedmi_detok_weights_t* w, const void* ids_i64, int batch, void* embeds, void* hidden,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!w || !ids_i64 || !embeds || batch <= 0) {
        seedmi_set_error("seedmi_detokenize: null argument or batch <= 0");
        return SEEDMI_E_SHAPE;
    }
    if (w->code_pad % 64 || w->code_pad < w->code_dim || w->dim % 64 || w->ffn % 64 || w->down1 % 64 || w->down2 % 64 ||
        (w->n_query * w->down3) % 64 || w->down3 % 8 || w->dim % w->heads || w->n_query > 288 || w->depth < 0) {
        seedmi_set_error("seedmi_detokenize: unsupported dims (code_pad %d dim %d ffn %d down %d/%d/%d n_query %d)", w->code_pad,
                         w->dim, w->ffn, w->down1, w->down2, w->down3, w->n_query);
        return SEEDMI_E_SHAPE;
    }
    const DetokWs t = carve(w, batch, workspace);
    if (!workspace || workspace_bytes < t.bytes) {
        seedmi_set_error("seedmi_detokenize: workspace too small (%zu < %zu)", workspace_bytes, t.bytes);
        return SEEDMI_E_SHAPE;
    }
    void* s = stream;
    const int nq = w->n_query, R = batch * nq, D = w->dim, F = w->ffn, H = w->heads, hd = D / H, cp = w->code_pad;

    // quantize.get_codebook_entry: nn.Embedding gather (qformer_quantizer.py:133)
    CK(seedmi_embed_rows(ids_i64, w->codebook_pad, cp, t.zq, cp, R, cp, w->n_embed, s));
    // decode_task_layer: Linear(32,32) -> Tanh -> Linear(32,768) (qformer_quantizer.py:225-229), then + pos_embed_image (:316-317)
    CK(seedmi_gemm_bf16(R, cp, cp, t.zq, cp, w->dec_w0, cp, w->dec_b0, nullptr, 0, SEEDMI_EPI_BIAS_TANH, t.t0, cp, 0, 0, s));
    CK(seedmi_gemm_bf16(R, D, cp, t.t0, cp, w->dec_w1, cp, w->dec_b1, w->pos_embed_image, D, SEEDMI_EPI_PATCH_EMBED, t.x, D, nq,
                        0, s));
    // blocks_image: x = x + attn(norm1(x)); x = x + mlp(norm2(x))  (vit.py:147-150)
    const float scale = 1.0f / sqrtf((float)hd);
    for (int l = 0; l < w->depth; ++l) {
        const seedmi_vit_layer_t& L = w->blocks[l];
        CK(seedmi_layernorm_bf16(t.x, D, L.ln1_w, L.ln1_b, 1e-6f, t.xn, D, R, D, s));
        CK(seedmi_gemm_bf16(R, 3 * D, D, t.xn, D, L.qkv_w, D, L.qkv_b, nullptr, 0, SEEDMI_EPI_BIAS, t.qkv, 3 * D, 0, 0, s));
        // (q @ k^T) * scale -> softmax -> @ v, all half (vit.py:93-103); 1/sqrt(64) is a power of two, so scaling before or
        // after the half rounding of q k^T gives the same bits
        CK(seedmi_attention_bf16(t.qkv, 3 * D, t.qkv + D, 3 * D, t.qkv + 2 * D, 3 * D, t.xn, D, batch, H, hd, nq, nq, scale, 0,
                                 1, s));
        CK(seedmi_gemm_bf16(R, D, D, t.xn, D, L.proj_w, D, L.proj_b, t.x, D, SEEDMI_EPI_BIAS_RESIDUAL, t.x, D, 0, 0, s));
        CK(seedmi_layernorm_bf16(t.x, D, L.ln2_w, L.ln2_b, 1e-6f, t.xn, D, R, D, s));
        CK(seedmi_gemm_bf16(R, F, D, t.xn, D, L.fc1_w, D, L.fc1_b, nullptr, 0, SEEDMI_EPI_BIAS_GELU, t.h, F, 0, 0, s));
        CK(seedmi_gemm_bf16(R, D, F, t.h, F, L.fc2_w, F, L.fc2_b, t.x, D, SEEDMI_EPI_BIAS_RESIDUAL, t.x, D, 0, 0, s));
    }
    if (hidden) {
        const hipError_t e = hipMemcpyAsync(hidden, t.x, (size_t)R * D * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e != hipSuccess) {
            seedmi_set_error("seedmi_detokenize: hidden copy: %s", hipGetErrorString(e));
            return SEEDMI_E_HIP;
        }
    }
    // image_down: Linear(768,256) ReLU Linear(256,128) ReLU Linear(128,32), all bias-free (qformer_quantizer.py:279-285)
    CK(seedmi_gemm_bf16(R, w->down1, D, t.x, D, w->down_w0, D, nullptr, nullptr, 0, SEEDMI_EPI_RELU, t.d1, w->down1, 0, 0, s));
    CK(seedmi_gemm_bf16(R, w->down2, w->down1, t.d1, w->down1, w->down_w1, w->down1, nullptr, nullptr, 0, SEEDMI_EPI_RELU, t.d2,
                        w->down2, 0, 0, s));
    CK(seedmi_gemm_bf16(R, w->down3, w->down2, t.d2, w->down2, w->down_w2, w->down2, nullptr, nullptr, 0, SEEDMI_EPI_NONE, t.d3,
                        w->down3, 0, 0, s));
    // reshape [B, n_query*32] is free (rows are contiguous); distill_image_proj (qformer_quantizer.py:334-336)
    const int KD = nq * w->down3;
    CK(seedmi_gemm_bf16(batch, w->out_dim, KD, t.d3, KD, w->distill_w, KD, w->distill_b, nullptr, 0, SEEDMI_EPI_BIAS, embeds,
                        w->out_dim, 0, 0, s));
    return SEEDMI_OK;
}
