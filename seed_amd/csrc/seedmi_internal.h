// Internal declarations shared by the seedmi translation units (not part of the public C ABI).
#pragma once
#include <stdint.h>

// epilogue selectors of seedmi_gemm_bf16 (mirrored in include/seedmi.h and seed_amd/lib.py)
enum {
    EPI_NONE = 0,           // C = A W^T
    EPI_BIAS = 1,           // + bias                       (nn.Linear)
    EPI_BIAS_GELU = 2,      // gelu_erf(half(. + bias))     (eva_vit.py:60-61, qformer_causual.py:321-322)
    EPI_BIAS_RESIDUAL = 3,  // half(. + bias) + residual    (eva_vit.py:201-202, qformer_causual.py:252-254, llama_xformer.py:316,322)
    EPI_BIAS_TANH = 4,      // tanh(half(. + bias))         (qformer_quantizer.py:219-221)
    EPI_SWIGLU = 5,         // silu(gate) * up on interleaved rows (llama_xformer.py:186)
    EPI_PATCH_EMBED = 6,    // conv bias + pos_embed, rows shifted past each image's cls slot (eva_vit.py:229,373-377)
    EPI_RELU = 7,           // relu(half(. [+ bias]))       (image_down, qformer_quantizer.py:279-285)
};

// seedmi_set_option("tokenize_streams", 1|2): sub-batch overlap inside seedmi_tokenize (tokenizer.hip)
int seedmi_tokenizer_set_streams(int n);
int seedmi_tokenizer_set_streamk(int v);
int seedmi_tokenizer_set_lnfold(int v);
int seedmi_tokenizer_set_split(int v);
int seedmi_tokenizer_set_vqhead(int v);
int seedmi_tokenizer_set_tilestats(int v);
// seedmi_set_option("skinny_nt" | "skinny_waves", v): decode GEMM experiments (llama.hip)
int seedmi_llama_set_option(const char* key, int value);
int seedmi_attn_set_option(const char* key, int value);
// attn_vit.hip: persistent ViT attention; returns 1 when the shape is not its own
int seedmi_attention_vit_try(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                             int batch, int heads, int head_dim, int nq, int nk, float scale, int causal, int round_scores,
                             void* stream);
int seedmi_attn_vit_set(int v);
int seedmi_attn_vit_small(int v);        // "attn_small": q-tile split of small ViT attention launches (1 auto, 2..8 fixed factor, 0 off)
int seedmi_attn_vit_xcd(int v);          // "attn_xcd": the staggered ViT kernel's XCD-aware item walk (1, default)
int seedmi_attn_vit_store_wait(int v);   // "attn_store_wait": the 16-wave ViT kernel's K / Q wait leaves the previous item's output stores in flight (1, default)

// per-device launch state (function attributes and the CU count belong to a device, not to the process): capi.hip
#define SEEDMI_MAX_DEVICES 64
// first launch of a decode step (norm_misc.hip): embedding rows -> row-major + fragment-major copies, n_zero 32-bit words cleared
int seedmi_embed_rows_decode(const void* ids_i64, const void* table, int ldt, void* out, int ldo, void* out_packed, int n, int cols,
                             int vocab, void* zero_words, int n_zero, void* stream);
int seedmi_current_device(void);          // hipGetDevice, clamped to [0, SEEDMI_MAX_DEVICES)
int seedmi_device_cus(int dev);           // multiProcessorCount of that device (cached)
