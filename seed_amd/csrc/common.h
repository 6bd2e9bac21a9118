// Shared device helpers for the seedmi gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The library is compiled once per 16-bit element type: bf16 (libseedmi.so, the default and what BASELINE.json's configs ask for) and
// IEEE fp16 (-DSEEDMI_F16 -> libseedmi_f16.so: the compute type the reference ships with, configs/tokenizer/seed_llama_tokenizer_hf.yaml:3,
// seed_llama_tokenizer.py:58-59,86-87).  Every conversion and every rounding point goes through the helpers below and every contraction
// through seedmi_mfma_16x16x32, so the two builds differ in nothing else: same kernels, same fp32 islands, same rounding PLACES
// (SURVEY Appendix A), v_mfma_f32_16x16x32_f16 instead of _bf16 at the same rate.  The historic names (bf16_t, *_bf16 entry points,
// f2bf / rbf / lo_bf ...) mean "the library's 16-bit element" in both builds.
typedef uint16_t bf16_t;                                            // storage type of 16-bit tensors
typedef __attribute__((ext_vector_type(8))) short bf16x8;           // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;            // one 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 hw_f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 hw_f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 hw_bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define SEEDMI_DEVINL __device__ __forceinline__

#ifdef SEEDMI_F16
#define SEEDMI_ELEM_NAME "fp16"
SEEDMI_DEVINL float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// round-to-nearest-even fp32 -> fp16 (v_cvt_f16_f32; values beyond 65504 become inf, as in the reference's fp16 tensors)
SEEDMI_DEVINL bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
SEEDMI_DEVINL uint32_t pack2bf(float lo, float hi) {
    f32x2 v = {lo, hi};
    hw_f16x2 r = __builtin_convertvector(v, hw_f16x2);
    return __builtin_bit_cast(uint32_t, r);
}
SEEDMI_DEVINL float lo_bf(uint32_t u) { return (float)__builtin_bit_cast(hw_f16x2, u)[0]; }
SEEDMI_DEVINL float hi_bf(uint32_t u) { return (float)__builtin_bit_cast(hw_f16x2, u)[1]; }
SEEDMI_DEVINL f32x4 seedmi_mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hw_f16x8, a), __builtin_bit_cast(hw_f16x8, b), c, 0, 0, 0);
}
#else
#define SEEDMI_ELEM_NAME "bf16"
SEEDMI_DEVINL float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even fp32 -> bf16 (lowers to v_cvt_pk_bf16_f32 on gfx950)
SEEDMI_DEVINL bf16_t f2bf(float f) {
    __bf16 r = (__bf16)f;
    return __builtin_bit_cast(bf16_t, r);
}
SEEDMI_DEVINL uint32_t pack2bf(float lo, float hi) {
    f32x2 v = {lo, hi};
    hw_bf16x2 r = __builtin_convertvector(v, hw_bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
SEEDMI_DEVINL float lo_bf(uint32_t u) { return __uint_as_float(u << 16); }
SEEDMI_DEVINL float hi_bf(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
SEEDMI_DEVINL f32x4 seedmi_mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif

// Full-line stores: (a, c) = the lane's two 16-byte pieces of row li (columns 16 g .. + 7 and 16 g + 8 .. + 15 of the wave's 64-column span).
// o1 = what the lane contributes to row (li & 7), o2 = to row 8 + (li & 7), at byte 32 g + 16 (li >> 3) of the span: lanes li < 8 keep their first
// piece and take the first piece of row li + 8, lanes li >= 8 keep their second piece and take the second piece of row li - 8 (a rotation by 8 inside
// every row of 16 lanes: one DPP move per register and direction, as many instructions as the permlane transposition they replace).
typedef unsigned seedmi_u32x4 __attribute__((ext_vector_type(4)));
SEEDMI_DEVINL void rows_to_full_lines(const unsigned (&a)[4], const unsigned (&c)[4], seedmi_u32x4& o1, seedmi_u32x4& o2) {
    unsigned x[4], y[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        x[d] = (unsigned)__builtin_amdgcn_update_dpp((int)a[d], (int)c[d], 0x128, 0xf, 0xc, false);     // row_ror:8 into lanes 8..15
        y[d] = (unsigned)__builtin_amdgcn_update_dpp((int)c[d], (int)a[d], 0x128, 0xf, 0x3, false);     // row_ror:8 into lanes 0..7
    }
    o1 = (seedmi_u32x4){x[0], x[1], x[2], x[3]};
    o2 = (seedmi_u32x4){y[0], y[1], y[2], y[3]};
}

// value rounded to the 16-bit element and widened again: the point where the reference materialises a half tensor
SEEDMI_DEVINL float rbf(float f) { return bf2f(f2bf(f)); }

// Exact-erf GELU (nn.GELU(), ACT2FN['gelu']).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-level)
// evaluated on |z| so that 1+erf(z) for z < 0 is formed without cancellation: ~14 VALU + v_exp + v_rcp instead of the
// ~40-instruction libm erff — the GEMM epilogue applies it to 128 values per lane and was 15 % of an fc1 tile.
SEEDMI_DEVINL float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc_z = p * t * __expf(-z * z);              // 1 - erf(|z|)
    const float one_plus_erf = x >= 0.f ? 2.0f - erfc_z : erfc_z;
    return 0.5f * x * one_plus_erf;
}
// (mean, rstd) of a row from its column sum and sum of squares: ONE expression for every place that finishes LayerNorm statistics (the
// finalize kernel, the 256x256 GEMM's in-tile finalize, the small-M consumers' epilogue) - written with the non-contracting intrinsics so
// that no compiler decision (a fused multiply-add here, none there) can make two of them differ in the last bit: an image must give the same
// ids whichever kernel its batch size selects.
SEEDMI_DEVINL float2 seedmi_ln_finish(float s1, float s2, float inv_n, float eps) {
    const float mean = __fmul_rn(s1, inv_n);
    const float var = fmaxf(__fsub_rn(__fmul_rn(s2, inv_n), __fmul_rn(mean, mean)), 0.f);
    return make_float2(mean, rsqrtf(__fadd_rn(var, eps)));
}

SEEDMI_DEVINL float silu(float x) { return x / (1.0f + __expf(-x)); }

SEEDMI_DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
SEEDMI_DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// lane id obtained where it is used (volatile: not hoisted).  Cold per-tile / per-item code of the persistent kernels (the 256x256 GEMM's tile address set-up, fold
// operand addresses and epilogue; the ViT attention's staging and side jobs) derives its lane-dependent values from this instead of from threadIdx up front, so that they do not occupy
// registers - or scratch slots, whose reloads are VM operations in the middle of the LDS-DMA pipeline - across the K loop.
SEEDMI_DEVINL int fresh_lane() {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
}

// status codes of the C ABI (include/seedmi.h)
#define SEEDMI_OK 0
#define SEEDMI_E_SHAPE (-1)
#define SEEDMI_E_DTYPE (-2)
#define SEEDMI_E_ALIGN (-3)
#define SEEDMI_E_ARCH (-4)
#define SEEDMI_E_HIP (-5)

void seedmi_set_error(const char* fmt, ...);
int seedmi_check_launch(const char* what);
