// Shared device helpers for the seedmi gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                            // storage type of bf16 tensors
typedef __attribute__((ext_vector_type(8))) short bf16x8;           // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;            // one 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define SEEDMI_DEVINL __device__ __forceinline__

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
// value rounded to bf16 and widened again: the point where the reference materialises a half tensor
SEEDMI_DEVINL float rbf(float f) { return bf2f(f2bf(f)); }

SEEDMI_DEVINL float lo_bf(uint32_t u) { return __uint_as_float(u << 16); }
SEEDMI_DEVINL float hi_bf(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

SEEDMI_DEVINL float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
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

// status codes of the C ABI (include/seedmi.h)
#define SEEDMI_OK 0
#define SEEDMI_E_SHAPE (-1)
#define SEEDMI_E_DTYPE (-2)
#define SEEDMI_E_ALIGN (-3)
#define SEEDMI_E_ARCH (-4)
#define SEEDMI_E_HIP (-5)

void seedmi_set_error(const char* fmt, ...);
int seedmi_check_launch(const char* what);
