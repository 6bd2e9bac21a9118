// Full-row softmax attention for short key sequences (<= 288 keys) on gfx950.
//
// Replaces  eva_vit.py:139-156 (ViT: 257 x 257, 16 heads x 88),
//           qformer_causual.py:189-236 self (32 x 32 causal, 12 x 64) and cross (32 x 257, 12 x 64).
//
// One workgroup (8 waves) per (image, head).  K [keys][hd] and V^T [hd][keys] of that head are staged once in
// LDS (zero padded to HDP x NKP); each wave then walks 16-query tiles:
//   S^T = K . Q^T            v_mfma_f32_16x16x32_bf16, A = K rows from LDS, B = Q rows straight from global
//   softmax over the whole key row held in registers (no online rescale: the row is <= 72 values per lane),
//   rounding points as the reference: q*scale -> half, S -> half, P normalised then -> half
//   O^T = V^T . P^T          A = V^T rows from LDS (two ds_read_b64), B = P, never leaves registers
// The S^T orientation makes P's accumulator layout *be* the next MFMA's operand layout: lane (i, g) holds
// keys {16t + 4g + r}, and V^T fragments are read with the same (g, j) -> key mapping, so no cross-lane
// movement or LDS round trip is needed for P.
#include <string.h>
#include <atomic>
#include "common.h"
#include "seedmi_internal.h"

namespace {

struct AttnParams {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
    int ldq, ldk, ldv, ldo;
    int nq, nk, heads;
    float scale;
};

template <int HD> struct AttnGeom {
    static constexpr int HDP = (HD + 31) / 32 * 32;
    static constexpr int KPITCH = (HDP == 96) ? 120 : HDP + 8;     // elements; 16-B chunks per row odd
    static constexpr int VRP = HDP + 16;                           // 224 B / 160 B / 288 B rows: ds_read_b64_tr_b16 conflict-free
};

template <int HD, int NKP, bool CAUSAL, bool ROUND_S, bool TRV>
__global__ __launch_bounds__(512) void attn_fullrow_kernel(AttnParams p) {
    constexpr int HDP = AttnGeom<HD>::HDP;
    constexpr int KPITCH = AttnGeom<HD>::KPITCH;
    constexpr int VPITCH = NKP + 8;                                 // V^T image: 8 * odd for NKP in {32, 288}
    constexpr int VRP = AttnGeom<HD>::VRP;                          // row-major V image (TRV): 32-byte * odd row pitch
    constexpr int NT = NKP / 16;                                    // key tiles
    constexpr int KS = HDP / 32;                                    // k-steps of QK^T
    constexpr int KK = NKP / 32;                                    // k-steps of PV
    constexpr int HT = HDP / 16;                                    // output tiles along hd
    constexpr int CH = HDP / 8;                                     // 16-B chunks per padded row

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ksm = (bf16_t*)smem;                                    // [NKP][KPITCH]
    bf16_t* Vt = Ksm + NKP * KPITCH;                                // [HDP][VPITCH]  or, TRV, row-major [NKP][VRP]

    const int tid = threadIdx.x;
    const int lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = tid >> 6;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const size_t krow0 = (size_t)b * p.nk;
    const size_t qrow0 = (size_t)b * p.nq;
    const int hoff = h * HD;

    // ---- stage K (row-major, zero padded)
    for (int idx = tid; idx < NKP * CH; idx += 512) {
        const int row = idx / CH, c = idx - row * CH;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < p.nk && 8 * c < HD) v = *(const uint4*)(p.K + (krow0 + row) * p.ldk + hoff + 8 * c);
        *(uint4*)(Ksm + row * KPITCH + 8 * c) = v;
    }
    if (TRV) {
        // ---- stage V row-major exactly like K (coalesced 16-B loads, ds_write_b128); the PV operand is produced by the
        //      hardware transpose read ds_read_b64_tr_b16, so no transposed image has to be built
        for (int idx = tid; idx < NKP * CH; idx += 512) {
            const int row = idx / CH, c = idx - row * CH;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < p.nk && 8 * c < HD) v = *(const uint4*)(p.V + (krow0 + row) * p.ldv + hoff + 8 * c);
            *(uint4*)(Vt + row * VRP + 8 * c) = v;
        }
    } else {
        // ---- stage V transposed: thread takes a key pair x 8 hd values, writes 8 packed dwords
        for (int idx = tid; idx < (NKP / 2) * CH; idx += 512) {
            const int c = idx / (NKP / 2), kp = idx - c * (NKP / 2);
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
            if (8 * c < HD) {
                if (2 * kp < p.nk) v0 = *(const uint4*)(p.V + (krow0 + 2 * kp) * p.ldv + hoff + 8 * c);
                if (2 * kp + 1 < p.nk) v1 = *(const uint4*)(p.V + (krow0 + 2 * kp + 1) * p.ldv + hoff + 8 * c);
            }
            const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w};
            const uint32_t d[4] = {v1.x, v1.y, v1.z, v1.w};
            uint32_t* dst = (uint32_t*)(Vt + (8 * c) * VPITCH + 2 * kp);
    #pragma unroll
            for (int i = 0; i < 4; ++i) {
                dst[(2 * i) * (VPITCH / 2)] = (a[i] & 0xffffu) | (d[i] << 16);
                dst[(2 * i + 1) * (VPITCH / 2)] = (a[i] >> 16) | (d[i] & 0xffff0000u);
            }
        }
    }
    __syncthreads();

    const int nqt = (p.nq + 15) / 16;
    for (int qt = wave; qt < nqt; qt += 8) {
        const int qrow = 16 * qt + li;                              // this lane's query row (MFMA column)
        // ---- Q fragments, pre-scaled and rounded like `q = q * self.scale` on a half tensor
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int c0 = 32 * ks + 8 * g;
            if (qrow < p.nq && c0 < HD) v = *(const uint4*)(p.Q + (qrow0 + qrow) * p.ldq + hoff + c0);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = pack2bf(lo_bf(w[i]) * p.scale, hi_bf(w[i]) * p.scale);
            uint4 o = make_uint4(w[0], w[1], w[2], w[3]);
            qf[ks] = __builtin_bit_cast(bf16x8, o);
        }
        // ---- S^T tiles
        f32x4 s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(Ksm + (16 * t + li) * KPITCH + 32 * ks + 8 * g);
                s[t] = seedmi_mfma_16x16x32(kf, qf[ks], s[t]);
            }
            // keep the scheduler from hoisting every LDS read of the unrolled tile loop (register blow-up)
            if (t & 1) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- softmax over keys {16t + 4g + r}
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * t + 4 * g + r;
                float v = ROUND_S ? rbf(s[t][r]) : s[t][r];
                if (CAUSAL || 16 * t + 15 >= p.nk) {                // wave-uniform: whole tiles of live keys skip the mask
                    const bool dead = (key >= p.nk) || (CAUSAL && key > qrow);
                    v = dead ? -INFINITY : v;
                }
                s[t][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[t][r] - mx);
                s[t][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        // ---- O^T = V^T P^T
        f32x4 o[HT];
#pragma unroll
        for (int n = 0; n < HT; ++n) o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            uint4 pw;
            pw.x = pack2bf(s[2 * kk][0] * inv, s[2 * kk][1] * inv);
            pw.y = pack2bf(s[2 * kk][2] * inv, s[2 * kk][3] * inv);
            pw.z = pack2bf(s[2 * kk + 1][0] * inv, s[2 * kk + 1][1] * inv);
            pw.w = pack2bf(s[2 * kk + 1][2] * inv, s[2 * kk + 1][3] * inv);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int n = 0; n < HT; ++n) {
                uint2 lo, hi;
                if (TRV) {
                    // 16-lane group g fetches the [4 keys x 16 cols] block of keys 32kk+4g.. (and +16): lane li supplies the
                    // address of row li>>2, cols 4*(li&3)..+3 and receives column li of the block = keys (g, j) of col 16n+li
                    typedef __attribute__((ext_vector_type(4))) short s16x4;
                    const bf16_t* vp = Vt + (32 * kk + 4 * g + (li >> 2)) * VRP + 16 * n + 4 * (li & 3);
                    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)vp);
                    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 16 * VRP));
                    lo = __builtin_bit_cast(uint2, a);
                    hi = __builtin_bit_cast(uint2, b);
                } else {
                    const bf16_t* vp = Vt + (16 * n + li) * VPITCH + 32 * kk + 4 * g;
                    lo = *(const uint2*)vp;
                    hi = *(const uint2*)(vp + 16);
                }
                const uint4 vw = make_uint4(lo.x, lo.y, hi.x, hi.y);
                o[n] = seedmi_mfma_16x16x32(__builtin_bit_cast(bf16x8, vw), pf, o[n]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- store: lane holds O[qrow][16n + 4g + r]
        if (qrow < p.nq) {
            bf16_t* op = p.O + (qrow0 + qrow) * p.ldo + hoff;
#pragma unroll
            for (int n = 0; n < HT; ++n) {
                const int c0 = 16 * n + 4 * g;
                if (c0 + 4 <= HD) {
                    uint2 w;
                    w.x = pack2bf(o[n][0], o[n][1]);
                    w.y = pack2bf(o[n][2], o[n][3]);
                    *(uint2*)(op + c0) = w;
                }
            }
        }
    }
}

std::atomic<int> g_attn_trv{1};     // seedmi_set_option("attn_trv", 0|1): hardware transpose read for V (1) or transposed LDS image (0)

template <int HD, int NKP, bool CAUSAL, bool ROUND_S, bool TRV>
int launch_attn_v(const AttnParams& p, int batch, hipStream_t stream) {
    constexpr int HDP = AttnGeom<HD>::HDP;
    constexpr int lds = (NKP * AttnGeom<HD>::KPITCH + (TRV ? NKP * AttnGeom<HD>::VRP : HDP * (NKP + 8))) * 2;
    static bool attr_set_dev[SEEDMI_MAX_DEVICES] = {};
    bool& attr_set = attr_set_dev[seedmi_current_device()];
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_fullrow_kernel<HD, NKP, CAUSAL, ROUND_S, TRV>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((attn_fullrow_kernel<HD, NKP, CAUSAL, ROUND_S, TRV>), dim3(batch * p.heads), dim3(512), lds, stream, p);
    return seedmi_check_launch("attn_fullrow");
}

template <int HD, int NKP, bool CAUSAL, bool ROUND_S>
int launch_attn(const AttnParams& p, int batch, hipStream_t stream) {
    return g_attn_trv ? launch_attn_v<HD, NKP, CAUSAL, ROUND_S, true>(p, batch, stream)
                      : launch_attn_v<HD, NKP, CAUSAL, ROUND_S, false>(p, batch, stream);
}

}  // namespace

int seedmi_attn_set_option(const char* key, int value) {
    if (!strcmp(key, "attn_trv") && (value == 0 || value == 1)) { g_attn_trv = value; return SEEDMI_OK; }
    // product values: 0 generic | 1 twelve-wave | 2 / 3 sixteen-wave (8- / 16-byte output stores) | 5 staggered sixteen-wave (default); all five give
    // the same bits on rows 0..255 (row 256's side path: to 1 ulp).  4 / 6 ("flash" normalisation: moves a rounding point) and 7 (5 without wave
    // priorities) are measurement arms: devtools build only
#ifdef SEEDMI_DEVTOOLS
    if (!strcmp(key, "attn_vit") && value >= 0 && value <= 7) return seedmi_attn_vit_set(value);
#else
    if (!strcmp(key, "attn_vit") && ((value >= 0 && value <= 3) || value == 5)) return seedmi_attn_vit_set(value);
#endif
    if (!strcmp(key, "attn_store_wait") && (value == 0 || value == 1)) return seedmi_attn_vit_store_wait(value);
    if (!strcmp(key, "attn_xcd") && (value == 0 || value == 1)) return seedmi_attn_vit_xcd(value);
    if (!strcmp(key, "attn_small") && value >= 0 && value <= 16) return seedmi_attn_vit_small(value);
    return SEEDMI_E_SHAPE;
}

// Q/K/V/O are [batch*n, ld] bf16 matrices whose columns [h*hd, (h+1)*hd) belong to head h.
extern "C" int seedmi_attention_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O,
                                     int ldo, int batch, int heads, int head_dim, int nq, int nk, float scale,
                                     int causal, int round_scores, void* stream) {
    if (batch <= 0 || heads <= 0 || nq <= 0 || nk <= 0) {
        seedmi_set_error("seedmi_attention_bf16: bad shape batch=%d heads=%d nq=%d nk=%d", batch, heads, nq, nk);
        return SEEDMI_E_SHAPE;
    }
    if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldo % 4) ||
        (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) || ((uintptr_t)O & 7)) {
        seedmi_set_error("seedmi_attention_bf16: Q/K/V need 16-byte aligned rows, O 8-byte");
        return SEEDMI_E_ALIGN;
    }
    AttnParams p;
    p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (bf16_t*)O;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.nq = nq; p.nk = nk; p.heads = heads; p.scale = scale;
    hipStream_t s = (hipStream_t)stream;
    const bool rs = round_scores != 0;
    {   // the persistent LDS-DMA pipeline handles the ViT shape (head_dim 88, 64..272 tokens, no mask)
        const int rc = seedmi_attention_vit_try(Q, ldq, K, ldk, V, ldv, O, ldo, batch, heads, head_dim, nq, nk, scale, causal,
                                                round_scores, stream);
        if (rc <= 0) return rc;
    }
    if (causal && nq != nk) {
        seedmi_set_error("seedmi_attention_bf16: causal needs nq == nk (got %d, %d)", nq, nk);
        return SEEDMI_E_SHAPE;
    }
#define SEEDMI_ATTN_CASE(HD_, NKP_)                                                              \
    if (head_dim == HD_ && nk <= NKP_) {                                                          \
        if (causal) return rs ? launch_attn<HD_, NKP_, true, true>(p, batch, s)                   \
                              : launch_attn<HD_, NKP_, true, false>(p, batch, s);                 \
        return rs ? launch_attn<HD_, NKP_, false, true>(p, batch, s)                              \
                  : launch_attn<HD_, NKP_, false, false>(p, batch, s);                            \
    }
    SEEDMI_ATTN_CASE(88, 32)
    SEEDMI_ATTN_CASE(88, 288)
    SEEDMI_ATTN_CASE(64, 32)
    SEEDMI_ATTN_CASE(64, 288)
#undef SEEDMI_ATTN_CASE
    seedmi_set_error("seedmi_attention_bf16: unsupported head_dim=%d / nk=%d (supported: hd 64|88, nk <= 288)", head_dim, nk);
    return SEEDMI_E_SHAPE;
}
