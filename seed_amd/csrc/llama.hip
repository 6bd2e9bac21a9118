// LLaMA-side kernels: decode GEMM (M <= 64), attention over the static KV cache, and the forward orchestrator.
//
//  seedmi_gemm_skinny_bf16        q/k/v/o_proj, gate/up/down_proj, lm_head at decode batch sizes
//                                 (llama_xformer.py:223-225,258,186,718) — HBM-bound weight streaming
//  seedmi_llama_attention_bf16    xformers.ops.memory_efficient_attention (llama_xformer.py:244-256):
//                                 prefill = causal, decode (T == 1) = all cached keys; scale 1/sqrt(128)
//  seedmi_llama_forward           LlamaForCausalLM.forward / LlamaModel.forward / LlamaDecoderLayer.forward
//                                 (llama_xformer.py:661-743, 496-627, 280-332)
#include "common.h"
#include "seedmi_internal.h"
#include "../../include/seedmi.h"

namespace {

// ------------------------------------------------------------------------------------------------ skinny GEMM
// One workgroup = 16 weight rows; its 4 waves split K, each streaming W fragments straight HBM -> VGPR
// (weights are read exactly once, so an LDS round trip would be pure overhead) and taking the activation
// fragments from L2.  MFMA A = W rows, B = activation rows: D col = m, D row = n.
struct SkinnyParams {
    int M, N, K;
    const bf16_t* A; int lda;
    const bf16_t* W; int ldw;
    const bf16_t* R; int ldr;
    bf16_t* C; int ldc;
};

// NW waves split K; every wave keeps two register sets of U k-steps in flight (the next batch is requested before
// the current one is consumed), i.e. up to 2*U*(1+MT) 16-byte loads per lane outstanding — what a one-workgroup-per-CU
// launch (N/16 = 256 workgroups for the 4096-row projections) needs to cover HBM latency.
template <int MT, int EPI, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(SkinnyParams p) {
    __shared__ float red[NW - 1][MT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const int kslice = p.K / NW;
    const int kbeg = wave * kslice;
    const int wrow = min(n0 + li, p.N - 1);
    const bf16_t* wp = p.W + (size_t)wrow * p.ldw + kbeg + 8 * g;
    const bf16_t* ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) ap[t] = p.A + (size_t)min(16 * t + li, p.M - 1) * p.lda + kbeg + 8 * g;

    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int U = 4;
    bf16x8 w0[U], a0[U][MT], w1[U], a1[U][MT];
    const int nb = (kslice + 32 * U - 1) / (32 * U);
    auto load = [&](bf16x8 (&wf)[U], bf16x8 (&af)[U][MT], int b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = min(32 * (U * b + u), kslice - 32);          // clamped: out-of-range steps are skipped below
            wf[u] = __builtin_nontemporal_load((const bf16x8*)(wp + kk));
#pragma unroll
            for (int t = 0; t < MT; ++t) af[u][t] = *(const bf16x8*)(ap[t] + kk);
        }
    };
    auto compute = [&](bf16x8 (&wf)[U], bf16x8 (&af)[U][MT], int b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (32 * (U * b + u) < kslice) {
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af[u][t], acc[t], 0, 0, 0);
            }
        }
    };
    load(w0, a0, 0);
    for (int b = 0; b < nb; b += 2) {
        if (b + 1 < nb) load(w1, a1, b + 1);
        compute(w0, a0, b);
        if (b + 2 < nb) load(w0, a0, b + 2);
        if (b + 1 < nb) compute(w1, a1, b + 1);
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][t][lane][r] = acc[t][r];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int w = 0; w < NW - 1; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] += red[w][t][lane][r];
        const int m = 16 * t + li;
        const int nb_ = n0 + 4 * g;
        if (m >= p.M || nb_ >= p.N) continue;
        float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
        if (EPI == EPI_BIAS_RESIDUAL) {
            const bf16_t* rp = p.R + (size_t)m * p.ldr + nb_;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (nb_ + r < p.N) v[r] = rbf(v[r]) + bf2f(rp[r]);
        }
        if (EPI == EPI_SWIGLU) {
            bf16_t* cp = p.C + (size_t)m * p.ldc + (nb_ >> 1);
            if (nb_ + 1 < p.N) cp[0] = f2bf(rbf(silu(rbf(v[0]))) * rbf(v[1]));
            if (nb_ + 3 < p.N) cp[1] = f2bf(rbf(silu(rbf(v[2]))) * rbf(v[3]));
        } else {
            bf16_t* cp = p.C + (size_t)m * p.ldc + nb_;
            if (nb_ + 4 <= p.N && (p.ldc % 4) == 0) {
                uint2 w;
                w.x = pack2bf(v[0], v[1]);
                w.y = pack2bf(v[2], v[3]);
                *(uint2*)cp = w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nb_ + r < p.N) cp[r] = f2bf(v[r]);
            }
        }
    }
}

template <int EPI, int NW>
int launch_skinny_nw(const SkinnyParams& p, hipStream_t s) {
    const int grid = (p.N + 15) / 16;
    const int mt = (p.M + 15) / 16;
    switch (mt) {
        case 1: hipLaunchKernelGGL((gemm_skinny_kernel<1, EPI, NW>), dim3(grid), dim3(64 * NW), 0, s, p); break;
        case 2: hipLaunchKernelGGL((gemm_skinny_kernel<2, EPI, NW>), dim3(grid), dim3(64 * NW), 0, s, p); break;
        case 3: hipLaunchKernelGGL((gemm_skinny_kernel<3, EPI, NW>), dim3(grid), dim3(64 * NW), 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_skinny_kernel<4, EPI, NW>), dim3(grid), dim3(64 * NW), 0, s, p); break;
    }
    return seedmi_check_launch("gemm_skinny");
}

template <int EPI>
int launch_skinny(const SkinnyParams& p, hipStream_t s) {
    // 8-way K split when each slice still holds at least two 128-deep batches, else 4-way
    if ((p.K % 256) == 0 && p.K >= 2048) return launch_skinny_nw<EPI, 8>(p, s);
    return launch_skinny_nw<EPI, 4>(p, s);
}

// ------------------------------------------------------------------------------------------------ decode attention
// T == 1: one workgroup per (batch, head).  Memory bound: K and V rows (256 B) are read once with 16-B accesses.
constexpr int DEC_HD = 128;

__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q, int ldq,
                                                          const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
                                                          bf16_t* __restrict__ out, int ldo, int H, int tmax, int kv_len,
                                                          float scale) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* sc = dsm;                                // [kv_len] scores -> probabilities
    float* part = dsm + ((kv_len + 3) & ~3);        // [16][128] partial outputs
    __shared__ float wred[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int c = tid & 15;                         // 16-B chunk of the head dim
    const int ks = tid >> 4;                        // key slot 0..15
    const bf16_t* kb = kc + ((size_t)b * H + h) * tmax * DEC_HD;
    const bf16_t* vb = vc + ((size_t)b * H + h) * tmax * DEC_HD;
    float qv[8];
    {
        const uint4 u = *(const uint4*)(q + (size_t)b * ldq + h * DEC_HD + 8 * c);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { qv[2 * i] = lo_bf(w[i]); qv[2 * i + 1] = hi_bf(w[i]); }
    }
    // scores
    float lmax = -INFINITY;
    for (int j = ks; j < kv_len; j += 16) {
        const uint4 u = *(const uint4*)(kb + (size_t)j * DEC_HD + 8 * c);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) d += qv[2 * i] * lo_bf(w[i]) + qv[2 * i + 1] * hi_bf(w[i]);
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64);
        d *= scale;
        if (c == 0) sc[j] = d;
        lmax = fmaxf(lmax, d);
    }
    lmax = wave_max(lmax);
    if (lane == 0) wred[wave] = lmax;
    __syncthreads();
    const float mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    __syncthreads();
    float lsum = 0.f;
    for (int j = tid; j < kv_len; j += 256) {
        const float e = __expf(sc[j] - mx);
        sc[j] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) wred[wave] = lsum;
    __syncthreads();
    const float inv = 1.0f / (wred[0] + wred[1] + wred[2] + wred[3]);
    // O = sum_j half(p_j) * v_j
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = ks; j < kv_len; j += 16) {
        const float pj = rbf(sc[j] * inv);
        const uint4 u = *(const uint4*)(vb + (size_t)j * DEC_HD + 8 * c);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] += pj * lo_bf(w[i]); o[2 * i + 1] += pj * hi_bf(w[i]); }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[ks * DEC_HD + 8 * c + i] = o[i];
    __syncthreads();
    if (tid < DEC_HD) {
        float a = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) a += part[s2 * DEC_HD + tid];
        out[(size_t)b * ldo + h * DEC_HD + tid] = f2bf(a);
    }
}

// ------------------------------------------------------------------------------------------------ prefill attention
// Workgroup = (batch, head, 64 queries); wave = 16 queries.  Two passes over the keys so that P can be normalised
// BEFORE it is rounded to bf16 (what a softmax -> half -> matmul sequence does): pass 1 row max / row sum from
// S^T = K Q^T with K fragments straight from the cache (L2), pass 2 recomputes S^T, forms P and accumulates
// O^T = V^T P^T with V^T tiles transposed through LDS.  The extra QK^T pass costs < 1 % of prefill FLOPs.
constexpr int PF_VP = 40;   // V^T tile pitch in elements (32 keys + 8): 80 B rows, conflict-free b64 reads

__global__ __launch_bounds__(256) void attn_prefill_kernel(const bf16_t* __restrict__ q, int ldq,
                                                           const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
                                                           bf16_t* __restrict__ out, int ldo, int T, int H, int tmax,
                                                           int past_len, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Vt[DEC_HD * PF_VP];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = tid >> 6;
    const int qblocks = (T + 63) / 64;
    const int qb = blockIdx.x % qblocks;
    const int bh = blockIdx.x / qblocks;
    const int b = bh / H, h = bh % H;
    const bf16_t* kb = kc + ((size_t)b * H + h) * tmax * DEC_HD;
    const bf16_t* vb = vc + ((size_t)b * H + h) * tmax * DEC_HD;
    const int qrow = 64 * qb + 16 * wave + li;                   // query position inside this forward
    const int qlim = past_len + qrow;                            // last key this query may see
    const int kv_len = past_len + T;
    const int wave_lim = min(kv_len - 1, past_len + 64 * qb + 16 * wave + 15);   // last key any lane of the wave sees
    const int blk_lim = min(kv_len - 1, past_len + 64 * qb + 63);

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (qrow < T) v = *(const uint4*)(q + ((size_t)b * T + qrow) * ldq + h * DEC_HD + 32 * ks + 8 * g);
        qf[ks] = __builtin_bit_cast(bf16x8, v);
    }
    // ---- pass 1: running max / sum per lane over its keys {16t + 4g + r}
    float m_run = -INFINITY, l_run = 0.f;
    for (int t0 = 0; t0 <= wave_lim; t0 += 16) {
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int krow = min(t0 + li, kv_len - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(kb + (size_t)krow * DEC_HD + 32 * ks + 8 * g);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        float tm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = t0 + 4 * g + r;
            s[r] = (key <= qlim && key < kv_len) ? s[r] * scale : -INFINITY;
            tm = fmaxf(tm, s[r]);
        }
        const float mn = fmaxf(m_run, tm);
        if (mn > -INFINITY) {
            float add = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) add += __expf(s[r] - mn);
            l_run = l_run * __expf(m_run - mn) + add;
            m_run = mn;
        }
    }
    float mx = fmaxf(m_run, __shfl_xor(m_run, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float lsum = (m_run > -INFINITY) ? l_run * __expf(m_run - mx) : 0.f;
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;

    // ---- pass 2
    f32x4 o[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 <= blk_lim; t0 += 32) {
        __syncthreads();                                           // previous tile fully consumed
        {   // stage V^T tile: thread = (key pair, 16-B chunk)
            const int kp = tid & 15, c = tid >> 4;
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
            const int k0 = t0 + 2 * kp;
            if (k0 < kv_len) v0 = *(const uint4*)(vb + (size_t)k0 * DEC_HD + 8 * c);
            if (k0 + 1 < kv_len) v1 = *(const uint4*)(vb + (size_t)(k0 + 1) * DEC_HD + 8 * c);
            const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w};
            const uint32_t d[4] = {v1.x, v1.y, v1.z, v1.w};
            uint32_t* dst = (uint32_t*)(Vt + (8 * c) * PF_VP + 2 * kp);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dst[(2 * i) * (PF_VP / 2)] = (a[i] & 0xffffu) | (d[i] << 16);
                dst[(2 * i + 1) * (PF_VP / 2)] = (a[i] >> 16) | (d[i] & 0xffff0000u);
            }
        }
        __syncthreads();
        if (t0 > wave_lim) continue;                               // tile entirely masked for this wave (wave-uniform)
        f32x4 s[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            s[half] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int krow = min(t0 + 16 * half + li, kv_len - 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kb + (size_t)krow * DEC_HD + 32 * ks + 8 * g);
                s[half] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[half], 0, 0, 0);
            }
        }
        float pv[8];
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t0 + 16 * half + 4 * g + r;
                pv[4 * half + r] = (key <= qlim && key < kv_len) ? __expf(s[half][r] * scale - mx) * inv : 0.f;
            }
        uint4 pw;
        pw.x = pack2bf(pv[0], pv[1]); pw.y = pack2bf(pv[2], pv[3]); pw.z = pack2bf(pv[4], pv[5]); pw.w = pack2bf(pv[6], pv[7]);
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const bf16_t* vp = Vt + (16 * n + li) * PF_VP + 4 * g;
            const uint2 lo = *(const uint2*)vp;
            const uint2 hi = *(const uint2*)(vp + 16);
            const uint4 vw = make_uint4(lo.x, lo.y, hi.x, hi.y);
            o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vw), pf, o[n], 0, 0, 0);
        }
    }
    if (qrow < T) {
        bf16_t* op = out + ((size_t)b * T + qrow) * ldo + h * DEC_HD;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            uint2 w;
            w.x = pack2bf(o[n][0], o[n][1]);
            w.y = pack2bf(o[n][2], o[n][3]);
            *(uint2*)(op + 16 * n + 4 * g) = w;
        }
    }
}

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
struct LlamaWs { bf16_t *x, *xn, *qkv, *q, *att, *act; size_t bytes; };
LlamaWs carve(const seedmi_llama_weights_t* w, int B, int T, void* ws) {
    const size_t M = (size_t)B * T, h = w->hidden, F = w->ffn;
    Carver c(ws);
    LlamaWs t;
    t.x = (bf16_t*)c.take(M * h * 2);
    t.xn = (bf16_t*)c.take(M * h * 2);
    t.qkv = (bf16_t*)c.take(M * 3 * h * 2);
    t.q = (bf16_t*)c.take(M * h * 2);
    t.att = (bf16_t*)c.take(M * h * 2);
    t.act = (bf16_t*)c.take(M * F * 2);
    t.bytes = c.off;
    return t;
}

#define CK(call)                          \
    do {                                  \
        const int rc_ = (call);           \
        if (rc_ != SEEDMI_OK) return rc_; \
    } while (0)

// dispatch: decode-sized M goes to the weight-streaming kernel, everything else to the 128x128 MFMA GEMM
int linear(int M, int N, int K, const void* A, int lda, const void* W, const void* R, int ldr, int epi, void* C, int ldc,
           void* s) {
    if (M <= 64 && (K % 128) == 0)
        return seedmi_gemm_skinny_bf16(M, N, K, A, lda, W, K, R, ldr, epi, C, ldc, s);
    return seedmi_gemm_bf16(M, N, K, A, lda, W, K, nullptr, R, ldr, epi, C, ldc, 0, 0, s);
}

}  // namespace

extern "C" int seedmi_gemm_skinny_bf16(int M, int N, int K, const void* A, int lda, const void* W, int ldw,
                                       const void* residual, int ldr, int epilogue, void* C, int ldc, void* stream) {
    if (M <= 0 || M > 64 || N <= 0 || K <= 0 || (K % 128)) {
        seedmi_set_error("seedmi_gemm_skinny_bf16: M=%d (1..64) N=%d K=%d (multiple of 128)", M, N, K);
        return SEEDMI_E_SHAPE;
    }
    if ((lda % 8) || (ldw % 8) || (((uintptr_t)A | (uintptr_t)W) & 15) || ((uintptr_t)C & 3)) {
        seedmi_set_error("seedmi_gemm_skinny_bf16: A/W need 16-byte aligned rows");
        return SEEDMI_E_ALIGN;
    }
    SkinnyParams p;
    p.M = M; p.N = N; p.K = K;
    p.A = (const bf16_t*)A; p.lda = lda;
    p.W = (const bf16_t*)W; p.ldw = ldw;
    p.R = (const bf16_t*)residual; p.ldr = ldr;
    p.C = (bf16_t*)C; p.ldc = ldc;
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case EPI_NONE: return launch_skinny<EPI_NONE>(p, s);
        case EPI_BIAS_RESIDUAL:
            if (!residual) { seedmi_set_error("seedmi_gemm_skinny_bf16: residual epilogue without residual"); return SEEDMI_E_SHAPE; }
            return launch_skinny<EPI_BIAS_RESIDUAL>(p, s);
        case EPI_SWIGLU: return launch_skinny<EPI_SWIGLU>(p, s);
        default:
            seedmi_set_error("seedmi_gemm_skinny_bf16: unsupported epilogue %d", epilogue);
            return SEEDMI_E_SHAPE;
    }
}

extern "C" int seedmi_llama_attention_bf16(const void* q, int ldq, const void* k_cache, const void* v_cache, void* out,
                                           int ldo, int B, int T, int H, int hd, int tmax, int past_len, float scale,
                                           void* stream) {
    if (hd != DEC_HD || B <= 0 || T <= 0 || H <= 0 || past_len < 0 || past_len + T > tmax) {
        seedmi_set_error("seedmi_llama_attention_bf16: B=%d T=%d H=%d hd=%d (must be 128) past=%d tmax=%d", B, T, H, hd, past_len, tmax);
        return SEEDMI_E_SHAPE;
    }
    if ((ldq % 8) || (ldo % 4) || (((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15) || ((uintptr_t)out & 7)) {
        seedmi_set_error("seedmi_llama_attention_bf16: alignment");
        return SEEDMI_E_ALIGN;
    }
    hipStream_t s = (hipStream_t)stream;
    if (T == 1) {
        const int kv_len = past_len + 1;
        const size_t lds = (size_t)(((kv_len + 3) & ~3) + 16 * DEC_HD) * sizeof(float);
        if (lds > 150 * 1024) {
            seedmi_set_error("seedmi_llama_attention_bf16: kv_len %d too long for the decode kernel", kv_len);
            return SEEDMI_E_SHAPE;
        }
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(attn_decode_kernel, dim3(B * H), dim3(256), lds, s, (const bf16_t*)q, ldq, (const bf16_t*)k_cache,
                           (const bf16_t*)v_cache, (bf16_t*)out, ldo, H, tmax, kv_len, scale);
        return seedmi_check_launch("attn_decode");
    }
    const int qblocks = (T + 63) / 64;
    hipLaunchKernelGGL(attn_prefill_kernel, dim3(B * H * qblocks), dim3(256), 0, s, (const bf16_t*)q, ldq,
                       (const bf16_t*)k_cache, (const bf16_t*)v_cache, (bf16_t*)out, ldo, T, H, tmax, past_len, scale);
    return seedmi_check_launch("attn_prefill");
}

extern "C" size_t seedmi_llama_workspace_bytes(const seedmi_llama_weights_t* w, int batch, int T) {
    if (!w || batch <= 0 || T <= 0) return 0;
    return carve(w, batch, T, nullptr).bytes;
}

extern "C" int seedmi_llama_forward(const seedmi_llama_weights_t* w, const void* ids_i64, const void* pos_i64, int batch,
                                    int T, int past_len, int last_only, void* logits, int ldl, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!w || !ids_i64 || !pos_i64 || !logits || batch <= 0 || T <= 0) {
        seedmi_set_error("seedmi_llama_forward: null argument or bad batch/T");
        return SEEDMI_E_SHAPE;
    }
    if (batch > w->batch_cap || past_len + T > w->tmax || past_len + T > w->max_pos) {
        seedmi_set_error("seedmi_llama_forward: batch %d (cap %d) / length %d (cache %d, rope table %d)", batch,
                         w->batch_cap, past_len + T, w->tmax, w->max_pos);
        return SEEDMI_E_SHAPE;
    }
    const LlamaWs t = carve(w, batch, T, workspace);
    if (!workspace || workspace_bytes < t.bytes || ((uintptr_t)workspace & 255)) {
        seedmi_set_error("seedmi_llama_forward: workspace %zu bytes (need %zu, 256-byte aligned)", workspace_bytes, t.bytes);
        return SEEDMI_E_ALIGN;
    }
    const int h = w->hidden, F = w->ffn, H = w->heads, hd = h / H;
    const int M = batch * T;
    const float scale = 1.0f / sqrtf((float)hd);
    CK(seedmi_embed_rows(ids_i64, w->embed, h, t.x, h, M, h, w->vocab, stream));
    for (int l = 0; l < w->layers; ++l) {
        const seedmi_llama_layer_t& L = w->layer[l];
        CK(seedmi_rmsnorm_bf16(t.x, h, L.ln1_w, w->rms_eps, t.xn, h, M, h, stream));
        CK(linear(M, 3 * h, h, t.xn, h, L.qkv_w, nullptr, 0, EPI_NONE, t.qkv, 3 * h, stream));
        CK(seedmi_rope_kv_append(t.qkv, 3 * h, pos_i64, w->cos_t, w->sin_t, t.q, h, L.k_cache, L.v_cache, batch, T, H, hd,
                                 w->tmax, past_len, stream));
        CK(seedmi_llama_attention_bf16(t.q, h, L.k_cache, L.v_cache, t.att, h, batch, T, H, hd, w->tmax, past_len, scale, stream));
        CK(linear(M, h, h, t.att, h, L.o_w, t.x, h, EPI_BIAS_RESIDUAL, t.x, h, stream));
        CK(seedmi_rmsnorm_bf16(t.x, h, L.ln2_w, w->rms_eps, t.xn, h, M, h, stream));
        CK(linear(M, 2 * F, h, t.xn, h, L.gate_up_w, nullptr, 0, EPI_SWIGLU, t.act, F, stream));
        CK(linear(M, h, F, t.act, F, L.down_w, t.x, h, EPI_BIAS_RESIDUAL, t.x, h, stream));
    }
    if (last_only) {
        // final norm + lm_head on the last position of every sequence only (decode fast path)
        CK(seedmi_rmsnorm_bf16(t.x + (size_t)(T - 1) * h, T * h, w->norm_w, w->rms_eps, t.xn, h, batch, h, stream));
        CK(linear(batch, w->vocab, h, t.xn, h, w->lm_head, nullptr, 0, EPI_NONE, logits, ldl, stream));
    } else {
        CK(seedmi_rmsnorm_bf16(t.x, h, w->norm_w, w->rms_eps, t.xn, h, M, h, stream));
        CK(linear(M, w->vocab, h, t.xn, h, w->lm_head, nullptr, 0, EPI_NONE, logits, ldl, stream));
    }
    return SEEDMI_OK;
}
