// LLaMA-side kernels: decode GEMM (M <= 64), attention over the static KV cache, and the forward orchestrator.
//
//  seedmi_gemm_skinny_bf16        q/k/v/o_proj, gate/up/down_proj, lm_head at decode batch sizes
//                                 (llama_xformer.py:223-225,258,186,718) — HBM-bound weight streaming
//  seedmi_llama_attention_bf16    xformers.ops.memory_efficient_attention (llama_xformer.py:244-256):
//                                 prefill = causal, decode (T == 1) = all cached keys; scale 1/sqrt(128)
//  seedmi_llama_forward           LlamaForCausalLM.forward / LlamaModel.forward / LlamaDecoderLayer.forward
//                                 (llama_xformer.py:661-743, 496-627, 280-332)
#include <string.h>
#include <atomic>
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
    int a_packed, c_packed;      // activations / SWIGLU output in the fragment-major layout (see norm_misc.hip PACK)
    // RMSNorm folded into the GEMM (decode chain): A holds the UN-normalised rows, W holds weight * gamma; the kernel sums the
    // squares of the activation fragments it streams anyway and scales its accumulators by rsqrt(mean(x^2) + eps) per row
    float norm_eps;              // > 0 enables it
    bf16_t* xp_out;              // optional second copy of a BIAS_RESIDUAL result in the fragment-major activation layout
#ifdef SEEDMI_DEVTOOLS
    int abl;                     // timing ablations (seedmi_set_option("skinny_ablate")): 1 no activation loads, 2 no MFMA / norm sums, 4 no reduction / epilogue
#endif
};

// stores of the persistent decode kernel's phase outputs: written through to memory (sc0 sc1), so that the other XCDs see them after
// the grid barrier without anybody writing back a whole L2
template <bool WT>
SEEDMI_DEVINL void st_b64(bf16_t* ptr, uint2 v) {
    if (WT) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(v) : "memory");
    else *(uint2*)ptr = v;
}
template <bool WT>
SEEDMI_DEVINL void st_b16(bf16_t* ptr, bf16_t v) {
    if (WT) asm volatile("global_store_short %0, %1, off sc0 sc1" ::"v"(ptr), "v"((uint32_t)v) : "memory");
    else *ptr = v;
}

// wave 0's epilogue: lane (li, g) owns rows m = 16 t + li and the four columns n0 + 16 r + 4 g .. +3
template <int MT, int EPI, int R, bool WT = false>
SEEDMI_DEVINL void skinny_epilogue(const SkinnyParams& p, f32x4 (&acc)[R][MT], int n0, int li, int g, const float (&rstd)[MT]) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = 16 * t + li;
        const int nb_ = n0 + 16 * r + 4 * g;
        if (m >= p.M || nb_ >= p.N) continue;
        float v[4] = {acc[r][t][0] * rstd[t], acc[r][t][1] * rstd[t], acc[r][t][2] * rstd[t], acc[r][t][3] * rstd[t]};
        if (EPI == EPI_BIAS_RESIDUAL) {
            const bf16_t* rp = p.R + (size_t)m * p.ldr + nb_;
            if (WT) {                                   // persistent kernel: x is re-written in place - read it past this CU's L1
                uint2 rw;
                asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(rw) : "v"(rp) : "memory");
                v[0] = rbf(v[0]) + lo_bf(rw.x); v[1] = rbf(v[1]) + hi_bf(rw.x);
                v[2] = rbf(v[2]) + lo_bf(rw.y); v[3] = rbf(v[3]) + hi_bf(rw.y);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (nb_ + e < p.N) v[e] = rbf(v[e]) + bf2f(rp[e]);
            }
        }
        if (EPI == EPI_SWIGLU) {
            const int kc = nb_ >> 1;                                     // output column of the first (gate, up) pair
            bf16_t* cp = p.c_packed
                ? p.C + ((size_t)((m >> 4) * (p.N >> 6) + (kc >> 5)) * 64 + ((kc >> 3) & 3) * 16 + (m & 15)) * 8 + (kc & 7)
                : p.C + (size_t)m * p.ldc + kc;
            if (nb_ + 1 < p.N) st_b16<WT>(cp, f2bf(rbf(silu(rbf(v[0]))) * rbf(v[1])));
            if (nb_ + 3 < p.N) st_b16<WT>(cp + 1, f2bf(rbf(silu(rbf(v[2]))) * rbf(v[3])));
        } else {
            bf16_t* cp = p.C + (size_t)m * p.ldc + nb_;
            if (nb_ + 4 <= p.N && (p.ldc % 4) == 0) {
                uint2 w;
                w.x = pack2bf(v[0], v[1]);
                w.y = pack2bf(v[2], v[3]);
                st_b64<WT>(cp, w);
                if (p.xp_out)                                             // the next GEMM's fragment-major A operand (row length N)
                    st_b64<WT>(p.xp_out + ((size_t)((m >> 4) * (p.N >> 5) + (nb_ >> 5)) * 64 + ((nb_ >> 3) & 3) * 16 + (m & 15)) * 8 + (nb_ & 7),
                               w);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (nb_ + e < p.N) st_b16<WT>(cp + e, f2bf(v[e]));
            }
        }
    }
}

// NW waves split K; every wave keeps two register sets of U k-steps in flight (the next batch is requested before
// the current one is consumed), i.e. up to 2*U*(1+MT) 16-byte loads per lane outstanding — what a one-workgroup-per-CU
// launch (N/16 = 256 workgroups for the 4096-row projections) needs to cover HBM latency.
template <int MT, int EPI, int NW, bool NT, bool PACKED, int R, bool WT = false>
SEEDMI_DEVINL void skinny_tile(const SkinnyParams& p, const int bidx) {
    // R weight row-tiles (16 rows each) per workgroup share every activation fragment: the activation loads (16 rows x 64 B
    // gathers out of L2) cost more TA cycles than the weight stream itself, so R = 2 where N leaves enough workgroups.
    __shared__ float red[NW - 1][R][MT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = tid >> 6;
    const int n0 = bidx * 16 * R;
#ifdef SEEDMI_DEVTOOLS
    const int abl = WT ? 0 : p.abl;
#else
    constexpr int abl = 0;
#endif
    unsigned xacc = 0;                                                // (ablations: keeps the loads alive)
    const int kslice = p.K / NW;
    const int kbeg = wave * kslice;
    // PACKED: fragment-major weights (seedmi_pack_skinny_weights): tile j, k-step s is one contiguous 1 KiB block holding
    // lane l's 16 bytes at l*16, so a wave streams a contiguous region instead of 16 rows x 64 B at a power-of-two stride
    // (which lands every row of a load on the same HBM channels).
    const bf16_t* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int wrow = min(n0 + 16 * r + li, p.N - 1);
        wp[r] = PACKED ? p.W + ((size_t)(bidx * R + r) * (p.K >> 5) + (kbeg >> 5)) * 512 + lane * 8
                       : p.W + (size_t)wrow * p.ldw + kbeg + 8 * g;
    }
    constexpr int WSTEP = PACKED ? 16 : 1;                           // element stride multiplier per k (32 k -> 512 elements)
    const bf16_t* ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
        ap[t] = p.a_packed ? p.A + ((size_t)(t * (p.K >> 5) + (kbeg >> 5)) * 64 + lane) * 8
                           : p.A + (size_t)min(16 * t + li, p.M - 1) * p.lda + kbeg + 8 * g;
    const int astep = p.a_packed ? 16 : 1;

    f32x4 acc[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // k-steps per register set (bounded by the VGPR budget).  Tried for the N = 4096 projections (R = 1, one workgroup per CU): U = 8,
    // 128 KB of weight loads in flight per CU instead of 64 - 18.5 vs 18.9 us per launch (profiles/r02 run 13): depth is not what holds
    // o_proj / down at 3.1-3.3 TB/s, so the smaller register set stays (two workgroups per CU where the grid has them).
    constexpr int U = (R * MT > 4) ? 2 : 4;
    bf16x8 w0[U][R], a0[U][MT], w1[U][R], a1[U][MT];
    const int nb = (kslice + 32 * U - 1) / (32 * U);
    auto load = [&](bf16x8 (&wf)[U][R], bf16x8 (&af)[U][MT], int b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = min(32 * (U * b + u), kslice - 32);          // clamped: out-of-range steps are skipped below
#pragma unroll
            for (int r = 0; r < R; ++r)
                wf[u][r] = NT ? __builtin_nontemporal_load((const bf16x8*)(wp[r] + kk * WSTEP)) : *(const bf16x8*)(wp[r] + kk * WSTEP);
            if (!(abl & 1)) {
#pragma unroll
                for (int t = 0; t < MT; ++t) af[u][t] = *(const bf16x8*)(ap[t] + kk * astep);
            }
        }
    };
    if (abl & 1) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const uint4 v = make_uint4(0x3c003c00u + lane, 0x3c803c80u, 0xbc003c00u, 0x3c00bc00u + t);
                a0[u][t] = a1[u][t] = __builtin_bit_cast(bf16x8, v);
            }
    }
    float ss[MT];                                                     // sum of squares of this wave's K slice, rows 16t + li
#pragma unroll
    for (int t = 0; t < MT; ++t) ss[t] = 0.f;
    const bool do_norm = p.norm_eps > 0.f;
    auto compute = [&](bf16x8 (&wf)[U][R], bf16x8 (&af)[U][MT], int b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (32 * (U * b + u) < kslice) {
                if (abl & 2) {
#pragma unroll
                    for (int r = 0; r < R; ++r) { const uint4 v = __builtin_bit_cast(uint4, wf[u][r]); xacc ^= v.x ^ v.y ^ v.z ^ v.w; }
#pragma unroll
                    for (int t = 0; t < MT; ++t) { const uint4 v = __builtin_bit_cast(uint4, af[u][t]); xacc ^= v.x ^ v.y ^ v.z ^ v.w; }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        acc[r][t] = seedmi_mfma_16x16x32(wf[u][r], af[u][t], acc[r][t]);
                if (do_norm) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const uint4 xw = __builtin_bit_cast(uint4, af[u][t]);
                        const uint32_t w4[4] = {xw.x, xw.y, xw.z, xw.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float lo = lo_bf(w4[i]), hi = hi_bf(w4[i]);
                            ss[t] = fmaf(lo, lo, fmaf(hi, hi, ss[t]));
                        }
                    }
                }
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
    __shared__ float red_ss[NW][MT][16];
    if (abl & 4) {
        if (xacc == 0x9e3779b9u || acc[0][0][0] == 1234.5f) p.C[lane] = (bf16_t)xacc;
        return;
    }
    if (abl & 2) acc[0][0][0] += (float)(xacc & 1);
    if (do_norm) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {                                // lanes li + 16 g hold the four k-quarters of row 16t + li
            ss[t] += __shfl_xor(ss[t], 16, 64);
            ss[t] += __shfl_xor(ss[t], 32, 64);
            if (g == 0) red_ss[wave][t][li] = ss[t];
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave - 1][r][t][lane][e] = acc[r][t][e];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int w = 0; w < NW - 1; ++w)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r][t][e] += red[w][r][t][lane][e];
        float rstd[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            rstd[t] = 1.f;
            if (do_norm) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += red_ss[w][t][li];        // fixed order: the same rstd in every workgroup
                rstd[t] = rsqrtf(tot / (float)p.K + p.norm_eps);             // LlamaRMSNorm: variance = mean(x^2) in fp32 (llama_xformer.py:108-110)
            }
        }
        skinny_epilogue<MT, EPI, R, WT>(p, acc, n0, li, g, rstd);
    }
}

template <int MT, int EPI, int NW, bool NT, bool PACKED, int R>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(SkinnyParams p) {
    skinny_tile<MT, EPI, NW, NT, PACKED, R>(p, (int)blockIdx.x);
}

std::atomic<int> g_skinny_nt{1}, g_skinny_nw{0};

std::atomic<int> g_skinny_r{0};
#ifdef SEEDMI_DEVTOOLS
std::atomic<int> g_skinny_abl{0};
#endif
std::atomic<int> g_prefill_tiled{1};             // seedmi_set_option("prefill_tiled", 0|1): LDS-tiled prefill attention (0 = first-round kernel)
std::atomic<int> g_ablate_norm{0};               // seedmi_set_option("decode_ablate_norm", 1): timing only, skips the decode RMSNorm launches
// seedmi_set_option("decode_attn_early", 0|1|2): cached rows requested ahead of the rotation (1 keys, 2 keys + values).  Interleaved graphs
// (tools/decode_ab.py, profiles/r03_decode_ab.json): 3.454 / 3.415 ms per 8B step for 0 / 1 (twice), 2 = level with 0 (128 VGPRs)
std::atomic<int> g_decode_attn_early{1};
std::atomic<int> g_decode_fused{1};              // seedmi_set_option("decode_fused", 0|1): RoPE + KV append folded into decode attention

template <int EPI, int NW, bool NT, bool PACKED, int R>
int launch_skinny_r(const SkinnyParams& p, hipStream_t s) {
    const int grid = (p.N + 16 * R - 1) / (16 * R);
    const int mt = (p.M + 15) / 16;
    switch (mt) {
        case 1: hipLaunchKernelGGL((gemm_skinny_kernel<1, EPI, NW, NT, PACKED, R>), dim3(grid), dim3(64 * NW), 0, s, p); break;
        case 2: hipLaunchKernelGGL((gemm_skinny_kernel<2, EPI, NW, NT, PACKED, R>), dim3(grid), dim3(64 * NW), 0, s, p); break;
        case 3: hipLaunchKernelGGL((gemm_skinny_kernel<3, EPI, NW, NT, PACKED, R>), dim3(grid), dim3(64 * NW), 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_skinny_kernel<4, EPI, NW, NT, PACKED, R>), dim3(grid), dim3(64 * NW), 0, s, p); break;
    }
    return seedmi_check_launch("gemm_skinny");
}

template <int EPI, int NW, bool NT, bool PACKED>
int launch_skinny_nw(const SkinnyParams& p, hipStream_t s) {
    // R weight row-tiles (16 rows each) per workgroup.  A memory-bound launch loses whatever part of its last round of workgroups
    // is empty, so R is chosen for the fullest rounds: R = 1 runs two workgroups per CU (116 VGPRs), R >= 2 one; ties go to the
    // larger R (each activation fragment is reused R times).  8B QKV: 768 tiles -> R = 3 = exactly one workgroup per CU (+19 %).
    const int tiles = (p.N + 15) / 16;
    const int n_cu = seedmi_device_cus(seedmi_current_device());
    int best_r = 1;
    if (g_skinny_r != 0) {
        best_r = g_skinny_r;
    } else {
        double best_eff = 0.0;
        for (int r = 1; r <= 3; ++r) {
            if (tiles % r) continue;
            if (r == 3 && p.M > 32) continue;                       // register budget (two activation row tiles at most)
            if (r == 2 && tiles < 1024) continue;                   // (measured: R = 2 only pays on the widest projections)
            const int wgs = tiles / r, slots = n_cu * (r == 1 ? 2 : 1);
            const double eff = (double)wgs / ((double)((wgs + slots - 1) / slots) * slots);
            if (eff > best_eff + 1e-9 || (eff > best_eff - 1e-9 && r > best_r)) { best_eff = eff; best_r = r; }
        }
    }
    if (best_r == 3 && (tiles % 3) == 0 && p.M <= 32) return launch_skinny_r<EPI, NW, NT, PACKED, 3>(p, s);
    if (best_r == 2 && (tiles % 2) == 0) return launch_skinny_r<EPI, NW, NT, PACKED, 2>(p, s);
    return launch_skinny_r<EPI, NW, NT, PACKED, 1>(p, s);
}

template <int EPI, bool PACKED>
int launch_skinny(const SkinnyParams& p, hipStream_t s) {
    // 8-way K split when each slice still holds at least two 128-deep batches, else 4-way
    const bool w8 = g_skinny_nw == 8 || (g_skinny_nw == 0 && (p.K % 256) == 0 && p.K >= 2048);
    // (the temporal-load forms - seedmi_set_option("skinny_nt", 0), an A/B switch of round 1 - exist in the devtools build only: they were
    // half of this file's 288 gemm_skinny_kernel instantiations and of its compile time)
#ifdef SEEDMI_DEVTOOLS
    if (!g_skinny_nt) return (w8 && (p.K % 256) == 0) ? launch_skinny_nw<EPI, 8, false, PACKED>(p, s) : launch_skinny_nw<EPI, 4, false, PACKED>(p, s);
#endif
    if (w8 && (p.K % 256) == 0) return launch_skinny_nw<EPI, 8, true, PACKED>(p, s);
    return launch_skinny_nw<EPI, 4, true, PACKED>(p, s);
}

// ------------------------------------------------------------------------------------------------ skinny GEMM, balanced split-K form
// What round 3's ablations (profiles/r03_call10_decode_gemm_ablations.log) say a 16R-row workgroup loses against a plain stream of the same
// bytes: the ACTIVATION fragments (every workgroup pulls all of A out of L2: 2 KiB per KiB of weights at R = 1 - 7.5 of down_proj's
// 25.6 us, 6.4 of gate/up's 45.3), the one-wave epilogue (5.4 us of gate/up) and partly filled last rounds (gate/up: 2.69 rounds of
// workgroups).  This form fixes the three together:
//   * 64 weight rows per tile (R = 4): every activation fragment meets four weight fragments - 0.5 KiB of A per KiB of W;
//   * the (tile, k-step) space is cut into gridDim.x EQUAL contiguous ranges, one resident workgroup each (stream-K): no rounds, no
//     tail round, any N; a range is at most [tail of a tile][whole tiles][head of a tile];
//   * inside a segment the 8 waves split the k-steps, reduce through LDS, and wave f finishes FRAGMENT f (row tile r = f / MT,
//     activation tile t = f % MT): the epilogue is spread over all waves.
// A tile cut by a range boundary is finished by the workgroup that holds its HEAD (k-step 0; always that workgroup's last segment): the
// others publish their fp32 image (+ their part of the RMSNorm row sums) write-through and raise a flag, the owner adds them in
// workgroup order - a fixed summation order, so results are reproducible run to run.  Flags are consumed and cleared by the owner
// (the flag area is all-zero between launches: graph replays need no epoch).  All workgroups must be resident: the launcher sizes the
// grid as CUs x WGS and verifies once per kernel and device, with the occupancy query, that WGS workgroups fit a CU; residency can still
// be lost to a concurrent kernel of another stream, so waits are bounded and a lost partner is recorded in the STICKY error word (the
// last word of the flag area: the per-step clear leaves it alone, seedmi_gemm_skinny_ws_status / seedmi_llama_decode_status report it).
// Logical workgroup order runs XCD by XCD (q = (id % 8) * (G / 8) + id / 8), so an image is produced and consumed on one L2.
#ifndef SEEDMI_SK_NT
// weight loads of the split-K kernel: 1 = non-temporal (`nt`), 0 = plain (A/B builds: python -m seed_amd.build --variant sknt0 -DSEEDMI_SK_NT=0).
// Measured (profiles/r03_call23_decode_weight_load_hint.log): nt 20.3 / 10.5 / 33.8-36.8 / 22.9 us for q/k/v, o, gate/up, down against
// 23.3 / 10.7 / 37.5-39.4 / 24.2 plain, 8B step 3.39-3.49 vs 3.64-3.65 ms - plain weight lines push the activations out of the L2s.
// (A bare stream of 64-200 MiB is the other way round, plain 5-12 % faster: tools/probes/cache_retention_probe.hip - nothing there to keep.)
#define SEEDMI_SK_NT 1
#endif
struct SkinnySk {
    float* slabs;                // [G][SK2_SLAB_FLOATS]
    unsigned* flags;             // [G] + error word at SK2_FLAG_WORDS - 1
    int tiles;                   // 64-row tiles
    int tiles16;                 // 16-row tiles of the packed weight (ceil(N / 16))
    int ks;                      // k-steps (32 deep) per tile
    int per, rem;                // the (tile, k-step) space in gridDim.x ranges: `per` units each, the first `rem` ranges one more
    int cut;                     // any tile shared between workgroups?
    float ks_inv;                // 1 / ks (unit -> tile without an integer divide ahead of the first request)
};
constexpr int SK2_SLAB_FLOATS = 8 * 256 + 64;                  // 8 fragments x (64 lanes x 4) + row sums [2][16] (+ pad)
constexpr int SK2_FLAG_WORDS = 1024;
// word SK2_FLAG_WORDS - 1: the sticky error word; word SK2_FLAG_WORDS - 2: the tag seedmi_gemm_skinny_workspace_init wrote - a status call on a
// workspace that was never initialised reports THAT instead of whatever the error word's bytes happen to hold
constexpr int SK2_LIVE_WORDS = SK2_FLAG_WORDS - 2;          // hand-off flags proper (one per workgroup)
constexpr unsigned SK2_TAG = 0x5eed514bu;
constexpr size_t SK2_WS_BYTES = (size_t)SK2_FLAG_WORDS * 4 + (size_t)(SK2_FLAG_WORDS - 1) * SK2_SLAB_FLOATS * 4;

template <int EPI>
SEEDMI_DEVINL void skinny_epilogue_frag(const SkinnyParams& p, const f32x4 a, const int m, const int nb_, const float rstd, const uint2 resid,
                                        const bool resid_ok) {
    if (m >= p.M || nb_ >= p.N) return;
    float v[4] = {a[0] * rstd, a[1] * rstd, a[2] * rstd, a[3] * rstd};
    if (EPI == EPI_BIAS_RESIDUAL) {
        if (resid_ok) {
            v[0] = rbf(v[0]) + lo_bf(resid.x); v[1] = rbf(v[1]) + hi_bf(resid.x);
            v[2] = rbf(v[2]) + lo_bf(resid.y); v[3] = rbf(v[3]) + hi_bf(resid.y);
        } else {
            const bf16_t* rp = p.R + (size_t)m * p.ldr + nb_;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (nb_ + e < p.N) v[e] = rbf(v[e]) + bf2f(rp[e]);
        }
    }
    if (EPI == EPI_SWIGLU) {
        const int kc = nb_ >> 1;                                     // output column of the first (gate, up) pair
        bf16_t* cp = p.c_packed
            ? p.C + ((size_t)((m >> 4) * (p.N >> 6) + (kc >> 5)) * 64 + ((kc >> 3) & 3) * 16 + (m & 15)) * 8 + (kc & 7)
            : p.C + (size_t)m * p.ldc + kc;
        if (nb_ + 3 < p.N && (p.c_packed || (p.ldc % 2) == 0)) {
            *(uint32_t*)cp = pack2bf(rbf(silu(rbf(v[0]))) * rbf(v[1]), rbf(silu(rbf(v[2]))) * rbf(v[3]));      // (kc is even: 4-byte aligned)
        } else {
            if (nb_ + 1 < p.N) *cp = f2bf(rbf(silu(rbf(v[0]))) * rbf(v[1]));
            if (nb_ + 3 < p.N) cp[1] = f2bf(rbf(silu(rbf(v[2]))) * rbf(v[3]));
        }
    } else {
        bf16_t* cp = p.C + (size_t)m * p.ldc + nb_;
        if (nb_ + 4 <= p.N && (p.ldc % 4) == 0) {
            uint2 w;
            w.x = pack2bf(v[0], v[1]);
            w.y = pack2bf(v[2], v[3]);
            *(uint2*)cp = w;
            if (p.xp_out)                                             // the next GEMM's fragment-major A operand (row length N)
                *(uint2*)(p.xp_out + ((size_t)((m >> 4) * (p.N >> 5) + (nb_ >> 5)) * 64 + ((nb_ >> 3) & 3) * 16 + (m & 15)) * 8 + (nb_ & 7)) = w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (nb_ + e < p.N) cp[e] = f2bf(v[e]);
        }
    }
}

template <int MT, int EPI, int R, int WGS = 1>
__global__ __launch_bounds__(512, WGS == 2 ? 4 : 1) void gemm_skinny_sk_kernel(const SkinnyParams p, const SkinnySk x) {
    // WGS = 2: two workgroups per CU (<= 128 VGPRs: one k-step per register set), so that one streams while the other reduces
    constexpr int NW = 8, U = WGS == 2 ? 1 : (R * MT > 4) ? 2 : 4, NF = R * MT;
    static_assert(NF <= NW, "one fragment per wave");
    __shared__ __attribute__((aligned(16))) float red[NW][NF][64][4];
    __shared__ float red_ss[NW][MT][16];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    // (32-bit index arithmetic, no integer divides ahead of the first request: ~130 instructions / 0.5 us of the first form's prologue)
    const bool cut = x.cut != 0;                                                                          // (uniform)
    const int q = (cut && (G % 8) == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;    // XCD-contiguous order
    auto range_start = [&](const int c) { return c * x.per + min(c, x.rem); };
    const int u_end = range_start(q + 1);
    const bool do_norm = p.norm_eps > 0.f;
    const int fr = wave / MT, ft = wave % MT;                        // the fragment this wave finishes (waves >= NF: none)
    // segment state: tile J, k-steps [ka, kb), this wave's share [s0, s0 + n) and its operand streams
    int J, ka, kb, n;
    const bf16_t* wp[R];
    const bf16_t* ap[MT];
    auto setup = [&](const int u) {
        J = (int)((float)u * x.ks_inv);                                // (u < 2^24: exact up to one, fixed below)
        J += ((J + 1) * x.ks <= u) - (J * x.ks > u);
        ka = u - J * x.ks;
        kb = (u_end - u < x.ks - ka) ? ka + (u_end - u) : x.ks;
        const int L = kb - ka;
        const int s0 = ka + L * wave / NW;
        n = ka + L * (wave + 1) / NW - s0;
#pragma unroll
        for (int r = 0; r < R; ++r)
            wp[r] = p.W + ((size_t)min(R * J + r, x.tiles16 - 1) * x.ks + s0) * 512;     // (rows past N: re-read the last tile, never stored)
#pragma unroll
        for (int t = 0; t < MT; ++t) ap[t] = p.A + ((size_t)t * x.ks + s0) * 512;               // (wave-uniform bases: scalar registers)
    };
    const int loff = lane * 8;
    bf16x8 w0[U][R], a0[U][MT], w1[U][R], a1[U][MT];
    auto load = [&](bf16x8 (&wf)[U][R], bf16x8 (&af)[U][MT], int b) {
#pragma unroll
        for (int u2 = 0; u2 < U; ++u2) {
            const int kk = min(U * b + u2, n - 1) * 512 + loff;          // clamped: out-of-range steps are skipped below
#pragma unroll
#if SEEDMI_SK_NT
            for (int r = 0; r < R; ++r) wf[u2][r] = __builtin_nontemporal_load((const bf16x8*)(wp[r] + kk));
#else
            for (int r = 0; r < R; ++r) wf[u2][r] = *(const bf16x8*)(wp[r] + kk);
#endif
#pragma unroll
            for (int t = 0; t < MT; ++t) af[u2][t] = *(const bf16x8*)(ap[t] + kk);
        }
    };
    int u = range_start(q);
    setup(u);
    if (n > 0) load(w0, a0, 0);
    for (;;) {
        const int cJ = J, cka = ka, ckb = kb;
        const bool owner = cka == 0;
        // the owner's residual values, requested before the stream starts (the epilogue used to wait for them at the very end)
        const int e_m = 16 * ft + li, e_n = 16 * R * cJ + 16 * fr + 4 * g;
        uint2 resid = make_uint2(0u, 0u);
        const bool resid_ok = EPI == EPI_BIAS_RESIDUAL && (p.ldr % 4) == 0 && ((uintptr_t)p.R & 7) == 0 && e_n + 4 <= p.N;
        if (EPI == EPI_BIAS_RESIDUAL && owner && wave < NF && resid_ok && e_m < p.M)
            resid = *(const uint2*)(p.R + (size_t)e_m * p.ldr + e_n);
        f32x4 acc[R][MT];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float ss[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) ss[t] = 0.f;
        if (n > 0) {
            const int nb = (n + U - 1) / U;
            auto compute = [&](bf16x8 (&wf)[U][R], bf16x8 (&af)[U][MT], int b) {
#pragma unroll
                for (int u2 = 0; u2 < U; ++u2) {
                    if (U * b + u2 < n) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int t = 0; t < MT; ++t)
                                acc[r][t] = seedmi_mfma_16x16x32(wf[u2][r], af[u2][t], acc[r][t]);
                        if (do_norm) {
#pragma unroll
                            for (int t = 0; t < MT; ++t) {
                                const uint4 xw = __builtin_bit_cast(uint4, af[u2][t]);
                                const uint32_t w4[4] = {xw.x, xw.y, xw.z, xw.w};
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float lo = lo_bf(w4[i]), hi = hi_bf(w4[i]);
                                    ss[t] = fmaf(lo, lo, fmaf(hi, hi, ss[t]));
                                }
                            }
                        }
                    }
                }
            };
            for (int b = 0; b < nb; b += 2) {                         // (batch 0 was requested before the previous segment's reduction)
                if (b + 1 < nb) load(w1, a1, b + 1);
                compute(w0, a0, b);
                if (b + 2 < nb) load(w0, a0, b + 2);
                if (b + 1 < nb) compute(w1, a1, b + 1);
            }
        }
        u += ckb - cka;
        const bool more = u < u_end;
        if (more) {                                                   // the next segment's first batch streams in behind the reduction below
            setup(u);
            if (n > 0) load(w0, a0, 0);
        }
        if (do_norm) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {                            // lanes li + 16 g hold the four k-quarters of row 16t + li
                ss[t] += __shfl_xor(ss[t], 16, 64);
                ss[t] += __shfl_xor(ss[t], 32, 64);
                if (g == 0) red_ss[wave][t][li] = ss[t];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int t = 0; t < MT; ++t) *(f32x4*)&red[wave][r * MT + t][lane][0] = acc[r][t];
        __syncthreads();
        f32x4 fa = (f32x4){0.f, 0.f, 0.f, 0.f};
        float tot = 0.f;
        if (wave < NF) {
            fa = *(const f32x4*)&red[0][wave][lane][0];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const f32x4 o = *(const f32x4*)&red[w][wave][lane][0];
#pragma unroll
                for (int e = 0; e < 4; ++e) fa[e] += o[e];
            }
            if (do_norm) {
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += red_ss[w][ft][li];
            }
        }
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        if (!owner) {
            // ---- tail of a tile whose head another workgroup holds: publish.  Protocol of the CDNA guide (G16 R1, write-through form):
            //      sc1 stores -> EVERY wave drains vmcnt -> workgroup barrier -> one relaxed agent-scope flag store.  (A release fence per
            //      workgroup + an acquire in the owner, this kernel's first form, wrote back / invalidated whole L2s: 4x slower.)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(x.slabs + (size_t)q * SK2_SLAB_FLOATS, 0, SK2_SLAB_FLOATS * 4,
                                                                                 0x00020000);
            if (wave < NF) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, fa), rs, (wave * 256 + lane * 4) * 4, 0, /*sc1*/ 16);
                if (do_norm && fr == 0 && g == 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tot), rs, (8 * 256 + ft * 16 + li) * 4, 0, /*sc1*/ 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(x.flags + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (ckb < x.ks) {
                // ---- head of a cut tile: add the later workgroups' images in workgroup order (relaxed poll by one lane, barrier, then
                //      sc1 loads: they pass this CU's L1 and see the write-through data wherever the publisher ran)
                const int tile_end = (cJ + 1) * x.ks;
                for (int c = q + 1; c < G && range_start(c) < tile_end; ++c) {
                    if (tid == 0) {
                        int spins = 0;
                        while (__hip_atomic_load(x.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1 << 22))
                            __builtin_amdgcn_s_sleep(2);
                        // consumed: cleared, so that the flag area is all-zero again when the launch ends (graph replays need no epoch)
                        if (spins < (1 << 22)) __hip_atomic_store(x.flags + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else __hip_atomic_store(x.flags + (SK2_FLAG_WORDS - 1), 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __syncthreads();
                    if (wave < NF) {
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(x.slabs + (size_t)c * SK2_SLAB_FLOATS, 0,
                                                                                             SK2_SLAB_FLOATS * 4, 0x00020000);
                        const f32x4 o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (wave * 256 + lane * 4) * 4, 0, /*sc1*/ 16));
                        float ot = 0.f;
                        if (do_norm) ot = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (8 * 256 + ft * 16 + li) * 4, 0, /*sc1*/ 16));
#pragma unroll
                        for (int e = 0; e < 4; ++e) fa[e] += o[e];
                        tot += ot;
                    }
                }
            }
            if (wave < NF) {
                // LlamaRMSNorm: variance = mean(x^2) in fp32 (llama_xformer.py:108-110)
                const float rstd = do_norm ? rsqrtf(tot / (float)p.K + p.norm_eps) : 1.f;
                skinny_epilogue_frag<EPI>(p, fa, e_m, e_n, rstd, resid, resid_ok);
            }
        }
        if (!more) break;
        __syncthreads();                                              // (the LDS images are rewritten by the next segment)
    }
}

// seedmi_set_option("skinny_splitk", 0..3): 0 = the round-1 kernel (one tile per workgroup, one-wave epilogue) always; 1 = this kernel, uncut
// where the shape divides, cut otherwise (see launch_skinny_sk); 2 = this kernel, always cut (R = 4); 3 = this kernel only where the round-1
// form's last round of workgroups would be under 95 % full (8B: gate/up, 2.69 rounds - 45.4 -> 35.6 us), the round-1 kernel elsewhere
std::atomic<int> g_skinny_sk{1};
// seedmi_set_option("prefill_streamk", 0|1|2): stream-K tail for the prefill's MFMA GEMMs - 0 off (default), 1 every GEMM, 2 only the GEMMs
// whose data-parallel walk is short and ends in a badly filled round (< 3 rounds of 256x256 tiles, the last one under 75 % full: o_proj and
// down at 8 x 649 tokens of 14B are 1.64 rounds).  Measured at that shape (profiles/r04_call4_prefill_streamk.log,
// r04_call5_prefill_streamk_selective.log): 107.9 / 112.0 ms without, 117.2 ms with 2, 124.9 / 127.0 ms with 1 - the hand-off (a 256 KiB
// fp32 image written through and read back per shared tile, a pipeline refill per segment) costs more than the badly filled round wastes,
// for the short GEMMs too.  Bit-identical logits in every mode.
// Devtools build only since round 5: the product workspace no longer carries the 64 MiB stream-K area, and nothing checked the GEMM's sticky
// stream-K error word on this path.
#ifdef SEEDMI_DEVTOOLS
std::atomic<int> g_prefill_streamk{0};
#endif

// best fill of the last round of workgroups over the row-tile counts launch_skinny_nw may pick (R = 1..3)
bool skinny_rounds_underfilled(int M, int N) {
    const int tiles = (N + 15) / 16, n_cu = seedmi_device_cus(seedmi_current_device());
    double best = 0.0;
    for (int r = 1; r <= 3; ++r) {
        if ((tiles % r) || (r == 3 && M > 32) || (r == 2 && tiles < 1024)) continue;
        const int wgs = tiles / r;
        const double fill = (double)wgs / ((double)((wgs + n_cu - 1) / n_cu) * n_cu);
        if (fill > best) best = fill;
    }
    return best < 0.95;
}

template <int EPI, int R, int WGS = 1>
int launch_skinny_sk_r(const SkinnyParams& p, void* ws, int grid, hipStream_t s) {
    SkinnySk x;
    x.flags = (unsigned*)ws;
    x.slabs = (float*)((char*)ws + SK2_FLAG_WORDS * 4);
    x.tiles16 = (p.N + 15) / 16;
    x.tiles = (x.tiles16 + R - 1) / R;
    x.ks = p.K / 32;
    const long long total = (long long)x.tiles * x.ks;
    if (grid > total) grid = (int)total;
    if (total >= (1ll << 24)) {                                      // (unit -> tile goes through an fp32 product; 8B lm_head: 629 x 128 = 80 512 units)
        seedmi_set_error("seedmi_gemm_skinny: N=%d K=%d is beyond the split-K kernel's index range", p.N, p.K);
        return SEEDMI_E_SHAPE;
    }
    x.per = (int)(total / grid);
    x.rem = (int)(total % grid);
    x.cut = (x.rem != 0 || (x.per % x.ks) != 0) ? 1 : 0;
    x.ks_inv = 1.0f / (float)x.ks;
    if (x.cut) {
        // a cut tile's owner waits for its partners: every workgroup of the grid must be resident.  Checked once per device for this
        // instantiation (the answer depends on the kernel's registers / LDS only)
        static std::atomic<int> fits[SEEDMI_MAX_DEVICES] = {};
        const int dev = seedmi_current_device();
        int ok = fits[dev].load(std::memory_order_relaxed);
        if (!ok) {
            int n1 = 0, n2 = 0;
            const hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, gemm_skinny_sk_kernel<1, EPI, R, WGS>, 512, 0);
            const hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, gemm_skinny_sk_kernel<2, EPI, R, WGS>, 512, 0);
            ok = (e1 == hipSuccess && e2 == hipSuccess && n1 >= WGS && n2 >= WGS) ? 1 : -1;
            fits[dev].store(ok, std::memory_order_relaxed);
        }
        if (ok < 0 || grid > WGS * seedmi_device_cus(dev)) {
            seedmi_set_error("seedmi_gemm_skinny: the split-K kernel's grid of %d workgroups is not resident on this device", grid);
            return SEEDMI_E_HIP;
        }
    }
    if (p.M <= 16) hipLaunchKernelGGL((gemm_skinny_sk_kernel<1, EPI, R, WGS>), dim3(grid), dim3(512), 0, s, p, x);
    else hipLaunchKernelGGL((gemm_skinny_sk_kernel<2, EPI, R, WGS>), dim3(grid), dim3(512), 0, s, p, x);
    return seedmi_check_launch("gemm_skinny_sk");
}

// Row tiles per workgroup tile.  A shape whose 16-row tiles divide into whole R-tiles with a whole number of them per workgroup runs
// UNCUT (no hand-off at all: 8B q/k/v = 768 tiles -> R = 3, one 48-row tile per CU; o_proj / down = 256 tiles -> R = 1): the kernel is then
// the one-tile form with the epilogue spread over the waves and the residual requested up front.  Everything else (8B gate/up: 1376
// tiles, lm_head: 2513) takes R = 4 and the balanced cut.  `force_cut` (seedmi_set_option("skinny_splitk", 2)): R = 4 for every shape.
template <int EPI>
int launch_skinny_sk(const SkinnyParams& p, void* ws, int mode, hipStream_t s) {
    const bool force_cut = mode == 2;
    int grid = seedmi_device_cus(seedmi_current_device());          // one 512-thread workgroup per CU (up to 182 VGPRs): every workgroup is resident
    if (grid > SK2_LIVE_WORDS) grid = SK2_LIVE_WORDS;
    const int tiles16 = (p.N + 15) / 16;
#ifdef SEEDMI_DEVTOOLS
    if (mode == 4 && 2 * grid <= SK2_LIVE_WORDS) return launch_skinny_sk_r<EPI, 4, 2>(p, ws, 2 * grid, s);      // (experiment: 2 workgroups per CU)
#endif
    if (!force_cut) {
        if ((tiles16 % 3) == 0 && ((tiles16 / 3) % grid) == 0) return launch_skinny_sk_r<EPI, 3>(p, ws, grid, s);
        if ((tiles16 % grid) == 0) return launch_skinny_sk_r<EPI, 1>(p, ws, grid, s);
    }
    return launch_skinny_sk_r<EPI, 4>(p, ws, grid, s);
}

// W [N, K] row-major -> fragment-major [ceil(N/16)][K/32][64 lanes][8]; rows beyond N are zero
__global__ void pack_skinny_kernel(const bf16_t* __restrict__ W, int ldw, int N, int K, bf16_t* __restrict__ out) {
    const long long total = (long long)((N + 15) / 16) * (K / 32) * 64;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long long blk = idx >> 6;
        const int s = (int)(blk % (K / 32));
        const int j = (int)(blk / (K / 32));
        const int row = 16 * j + (lane & 15), col = 32 * s + 8 * (lane >> 4);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < N) v = *(const uint4*)(W + (size_t)row * ldw + col);
        *(uint4*)(out + idx * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------ decode attention
// T == 1: one workgroup per (batch, head).  Memory bound: K and V rows (256 B) are read once with 16-B accesses.
constexpr int DEC_HD = 128;

__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q, int ldq,
                                                          const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
                                                          bf16_t* __restrict__ out, int ldo, int H, int tmax, int kv_len_arg,
                                                          float scale, int out_packed, int lds_len,
                                                          const int* __restrict__ past_dev) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int kv_len = past_dev ? *past_dev + 1 : kv_len_arg;     // device-resident length: graph-replayable decode step
    float* sc = dsm;                                // [kv_len] scores -> probabilities
    float* part = dsm + lds_len;                    // [16][128] partial outputs
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
        const int kcol = h * DEC_HD + tid;
        const size_t off = out_packed
            ? ((size_t)((b >> 4) * ((H * DEC_HD) >> 5) + (kcol >> 5)) * 64 + ((kcol >> 3) & 3) * 16 + (b & 15)) * 8 + (kcol & 7)
            : (size_t)b * ldo + kcol;
        out[off] = f2bf(a);
    }
}

// Decode step with RoPE and the KV-cache append folded in (T == 1): the workgroup of (b, h) rotates its own q and the new
// key (same half-precision expression and rounding points as rope_kv_append_kernel, llama_xformer.py:147-168), stores the new
// key / value row in the cache for the following steps, and attends over the cached rows plus the new one taken from
// registers.  One launch per layer instead of two and no q round trip; results are bit-identical to the two-kernel form.
// One (batch row, head) item of the fused decode attention, run by 256 threads (tid = 0..255) that share `dsm` (lds_len + 16 * 128
// floats) and `wred` (4 floats).  Called by the stand-alone kernel (one item per workgroup) and by the persistent decode kernel (two
// 256-thread halves of a workgroup, each with its own LDS slices; every thread of the workgroup reaches the four barriers whether its
// half has an item (`active`) or not).
// EARLY (stand-alone kernel, seedmi_set_option("decode_attn_early")): 1 = the first batch of cached KEY rows is requested before the
// rotation (whose operands come from a different buffer - the kernel used to pay that round trip first, and the cache append stores that
// follow the rotation kept hipcc from hoisting the loads itself); 2 = the first batch of VALUE rows too.  Same arithmetic, same order.
template <bool WT = false, int EARLY = 0>
SEEDMI_DEVINL void attn_decode_item(const int item, const int tid, float* dsm, float* wred, const bool active,
                                    const bf16_t* __restrict__ qkv, int ldqkv, const long long* __restrict__ pos_ids,
                                    const bf16_t* __restrict__ cos_t, const bf16_t* __restrict__ sin_t, bf16_t* __restrict__ kc,
                                    bf16_t* __restrict__ vc, bf16_t* __restrict__ out, int ldo, int H, int tmax, int past_arg, float scale,
                                    int out_packed, int lds_len, const int* __restrict__ past_dev, int max_pos, int past_stride) {
    // (a graph replayed past the cache capacity keeps rewriting the last row instead of leaving the allocation)
    // past_stride 0: one cache length for the whole batch; 1: one per row (continuous batching: every slot at its own position)
    const int past = past_dev ? min(past_dev[(item / H) * past_stride], tmax - 1) : past_arg;
    const int kv_len = active ? past + 1 : 0;                     // (an idle half walks the barriers with empty loops)
    float* sc = dsm;
    float* part = dsm + lds_len;
    const int lane = tid & 63, wave = tid >> 6;
    const int b = item / H, h = item % H;
    const int c = tid & 15;                         // 16-B chunk of the head dim: elements 8c .. 8c+7
    const int ks = tid >> 4;                        // key slot 0..15
    bf16_t* kb = kc + ((size_t)b * H + h) * tmax * DEC_HD;
    bf16_t* vb = vc + ((size_t)b * H + h) * tmax * DEC_HD;
    long long pos = pos_ids ? pos_ids[b] : (long long)past;
    pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);           // cos/sin tables have max_pos rows
    constexpr int DU = 8;
    uint4 kr0[DU], vr0[DU];                         // EARLY: the first batch of this thread's key / value rows (j = ks + 16 u < past)
    if (EARLY >= 1) {
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int j = ks + 16 * u;
            if (j < kv_len - 1) kr0[u] = *(const uint4*)(kb + (size_t)j * DEC_HD + 8 * c);
        }
    }
    if (EARLY >= 2) {
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int j = ks + 16 * u;
            if (j < kv_len - 1) vr0[u] = *(const uint4*)(vb + (size_t)j * DEC_HD + 8 * c);
        }
    }
    // rotate q and the new key: element i pairs with i +- 64, i.e. chunk c with chunk c ^ 8
    float qv[8], kn[8];
    uint4 vnew = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = kn[i] = 0.f;
    if (active) {
        const bf16_t* row = qkv + (size_t)b * ldqkv + h * DEC_HD;
        const bool lo = c < 8;                      // first half: x*cos + (-partner)*sin ; second half: x*cos + partner*sin
        const uint4 uc = *(const uint4*)(cos_t + pos * DEC_HD + 8 * c);
        const uint4 us = *(const uint4*)(sin_t + pos * DEC_HD + 8 * c);
        const uint32_t cw[4] = {uc.x, uc.y, uc.z, uc.w}, sw[4] = {us.x, us.y, us.z, us.w};
        auto rot = [&](const bf16_t* x, float (&dst)[8]) {
            const uint4 a = *(const uint4*)(x + 8 * c), p2 = *(const uint4*)(x + 8 * (c ^ 8));
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, pw[4] = {p2.x, p2.y, p2.z, p2.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x0 = lo_bf(aw[i]), x1 = hi_bf(aw[i]);
                const float y0 = lo ? -lo_bf(pw[i]) : lo_bf(pw[i]), y1 = lo ? -hi_bf(pw[i]) : hi_bf(pw[i]);
                dst[2 * i] = rbf(rbf(x0 * lo_bf(cw[i])) + rbf(y0 * lo_bf(sw[i])));
                dst[2 * i + 1] = rbf(rbf(x1 * hi_bf(cw[i])) + rbf(y1 * hi_bf(sw[i])));
            }
        };
        rot(row, qv);
        rot(row + H * DEC_HD, kn);
        vnew = *(const uint4*)(row + 2 * H * DEC_HD + 8 * c);
        if (ks == 0) {                              // append for the following steps
            uint4 kw;
            kw.x = pack2bf(kn[0], kn[1]); kw.y = pack2bf(kn[2], kn[3]); kw.z = pack2bf(kn[4], kn[5]); kw.w = pack2bf(kn[6], kn[7]);
            *(uint4*)(kb + (size_t)past * DEC_HD + 8 * c) = kw;
            *(uint4*)(vb + (size_t)past * DEC_HD + 8 * c) = vnew;
        }
    }
    // scores: cached keys from HBM, the new key (j == past) from registers.  The key rows of a thread (j = ks, ks + 16, ...) are
    // requested DU at a time before any of them is used: one exposed memory latency per DU rows instead of one per row (at ctx 123 the
    // un-batched loop was 8 dependent round trips for QK^T and 8 more for PV - most of this kernel's 20 us)
    // (round 2: requesting the first DU value rows together with the key rows, AFTER the rotation, was tried: 16.4 vs 14.4 us per launch at
    // ctx 123; 116 instead of 78 VGPRs and twice the loads ahead of the first use)
    float lmax = -INFINITY;
    for (int j0 = ks; j0 < kv_len; j0 += 16 * DU) {
        uint4 kr[DU];
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int j = j0 + 16 * u;
            if (EARLY >= 1 && j0 == ks) kr[u] = kr0[u];
            else if (j < past) kr[u] = *(const uint4*)(kb + (size_t)j * DEC_HD + 8 * c);
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int j = j0 + 16 * u;
            if (j >= kv_len) break;
            float d = 0.f;
            if (j < past) {
                const uint32_t w[4] = {kr[u].x, kr[u].y, kr[u].z, kr[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) d += qv[2 * i] * lo_bf(w[i]) + qv[2 * i + 1] * hi_bf(w[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) d += qv[2 * i] * kn[2 * i] + qv[2 * i + 1] * kn[2 * i + 1];
            }
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64);
            d *= scale;
            if (c == 0) sc[j] = d;
            lmax = fmaxf(lmax, d);
        }
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
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j0 = ks; j0 < kv_len; j0 += 16 * DU) {
        uint4 vr[DU];
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int j = j0 + 16 * u;
            vr[u] = vnew;
            if (EARLY >= 2 && j0 == ks) { if (j < past) vr[u] = vr0[u]; }
            else if (j < past) vr[u] = *(const uint4*)(vb + (size_t)j * DEC_HD + 8 * c);
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int j = j0 + 16 * u;
            if (j >= kv_len) break;
            const float pj = rbf(sc[j] * inv);
            const uint32_t w[4] = {vr[u].x, vr[u].y, vr[u].z, vr[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { o[2 * i] += pj * lo_bf(w[i]); o[2 * i + 1] += pj * hi_bf(w[i]); }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[ks * DEC_HD + 8 * c + i] = o[i];
    __syncthreads();
    if (active && tid < DEC_HD) {
        float a = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) a += part[s2 * DEC_HD + tid];
        const int kcol = h * DEC_HD + tid;
        const size_t off = out_packed
            ? ((size_t)((b >> 4) * ((H * DEC_HD) >> 5) + (kcol >> 5)) * 64 + ((kcol >> 3) & 3) * 16 + (b & 15)) * 8 + (kcol & 7)
            : (size_t)b * ldo + kcol;
        st_b16<WT>(out + off, f2bf(a));
    }
}

template <int EARLY>
__global__ __launch_bounds__(256) void attn_decode_rope_kernel(const bf16_t* __restrict__ qkv, int ldqkv,
                                                               const long long* __restrict__ pos_ids,
                                                               const bf16_t* __restrict__ cos_t, const bf16_t* __restrict__ sin_t,
                                                               bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                               bf16_t* __restrict__ out, int ldo, int H, int tmax, int past_arg,
                                                               float scale, int out_packed, int lds_len,
                                                               const int* __restrict__ past_dev, int max_pos, int past_stride) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    __shared__ float wred[4];
    attn_decode_item<false, EARLY>((int)blockIdx.x, (int)threadIdx.x, dsm, wred, true, qkv, ldqkv, pos_ids, cos_t, sin_t, kc, vc, out, ldo, H,
                                   tmax, past_arg, scale, out_packed, lds_len, past_dev, max_pos, past_stride);
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
            s = seedmi_mfma_16x16x32(kf, qf[ks], s);
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
                s[half] = seedmi_mfma_16x16x32(kf, qf[ks], s[half]);
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
            o[n] = seedmi_mfma_16x16x32(__builtin_bit_cast(bf16x8, vw), pf, o[n]);
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
#ifdef SEEDMI_DEVTOOLS
// ------------------------------------------------------------------------------------------------ persistent decode layers
// (-DSEEDMI_DEVTOOLS build only: measured slower than one launch per phase, see DESIGN.md "persistent decode layers")
// All decoder layers of one decode step (T == 1, M <= 32, folded RMSNorm, fragment-major operands) in ONE launch of one 512-thread
// workgroup per CU.  Per layer five phases - [norm1 + QKV] -> [RoPE + append + attention] -> [o_proj + residual] -> [norm2 + gate/up +
// SwiGLU] -> [down + residual] (llama_xformer.py:280-332) - separated by a grid barrier instead of a kernel boundary: the same tile
// functions as the stand-alone kernels (skinny_tile, attn_decode_item), so the arithmetic of a step is bit-identical to the five-launch
// form.  What the barrier buys over a kernel boundary: a workgroup issues the FIRST weight loads of its next phase before it waits
// (weights do not depend on the previous phase; only the activations do), so the HBM stream does not restart from an empty pipeline
// 160 times per 8B step.
// Visibility: the phase outputs (qkv, att, xn, act) are written by one XCD and read by all.  They are stored write-through (sc0 sc1)
// into buffers of their own PER LAYER, and every wave drains vmcnt before the barrier: an address that is written once per launch has
// no stale copy in another XCD's L2 or in any L1, so no phase needs a cache invalidate (agent-scope release + acquire fences in every
// wave cost 52 us per barrier - 2048 L2 write-backs; one acquiring wave per workgroup still 12 us - 32 L2 invalidates per XCD).
// The residual stream x is the exception: re-written in place, but column block j is only ever touched by workgroup j (o_proj and
// down use the same tiling), which reads it past its L1 (sc0).  The barrier is a
// monotonic arrival counter in the workspace (zeroed by a memset node ahead of the launch), spins are bounded: a lost workgroup leaves
// an error flag and wrong logits, not a hang.
struct MegaLayer { const bf16_t *qkv_wp, *o_wp, *gu_wp, *down_wp; bf16_t *kc, *vc; };
constexpr int MEGA_MAX_LAYERS = 64;
struct MegaParams {
    int M, h, F, H, tmax, max_pos, layers, past_stride, past_arg, lds_len;
    int probe;                           // timing probes: 2 = barriers only, 3 = phases without barriers (results invalid)
    float eps, scale;
    bf16_t *x, *xn;                      // residual stream (row-major) and its fragment-major copy as layer 0 reads it
    char* act_base;                      // per-layer buffers: [qkv | att | xn_mid | act | xn_out], act_stride bytes per layer
    size_t act_stride;
    const bf16_t *cos_t, *sin_t;
    const long long* pos_ids;
    const int* past_dev;
    unsigned* bar;
    MegaLayer layer[MEGA_MAX_LAYERS];
};

SEEDMI_DEVINL void grid_barrier(unsigned* bar, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's write-through stores have reached memory
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1 << 22)) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

template <int MT>
__global__ __launch_bounds__(512) void decode_layers_kernel(const MegaParams p) {
    extern __shared__ __attribute__((aligned(16))) float mega_dsm[];
    __shared__ float mega_wred[2][4];
    const int tid = threadIdx.x, half = tid >> 8, tid2 = tid & 255;
    const int G = gridDim.x;
    unsigned phase = 0;
    float* dsm_half = mega_dsm + half * (p.lds_len + 16 * DEC_HD);
    SkinnyParams sp;
    sp.M = p.M; sp.lda = 0; sp.ldw = 8; sp.a_packed = 1; sp.abl = 0;
    const size_t Mp = (size_t)((p.M + 15) / 16 * 16);
    const bf16_t* xn_in = p.xn;
    for (int l = 0; l < p.layers; ++l) {
        const MegaLayer& L = p.layer[l];
        bf16_t* const qkv = (bf16_t*)(p.act_base + (size_t)l * p.act_stride);
        bf16_t* const att = qkv + Mp * 3 * p.h;
        bf16_t* const xn_mid = att + Mp * p.h;
        bf16_t* const act = xn_mid + Mp * p.h;
        bf16_t* const xn_out = act + Mp * p.F;
        // ---- input_layernorm + q/k/v projections (llama_xformer.py:298-299, 223-225)
        sp.N = 3 * p.h; sp.K = p.h; sp.A = xn_in; sp.W = L.qkv_wp; sp.R = nullptr; sp.ldr = 0; sp.C = qkv; sp.ldc = 3 * p.h;
        sp.c_packed = 0; sp.norm_eps = p.eps; sp.xp_out = nullptr;
        for (int j = blockIdx.x; j < sp.N / 48 && p.probe != 2; j += G) { skinny_tile<MT, EPI_NONE, 8, true, true, 3, true>(sp, j); __syncthreads(); }
        if (p.probe != 3) grid_barrier(p.bar, ++phase * G);
        // ---- rotary embedding, cache append, attention over the cache (llama_xformer.py:228-256)
        {
            const int items = p.M * p.H;
            for (int i0 = 2 * blockIdx.x; i0 < items && p.probe != 2; i0 += 2 * G) {
                const int item = i0 + half;
                const bool active = item < items;
                attn_decode_item<true>(active ? item : 0, tid2, dsm_half, mega_wred[half], active, qkv, 3 * p.h, p.pos_ids, p.cos_t, p.sin_t,
                                       L.kc, L.vc, att, p.h, p.H, p.tmax, p.past_arg, p.scale, 1, p.lds_len, p.past_dev, p.max_pos,
                                       p.past_stride);
                __syncthreads();                                   // (the LDS slices are reused by the next item)
            }
        }
        if (p.probe != 3) grid_barrier(p.bar, ++phase * G);
        // ---- o_proj + residual (llama_xformer.py:258, 316); also writes the fragment-major copy of the new residual stream
        sp.N = p.h; sp.K = p.h; sp.A = att; sp.W = L.o_wp; sp.R = p.x; sp.ldr = p.h; sp.C = p.x; sp.ldc = p.h; sp.c_packed = 0;
        sp.norm_eps = 0.f; sp.xp_out = xn_mid;
        for (int j = blockIdx.x; j < sp.N / 16 && p.probe != 2; j += G) { skinny_tile<MT, EPI_BIAS_RESIDUAL, 8, true, true, 1, true>(sp, j); __syncthreads(); }
        if (p.probe != 3) grid_barrier(p.bar, ++phase * G);
        // ---- post_attention_layernorm + gate/up + SwiGLU (llama_xformer.py:320-321, 186)
        sp.N = 2 * p.F; sp.K = p.h; sp.A = xn_mid; sp.W = L.gu_wp; sp.R = nullptr; sp.ldr = 0; sp.C = act; sp.ldc = p.F; sp.c_packed = 1;
        sp.norm_eps = p.eps; sp.xp_out = nullptr;
        for (int j = blockIdx.x; j < sp.N / 32 && p.probe != 2; j += G) { skinny_tile<MT, EPI_SWIGLU, 8, true, true, 2, true>(sp, j); __syncthreads(); }
        if (p.probe != 3) grid_barrier(p.bar, ++phase * G);
        // ---- down_proj + residual (llama_xformer.py:186, 322)
        sp.N = p.h; sp.K = p.F; sp.A = act; sp.W = L.down_wp; sp.R = p.x; sp.ldr = p.h; sp.C = p.x; sp.ldc = p.h; sp.c_packed = 0;
        sp.norm_eps = 0.f; sp.xp_out = (l + 1 < p.layers) ? xn_out : p.xn;           // (the last layer's copy is what lm_head reads)
        for (int j = blockIdx.x; j < sp.N / 16 && p.probe != 2; j += G) { skinny_tile<MT, EPI_BIAS_RESIDUAL, 8, true, true, 1, true>(sp, j); __syncthreads(); }
        if (p.probe != 3) grid_barrier(p.bar, ++phase * G);
        xn_in = xn_out;
    }
}

std::atomic<int> g_decode_mega{0};               // seedmi_set_option("decode_persistent", 0|1): all layers of a decode step in one persistent launch
#endif

struct LlamaWs { bf16_t *x, *xn, *qkv, *q, *att, *act; unsigned* bar; void* sk; void* gsk; size_t gsk_bytes; bf16_t* mega; size_t mega_stride; size_t bytes; };
LlamaWs carve(const seedmi_llama_weights_t* w, int B, int T, void* ws) {
    const size_t M = (size_t)B * T, h = w->hidden, F = w->ffn;
    const size_t Mp = (M + 15) / 16 * 16;            // fragment-major buffers hold whole 16-row tiles
    Carver c(ws);
    LlamaWs t;
    // split-K decode GEMMs: flag words + fp32 images of the cut tiles.  FIRST and at the same offset for every (B, T): a workspace that
    // serves prefills and decode steps alike (LlamaEngine's) must not have a prefill's activations land on the flag words - the sticky
    // error word among them is cleared by nobody but seedmi_llama_decode_status (zeroed once by the caller after allocation)
    void* const sk_area = c.take(SK2_WS_BYTES);
    t.gsk = nullptr; t.gsk_bytes = 0;
    t.x = (bf16_t*)c.take(M * h * 2);
    t.xn = (bf16_t*)c.take(Mp * h * 2);
    t.qkv = (bf16_t*)c.take(M * 3 * h * 2);
    t.q = (bf16_t*)c.take(M * h * 2);
    t.att = (bf16_t*)c.take(Mp * h * 2);
    t.act = (bf16_t*)c.take(Mp * F * 2);
    t.bar = (unsigned*)c.take(256);                  // persistent decode kernel: arrival counter, error flag
    t.sk = (T == 1 && M <= 32) ? sk_area : nullptr;
    // ... and its per-layer activation buffers (qkv | att | xn after o_proj | act | xn after down), decode steps only: a buffer that
    // is written once per launch never has a stale copy in another XCD's L2, so no phase needs a cache invalidate
    t.mega_stride = 0;
    t.mega = nullptr;
#ifdef SEEDMI_DEVTOOLS
    if (T == 1 && M <= 32) {
        t.mega_stride = (Mp * (3 * h + h + h + F + h) * 2 + 255) / 256 * 256;
        t.mega = (bf16_t*)c.take(t.mega_stride * (size_t)w->layers);
    }
    // devtools only (measured slower and left out of the product library, see g_prefill_streamk): the stream-K workspace of the prefill's MFMA
    // GEMMs (seedmi_gemm_bf16_ws: flag words + one fp32 accumulator image per CU, ~64 MiB on 256 CUs), LAST and only for prefill shapes, so
    // that no other buffer's offset depends on the device's CU count
    if (M > 64) {
        t.gsk_bytes = seedmi_gemm_workspace_bytes();
        t.gsk = c.take(t.gsk_bytes);
    }
#endif
    t.bytes = c.off;
    return t;
}


#ifdef SEEDMI_DEVTOOLS
// host side of decode_layers_kernel; returns SEEDMI_OK + *done = 1 when it ran, *done = 0 when the shape is not covered (caller falls
// back to one launch per phase)
int decode_layers_launch(const seedmi_llama_weights_t* w, const LlamaWs& t, int batch, const void* pos_i64, int past_len,
                         const void* past_len_dev, int past_stride, float scale, hipStream_t stream, int* done) {
    *done = 0;
    const int h = w->hidden, F = w->ffn, H = w->heads;
    // (the shapes for which the stand-alone launcher picks the same tiling: 8-way K split, 48 / 16 / 32 / 16 weight rows per tile)
    if (!g_decode_mega || !t.mega || batch > 32 || w->layers > MEGA_MAX_LAYERS || h / H != DEC_HD || (3 * h) % 48 || (2 * F) % 32 || h % 16 ||
        (h % 256) || h < 2048 || (F % 256) || F < 2048 || g_skinny_nw == 4 || g_skinny_r != 0 || !g_skinny_nt)
        return SEEDMI_OK;
    const int lds_len = ((past_len_dev ? w->tmax : past_len + 1) + 3) & ~3;
    const size_t dyn = 2 * (size_t)(lds_len + 16 * DEC_HD) * sizeof(float);
    if (dyn > 60 * 1024) return SEEDMI_OK;
    MegaParams p;
    p.M = batch; p.h = h; p.F = F; p.H = H; p.tmax = w->tmax; p.max_pos = w->max_pos; p.layers = w->layers; p.past_stride = past_stride;
    p.past_arg = past_len; p.lds_len = lds_len; p.eps = w->rms_eps; p.scale = scale; p.probe = g_decode_mega;
    p.x = t.x; p.xn = t.xn; p.act_base = (char*)t.mega; p.act_stride = t.mega_stride;
    p.cos_t = (const bf16_t*)w->cos_t; p.sin_t = (const bf16_t*)w->sin_t;
    p.pos_ids = past_len_dev ? nullptr : (const long long*)pos_i64;
    p.past_dev = (const int*)past_len_dev;
    p.bar = t.bar;
    for (int l = 0; l < w->layers; ++l) {
        const seedmi_llama_layer_t& L = w->layer[l];
        p.layer[l] = {(const bf16_t*)L.qkv_wp, (const bf16_t*)L.o_wp, (const bf16_t*)L.gate_up_wp, (const bf16_t*)L.down_wp, (bf16_t*)L.k_cache,
                      (bf16_t*)L.v_cache};
    }
    if (hipMemsetAsync(t.bar, 0, 256, stream) != hipSuccess) {
        seedmi_set_error("seedmi_llama_forward: clearing the decode barrier failed");
        return SEEDMI_E_HIP;
    }
    const int dev = seedmi_current_device();
    const int grid = seedmi_device_cus(dev);
    static bool attr_set[2][SEEDMI_MAX_DEVICES] = {};
    const int mt = (batch + 15) / 16;
    if (!attr_set[mt - 1][dev]) {
        if (mt == 1) (void)hipFuncSetAttribute((const void*)decode_layers_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        else (void)hipFuncSetAttribute((const void*)decode_layers_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        attr_set[mt - 1][dev] = true;
    }
    if (mt == 1) hipLaunchKernelGGL(decode_layers_kernel<1>, dim3(grid), dim3(512), dyn, stream, p);
    else hipLaunchKernelGGL(decode_layers_kernel<2>, dim3(grid), dim3(512), dyn, stream, p);
    *done = 1;
    return seedmi_check_launch("decode_layers");
}
#endif

#define CK(call)                          \
    do {                                  \
        const int rc_ = (call);           \
        if (rc_ != SEEDMI_OK) return rc_; \
    } while (0)

// dispatch: decode-sized M goes to the weight-streaming kernel, everything else to the 128x128 MFMA GEMM
int linear(int M, int N, int K, const void* A, int lda, const void* W, const void* Wp, const void* R, int ldr, int epi,
           void* C, int ldc, void* s, int a_packed = 0, int c_packed = 0, void* sk_ws = nullptr, void* gemm_ws = nullptr,
           size_t gemm_ws_bytes = 0) {
    if (M <= 64 && (K % 128) == 0) {
        if (Wp && a_packed && sk_ws)
            return seedmi_gemm_skinny_norm_ws_bf16(M, N, K, A, 1, Wp, 0.f, R, ldr, epi, C, ldc, c_packed, nullptr, sk_ws, SK2_WS_BYTES, s);
        if (Wp) return seedmi_gemm_skinny_packed_bf16(M, N, K, A, lda, Wp, R, ldr, epi, C, ldc, a_packed, c_packed, s);
        return seedmi_gemm_skinny_bf16(M, N, K, A, lda, W, K, R, ldr, epi, C, ldc, s);
    }
    // (with a workspace the persistent 256x256 kernel balances a partial last round of tiles by stream-K; NULL = data-parallel walk)
#ifdef SEEDMI_DEVTOOLS
    if (gemm_ws && g_prefill_streamk.load(std::memory_order_relaxed) == 2) {
        const long long tiles = (long long)((M + 255) / 256) * ((N + 255) / 256);
        const int n_cu = seedmi_device_cus(seedmi_current_device());
        const long long last = tiles % n_cu;
        if (!(tiles < 3LL * n_cu && last != 0 && 4 * last < 3LL * n_cu)) gemm_ws = nullptr;
    }
#endif
    return seedmi_gemm_bf16_ws(M, N, K, A, lda, W, K, nullptr, R, ldr, epi, C, ldc, 0, 0, gemm_ws, gemm_ws_bytes, s);
}

// ------------------------------------------------------------------------------------------------ prefill attention, tiled
// Workgroup = (batch, head, 128 queries), wave = 32 queries (two 16-query MFMA tiles sharing every K / V fragment).  Keys stream
// through LDS in 64-key tiles: K and V rows (256 B) land by LDS-DMA one tile ahead, double buffered, one barrier per tile; the
// 16-byte chunk index of a row is XOR-ed with (row & 15) on the DMA source address, so both the K fragment reads (16 rows x
// 16 B) and the transposing V reads (ds_read_b64_tr_b16, 4 rows x 32 B per lane group) touch every bank equally.
// S^T = K Q^T puts a lane's scores on keys {4g + r}: two key tiles concatenate into the 8-deep k index of the P^T operand,
// and the V^T operand reads the same key permutation, so P never leaves registers.  Two passes like the kernel above
// (P normalised before the bf16 rounding), both from LDS.
typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int PT = 64;                       // keys per tile
constexpr int PT_BYTES = PT * DEC_HD * 2;    // 16 KiB

SEEDMI_DEVINL void glds16_l(const bf16_t* gptr, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void attn_prefill_tiled_kernel(const bf16_t* __restrict__ q, int ldq,
                                                                    const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
                                                                    bf16_t* __restrict__ out, int ldo, int T, int H, int tmax,
                                                                    int past_len, float scale) {
    __shared__ __attribute__((aligned(16))) char sm[4 * PT_BYTES];          // K0 K1 V0 V1
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qblocks = (T + 127) / 128;
    const int qb = blockIdx.x % qblocks;
    const int bh = blockIdx.x / qblocks;
    const int b = bh / H, h = bh % H;
    const bf16_t* kb = kc + ((size_t)b * H + h) * tmax * DEC_HD;
    const bf16_t* vb = vc + ((size_t)b * H + h) * tmax * DEC_HD;
    const int kv_len = past_len + T;
    const int q0 = 128 * qb + 32 * wave;                                     // first query of this wave
    const int wave_lim = min(kv_len - 1, past_len + q0 + 31);                // last key any lane of the wave sees
    const int blk_lim = min(kv_len - 1, past_len + 128 * qb + 127);
    const int ntile = blk_lim / PT + 1;

    bf16x8 qf[2][4];
    int qlim[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qrow = q0 + 16 * u + li;
        qlim[u] = past_len + qrow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (qrow < T) v = *(const uint4*)(q + ((size_t)b * T + qrow) * ldq + h * DEC_HD + 32 * ks + 8 * g);
            qf[u][ks] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // LDS-DMA of one 64-key tile of K or V: wave w copies pieces 4w .. 4w+3 (4 rows of 256 B each)
    auto stage = [&](const bf16_t* base, char* dst, int t) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int piece = 4 * wave + pc;
            const int row = 4 * piece + (lane >> 4);
            const int key = min(PT * t + row, kv_len - 1);
            const int chunk = (lane & 15) ^ (row & 15);
            glds16_l(base + (size_t)key * DEC_HD + 8 * chunk, dst + piece * 1024);
        }
    };
    // S^T of this wave's two query tiles against key tile kt4 (16 keys) of the K buffer
    auto scores = [&](const char* kbuf, int kt4, f32x4 (&s)[2]) {
        bf16x8 kf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            kf[ks] = *(const bf16x8*)(kbuf + (16 * kt4 + li) * 256 + (((4 * ks + g) ^ li) << 4));
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            s[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s[u] = seedmi_mfma_16x16x32(kf[ks], qf[u][ks], s[u]);
        }
    };

    // ---- pass 1: per-lane running max / sum over its keys {16 kt + 4g + r}
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    stage(kb, sm, 0);
    for (int t = 0; t < ntile; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < ntile) stage(kb, sm + ((t + 1) & 1) * PT_BYTES, t + 1);
        if (PT * t > wave_lim) continue;                                     // (wave-uniform) fully masked for this wave
        const char* kbuf = sm + (t & 1) * PT_BYTES;
#pragma unroll
        for (int kt4 = 0; kt4 < 4; ++kt4) {
            f32x4 s[2];
            scores(kbuf, kt4, s);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float tm = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = PT * t + 16 * kt4 + 4 * g + r;
                    s[u][r] = (key <= qlim[u] && key < kv_len) ? s[u][r] * scale : -INFINITY;
                    tm = fmaxf(tm, s[u][r]);
                }
                const float mn = fmaxf(m_run[u], tm);
                if (mn > -INFINITY) {
                    float add = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) add += __expf(s[u][r] - mn);
                    l_run[u] = l_run[u] * __expf(m_run[u] - mn) + add;
                    m_run[u] = mn;
                }
            }
        }
    }
    float mx[2], inv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float m = fmaxf(m_run[u], __shfl_xor(m_run[u], 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float lsum = (m_run[u] > -INFINITY) ? l_run[u] * __expf(m_run[u] - m) : 0.f;
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        mx[u] = m;
        inv[u] = lsum > 0.f ? 1.0f / lsum : 0.f;
    }

    // ---- pass 2: P = exp(S - max) / sum rounded to bf16, O^T += V^T P^T
    f32x4 o[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int n = 0; n < 8; ++n) o[u][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                                         // pass 1's last tile fully consumed
    stage(kb, sm, 0);
    stage(vb, sm + 2 * PT_BYTES, 0);
    for (int t = 0; t < ntile; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < ntile) {
            stage(kb, sm + ((t + 1) & 1) * PT_BYTES, t + 1);
            stage(vb, sm + (2 + ((t + 1) & 1)) * PT_BYTES, t + 1);
        }
        if (PT * t > wave_lim) continue;
        const char* kbuf = sm + (t & 1) * PT_BYTES;
        const char* vbuf = sm + (2 + (t & 1)) * PT_BYTES;
        bf16x8 pf[2][2];                                                     // [query tile][32-key step]
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float pv[2][8];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 s[2];
                scores(kbuf, 2 * kk + hf, s);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = PT * t + 32 * kk + 16 * hf + 4 * g + r;
                        pv[u][4 * hf + r] = (key <= qlim[u] && key < kv_len) ? __expf(s[u][r] * scale - mx[u]) * inv[u] : 0.f;
                    }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint4 pw;
                pw.x = pack2bf(pv[u][0], pv[u][1]); pw.y = pack2bf(pv[u][2], pv[u][3]);
                pw.z = pack2bf(pv[u][4], pv[u][5]); pw.w = pack2bf(pv[u][6], pv[u][7]);
                pf[u][kk] = __builtin_bit_cast(bf16x8, pw);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                // V^T fragment: hd rows 16n + li, keys (g, j) = 32kk + 16 (j >> 2) + 4g + (j & 3): two transposing 8-byte reads
                const int row = 32 * kk + 4 * g + (li >> 2);
                const int c = 2 * n + ((li & 3) >> 1);
                const char* vp0 = vbuf + row * 256 + ((c ^ (row & 15)) << 4) + ((li & 1) << 3);
                const char* vp1 = vbuf + (row + 16) * 256 + ((c ^ ((row + 16) & 15)) << 4) + ((li & 1) << 3);
                const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)vp0);
                const s16x4 cc = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)vp1);
                const uint2 lo = __builtin_bit_cast(uint2, a), hi = __builtin_bit_cast(uint2, cc);
                const bf16x8 vf = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
                for (int u = 0; u < 2; ++u) o[u][n] = seedmi_mfma_16x16x32(vf, pf[u][kk], o[u][n]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qrow = q0 + 16 * u + li;
        if (qrow < T) {
            bf16_t* op = out + ((size_t)b * T + qrow) * ldo + h * DEC_HD;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                uint2 w;
                w.x = pack2bf(o[u][n][0], o[u][n][1]);
                w.y = pack2bf(o[u][n][2], o[u][n][3]);
                *(uint2*)(op + 16 * n + 4 * g) = w;
            }
        }
    }
}

}  // namespace

int seedmi_llama_set_option(const char* key, int value) {
#ifdef SEEDMI_DEVTOOLS
    // A/B knobs of the decode GEMM (workgroup shape, temporal weight loads): tools only
    if (!strcmp(key, "skinny_nt") && (value == 0 || value == 1)) { g_skinny_nt = value; return SEEDMI_OK; }
    if (!strcmp(key, "skinny_waves") && (value == 0 || value == 4 || value == 8)) { g_skinny_nw = value; return SEEDMI_OK; }
    if (!strcmp(key, "skinny_rows") && (value >= 0 && value <= 3)) { g_skinny_r = value; return SEEDMI_OK; }
    if (!strcmp(key, "decode_ablate_norm") && (value == 0 || value == 1)) { g_ablate_norm = value; return SEEDMI_OK; }
#endif
    if (!strcmp(key, "skinny_splitk") && (value >= 0 && value <= 3)) { g_skinny_sk = value; return SEEDMI_OK; }
#ifdef SEEDMI_DEVTOOLS
    if (!strcmp(key, "prefill_streamk") && (value >= 0 && value <= 2)) { g_prefill_streamk = value; return SEEDMI_OK; }
    if (!strcmp(key, "skinny_splitk") && value == 4) { g_skinny_sk = value; return SEEDMI_OK; }
#endif
    if (!strcmp(key, "decode_fused") && (value == 0 || value == 1)) { g_decode_fused = value; return SEEDMI_OK; }
    if (!strcmp(key, "decode_attn_early") && (value >= 0 && value <= 2)) { g_decode_attn_early = value; return SEEDMI_OK; }
#ifdef SEEDMI_DEVTOOLS
    if (!strcmp(key, "decode_persistent") && (value >= 0 && value <= 3)) { g_decode_mega = value; return SEEDMI_OK; }
    if (!strcmp(key, "skinny_ablate") && (value >= 0 && value <= 7)) { g_skinny_abl = value; return SEEDMI_OK; }
#endif
    if (!strcmp(key, "prefill_tiled") && (value == 0 || value == 1)) { g_prefill_tiled = value; return SEEDMI_OK; }
    return SEEDMI_E_SHAPE;
}

static int skinny_entry(bool packed, int M, int N, int K, const void* A, int lda, const void* W, int ldw,
                        const void* residual, int ldr, int epilogue, void* C, int ldc, int a_packed, int c_packed,
                        void* stream, float norm_eps = 0.f, void* xp_out = nullptr, void* sk_ws = nullptr, size_t sk_ws_bytes = 0) {
    if (sk_ws && (sk_ws_bytes < SK2_WS_BYTES || ((uintptr_t)sk_ws & 255))) {
        seedmi_set_error("seedmi_gemm_skinny: workspace %zu bytes (need %zu, 256-byte aligned)", sk_ws_bytes, SK2_WS_BYTES);
        return SEEDMI_E_ALIGN;
    }
    if (xp_out && (epilogue != EPI_BIAS_RESIDUAL || (N % 32) || (ldc % 4))) {
        seedmi_set_error("seedmi_gemm_skinny: the fragment-major copy needs the BIAS_RESIDUAL epilogue, N %% 32 == 0 and ldc %% 4 == 0");
        return SEEDMI_E_SHAPE;
    }
    if ((a_packed && (K % 32)) || (c_packed && (epilogue != EPI_SWIGLU || (N % 64)))) {
        seedmi_set_error("seedmi_gemm_skinny: packed activations need K %% 32 == 0; packed output is SWIGLU-only with N %% 64 == 0");
        return SEEDMI_E_SHAPE;
    }
    if (M <= 0 || M > 64 || N <= 0 || K <= 0 || (K % 128)) {
        seedmi_set_error("seedmi_gemm_skinny: M=%d (1..64) N=%d K=%d (multiple of 128)", M, N, K);
        return SEEDMI_E_SHAPE;
    }
    if ((lda % 8) || (ldw % 8) || (((uintptr_t)A | (uintptr_t)W) & 15) || ((uintptr_t)C & 3)) {
        seedmi_set_error("seedmi_gemm_skinny: A/W need 16-byte aligned rows");
        return SEEDMI_E_ALIGN;
    }
    SkinnyParams p;
    p.M = M; p.N = N; p.K = K;
    p.A = (const bf16_t*)A; p.lda = lda;
    p.W = (const bf16_t*)W; p.ldw = ldw;
    p.R = (const bf16_t*)residual; p.ldr = ldr;
    p.C = (bf16_t*)C; p.ldc = ldc;
    p.a_packed = a_packed; p.c_packed = c_packed;
    p.norm_eps = norm_eps; p.xp_out = (bf16_t*)xp_out;
#ifdef SEEDMI_DEVTOOLS
    p.abl = g_skinny_abl;
#endif
    hipStream_t s = (hipStream_t)stream;
    // balanced split-K form: fragment-major W and A, at most two activation row tiles, a caller-owned workspace for the cut tiles
    const int sk_mode = g_skinny_sk.load(std::memory_order_relaxed);
    const bool sk = packed && a_packed && sk_ws && M <= 32 && (K % 32) == 0 && sk_mode != 0 && (sk_mode != 3 || skinny_rounds_underfilled(M, N)) &&
                    (long long)((N + 15) / 16) * (K / 32) < (1ll << 24);                       // (the kernel's index range)
    if (sk && epilogue == EPI_BIAS_RESIDUAL && !residual) { seedmi_set_error("seedmi_gemm_skinny: residual epilogue without residual"); return SEEDMI_E_SHAPE; }
    if (sk) {
        switch (epilogue) {
            case EPI_NONE: return launch_skinny_sk<EPI_NONE>(p, sk_ws, sk_mode, s);
            case EPI_BIAS_RESIDUAL: return launch_skinny_sk<EPI_BIAS_RESIDUAL>(p, sk_ws, sk_mode, s);
            case EPI_SWIGLU: return launch_skinny_sk<EPI_SWIGLU>(p, sk_ws, sk_mode, s);
            default: break;
        }
    }
    switch (epilogue) {
        case EPI_NONE: return packed ? launch_skinny<EPI_NONE, true>(p, s) : launch_skinny<EPI_NONE, false>(p, s);
        case EPI_BIAS_RESIDUAL:
            if (!residual) { seedmi_set_error("seedmi_gemm_skinny: residual epilogue without residual"); return SEEDMI_E_SHAPE; }
            return packed ? launch_skinny<EPI_BIAS_RESIDUAL, true>(p, s) : launch_skinny<EPI_BIAS_RESIDUAL, false>(p, s);
        case EPI_SWIGLU: return packed ? launch_skinny<EPI_SWIGLU, true>(p, s) : launch_skinny<EPI_SWIGLU, false>(p, s);
        default:
            seedmi_set_error("seedmi_gemm_skinny: unsupported epilogue %d", epilogue);
            return SEEDMI_E_SHAPE;
    }
}

extern "C" int seedmi_gemm_skinny_bf16(int M, int N, int K, const void* A, int lda, const void* W, int ldw,
                                       const void* residual, int ldr, int epilogue, void* C, int ldc, void* stream) {
    return skinny_entry(false, M, N, K, A, lda, W, ldw, residual, ldr, epilogue, C, ldc, 0, 0, stream);
}

extern "C" int seedmi_gemm_skinny_packed_bf16(int M, int N, int K, const void* A, int lda, const void* W_packed,
                                              const void* residual, int ldr, int epilogue, void* C, int ldc, int a_packed,
                                              int c_packed, void* stream) {
    return skinny_entry(true, M, N, K, A, lda, W_packed, 8, residual, ldr, epilogue, C, ldc, a_packed, c_packed, stream);
}

extern "C" int seedmi_gemm_skinny_norm_bf16(int M, int N, int K, const void* A, int a_packed, const void* W_packed, float rms_eps,
                                            const void* residual, int ldr, int epilogue, void* C, int ldc, int c_packed,
                                            void* x_packed_out, void* stream) {
    return skinny_entry(true, M, N, K, A, K, W_packed, 8, residual, ldr, epilogue, C, ldc, a_packed, c_packed, stream, rms_eps,
                        x_packed_out);
}

extern "C" size_t seedmi_gemm_skinny_workspace_bytes(void) { return SK2_WS_BYTES; }

extern "C" int seedmi_gemm_skinny_norm_ws_bf16(int M, int N, int K, const void* A, int a_packed, const void* W_packed, float rms_eps,
                                               const void* residual, int ldr, int epilogue, void* C, int ldc, int c_packed,
                                               void* x_packed_out, void* workspace, size_t workspace_bytes, void* stream) {
    return skinny_entry(true, M, N, K, A, K, W_packed, 8, residual, ldr, epilogue, C, ldc, a_packed, c_packed, stream, rms_eps,
                        x_packed_out, workspace, workspace_bytes);
}

extern "C" size_t seedmi_pack_skinny_weights_bytes(int N, int K) {
    return (size_t)((N + 15) / 16) * 16 * (size_t)K * 2;
}

extern "C" int seedmi_pack_skinny_weights(const void* W, int ldw, int N, int K, void* out, void* stream) {
    if (N <= 0 || K <= 0 || (K % 32) || (ldw % 8) || (((uintptr_t)W | (uintptr_t)out) & 15)) {
        seedmi_set_error("seedmi_pack_skinny_weights: N=%d K=%d ldw=%d (K multiple of 32, 16-byte aligned)", N, K, ldw);
        return SEEDMI_E_SHAPE;
    }
    const long long total = (long long)((N + 15) / 16) * (K / 32) * 64;
    long long grid = (total + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(pack_skinny_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)W, ldw, N, K,
                       (bf16_t*)out);
    return seedmi_check_launch("pack_skinny_weights");
}

static int decode_attention_launch(const void* qkv, int ldqkv, const void* pos_ids_i64, const void* cos_t, const void* sin_t,
                                   void* k_cache, void* v_cache, void* out, int ldo, int B, int H, int hd, int tmax, int past_len,
                                   float scale, int out_packed, const void* past_len_dev, int past_stride, int max_pos, void* stream);

extern "C" int seedmi_llama_decode_attention_bf16(const void* qkv, int ldqkv, const void* pos_ids_i64, const void* cos_t,
                                                  const void* sin_t, void* k_cache, void* v_cache, void* out, int ldo, int B,
                                                  int H, int hd, int tmax, int past_len, float scale, int out_packed,
                                                  const void* past_len_dev, int max_pos, void* stream) {
    return decode_attention_launch(qkv, ldqkv, pos_ids_i64, cos_t, sin_t, k_cache, v_cache, out, ldo, B, H, hd, tmax, past_len, scale,
                                   out_packed, past_len_dev, 0, max_pos, stream);
}

static int decode_attention_launch(const void* qkv, int ldqkv, const void* pos_ids_i64, const void* cos_t, const void* sin_t,
                                   void* k_cache, void* v_cache, void* out, int ldo, int B, int H, int hd, int tmax, int past_len,
                                   float scale, int out_packed, const void* past_len_dev, int past_stride, int max_pos, void* stream) {
    if (hd != DEC_HD || B <= 0 || H <= 0 || past_len < 0 || past_len + 1 > tmax || (ldqkv % 8) || max_pos <= 0 ||
        (((uintptr_t)qkv | (uintptr_t)cos_t | (uintptr_t)sin_t | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15)) {
        seedmi_set_error("seedmi_llama_decode_attention_bf16: B=%d H=%d hd=%d (must be 128) past=%d tmax=%d ldqkv=%d", B, H, hd,
                         past_len, tmax, ldqkv);
        return SEEDMI_E_SHAPE;
    }
    const int lds_len = ((past_len_dev ? tmax : past_len + 1) + 3) & ~3;
    const size_t lds = (size_t)(lds_len + 16 * DEC_HD) * sizeof(float);
    if (lds > 150 * 1024) {
        seedmi_set_error("seedmi_llama_decode_attention_bf16: kv_len %d too long for the decode kernel", lds_len);
        return SEEDMI_E_SHAPE;
    }
    static bool attr_set_dev[SEEDMI_MAX_DEVICES] = {};
    bool& attr_set = attr_set_dev[seedmi_current_device()];
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_decode_rope_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_decode_rope_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_decode_rope_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
#define SEEDMI_LAUNCH_DECODE_ATTN(E)                                                                                                   \
    hipLaunchKernelGGL(attn_decode_rope_kernel<E>, dim3(B * H), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)qkv, ldqkv,        \
                       (const long long*)pos_ids_i64, (const bf16_t*)cos_t, (const bf16_t*)sin_t, (bf16_t*)k_cache,                   \
                       (bf16_t*)v_cache, (bf16_t*)out, ldo, H, tmax, past_len, scale, out_packed, lds_len,                            \
                       (const int*)past_len_dev, max_pos, past_stride)
    switch (g_decode_attn_early.load(std::memory_order_relaxed)) {
        case 0: SEEDMI_LAUNCH_DECODE_ATTN(0); break;
        case 2: SEEDMI_LAUNCH_DECODE_ATTN(2); break;
        default: SEEDMI_LAUNCH_DECODE_ATTN(1); break;
    }
#undef SEEDMI_LAUNCH_DECODE_ATTN
    return seedmi_check_launch("attn_decode_rope");
}

extern "C" int seedmi_llama_attention_bf16(const void* q, int ldq, const void* k_cache, const void* v_cache, void* out,
                                           int ldo, int B, int T, int H, int hd, int tmax, int past_len, float scale,
                                           int out_packed, const void* past_len_dev, void* stream) {
    if (past_len_dev && T != 1) {
        seedmi_set_error("seedmi_llama_attention_bf16: device-resident past_len is a decode (T == 1) feature");
        return SEEDMI_E_SHAPE;
    }
    if (out_packed && T != 1) {
        seedmi_set_error("seedmi_llama_attention_bf16: packed output is a decode (T == 1) feature");
        return SEEDMI_E_SHAPE;
    }
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
        // with a device-resident length the launch must cover any length up to the cache capacity
        const int lds_len = ((past_len_dev ? tmax : kv_len) + 3) & ~3;
        const size_t lds = (size_t)(lds_len + 16 * DEC_HD) * sizeof(float);
        if (lds > 150 * 1024) {
            seedmi_set_error("seedmi_llama_attention_bf16: kv_len %d too long for the decode kernel", kv_len);
            return SEEDMI_E_SHAPE;
        }
        static bool attr_set_dev[SEEDMI_MAX_DEVICES] = {};
        bool& attr_set = attr_set_dev[seedmi_current_device()];
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(attn_decode_kernel, dim3(B * H), dim3(256), lds, s, (const bf16_t*)q, ldq, (const bf16_t*)k_cache,
                           (const bf16_t*)v_cache, (bf16_t*)out, ldo, H, tmax, kv_len, scale, out_packed, lds_len,
                           (const int*)past_len_dev);
        return seedmi_check_launch("attn_decode");
    }
    if (g_prefill_tiled) {
        const int qb128 = (T + 127) / 128;
        hipLaunchKernelGGL(attn_prefill_tiled_kernel, dim3(B * H * qb128), dim3(256), 0, s, (const bf16_t*)q, ldq,
                           (const bf16_t*)k_cache, (const bf16_t*)v_cache, (bf16_t*)out, ldo, T, H, tmax, past_len, scale);
        return seedmi_check_launch("attn_prefill_tiled");
    }
    const int qblocks = (T + 63) / 64;
    hipLaunchKernelGGL(attn_prefill_kernel, dim3(B * H * qblocks), dim3(256), 0, s, (const bf16_t*)q, ldq,
                       (const bf16_t*)k_cache, (const bf16_t*)v_cache, (bf16_t*)out, ldo, T, H, tmax, past_len, scale);
    return seedmi_check_launch("attn_prefill");
}

// The sticky error word of a split-K workspace: 0, or 1 + the id of a workgroup whose bounded wait for a partner's image ran out (its
// tile of C is then wrong).  Synchronises `stream`, reads the word and clears it once reported.
extern "C" int seedmi_gemm_skinny_ws_status(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace || workspace_bytes < SK2_WS_BYTES) {
        seedmi_set_error("seedmi_gemm_skinny_ws_status: not a split-K workspace (seedmi_gemm_skinny_workspace_bytes())");
        return SEEDMI_E_SHAPE;
    }
    unsigned words[2] = {0, 0};                       // tag, error word
    unsigned* dev = (unsigned*)workspace + (SK2_FLAG_WORDS - 1);
    if (hipMemcpyAsync(words, dev - 1, 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        seedmi_set_error("seedmi_gemm_skinny_ws_status: reading the error word failed");
        return SEEDMI_E_HIP;
    }
    if (words[0] != SK2_TAG) {
        seedmi_set_error("seedmi_gemm_skinny_ws_status: this workspace was never initialised (seedmi_gemm_skinny_workspace_init / "
                         "seedmi_llama_workspace_init after allocation): its flag and error words are whatever the allocation held");
        return SEEDMI_E_SHAPE;
    }
    const unsigned word = words[1];
    if (word == 0) return SEEDMI_OK;
    (void)hipMemsetAsync(dev, 0, 4, (hipStream_t)stream);
    seedmi_set_error("split-K decode GEMM: workgroup %u gave up waiting for a partner's partial tile (the grid was not fully resident, "
                     "e.g. another stream's kernel held CUs): results computed through this workspace since the last check are invalid", word - 1);
    return SEEDMI_E_HIP;
}

// Once after allocation: hand-off flags and error word zero, tag set (stream-ordered; no synchronisation)
extern "C" int seedmi_gemm_skinny_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace || workspace_bytes < SK2_WS_BYTES || ((uintptr_t)workspace & 255)) {
        seedmi_set_error("seedmi_gemm_skinny_workspace_init: need seedmi_gemm_skinny_workspace_bytes() bytes, 256-byte aligned");
        return SEEDMI_E_SHAPE;
    }
    static const unsigned tail[2] = {SK2_TAG, 0u};
    if (hipMemsetAsync(workspace, 0, SK2_LIVE_WORDS * 4, (hipStream_t)stream) != hipSuccess ||
        hipMemcpyAsync((unsigned*)workspace + SK2_LIVE_WORDS, tail, 8, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        seedmi_set_error("seedmi_gemm_skinny_workspace_init: clearing the flag words failed");
        return SEEDMI_E_HIP;
    }
    return SEEDMI_OK;
}

extern "C" int seedmi_llama_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    // (the split-K area is the FIRST region of every llama workspace, whatever (batch, T) it was sized for: carve())
    if (workspace && workspace_bytes < SK2_WS_BYTES) return SEEDMI_OK;      // too small to hold decode steps' split-K area: nothing to initialise
    return seedmi_gemm_skinny_workspace_init(workspace, workspace_bytes, stream);
}

extern "C" int seedmi_llama_decode_status(const seedmi_llama_weights_t* w, int batch, void* workspace, size_t workspace_bytes, void* stream) {
    if (!w || batch <= 0 || !workspace) {
        seedmi_set_error("seedmi_llama_decode_status: bad arguments");
        return SEEDMI_E_SHAPE;
    }
    const LlamaWs t = carve(w, batch, 1, workspace);
    if (workspace_bytes < t.bytes) {
        seedmi_set_error("seedmi_llama_decode_status: workspace smaller than seedmi_llama_workspace_bytes(w, batch, 1)");
        return SEEDMI_E_SHAPE;
    }
    if (!t.sk) return SEEDMI_OK;                       // this batch does not take the split-K kernels
    return seedmi_gemm_skinny_ws_status(t.sk, SK2_WS_BYTES, stream);
}

extern "C" size_t seedmi_llama_workspace_bytes(const seedmi_llama_weights_t* w, int batch, int T) {
    if (!w || batch <= 0 || T <= 0) return 0;
    return carve(w, batch, T, nullptr).bytes;
}

extern "C" int seedmi_llama_forward(const seedmi_llama_weights_t* w, const void* ids_i64, const void* pos_i64, int batch,
                                    int T, int past_len, int last_only, void* logits, int ldl, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return seedmi_llama_forward_ex(w, ids_i64, pos_i64, batch, T, past_len, nullptr, last_only, logits, ldl, workspace,
                                   workspace_bytes, stream);
}

extern "C" int seedmi_llama_forward_ex(const seedmi_llama_weights_t* w, const void* ids_i64, const void* pos_i64, int batch,
                                       int T, int past_len, const void* past_len_dev, int last_only, void* logits, int ldl,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    return seedmi_llama_forward_io(w, ids_i64, nullptr, pos_i64, batch, T, past_len, past_len_dev, last_only, logits, ldl, nullptr,
                                   workspace, workspace_bytes, stream);
}

static int llama_forward_impl(const seedmi_llama_weights_t* w, const void* ids_i64, const void* inputs_embeds, const void* pos_i64,
                              int batch, int T, int past_len, const void* past_len_dev, int past_stride, int last_only,
                              void* logits, int ldl, void* hidden_states, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int seedmi_llama_forward_io(const seedmi_llama_weights_t* w, const void* ids_i64, const void* inputs_embeds,
                                       const void* pos_i64, int batch, int T, int past_len, const void* past_len_dev,
                                       int last_only, void* logits, int ldl, void* hidden_states, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return llama_forward_impl(w, ids_i64, inputs_embeds, pos_i64, batch, T, past_len, past_len_dev, 0, last_only, logits, ldl,
                              hidden_states, workspace, workspace_bytes, stream);
}

// One decode step of `batch` independent SLOTS, each at its own cache length lens_i32[b] (device): the continuous-batching step.
// Row b appends its token at position lens[b] of its cache row and attends over lens[b] + 1 keys; everything else is the ordinary
// single-token step (every kernel of it is row-independent).  Graph-replayable: nothing but device memory changes between steps.
extern "C" int seedmi_llama_decode_slots(const seedmi_llama_weights_t* w, const void* tok_i64, const void* lens_i32, int batch,
                                         void* logits, int ldl, void* workspace, size_t workspace_bytes, void* stream) {
    if (!lens_i32 || !g_decode_fused) {
        seedmi_set_error("seedmi_llama_decode_slots: needs per-slot lengths and the fused decode attention");
        return SEEDMI_E_SHAPE;
    }
    return llama_forward_impl(w, tok_i64, nullptr, nullptr, batch, 1, w ? w->tmax - 1 : 0, lens_i32, 1, 1, logits, ldl, nullptr,
                              workspace, workspace_bytes, stream);
}

static int llama_forward_impl(const seedmi_llama_weights_t* w, const void* ids_i64, const void* inputs_embeds, const void* pos_i64,
                              int batch, int T, int past_len, const void* past_len_dev, int past_stride, int last_only,
                              void* logits, int ldl, void* hidden_states, void* workspace, size_t workspace_bytes, void* stream) {
    if (!w || (!ids_i64 && !inputs_embeds) || (ids_i64 && inputs_embeds) || (!pos_i64 && !past_len_dev) || !logits || batch <= 0 ||
        T <= 0) {
        seedmi_set_error("seedmi_llama_forward: null argument, both/neither of ids and inputs_embeds, or bad batch/T");
        return SEEDMI_E_SHAPE;
    }
    if (hidden_states && last_only) {
        seedmi_set_error("seedmi_llama_forward_io: hidden_states are produced for all positions (last_only must be 0)");
        return SEEDMI_E_SHAPE;
    }
    if (past_len_dev && T != 1) {
        seedmi_set_error("seedmi_llama_forward_ex: a device-resident past_len is only valid for single-token decode steps");
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
    // decode steps (T == 1, M <= 64) keep every GEMM operand in the fragment-major layout end to end:
    // rmsnorm -> [QKV], attention -> [o_proj], rmsnorm -> [gate|up] -> SwiGLU -> [down]; the residual stream stays row-major
    const bool pk = (T == 1 && M <= 64 && (h % 128) == 0 && (F % 128) == 0 && w->layer[0].qkv_wp && w->layer[0].o_wp &&
                     w->layer[0].gate_up_wp && w->layer[0].down_wp);
    // folded RMSNorm (w->norm_folded: the fragment-major qkv / gate_up / lm_head copies carry weight * gamma): the GEMM that
    // consumes a norm computes the row scale itself from the activation fragments it streams, and the GEMM that produces the
    // residual stream also writes its fragment-major copy - no norm launches, no extra pass over x (64 launches per 8B step)
    const bool fold = pk && w->norm_folded && !g_ablate_norm;
    void* const sk = (pk && t.sk && g_skinny_sk.load(std::memory_order_relaxed)) ? t.sk : nullptr;
    const bool fused_first = fold && !inputs_embeds;     // embedding rows, their fragment-major copy and the flag clear in ONE launch
    if (inputs_embeds) {                                 // LlamaModel.forward with inputs_embeds (llama_xformer.py:543-544 skipped)
        if (hipMemcpyAsync(t.x, inputs_embeds, (size_t)M * h * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
            seedmi_set_error("seedmi_llama_forward_io: copying inputs_embeds failed");
            return SEEDMI_E_HIP;
        }
    } else if (fused_first) {
        CK(seedmi_embed_rows_decode(ids_i64, w->embed, h, t.x, h, t.xn, M, h, w->vocab, sk, sk ? SK2_LIVE_WORDS : 0, stream));   // (not the tag, not the sticky error word)
    } else {
        CK(seedmi_embed_rows(ids_i64, w->embed, h, t.x, h, M, h, w->vocab, stream));
    }
    // output_hidden_states (llama_xformer.py:569-570, 613-617): the input of every layer, then the final-norm output
    auto tap_hidden = [&](int idx, const bf16_t* src) -> int {
        if (!hidden_states) return SEEDMI_OK;
        if (hipMemcpyAsync((bf16_t*)hidden_states + (size_t)idx * M * h, src, (size_t)M * h * 2, hipMemcpyDeviceToDevice,
                           (hipStream_t)stream) != hipSuccess) {
            seedmi_set_error("seedmi_llama_forward_io: copying a hidden state failed");
            return SEEDMI_E_HIP;
        }
        return SEEDMI_OK;
    };
    // stream-K tail of the MFMA GEMMs (prefill, M > 64): its flag words are cleared once per call, so the workspace's history (another
    // shape's activations may have lived there before this area moved to the front) never matters
#ifdef SEEDMI_DEVTOOLS
    void* const gsk = (M > 64 && g_prefill_streamk.load(std::memory_order_relaxed)) ? t.gsk : nullptr;
    // (all flag words but the LAST one: that is gemm256's sticky "lost partner" word, which only a status reader may clear)
    if (gsk && hipMemsetAsync(gsk, 0, 4096 - 4, (hipStream_t)stream) != hipSuccess) {
        seedmi_set_error("seedmi_llama_forward: clearing the stream-K flag words failed");
        return SEEDMI_E_HIP;
    }
#else
    void* const gsk = nullptr;
#endif
    if (sk && !fused_first && hipMemsetAsync(sk, 0, SK2_LIVE_WORDS * 4, (hipStream_t)stream) != hipSuccess) {     // (the kernels leave the
        seedmi_set_error("seedmi_llama_forward: clearing the split-K flag words failed");                       //  words zero: safety net)
        return SEEDMI_E_HIP;
    }
    if (fold && !fused_first) CK(seedmi_pack_activations_bf16(t.x, h, t.xn, M, h, stream));
    int mega_done = 0;
#ifdef SEEDMI_DEVTOOLS
    if (fold && g_decode_fused && !hidden_states)
        CK(decode_layers_launch(w, t, batch, pos_i64, past_len, past_len_dev, past_stride, scale, (hipStream_t)stream, &mega_done));
#endif
    for (int l = 0; l < w->layers && !mega_done; ++l) {
        const seedmi_llama_layer_t& L = w->layer[l];
        CK(tap_hidden(l, t.x));
        if (fold) {
            CK(seedmi_gemm_skinny_norm_ws_bf16(M, 3 * h, h, t.xn, 1, L.qkv_wp, w->rms_eps, nullptr, 0, EPI_NONE, t.qkv, 3 * h, 0, nullptr,
                                               sk, sk ? SK2_WS_BYTES : 0, stream));
        } else {
            if (pk && g_ablate_norm) {}                      // timing ablation: stale xn
            else if (pk) CK(seedmi_rmsnorm_packed_bf16(t.x, h, L.ln1_w, w->rms_eps, t.xn, M, h, stream));
            else CK(seedmi_rmsnorm_bf16(t.x, h, L.ln1_w, w->rms_eps, t.xn, h, M, h, stream));
            CK(linear(M, 3 * h, h, t.xn, h, L.qkv_w, (pk || !w->norm_folded) ? L.qkv_wp : nullptr, nullptr, 0, EPI_NONE, t.qkv, 3 * h,
                      stream, pk, 0, sk, gsk, t.gsk_bytes));
        }
        if (T == 1 && g_decode_fused) {
            // RoPE + cache append + attention in one launch (bit-identical to the two-kernel form below)
            CK(decode_attention_launch(t.qkv, 3 * h, past_len_dev ? nullptr : pos_i64, w->cos_t, w->sin_t, L.k_cache, L.v_cache, t.att, h,
                                       batch, H, hd, w->tmax, past_len, scale, pk, past_len_dev, past_stride, w->max_pos, stream));
        } else {
            CK(seedmi_rope_kv_append(t.qkv, 3 * h, past_len_dev ? nullptr : pos_i64, w->cos_t, w->sin_t, t.q, h, L.k_cache,
                                     L.v_cache, batch, T, H, hd, w->tmax, past_len, past_len_dev, w->max_pos, stream));
            CK(seedmi_llama_attention_bf16(t.q, h, L.k_cache, L.v_cache, t.att, h, batch, T, H, hd, w->tmax, past_len, scale,
                                           pk, past_len_dev, stream));
        }
        if (fold) {
            const size_t skb = sk ? SK2_WS_BYTES : 0;
            CK(seedmi_gemm_skinny_norm_ws_bf16(M, h, h, t.att, 1, L.o_wp, 0.f, t.x, h, EPI_BIAS_RESIDUAL, t.x, h, 0, t.xn, sk, skb, stream));
            CK(seedmi_gemm_skinny_norm_ws_bf16(M, 2 * F, h, t.xn, 1, L.gate_up_wp, w->rms_eps, nullptr, 0, EPI_SWIGLU, t.act, F, 1, nullptr,
                                               sk, skb, stream));
            CK(seedmi_gemm_skinny_norm_ws_bf16(M, h, F, t.act, 1, L.down_wp, 0.f, t.x, h, EPI_BIAS_RESIDUAL, t.x, h, 0, t.xn, sk, skb, stream));
        } else {
            CK(linear(M, h, h, t.att, h, L.o_w, L.o_wp, t.x, h, EPI_BIAS_RESIDUAL, t.x, h, stream, pk, 0, sk, gsk, t.gsk_bytes));
            if (pk && g_ablate_norm) {}
            else if (pk) CK(seedmi_rmsnorm_packed_bf16(t.x, h, L.ln2_w, w->rms_eps, t.xn, M, h, stream));
            else CK(seedmi_rmsnorm_bf16(t.x, h, L.ln2_w, w->rms_eps, t.xn, h, M, h, stream));
            CK(linear(M, 2 * F, h, t.xn, h, L.gate_up_w, (pk || !w->norm_folded) ? L.gate_up_wp : nullptr, nullptr, 0, EPI_SWIGLU, t.act,
                      F, stream, pk, pk, sk, gsk, t.gsk_bytes));
            CK(linear(M, h, F, t.act, F, L.down_w, L.down_wp, t.x, h, EPI_BIAS_RESIDUAL, t.x, h, stream, pk, 0, sk, gsk, t.gsk_bytes));
        }
    }
    const bool pk_head = (batch <= 64 && (h % 128) == 0 && w->lm_head_p);
    if (last_only) {
        // final norm + lm_head on the last position of every sequence only (decode fast path)
        if (pk_head && w->norm_folded && !g_ablate_norm) {
            if (!fold) CK(seedmi_pack_activations_bf16(t.x + (size_t)(T - 1) * h, T * h, t.xn, batch, h, stream));
            CK(seedmi_gemm_skinny_norm_ws_bf16(batch, w->vocab, h, t.xn, 1, w->lm_head_p, w->rms_eps, nullptr, 0, EPI_NONE, logits, ldl, 0,
                                               nullptr, sk, sk ? SK2_WS_BYTES : 0, stream));
        } else {
            if (pk_head) CK(seedmi_rmsnorm_packed_bf16(t.x + (size_t)(T - 1) * h, T * h, w->norm_w, w->rms_eps, t.xn, batch, h, stream));
            else CK(seedmi_rmsnorm_bf16(t.x + (size_t)(T - 1) * h, T * h, w->norm_w, w->rms_eps, t.xn, h, batch, h, stream));
            CK(linear(batch, w->vocab, h, t.xn, h, w->lm_head, w->lm_head_p, nullptr, 0, EPI_NONE, logits, ldl, stream, pk_head, 0));
        }
    } else {
        CK(seedmi_rmsnorm_bf16(t.x, h, w->norm_w, w->rms_eps, t.xn, h, M, h, stream));
        CK(tap_hidden(w->layers, t.xn));
        CK(linear(M, w->vocab, h, t.xn, h, w->lm_head, w->norm_folded ? nullptr : w->lm_head_p, nullptr, 0, EPI_NONE, logits, ldl, stream, 0, 0,
                  nullptr, gsk, t.gsk_bytes));
    }
    return SEEDMI_OK;
}
