// bf16 GEMM with fused epilogues for gfx950:  C[M,N] = epi(A[M,K] * W[N,K]^T + bias)
//
// Replaces every nn.Linear / F.linear on the hot path (eva_vit.py:135,157,60,64;
// qformer_causual.py:165-179,252,321,334; qformer_quantizer.py:219-223; llama_xformer.py:223-225,258,186,718)
// and the patch-embed conv-as-GEMM (eva_vit.py:229).  A and W are both K-contiguous (activation rows,
// nn.Linear weight rows), fp32 accumulation on the MFMA pipe, one rounding to bf16 after the bias
// (what cuBLAS/oneDNN do for the reference), further roundings where the reference materialises
// another half tensor (GELU output, residual sum, ...).
//
// Kernel "gemm128": 128x128x64 block tile, 4 waves (2x2), 64x64 per wave = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16.  Operands are staged HBM->LDS with global_load_lds_dwordx4 (no VGPR round
// trip), double buffered, one barrier per K-tile.  LDS rows are 128 B; the 16-B chunk index is XOR-swizzled
// so each ds_read_b128 lane group touches 16 distinct 16-B slots (the swizzle is applied on the per-lane
// *global source* address and on the read address; the LDS image itself must stay lane-linear for LDS-DMA).
// MFMA orientation is swapped (weights are the A operand) and the weight rows feeding one MFMA are
// {16a + 4*ni + b}, so after the K loop every lane owns 16 *contiguous* output columns of 4 rows:
// the epilogue issues 16-byte loads/stores only.
//
// Workgroups are remapped so that each XCD (private 4 MiB L2) walks a contiguous range of tiles in
// grouped (4 m-tiles x all n-tiles) order.
#include <string.h>
#include <type_traits>
#include "common.h"
#include "seedmi_internal.h"

// s_setprio(1) around the MFMA runs of the 256x256 kernel: measured neutral-to-negative (-0.5 % end to end), off by default
#ifndef SEEDMI_GEMM_PRIO
#define SEEDMI_GEMM_PRIO 0
#endif
#ifndef SEEDMI_GEMM256_DEFAULT
#define SEEDMI_GEMM256_DEFAULT 1
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // A tile + W tile
int g_gemm_ablate = 0;             // seedmi_set_option("gemm_ablate", mask): timing-only ablations of the 255 kernel
int g_group_m = 4;                 // seedmi_set_option("gemm_group_m", v): m-tiles per L2 tile group

struct GemmParams {
    int M, N, K;
    const bf16_t* A; int lda;
    const bf16_t* W; int ldw;
    const bf16_t* bias;
    const bf16_t* R; int ldr;
    bf16_t* C; int ldc;
    int tiles_m, tiles_n;
    int group_m;                // m-tiles walked per group of the tile order (L2 locality)
    int skip_epilogue;          // timing ablations, seedmi_set_option("gemm_ablate", 32|33|34): 1 = no epilogue, 2 = epilogue without
                                // its stores, 3 = ordinary instead of streaming stores, 4 = streaming stores without the lane transposition
    int row_group, row_extra;   // patch-embed: out_row = m + (m / row_group) * row_extra + row_extra ; res_row = m % row_group + row_extra
};

SEEDMI_DEVINL int swzA(int row) { return (row >> 1) & 7; }
SEEDMI_DEVINL int swzW(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

SEEDMI_DEVINL void glds16(const bf16_t* gptr, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}


// ---- shared epilogue: the lane owns rows mrow0 + 16*mi + li (mi < MT) and the 16 contiguous columns nb..nb+15
// LANE4 (lane = li + 16 g, the four lanes of a row own adjacent 16-column groups): when the wave's whole 64-column span lies
// inside N the 16-byte halves of the four lanes are transposed with v_permlane16_swap / v_permlane32_swap so that each store
// instruction writes 64 contiguous bytes of a row instead of four 16-byte pieces at a 32-byte stride (whole 32-byte sectors
// instead of half sectors: -7 % on the ViT QKV GEMM).
template <int EPI, int MT, bool LANE4 = true>
SEEDMI_DEVINL void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[MT][4], int mrow0, int nb, int li) {
    const int span0 = nb & ~63;                                   // first column of the wave's 64-column span (wave-uniform)
    const bool span_full = LANE4 && EPI != EPI_SWIGLU && (span0 + 64 <= p.N) && p.skip_epilogue == 0;
    if (nb >= p.N) return;
    const bool full = (nb + 16 <= p.N);
    float bias[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bias[i] = 0.f;
    if (EPI != EPI_NONE && p.bias) {
        if (full) {
            const uint4 b0 = *(const uint4*)(p.bias + nb);
            const uint4 b1 = *(const uint4*)(p.bias + nb + 8);
            const uint32_t bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) { bias[2 * i] = lo_bf(bw[i]); bias[2 * i + 1] = hi_bf(bw[i]); }
        } else {
            for (int i = 0; i < 16; ++i) if (nb + i < p.N) bias[i] = bf2f(p.bias[nb + i]);
        }
    }
    // residual / pos_embed rows of ALL the lane's rows are requested up front: one exposed HBM latency per tile instead
    // of one per row (the per-row form serialised 8 round trips and cost the proj GEMM 25 %)
    uint4 rr[MT][2];
    if (EPI == EPI_BIAS_RESIDUAL || EPI == EPI_PATCH_EMBED) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int m = min(mrow0 + 16 * mi + li, p.M - 1);
            int res_row = m;
            if (EPI == EPI_PATCH_EMBED) res_row = m - (m / p.row_group) * p.row_group + p.row_extra;
            const bf16_t* rp = p.R + (size_t)res_row * p.ldr + nb;
            if (full) {
                rr[mi][0] = *(const uint4*)rp;
                rr[mi][1] = *(const uint4*)(rp + 8);
            }
        }
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int m = mrow0 + 16 * mi + li;
        if (!span_full && m >= p.M) continue;                     // (the transposing path keeps every lane of the row alive)
        float v[16];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * ni + r] = acc[mi][ni][r] + bias[4 * ni + r];

        int out_row = m;
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = gelu_erf(rbf(v[i]));      // GELU of the half fc1 output
        } else if (EPI == EPI_BIAS_TANH) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = tanhf(rbf(v[i]));
        } else if (EPI == EPI_RELU) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaxf(rbf(v[i]), 0.f);
        } else if (EPI == EPI_BIAS_RESIDUAL || EPI == EPI_PATCH_EMBED) {
            int res_row = m;
            if (EPI == EPI_PATCH_EMBED) {
                const int img = m / p.row_group;
                res_row = m - img * p.row_group + p.row_extra;            // pos_embed row (skip cls)
                out_row = m + (img + 1) * p.row_extra;                    // leave room for one cls row per image
            }
            const bf16_t* rp = p.R + (size_t)res_row * p.ldr + nb;
            if (full) {
                const uint4 r0 = rr[mi][0];
                const uint4 r1 = rr[mi][1];
                const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[2 * i] = rbf(v[2 * i]) + lo_bf(rw[i]);               // half GEMM output + half residual
                    v[2 * i + 1] = rbf(v[2 * i + 1]) + hi_bf(rw[i]);
                }
            } else {
                for (int i = 0; i < 16; ++i) if (nb + i < p.N) v[i] = rbf(v[i]) + bf2f(rp[i]);
            }
        }

        if (EPI == EPI_SWIGLU) {
            // interleaved rows: even = gate_proj, odd = up_proj  ->  out[m][n/2] = silu(gate) * up
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = rbf(silu(rbf(v[2 * i]))) * rbf(v[2 * i + 1]);
            bf16_t* cp = p.C + (size_t)out_row * p.ldc + (nb >> 1);
            if (full) {
                uint4 s;
                s.x = pack2bf(o[0], o[1]); s.y = pack2bf(o[2], o[3]); s.z = pack2bf(o[4], o[5]); s.w = pack2bf(o[6], o[7]);
                *(uint4*)cp = s;
            } else {
                for (int i = 0; i < 8; ++i) if (nb + 2 * i + 1 < p.N) cp[i] = f2bf(o[i]);
            }
        } else {
            bf16_t* cp = p.C + (size_t)out_row * p.ldc + nb;
            if (full) {
                uint4 s0, s1;
                s0.x = pack2bf(v[0], v[1]); s0.y = pack2bf(v[2], v[3]); s0.z = pack2bf(v[4], v[5]); s0.w = pack2bf(v[6], v[7]);
                s1.x = pack2bf(v[8], v[9]); s1.y = pack2bf(v[10], v[11]); s1.z = pack2bf(v[12], v[13]); s1.w = pack2bf(v[14], v[15]);
                if (span_full) {
                    // rows of 16 lanes = column groups g: s0 = pieces [0,2,4,6], s1 = [1,3,5,7] of the row's eight 16-byte pieces;
                    // permlane16_swap -> [0,1,4,5] / [2,3,6,7]; permlane32_swap -> [0,1,2,3] / [4,5,6,7]
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    unsigned a[4] = {s0.x, s0.y, s0.z, s0.w}, c[4] = {s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const auto t1 = __builtin_amdgcn_permlane16_swap(a[d], c[d], false, false);
                        const auto t2 = __builtin_amdgcn_permlane32_swap(t1[0], t1[1], false, false);
                        a[d] = t2[0];
                        c[d] = t2[1];
                    }
                    if (m < p.M) {
                        bf16_t* wp = p.C + (size_t)out_row * p.ldc + span0 + 8 * ((nb >> 4) & 3);
                        __builtin_nontemporal_store((u32x4_t){a[0], a[1], a[2], a[3]}, (u32x4_t*)wp);
                        __builtin_nontemporal_store((u32x4_t){c[0], c[1], c[2], c[3]}, (u32x4_t*)(wp + 32));
                    }
                } else if (p.skip_epilogue == 2) {                       // timing ablation: everything but the stores themselves
                    if ((s0.x ^ s1.w) == 0x12345678u) *(uint4*)cp = s0;
                } else if (p.skip_epilogue == 3) {                // A/B: ordinary (L2-allocating) stores
                    *(uint4*)cp = s0;
                    *(uint4*)(cp + 8) = s1;
                } else {
                    // streaming stores: C is written once and is far larger than L2 (0.56 GB for the ViT QKV), so letting it
                    // allocate there only evicts the A / W panels the other tiles of the XCD are about to re-read (+7 % on QKV)
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((u32x4_t){s0.x, s0.y, s0.z, s0.w}, (u32x4_t*)cp);
                    __builtin_nontemporal_store((u32x4_t){s1.x, s1.y, s1.z, s1.w}, (u32x4_t*)(cp + 8));
                }
            } else {
                for (int i = 0; i < 16; ++i) if (nb + i < p.N) cp[i] = f2bf(v[i]);
            }
        }
    }
}

// XCD-contiguous, grouped (GROUP_M m-tiles x all n-tiles) workgroup -> tile map
SEEDMI_DEVINL void tile_of_block(const GemmParams& p, int& tm, int& tn) {
    const int nt = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int gsize = p.group_m * p.tiles_n;
    const int gid = t / gsize;
    const int first_m = gid * p.group_m;
    const int gm = min(p.tiles_m - first_m, p.group_m);
    const int in_g = t - gid * gsize;
    tm = first_m + in_g % gm;
    tn = in_g / gm;
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, g = lane >> 4;

    int tm, tn;
    tile_of_block(p, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: wave w copies rows [32w, 32w+32) of both tiles, 4 LDS-DMA pieces of 8 rows each
    int offA[4], offW[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 32 * wave + 8 * j + (lane >> 3);
        const int cs = lane & 7;                                  // chunk slot this lane fills in LDS
        const int ra = min(m0 + row, p.M - 1);
        const int rw = min(n0 + row, p.N - 1);
        offA[j] = ra * p.lda + 8 * (cs ^ swzA(row));
        offW[j] = rw * p.ldw + 8 * (cs ^ swzW(row));
    }
    // ---- fragment read addresses (byte offsets inside a stage), k-step 0; k-step 1 = ^64
    int rdA[4], rdW[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = 64 * wm + 16 * t + li;                     // activation row feeding MFMA column li
        rdA[t] = ra * 128 + ((g ^ swzA(ra)) << 4);
        const int rw = 64 * wn + 16 * (li >> 2) + 4 * t + (li & 3);   // weight row feeding MFMA row li
        rdW[t] = TILE_BYTES + rw * 128 + ((g ^ swzW(rw)) << 4);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    auto stage = [&](int s, int kt) {
        char* base = smem + s * STAGE_BYTES + wave * 4096;
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(p.A + (size_t)(offA[j] + k0), base + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(p.W + (size_t)(offW[j] + k0), base + TILE_BYTES + j * 1024);
    };

    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sb = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], w[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = *(const bf16x8*)(sb + (rdA[t] ^ (ks << 6)));
#pragma unroll
            for (int t = 0; t < 4; ++t) w[t] = *(const bf16x8*)(sb + (rdW[t] ^ (ks << 6)));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ni], a[mi], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }

    gemm_epilogue<EPI, 4>(p, acc, m0 + 64 * wm, n0 + 64 * wn + 16 * g, li);
}


// ======================================================================================================
// Kernel "gemm256": 256x256x64 block tile, 8 waves (2 along M x 4 along N), 128x64 per wave.
//
// Deep-pipelined schedule for one workgroup per CU (128 KiB LDS, 2 waves per SIMD):
//  * LDS holds two K-tiles (parity = kt & 1), each as four 16 KiB half-tiles A0|A1|W0|W1 (128 rows x 64 k,
//    same XOR-swizzled 128-B rows as gemm128).  A wave only ever reads A_{wm} and W_{wn>>1}.
//  * a K-tile is computed in 4 phases of 16 MFMAs (64x32 quadrants): P1 (mh0,nh0) P2 (mh0,nh1) P3 (mh1,nh1)
//    P4 (mh1,nh0); fragments are read in the phase's LOAD section: P1 W(nh0)+A(mh0), P2 W(nh1), P3 A(mh1).
//  * the two wave groups (wm = 0 / 1, one wave of each per SIMD) run one barrier apart: while one group is in
//    its MFMA section the other issues its LDS reads and LDS-DMA, so the matrix pipe and the LDS/TA alternate
//    owners instead of idling together.  Every phase is  LOAD | s_barrier | MFMA | s_barrier.
//  * LDS-DMA is issued two half-tiles at a time, far ahead: A(kt+1) in P1 (slots last read in P3 of kt-1),
//    W(kt+2) in P4 (slots last read in P2 of kt); one s_waitcnt vmcnt(0) per K-tile, in P4 *before* the new
//    issue, retires loads that have been in flight for 3-4 phases.  Hazard rules (derived for the one-barrier
//    stagger): a slot is read no earlier than the phase after the wait that retires it, and restaged no earlier
//    than two phases after its last read.
// Raw s_barrier (not __syncthreads) so LDS-DMA stays in flight across barriers; waits are explicit.
int g_gemm_persist = 1;          // seedmi_set_option("gemm_persist", 0|1)
constexpr int B2 = 256;
constexpr int HALF_BYTES = 128 * BK * 2;          // 16 KiB
constexpr int KT_BYTES = 4 * HALF_BYTES;          // 64 KiB per K-tile

#define SEEDMI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)


typedef __attribute__((ext_vector_type(16))) float f32x16;

// epilogue for the 32x32 accumulator layout: the lane owns row (mrow + 32*mt) and, per n-tile, 16 contiguous columns
template <int EPI>
SEEDMI_DEVINL void gemm_epilogue32(const GemmParams& p, f32x16 (&acc)[4][2], int mrow, int nb0) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int nb = nb0 + 32 * nt;
        if (nb >= p.N) continue;
        f32x4 a4[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) a4[mt][q][r] = acc[mt][nt][4 * q + r];
        // rows mrow + 32*mt: reuse the generic epilogue one row group at a time (MT = 1, row stride handled by the base)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 one[1][4] = {{a4[mt][0], a4[mt][1], a4[mt][2], a4[mt][3]}};
            gemm_epilogue<EPI, 1, false>(p, one, mrow + 32 * mt, nb, 0);
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmParams p) {
    constexpr bool PRIO = SEEDMI_GEMM_PRIO;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, g = lane >> 4;

    // ---- persistent tile walk: the launch holds one workgroup per CU; workgroup b (XCD b % 8) takes every
    //      (workgroups-on-that-XCD)-th tile of its XCD's contiguous chunk of the grouped tile order, so the tiles
    //      resident on an XCD at any time are neighbours sharing A / W panels in its L2.
    const int nt = p.tiles_m * p.tiles_n;
    int t_cur, t_end, t_stride;
    {
        const int bid = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
        const int cs = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q);
        t_stride = ((int)gridDim.x + 7 - xcd) >> 3;
        t_cur = cs + idx;
        t_end = cs + q + (xcd < r ? 1 : 0);
    }
    if (t_cur >= t_end) return;                                        // uniform for the whole workgroup

    int m0 = 0, n0 = 0;
    int offA[2][2], offW[2][2];
    // LDS-DMA sources of a tile: every wave copies rows [16w, 16w+16) of each half-tile (2 pieces of 8 rows)
    auto set_tile = [&](int t) {
        const int gsize = p.group_m * p.tiles_n;
        const int gid = t / gsize;
        const int first_m = gid * p.group_m;
        const int gm = min(p.tiles_m - first_m, p.group_m);
        const int in_g = t - gid * gsize;
        m0 = (first_m + in_g % gm) * B2;
        n0 = (in_g / gm) * B2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 128 * h + 16 * wave + 8 * j + (lane >> 3);     // row inside the 256-row tile
                const int cs = lane & 7;
                offA[h][j] = min(m0 + row, p.M - 1) * p.lda + 8 * (cs ^ swzA(row));
                offW[h][j] = min(n0 + row, p.N - 1) * p.ldw + 8 * (cs ^ swzW(row));
            }
    };
    // ---- fragment read bases (byte offsets inside a K-tile buffer); tile index adds an immediate
    //   A: row = 16*mi + li inside half wm   (swizzle depends on li only)
    //   W: row = 64*wn + 16*(li>>2) + 4*ni + (li&3) inside the 256-row tile, half wn>>1 (swizzle independent of ni)
    const int rowW0 = 64 * wn + 16 * (li >> 2) + (li & 3);
    const int rdA0 = wm * HALF_BYTES + li * 128 + ((g ^ swzA(li)) << 4);
    const int rdW0 = 2 * HALF_BYTES + (wn >> 1) * HALF_BYTES + (rowW0 & 127) * 128 + ((g ^ swzW(rowW0)) << 4);

    f32x4 acc[8][4];
    const int nk = p.K / BK;
    auto stageA = [&](int kt) {            // both A half-tiles of K-tile kt
        char* base = smem + (kt & 1) * KT_BYTES + wave * 2048;
        const int k0 = kt * BK;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(p.A + (size_t)(offA[h][j] + k0), base + h * HALF_BYTES + j * 1024);
    };
    auto stageW = [&](int kt) {
        char* base = smem + (kt & 1) * KT_BYTES + 2 * HALF_BYTES + wave * 2048;
        const int k0 = kt * BK;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(p.W + (size_t)(offW[h][j] + k0), base + h * HALF_BYTES + j * 1024);
    };

    // prologue loads of a tile: K-tile 0 (A and W) and, already in flight behind it, W(1)
    auto issue_prologue = [&]() {
        stageA(0);
        stageW(0);
        if (nk > 1) stageW(1);
    };
    set_tile(t_cur);
    issue_prologue();

    bf16x8 fa[8], fw0[4], fw1[4];                       // A(mh) k0/k1 x 4 tiles ; W(nh0), W(nh1): k0/k1 x 2 tiles
    for (;;) {
    const int em0 = m0, en0 = n0;                       // this tile's output origin (m0/n0 move on to the next tile early)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K-tile 0 complete (the up-to-4 youngest VM ops are W(1)'s LDS-DMA or the previous tile's epilogue stores)
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // stagger the second wave group by one barrier
    SEEDMI_SCHED_FENCE();

    for (int kt = 0; kt < nk; ++kt) {
        const char* sb = smem + (kt & 1) * KT_BYTES;
        const char* pa0 = sb + rdA0;                    // k-step 0
        const char* pa1 = sb + (rdA0 ^ 64);             // k-step 1
        const char* pw0 = sb + rdW0;
        const char* pw1 = sb + (rdW0 ^ 64);

        // ================= P1: (mh0, nh0) =================
#pragma unroll
        for (int t = 0; t < 2; ++t) { fw0[t] = *(const bf16x8*)(pw0 + t * 512); fw0[2 + t] = *(const bf16x8*)(pw1 + t * 512); }
#pragma unroll
        for (int t = 0; t < 4; ++t) { fa[t] = *(const bf16x8*)(pa0 + t * 2048); fa[4 + t] = *(const bf16x8*)(pa1 + t * 2048); }
        if (kt + 1 < nk) stageA(kt + 1);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw0[2 * ks + ni], fa[4 * ks + mi], acc[mi][ni], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();

        // ================= P2: (mh0, nh1) =================
#pragma unroll
        for (int t = 0; t < 2; ++t) { fw1[t] = *(const bf16x8*)(pw0 + (2 + t) * 512); fw1[2 + t] = *(const bf16x8*)(pw1 + (2 + t) * 512); }
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[2 * ks + ni], fa[4 * ks + mi], acc[mi][2 + ni], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();

        // ================= P3: (mh1, nh1) =================
#pragma unroll
        for (int t = 0; t < 4; ++t) { fa[t] = *(const bf16x8*)(pa0 + (4 + t) * 2048); fa[4 + t] = *(const bf16x8*)(pa1 + (4 + t) * 2048); }
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[4 + mi][2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[2 * ks + ni], fa[4 * ks + mi], acc[4 + mi][2 + ni], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();

        // ================= P4: (mh1, nh0) =================
        // K-tile kt+1 must be complete before anyone reads it in the next P1; its loads are 3-4 phases old.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (kt + 2 < nk) stageW(kt + 2);               // W slots of this parity were last read in P2
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        SEEDMI_SCHED_FENCE();
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[4 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw0[2 * ks + ni], fa[4 * ks + mi], acc[4 + mi][ni], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();          // matches the extra barrier the other group took up front
    SEEDMI_SCHED_FENCE();

    // every LDS read of this tile is done: start the next tile's prologue loads now so that their latency (and the
    // epilogue's own loads and stores) overlap instead of opening the next tile with an empty pipeline
    const int t_next = t_cur + t_stride;
    const bool more = t_next < t_end;
    if (more) {
        set_tile(t_next);
        issue_prologue();
    }
    if (p.skip_epilogue != 1) gemm_epilogue<EPI, 8>(p, acc, em0 + 128 * wm, en0 + 64 * wn + 16 * g, li);
    else if (acc[0][0][0] == 123.456f) p.C[0] = 0;      // keep the accumulators alive
    if (!more) break;
    t_cur = t_next;
    }
}

// ABL (timing ablations only, results are wrong): 1 = no fragment reads in the loop, 2 = no LDS-DMA in the loop,
// 4 = no barrier / vmcnt wait in the loop
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm256f_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, g = lane >> 4;

    // ---- persistent tile walk: the launch holds one workgroup per CU; workgroup b (XCD b % 8) takes every
    //      (workgroups-on-that-XCD)-th tile of its XCD's contiguous chunk of the grouped tile order, so the tiles
    //      resident on an XCD at any time are neighbours sharing A / W panels in its L2.
    const int nt = p.tiles_m * p.tiles_n;
    int t_cur, t_end, t_stride;
    {
        const int bid = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
        const int cs = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q);
        t_stride = ((int)gridDim.x + 7 - xcd) >> 3;
        t_cur = cs + idx;
        t_end = cs + q + (xcd < r ? 1 : 0);
    }
    if (t_cur >= t_end) return;                                        // uniform for the whole workgroup

    int m0 = 0, n0 = 0;
    int offA[2][2], offW[2][2];
    // LDS-DMA sources of a tile: every wave copies rows [16w, 16w+16) of each half-tile (2 pieces of 8 rows)
    auto set_tile = [&](int t) {
        const int gsize = p.group_m * p.tiles_n;
        const int gid = t / gsize;
        const int first_m = gid * p.group_m;
        const int gm = min(p.tiles_m - first_m, p.group_m);
        const int in_g = t - gid * gsize;
        m0 = (first_m + in_g % gm) * B2;
        n0 = (in_g / gm) * B2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 128 * h + 16 * wave + 8 * j + (lane >> 3);     // row inside the 256-row tile
                const int cs = lane & 7;
                offA[h][j] = min(((ABL & 8) ? 0 : m0) + row, p.M - 1) * p.lda + 8 * (cs ^ swzA(row));
                offW[h][j] = min(((ABL & 8) ? 0 : n0) + row, p.N - 1) * p.ldw + 8 * (cs ^ swzW(row));
            }
    };
    // ---- fragment read bases (byte offsets inside a K-tile buffer); tile index adds an immediate
    //   A: row = 16*mi + li inside half wm   (swizzle depends on li only)
    //   W: row = 64*wn + 16*(li>>2) + 4*ni + (li&3) inside the 256-row tile, half wn>>1 (swizzle independent of ni)
    const int rowW0 = 64 * wn + 16 * (li >> 2) + (li & 3);
    const int rdA0 = wm * HALF_BYTES + li * 128 + ((g ^ swzA(li)) << 4);
    const int rdW0 = 2 * HALF_BYTES + (wn >> 1) * HALF_BYTES + (rowW0 & 127) * 128 + ((g ^ swzW(rowW0)) << 4);

    f32x4 acc[8][4];
    const int nk = p.K / BK;
    auto stageA = [&](int kt) {            // both A half-tiles of K-tile kt
        char* base = smem + (kt & 1) * KT_BYTES + wave * 2048;
        const int k0 = (ABL & 8) ? 0 : kt * BK;            // ablation 8: every K-tile re-reads the same (L2-hot) 64 KiB
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(p.A + (size_t)(offA[h][j] + k0), base + h * HALF_BYTES + j * 1024);
    };
    auto stageW = [&](int kt) {
        char* base = smem + (kt & 1) * KT_BYTES + 2 * HALF_BYTES + wave * 2048;
        const int k0 = (ABL & 8) ? 0 : kt * BK;            // ablation 8: every K-tile re-reads the same (L2-hot) 64 KiB
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(p.W + (size_t)(offW[h][j] + k0), base + h * HALF_BYTES + j * 1024);
    };

    // prologue loads of a tile: K-tiles 0 and 1 (A and W each)
    auto issue_prologue = [&]() {
        stageA(0);
        stageW(0);
        if (nk > 1) { stageA(1); stageW(1); }
    };
    set_tile(t_cur);
    issue_prologue();

    // "free-running" schedule: ONE barrier per K-tile.  Fragments are double-buffered by k-step (32 deep): while the 32
    // MFMAs of one k-step issue from one register set, the 12 ds_read_b128 of the next k-step fill the other.  The barrier
    // sits after the last LDS read of K-tile t; behind it the wave requests the LDS-DMA of K-tile t+2 into the buffer just
    // vacated and the first fragments of K-tile t+1 (whose DMA was requested a whole K-tile earlier, so vmcnt(0) at the
    // barrier costs nothing).  No stagger: the two waves of a SIMD drift apart by themselves and share the matrix pipe.
    bf16x8 fa0[8], fw0[4], fa1[8], fw1[4];
    auto rd = [&](bf16x8 (&fa)[8], bf16x8 (&fw)[4], const char* sb, int ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fw[t] = *(const bf16x8*)(sb + ((rdW0 ^ (ks << 6)) + t * 512));
#pragma unroll
        for (int t = 0; t < 8; ++t) fa[t] = *(const bf16x8*)(sb + ((rdA0 ^ (ks << 6)) + t * 2048));
    };
    auto mm = [&](bf16x8 (&fa)[8], bf16x8 (&fw)[4]) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[mi][ni], 0, 0, 0);
    };
    // pin the issue order inside a k-step: 4 MFMAs, then 1-2 of the next k-step's fragment reads, and so on, so that the
    // wait in front of the first MFMA covers only the fragments requested a whole k-step ago
    auto interleave = [&]() {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    };
    for (;;) {
    const int em0 = m0, en0 = n0;                       // this tile's output origin (m0/n0 move on to the next tile early)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K-tile 0 complete (the up-to-8 youngest VM ops are K-tile 1's LDS-DMA or the previous tile's epilogue stores)
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SEEDMI_SCHED_FENCE();
    rd(fa0, fw0, smem, 0);
    if (ABL & 1) rd(fa1, fw1, smem, 1);

    for (int kt = 0; kt < nk; ++kt) {
        const char* sb = smem + (kt & 1) * KT_BYTES;
        // ---- k-step 0 of K-tile kt issues while k-step 1's fragments are read
        if (!(ABL & 1)) rd(fa1, fw1, sb, 1);
        mm(fa0, fw0);
        interleave();
        SEEDMI_SCHED_FENCE();
        // ---- all LDS reads of K-tile kt are complete; K-tile kt+1 has landed (requested one K-tile ago)
        if (!(ABL & 4)) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        SEEDMI_SCHED_FENCE();
        if (ABL & 16) {
            // branch-free request of K-tile kt+2 (a dead request lands in the 1 KiB scratch block), spread between the MFMAs
            const bool live = kt + 2 < nk;
            char* base = live ? smem + (kt & 1) * KT_BYTES + wave * 2048 : smem + 2 * KT_BYTES;
            const int hs = live ? HALF_BYTES : 0, js = live ? 1024 : 0, ws = live ? 2 * HALF_BYTES : 0;
            const int k0 = live ? (kt + 2) * BK : 0;
            rd(fa0, fw0, smem + ((kt + 1) & 1) * KT_BYTES, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    glds16(p.A + (size_t)(offA[h][j] + k0), base + h * hs + j * js);
                    glds16(p.W + (size_t)(offW[h][j] + k0), base + ws + h * hs + j * js);
                }
            mm(fa1, fw1);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        } else {
            if (!(ABL & 2) && kt + 2 < nk) { stageA(kt + 2); stageW(kt + 2); }      // into the buffer K-tile kt just vacated
            if (!(ABL & 1)) rd(fa0, fw0, smem + ((kt + 1) & 1) * KT_BYTES, 0);     // (after the last K-tile: stale LDS, unused)
            mm(fa1, fw1);
            interleave();
        }
        SEEDMI_SCHED_FENCE();
    }
    if (ABL & 16) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave has issued its last LDS read
    SEEDMI_SCHED_FENCE();

    // every LDS read of this tile is done: start the next tile's prologue loads now so that their latency (and the
    // epilogue's own loads and stores) overlap instead of opening the next tile with an empty pipeline
    const int t_next = t_cur + t_stride;
    const bool more = t_next < t_end;
    if (more) {
        set_tile(t_next);
        issue_prologue();
    }
    gemm_epilogue<EPI, 8>(p, acc, em0 + 128 * wm, en0 + 64 * wn + 16 * g, li);
    if (!more) break;
    t_cur = t_next;
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256x_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, g = lane >> 4;

    // ---- persistent tile walk: the launch holds one workgroup per CU; workgroup b (XCD b % 8) takes every
    //      (workgroups-on-that-XCD)-th tile of its XCD's contiguous chunk of the grouped tile order, so the tiles
    //      resident on an XCD at any time are neighbours sharing A / W panels in its L2.
    const int nt = p.tiles_m * p.tiles_n;
    int t_cur, t_end, t_stride;
    {
        const int bid = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
        const int cs = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q);
        t_stride = ((int)gridDim.x + 7 - xcd) >> 3;
        t_cur = cs + idx;
        t_end = cs + q + (xcd < r ? 1 : 0);
    }
    if (t_cur >= t_end) return;                                        // uniform for the whole workgroup

    int m0 = 0, n0 = 0;
    int offA[2][2], offW[2][2];
    // LDS-DMA sources of a tile: every wave copies rows [16w, 16w+16) of each half-tile (2 pieces of 8 rows)
    auto set_tile = [&](int t) {
        const int gsize = p.group_m * p.tiles_n;
        const int gid = t / gsize;
        const int first_m = gid * p.group_m;
        const int gm = min(p.tiles_m - first_m, p.group_m);
        const int in_g = t - gid * gsize;
        m0 = (first_m + in_g % gm) * B2;
        n0 = (in_g / gm) * B2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 128 * h + 16 * wave + 8 * j + (lane >> 3);     // row inside the 256-row tile
                const int cs = lane & 7;
                offA[h][j] = min(m0 + row, p.M - 1) * p.lda + 8 * (cs ^ swzA(row));
                offW[h][j] = min(n0 + row, p.N - 1) * p.ldw + 8 * (cs ^ swzA(row));
            }
    };
    // ---- fragment read bases for v_mfma_f32_32x32x16_bf16 (lane = row i in 0..31, k-half hl = lane >> 5; 16 k per step)
    //   A (activations, MFMA B operand): row = 32*mt + i inside half wm; chunk = 2*ks + hl
    //   W (weights, MFMA A operand): MFMA row rho = i feeds weight row 16*((rho>>2)&1) + 4*(rho>>3) + (rho&3) of the 32-row
    //   n-tile, so that the accumulator registers of a lane (rho = 4*hl + (reg&3) + 8*(reg>>2)) are 16 CONTIGUOUS columns.
    const int i32 = lane & 31, hl = lane >> 5;
    const int rowW0 = 64 * wn + 16 * ((i32 >> 2) & 1) + 4 * (i32 >> 3) + (i32 & 3);
    const int rdA0 = wm * HALF_BYTES + i32 * 128 + ((hl ^ swzA(i32)) << 4);
    const int rdW0 = 2 * HALF_BYTES + (wn >> 1) * HALF_BYTES + (rowW0 & 127) * 128 + ((hl ^ swzA(rowW0)) << 4);

    f32x16 acc[4][2];                                    // [m-tile of 32][n-tile of 32]
    const int nk = p.K / BK;
    auto stageA = [&](int kt) {            // both A half-tiles of K-tile kt
        char* base = smem + (kt & 1) * KT_BYTES + wave * 2048;
        const int k0 = kt * BK;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(p.A + (size_t)(offA[h][j] + k0), base + h * HALF_BYTES + j * 1024);
    };
    auto stageW = [&](int kt) {
        char* base = smem + (kt & 1) * KT_BYTES + 2 * HALF_BYTES + wave * 2048;
        const int k0 = kt * BK;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(p.W + (size_t)(offW[h][j] + k0), base + h * HALF_BYTES + j * 1024);
    };

    // prologue loads of a tile: K-tile 0 (A and W) and, already in flight behind it, W(1)
    auto issue_prologue = [&]() {
        stageA(0);
        stageW(0);
        if (nk > 1) stageW(1);
    };
    set_tile(t_cur);
    issue_prologue();

    bf16x8 fa[8], fw0[4], fw1[4];                       // A(mh): 2 m-tiles x 4 k-steps ; W(nh0), W(nh1): 4 k-steps each
    for (;;) {
    const int em0 = m0, en0 = n0;                       // this tile's output origin (m0/n0 move on to the next tile early)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // K-tile 0 complete (the up-to-4 youngest VM ops are W(1)'s LDS-DMA or the previous tile's epilogue stores)
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // stagger the second wave group by one barrier
    SEEDMI_SCHED_FENCE();

    for (int kt = 0; kt < nk; ++kt) {
        const char* sb = smem + (kt & 1) * KT_BYTES;
        // k-step ks reads chunk (2*ks + hl) ^ swz  =  base ^ (ks << 5)
#define PA(ks, mt) (*(const bf16x8*)(sb + ((rdA0 ^ ((ks) << 5)) + (mt) * 4096)))
#define PW(ks, nt) (*(const bf16x8*)(sb + ((rdW0 ^ ((ks) << 5)) + (nt) * 4096)))

        // ================= P1: (mh0, nh0) =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fw0[ks] = PW(ks, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa[ks] = PA(ks, 0); fa[4 + ks] = PA(ks, 1); }
        if (kt + 1 < nk) stageA(kt + 1);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0[ks], fa[4 * mt + ks], acc[mt][0], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();

        // ================= P2: (mh0, nh1) =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fw1[ks] = PW(ks, 1);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1[ks], fa[4 * mt + ks], acc[mt][1], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();

        // ================= P3: (mh1, nh1) =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa[ks] = PA(ks, 2); fa[4 + ks] = PA(ks, 3); }
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[2 + mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1[ks], fa[4 * mt + ks], acc[2 + mt][1], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();

        // ================= P4: (mh1, nh0) =================
        // K-tile kt+1 must be complete before anyone reads it in the next P1; its loads are 3-4 phases old.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (kt + 2 < nk) stageW(kt + 2);               // W slots of this parity were last read in P2
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[2 + mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0[ks], fa[4 * mt + ks], acc[2 + mt][0], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();          // matches the extra barrier the other group took up front
    SEEDMI_SCHED_FENCE();

    // every LDS read of this tile is done: start the next tile's prologue loads now so that their latency (and the
    // epilogue's own loads and stores) overlap instead of opening the next tile with an empty pipeline
    const int t_next = t_cur + t_stride;
    const bool more = t_next < t_end;
    if (more) {
        set_tile(t_next);
        issue_prologue();
    }
    gemm_epilogue32<EPI>(p, acc, em0 + 128 * wm + i32, en0 + 64 * wn + 16 * hl);
    if (!more) break;
    t_cur = t_next;
    }
#undef PA
#undef PW
}

template <int EPI>
int launch_gemm256(GemmParams p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KT_BYTES);
        attr_set = true;
    }
    p.tiles_m = (p.M + B2 - 1) / B2;
    p.tiles_n = (p.N + B2 - 1) / B2;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int nt = p.tiles_m * p.tiles_n;
    const int grid = (g_gemm_persist && nt > n_cu) ? n_cu : nt;
    hipLaunchKernelGGL(gemm256_kernel<EPI>, dim3(grid), dim3(512), 2 * KT_BYTES, stream, p);
    return seedmi_check_launch("gemm256");
}

template <int EPI>
int launch_gemm256f(GemmParams p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm256f_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KT_BYTES);
        attr_set = true;
    }
    p.tiles_m = (p.M + B2 - 1) / B2;
    p.tiles_n = (p.N + B2 - 1) / B2;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int nt = p.tiles_m * p.tiles_n;
    const int grid = (g_gemm_persist && nt > n_cu) ? n_cu : nt;
    if (EPI == EPI_BIAS && g_gemm_ablate) {
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KT_BYTES + 1024);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * KT_BYTES + 1024, stream, p);
        };
        switch (g_gemm_ablate) {
            case 1: go(gemm256f_kernel<EPI_BIAS, 1>); break;
            case 2: go(gemm256f_kernel<EPI_BIAS, 2>); break;
            case 3: go(gemm256f_kernel<EPI_BIAS, 3>); break;
            case 4: go(gemm256f_kernel<EPI_BIAS, 4>); break;
            case 6: go(gemm256f_kernel<EPI_BIAS, 6>); break;
            case 8: go(gemm256f_kernel<EPI_BIAS, 8>); break;
            case 16: go(gemm256f_kernel<EPI_BIAS, 16>); break;
            case 24: go(gemm256f_kernel<EPI_BIAS, 24>); break;
            case 12: go(gemm256f_kernel<EPI_BIAS, 12>); break;
            default: go(gemm256f_kernel<EPI_BIAS, 7>); break;
        }
        return seedmi_check_launch("gemm256f(ablation)");
    }
    hipLaunchKernelGGL(gemm256f_kernel<EPI>, dim3(grid), dim3(512), 2 * KT_BYTES, stream, p);
    return seedmi_check_launch("gemm256f");
}

template <int EPI>
int launch_gemm256x(GemmParams p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm256x_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * KT_BYTES);
        attr_set = true;
    }
    p.tiles_m = (p.M + B2 - 1) / B2;
    p.tiles_n = (p.N + B2 - 1) / B2;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int nt = p.tiles_m * p.tiles_n;
    const int grid = (g_gemm_persist && nt > n_cu) ? n_cu : nt;
    hipLaunchKernelGGL(gemm256x_kernel<EPI>, dim3(grid), dim3(512), 2 * KT_BYTES, stream, p);
    return seedmi_check_launch("gemm256x");
}

template <int EPI>
int launch_gemm128(const GemmParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm128_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
        attr_set = true;
    }
    const int grid = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL(gemm128_kernel<EPI>, dim3(grid), dim3(256), 2 * STAGE_BYTES, stream, p);
    return seedmi_check_launch("gemm128");
}

int g_gemm_variant = 0;      // 0 = auto, 128 / 256 = force a kernel (seedmi_set_option("gemm", v))
int g_gemm_min_tiles = 160;    // seedmi_set_option("gemm_min_tiles", n)

template <int EPI>
int launch_gemm(const GemmParams& p, hipStream_t s) {
    // the 256x256 kernel wants at least g_gemm_min_tiles tiles (one per CU is 256): below that the 128x128 kernel's four times
    // finer tiling fills the chip better (Q-Former GEMMs at M = B*32)
    const long long tiles256 = (long long)((p.M + B2 - 1) / B2) * ((p.N + B2 - 1) / B2);
    const bool big = p.M >= 1024 && p.N >= 256 && tiles256 >= g_gemm_min_tiles;
    if (g_gemm_variant == 232) return launch_gemm256x<EPI>(p, s);          // 256x256 tile on v_mfma_f32_32x32x16_bf16
    if (g_gemm_variant == 255) return launch_gemm256f<EPI>(p, s);          // 256x256, one barrier per K-tile, free-running waves
    const bool use256 = g_gemm_variant == 256 || (g_gemm_variant == 0 && big && SEEDMI_GEMM256_DEFAULT);
    return use256 ? launch_gemm256<EPI>(p, s) : launch_gemm128<EPI>(p, s);
}

}  // namespace

extern "C" int seedmi_set_option(const char* key, int value) {
    if (key && !strcmp(key, "gemm") && (value == 0 || value == 128 || value == 256 || value == 232 || value == 255)) {
        g_gemm_variant = value;
        return SEEDMI_OK;
    }
    if (key && !strcmp(key, "gemm_group_m") && value >= 1 && value <= 64) {
        g_group_m = value;
        return SEEDMI_OK;
    }
    if (key && !strcmp(key, "gemm_min_tiles") && value >= 1 && value <= 4096) {
        g_gemm_min_tiles = value;
        return SEEDMI_OK;
    }
    if (key && !strcmp(key, "gemm_ablate") && value >= 0 && value <= 35) {
        g_gemm_ablate = value;
        return SEEDMI_OK;
    }
    if (key && !strcmp(key, "gemm_persist") && (value == 0 || value == 1)) {
        g_gemm_persist = value;
        return SEEDMI_OK;
    }
    if (key && !strcmp(key, "tokenize_streams") && seedmi_tokenizer_set_streams(value) == SEEDMI_OK) return SEEDMI_OK;
    if (key && seedmi_llama_set_option(key, value) == SEEDMI_OK) return SEEDMI_OK;
    if (key && seedmi_attn_set_option(key, value) == SEEDMI_OK) return SEEDMI_OK;
    seedmi_set_error("seedmi_set_option: unknown option/value %s=%d", key ? key : "(null)", value);
    return SEEDMI_E_SHAPE;
}

extern "C" int seedmi_gemm_bf16(int M, int N, int K, const void* A, int lda, const void* W, int ldw, const void* bias,
                                const void* residual, int ldr, int epilogue, void* C, int ldc, int row_group,
                                int row_extra, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0) {
        seedmi_set_error("seedmi_gemm_bf16: bad shape M=%d N=%d K=%d (K must be a positive multiple of %d)", M, N, K, BK);
        return SEEDMI_E_SHAPE;
    }
    if ((lda % 8) || (ldw % 8) || (ldc % 8) || (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) & 15) ||
        (bias && ((uintptr_t)bias & 15)) || (residual && (((uintptr_t)residual & 15) || (ldr % 8)))) {
        seedmi_set_error("seedmi_gemm_bf16: pointers must be 16-byte aligned and leading dimensions multiples of 8");
        return SEEDMI_E_ALIGN;
    }
    // operand offsets are 32-bit element indices inside the kernels (LDS-DMA sources): keep every matrix below 2^31 elements
    const long long lim = 0x7fffffffLL;
    if ((long long)M * lda > lim || (long long)N * ldw > lim || ((long long)M + (long long)row_extra * (M / (row_group > 0 ? row_group : 1) + 1)) * ldc > lim ||
        (residual && (long long)M * ldr > lim)) {
        seedmi_set_error("seedmi_gemm_bf16: a matrix exceeds 2^31 elements (M=%d N=%d K=%d): split the batch", M, N, K);
        return SEEDMI_E_SHAPE;
    }
    if ((epilogue == EPI_BIAS_RESIDUAL || epilogue == EPI_PATCH_EMBED) && !residual) {
        seedmi_set_error("seedmi_gemm_bf16: residual epilogue without a residual pointer");
        return SEEDMI_E_SHAPE;
    }
    if (epilogue == EPI_SWIGLU && (N % 2)) {
        seedmi_set_error("seedmi_gemm_bf16: SWIGLU needs an even N (interleaved gate/up rows)");
        return SEEDMI_E_SHAPE;
    }
    GemmParams p;
    p.M = M; p.N = N; p.K = K;
    p.A = (const bf16_t*)A; p.lda = lda;
    p.W = (const bf16_t*)W; p.ldw = ldw;
    p.bias = (const bf16_t*)bias;
    p.R = (const bf16_t*)residual; p.ldr = ldr;
    p.C = (bf16_t*)C; p.ldc = ldc;
    p.tiles_m = (M + BM - 1) / BM;
    p.tiles_n = (N + BN - 1) / BN;
    p.group_m = g_group_m;
    p.skip_epilogue = (g_gemm_ablate == 32) ? 1 : (g_gemm_ablate == 33 ? 2 : (g_gemm_ablate == 34 ? 3 : (g_gemm_ablate == 35 ? 4 : 0)));
    p.row_group = row_group > 0 ? row_group : 1;
    p.row_extra = row_extra;
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case EPI_NONE: return launch_gemm<EPI_NONE>(p, s);
        case EPI_BIAS: return launch_gemm<EPI_BIAS>(p, s);
        case EPI_BIAS_GELU: return launch_gemm<EPI_BIAS_GELU>(p, s);
        case EPI_BIAS_RESIDUAL: return launch_gemm<EPI_BIAS_RESIDUAL>(p, s);
        case EPI_BIAS_TANH: return launch_gemm<EPI_BIAS_TANH>(p, s);
        case EPI_SWIGLU: return launch_gemm<EPI_SWIGLU>(p, s);
        case EPI_PATCH_EMBED: return launch_gemm<EPI_PATCH_EMBED>(p, s);
        case EPI_RELU: return launch_gemm<EPI_RELU>(p, s);
        default:
            seedmi_set_error("seedmi_gemm_bf16: unknown epilogue %d", epilogue);
            return SEEDMI_E_SHAPE;
    }
}
