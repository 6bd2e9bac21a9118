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
#include <atomic>
#include <type_traits>
#include "common.h"
#include "seedmi_internal.h"
#include "../../include/seedmi.h"

// s_setprio(1) around the MFMA runs of the 256x256 kernel: measured neutral-to-negative (-0.5 % end to end), off by default
#ifndef SEEDMI_GEMM_PRIO
#define SEEDMI_GEMM_PRIO 0
#endif
// 1: the persistent kernel issues the next tile's LDS-DMA after the epilogue's own loads have landed instead of before the epilogue
#ifndef SEEDMI_LATE_PROLOGUE
#define SEEDMI_LATE_PROLOGUE 0
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // A tile + W tile
// Process-wide tuning overrides (seedmi_set_option).  They select between kernels that compute the same result and are not part of
// the per-stream thread-safety contract: set them before concurrent use.  Relaxed atomics keep concurrent reads race-free.
#ifdef SEEDMI_DEVTOOLS
std::atomic<int> g_gemm_ablate{0};   // "gemm_ablate": timing-only ablations (devtools build only)
#endif
std::atomic<int> g_group_m{0};       // "gemm_group_m": m-tiles per L2 tile group (0 = by shape, see auto_group_m)
std::atomic<int> g_gemm_persist{1};  // "gemm_persist": persistent one-workgroup-per-CU launch of the 256x256 kernel
std::atomic<int> g_gemm_streamk{1};  // "gemm_streamk": stream-K tail when the caller passes a workspace
std::atomic<int> g_gemm_prefetch_r{0};   // "gemm_prefetch_residual": residual tile touched during the K loop (measured neutral: off)
std::atomic<int> g_gemm_residual_nt{1};  // "gemm_residual_nt": streaming stores for the BIAS_RESIDUAL output
// "gemm_store": how the 256x256 kernel's epilogues lay a row group's 16 rows x 128 bytes (the wave's 64-column span) over its two store instructions.
//   64: each instruction writes 64 contiguous bytes of all 16 rows (lane transposition by v_permlane16 / 32_swap; rounds 1-4)
//  128: each instruction writes the FULL 128-byte span of 8 rows (two DPP row rotations per register): half as many cache lines touched per
//       instruction, every line written by one instruction instead of two halves by two
std::atomic<int> g_gemm_store{128};
std::atomic<int> g_gemm_variant{0};  // "gemm": 0 = auto, 128 / 256 = force a kernel
// "gemm_sched": schedule variant of the 256x256 kernel (bits: see gemm256_kernel) for the three ViT epilogues.  31 = every measured gain of
// round 3 (epilogue-side wait, W pre-read, two LDS-DMA requests per phase with counted waits, early residual requests): bit-identical to
// schedule 0, +3..5 % per GEMM, +3.4 % end to end (profiles/r03_gemm_sched_ab_call*.json).  0 = the round-2 schedule.
constexpr int GEMM_SCHED_DEFAULT = 8273;         // two-phase K-tile + counted waits across tile boundaries + early residual rows + position-free body (see gemm256_kernel)
std::atomic<int> g_gemm_sched{GEMM_SCHED_DEFAULT};
#ifdef SEEDMI_DEVTOOLS
unsigned long long* g_gemm_dbg = nullptr;   // seedmi_gemm_phase_timing: device buffer for the phase clock stamps
#endif
std::atomic<int> g_gemm_min_tiles{160};   // "gemm_min_tiles": fewer 256x256 tiles than this -> 128x128 kernel

struct GemmParams {
    int M, N, K;
    const bf16_t* A; int lda;
    const bf16_t* W; int ldw;
    const bf16_t* bias;
    const bf16_t* R; int ldr;
    bf16_t* C; int ldc;
    int tiles_m, tiles_n;
    int group_m;                // m-tiles walked per group of the tile order (L2 locality)
    int xcd_map;                // 64x64 kernel: 1 = XCD-contiguous tile order ("gemm64_xcd", default), 0 = workgroup b takes tile b
#ifdef SEEDMI_DEVTOOLS
    unsigned long long* dbg;    // phase clock stamps of workgroup 0 (tools/gemm_phase_times.py), or null
    int skip_epilogue;          // timing ablations, seedmi_set_option("gemm_ablate", 32|33|34): 1 = no epilogue, 2 = epilogue without
                                // its stores, 3 = ordinary instead of streaming stores, 4 = streaming stores without the lane transposition
#else
    static constexpr int skip_epilogue = 0;
#endif
    // stream-K tail (gemm256 only; null = data-parallel walk): per-workgroup fp32 partial-tile slabs + one flag word each
    float* sk_slabs;
    unsigned* sk_flags;
    unsigned sk_epoch;
    // LayerNorm folded into the GEMM (eva_vit.py:199-202: norm1 -> qkv, norm2 -> fc1).  Consumer side (BIAS / BIAS_GELU): A holds the
    // UN-normalised rows x, W holds half(weight * gamma); with (mean, rstd) of every row in ln_stats, s_n = sum_k W'[n][k] in ln_colsum
    // and b'_n = bias_n + sum_k beta_k weight[n][k] in bias_f32 the epilogue forms  rstd * (acc - mean * s_n) + b'_n  =  LN(x) W^T + bias
    // with ONE rounding (the reference rounds LN(x) to half first).  Producer side (BIAS_RESIDUAL): stats_out receives, per row and
    // 64-column span, (sum, sum of squares) of the half outputs; seedmi_layernorm_stats_finalize turns them into (mean, rstd).
    const float* ln_stats;
    const float* ln_colsum;
    const float* bias_f32;
    float* stats_out;
    int stats_ld;               // rows per plane
    // statistics by 256-column tile (256x256 kernel only, see seedmi_gemm_ext_t): the producer writes one pair per row and n-tile; the consumer
    // receives ln_planes partial planes in ln_stats (ln_ld rows each) and forms (mean, rstd) itself
    int stats_by_tile;
    int ln_planes, ln_ld;
    float ln_inv_cols, ln_eps;
    int prefetch_residual;      // BIAS_RESIDUAL on the 256x256 kernel: touch the residual tile during the K loop ("gemm_prefetch_residual")
    int residual_nt;            // BIAS_RESIDUAL output stores: 1 = streaming (non-temporal), 0 = ordinary ("gemm_residual_nt")
    int store128;               // 256x256 kernel's full-span epilogues: 1 = 128 contiguous bytes of 8 rows per store instruction ("gemm_store")
    int row_group, row_extra;   // patch-embed: out_row = m + (m / row_group) * row_extra + row_extra ; res_row = m % row_group + row_extra
};

SEEDMI_DEVINL int swzA(int row) { return (row >> 1) & 7; }
SEEDMI_DEVINL int swzW(int row) { return ((row >> 1) & 1) | (((row >> 4) & 3) << 1); }

SEEDMI_DEVINL void glds16(const bf16_t* gptr, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same 16-byte-per-lane LDS-DMA request as a BUFFER load (buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds): the lane's byte offset
// inside the matrix sits in ONE VGPR for the whole tile, the K advance in the scalar offset - a request costs no VALU instruction.  The
// flat form above costs v_add_u32 + v_lshl_add_u64 per request (hipcc does not split a flat LDS-DMA address into SGPR base + VGPR offset):
// 24 of the 48 VALU instructions of two K-tiles of the 256x256 kernel, each of which takes the SIMD's VALU/MFMA issue port from the partner
// wave's MFMA stream for ~4 cycles.  Destination semantics are those of global_load_lds_dwordx4 (M0 base + 16 * lane, bases above 64 KiB
// included): tools/probes/buffer_lds_probe.hip.
SEEDMI_DEVINL void glds16_buf(const __amdgpu_buffer_rsrc_t rsrc, uint32_t lane_byte_off, uint32_t uniform_byte_off, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)lane_byte_off,
                                             (int)uniform_byte_off, 0, 0);
}

// ---- nn.GELU() on the half fc1 output as a bf16 -> bf16 table (tools/gen_gelu_lut.py): 5120 entries for 2^-16 <= |x| < 16, held
// in LDS behind the operand ring.  Per value: one rounding to bf16 (the reference's half fc1 output), ~6 integer VALU and one
// ds_read_u16 instead of ~20 VALU + v_rcp + v_exp — the erf epilogue was ~30 % of an ideal fc1 tile's MFMA time; the table is what
// torch returns for each input, so the activation is bit-identical to the reference's.  Values outside the table's range (tiny:
// 0.5x; huge: relu(x)) are rare: a wave-uniform branch sends such rows through the polynomial form.
// (fp16 build: fp16 has 1024 mantissa steps per binade - the same range would be a 40 KB table; there the activation is the arithmetic
//  gelu_erf of common.h on the half fc1 output, |abs err| <= 1.5e-7 against fp16's 2^-11 relative resolution)
#ifdef SEEDMI_F16
constexpr bool GELU_BY_TABLE = false;
#else
constexpr bool GELU_BY_TABLE = true;
#endif
constexpr int GELU_E_MIN = 111, GELU_N = 2560, GELU_LUT_BYTES = 2 * GELU_N * 2;
__device__ const uint16_t g_gelu_lut[2 * GELU_N] = {
#include "gelu_lut.inc"
};

SEEDMI_DEVINL void load_gelu_lut(char* lds, int tid, int nthreads) {
    const uint4* src = (const uint4*)g_gelu_lut;
    for (int i = tid; i < GELU_LUT_BYTES / 16; i += nthreads) *(uint4*)(lds + 16 * i) = src[i];
}

// two bf16 values packed in w -> their GELUs packed the same way; bad is set when either is outside the table
SEEDMI_DEVINL uint32_t gelu_lut_pair(const char* lut, uint32_t w, bool& bad) {
    const uint32_t lo = w & 0xffffu, hi = w >> 16;
    uint32_t il = (lo & 0x7fffu) - (GELU_E_MIN << 7), ih = (hi & 0x7fffu) - (GELU_E_MIN << 7);
    bad |= (il >= (uint32_t)GELU_N) | (ih >= (uint32_t)GELU_N);
    il = min(il, (uint32_t)(GELU_N - 1)) + (lo >> 15) * GELU_N;
    ih = min(ih, (uint32_t)(GELU_N - 1)) + (hi >> 15) * GELU_N;
    const uint32_t rl = *(const uint16_t*)(lut + 2 * il), rh = *(const uint16_t*)(lut + 2 * ih);
    return rl | (rh << 16);
}
SEEDMI_DEVINL bool gelu_in_table(uint32_t h) { return ((h & 0x7fffu) - (GELU_E_MIN << 7)) < (uint32_t)GELU_N; }

// ---- shared epilogue: the lane owns rows mrow0 + 16*mi + li (mi < MT) and the 16 contiguous columns nb..nb+15
// LANE4 (lane = li + 16 g, the four lanes of a row own adjacent 16-column groups): when the wave's whole 64-column span lies
// inside N the 16-byte pieces of the four lanes are exchanged so that a store instruction writes whole cache lines: rounds 1-4 64 contiguous
// bytes of each of 16 rows (v_permlane16_swap / v_permlane32_swap; whole 32-byte sectors instead of half sectors: -7 % on the ViT QKV GEMM),
// round 5 the full 128-byte span of 8 rows (rows_to_full_lines below: a further +1 ... 4.5 % per ViT GEMM, +2.4 % per tokenize pass).
struct NoHook { SEEDMI_DEVINL void operator()() const {} };

// (the product build knows only the full-line layout; the devtools build keeps rounds 1-4's half-line layout behind "gemm_store" = 64 for the A/B
// in profiles/r05_store128_*.json)
#ifdef SEEDMI_DEVTOOLS
#define SEEDMI_FULL_LINES(p) ((p).store128 != 0)
#else
#define SEEDMI_FULL_LINES(p) true
#endif
// rounds 1-4: pieces [0,2,4,6] / [1,3,5,7] of the row's eight -> permlane16_swap [0,1,4,5] / [2,3,6,7] -> permlane32_swap [0,1,2,3] / [4,5,6,7]
SEEDMI_DEVINL void rows_to_half_lines(const unsigned (&a)[4], const unsigned (&c)[4], seedmi_u32x4& o1, seedmi_u32x4& o2) {
    unsigned x[4], y[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const auto t1 = __builtin_amdgcn_permlane16_swap(a[d], c[d], false, false);
        const auto t2 = __builtin_amdgcn_permlane32_swap(t1[0], t1[1], false, false);
        x[d] = t2[0];
        y[d] = t2[1];
    }
    o1 = (seedmi_u32x4){x[0], x[1], x[2], x[3]};
    o2 = (seedmi_u32x4){y[0], y[1], y[2], y[3]};
}
// where the two registers of a lane go: row offsets inside the 16-row group and element columns, for either layout
struct SpanStoreLane {
    int r1, r2;
    uint32_t c1, c2;
    SEEDMI_DEVINL SpanStoreLane(bool full_lines, int nb, int li) {
        const uint32_t wcol = (uint32_t)((nb & ~63) + 8 * ((nb >> 4) & 3));
        r1 = full_lines ? (li & 7) : li;
        r2 = full_lines ? (li & 7) + 8 : li;
        c1 = full_lines ? (uint32_t)(nb + 8 * (li >> 3)) : wcol;
        c2 = full_lines ? c1 : wcol + 32;
    }
};

// producer side of the LayerNorm fold: (sum, sum of squares) of one row's 16 packed half outputs of this lane, reduced over the four
// lanes (li + 16 g) that share the row's 64-column span; every lane of the span returns the span's totals
SEEDMI_DEVINL float2 row_stats(const uint32_t (&pk)[8], int nvalid) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float a = (2 * i < nvalid) ? lo_bf(pk[i]) : 0.f, b = (2 * i + 1 < nvalid) ? hi_bf(pk[i]) : 0.f;
        s1 += a + b;
        s2 = fmaf(a, a, fmaf(b, b, s2));
    }
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    return make_float2(s1, s2);
}
// span-major partials, [span][stats_ld rows]: the 16 lanes li of a store write 128 contiguous bytes, and the finalize kernel (one thread per
// row) reads every span plane coalesced.  Written by the span's first column group only.
SEEDMI_DEVINL void store_row_stats(const GemmParams& p, float2 st, int m, int nb) {
    if ((nb & 63) == 0 && m < p.M) *(float2*)(p.stats_out + ((size_t)(nb >> 6) * p.stats_ld + m) * 2) = st;
}

// LayerNorm fold, consumer side of the 256x256 kernel: where this lane finds the tile's fold operands in LDS (put there by the
// tile's prologue LDS-DMA): cs = column sums of the lane's 16 columns (the folded bias 1 KiB behind them), st = (mean, rstd) of row li of
// the wave's 128 rows (row 16 mi + li: + 128 mi bytes)
struct FoldLds { const char* cs = nullptr; const char* st = nullptr; };
// acc <- rstd * acc + (bias - mean * rstd * colsum), in place (the consumer epilogue then treats the accumulators as finished sums)
SEEDMI_DEVINL void fold_accumulators(f32x4 (&acc)[8][4], const FoldLds& fold) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const f32x4 c4 = *(const f32x4*)(fold.cs + 16 * ni), b4 = *(const f32x4*)(fold.cs + 1024 + 16 * ni);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const float2 st = *(const float2*)(fold.st + 128 * mi);
            const float nmr = -st.x * st.y;                                // -mean * rstd
            const f32x4 t = __builtin_elementwise_fma((f32x4){nmr, nmr, nmr, nmr}, c4, b4);
            acc[mi][ni] = __builtin_elementwise_fma((f32x4){st.y, st.y, st.y, st.y}, acc[mi][ni], t);
        }
    }
}

// after_loads: called once, after the epilogue's up-front loads have been issued AND waited for and before its first store (the
// persistent kernel starts the next tile's LDS-DMA there: hipcc waits vmcnt(0) for every ordinary load while LDS-DMA is in flight,
// so DMA issued ahead of the bias / residual loads puts its own latency into the epilogue's critical path)
// row_end: rows >= row_end are computed but not stored (the 256x256 kernel's re-divided ragged tiles own 64 of the 128 rows); < 0 = p.M
template <int EPI, int MT, bool LANE4 = true, typename Hook = NoHook, bool LNF = false>
SEEDMI_DEVINL void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[MT][4], int mrow0, int nb, int li, const char* lut = nullptr,
                                 Hook after_loads = Hook(), int row_end = -1) {
    const int Mend = row_end < 0 ? p.M : row_end;
    const int span0 = nb & ~63;                                   // first column of the wave's 64-column span (wave-uniform)
    // (the LayerNorm-fold variants require N % 64 == 0: a wave's span is inside N or outside it as a whole, no ragged code)
    const bool span_full = LANE4 && EPI != EPI_SWIGLU && (LNF || span0 + 64 <= p.N) && p.skip_epilogue == 0;
    if (nb >= p.N) {
        after_loads();
        return;
    }
    const bool full = LNF || (nb + 16 <= p.N);
    float bias[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bias[i] = 0.f;
    // LayerNorm fold, consumer side (BIAS / BIAS_GELU).  256x256 kernel (MT = 8): the tile's column sums, folded bias and (mean, rstd)
    // pairs were brought into LDS by the tile's own prologue LDS-DMA (FoldLds): the epilogue has no global load at all and holds 10
    // transient registers for them (held in registers - 32 column values + 16 row values next to 128 accumulators - the GELU variant
    // spilled 130-210 B per lane, with reloads behind the tile's stores).  128x128 kernel (MT = 4): registers, loaded here.
    constexpr bool CONSUME = LNF && EPI != EPI_BIAS_RESIDUAL;
    constexpr bool FOLD_REGS = CONSUME && MT <= 4;
    float csum[FOLD_REGS ? 16 : 1];
    float2 rstat[FOLD_REGS ? MT : 1];
    if (FOLD_REGS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // (unsigned 32-bit offsets from the kernel-argument bases: one offset register, not eight hoisted pointers)
            const float4 c4 = *(const float4*)((const char*)p.ln_colsum + ((uint32_t)nb * 4u + 16u * i));
            const float4 b4 = *(const float4*)((const char*)p.bias_f32 + ((uint32_t)nb * 4u + 16u * i));
            csum[4 * i] = c4.x; csum[4 * i + 1] = c4.y; csum[4 * i + 2] = c4.z; csum[4 * i + 3] = c4.w;
            bias[4 * i] = b4.x; bias[4 * i + 1] = b4.y; bias[4 * i + 2] = b4.z; bias[4 * i + 3] = b4.w;
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const uint32_t row = (uint32_t)min(mrow0 + 16 * mi + li, p.M - 1);
            if (p.ln_planes > 0) {
                // partial planes (the producer GEMM's span-major (sum, sum of squares) pairs) instead of finished statistics: the row's pairs
                // summed in plane order and finished exactly as seedmi_layernorm_stats_finalize does - the small-M kernels (64x64, 128x128)
                // then need no finalize launch between two GEMMs (one image: 77 launches of ~6.7 us)
                // (16 planes requested at a time, THEN summed in plane order: written as one load + one add per plane hipcc made it a loop of
                //  `global_load; s_waitcnt vmcnt(0); add` - 22 serial L2 round trips per epilogue, most of a one-image QKV / fc1 launch)
                float s1 = 0.f, s2 = 0.f;
                // (requested 16 / 4 / 1 planes at a time and THEN summed in plane order - 22 planes = 16 + 4 + 1 + 1: four L2 round trips; written
                //  as one load + one add per plane hipcc emits `global_load; s_waitcnt vmcnt(0); add` per plane, 22 serial round trips per
                //  epilogue.  No masked lanes: a clamped-and-masked 16-wide form measured NOT bit-equal to the finalize kernel's chain.)
                const char* pb = (const char*)p.ln_stats + 8 * (size_t)row;
                const size_t pstride = 8 * (size_t)p.ln_ld;
                int pl = 0;
                auto take = [&](auto nconst) {
                    constexpr int NP = decltype(nconst)::value;
                    float2 t[NP];
#pragma unroll
                    for (int j = 0; j < NP; ++j) t[j] = *(const float2*)(pb + (size_t)(pl + j) * pstride);
#pragma unroll
                    for (int j = 0; j < NP; ++j) { s1 += t[j].x; s2 += t[j].y; }
                    pl += NP;
                };
                while (pl + 16 <= p.ln_planes) take(std::integral_constant<int, 16>());
                while (pl + 4 <= p.ln_planes) take(std::integral_constant<int, 4>());
                while (pl < p.ln_planes) take(std::integral_constant<int, 1>());
                rstat[mi] = seedmi_ln_finish(s1, s2, p.ln_inv_cols, p.ln_eps);
            } else {
                rstat[mi] = *(const float2*)((const char*)p.ln_stats + 8u * row);
            }
        }
    } else if (!CONSUME && EPI != EPI_NONE && p.bias) {
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
    if (!(EPI == EPI_BIAS_RESIDUAL || EPI == EPI_PATCH_EMBED)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(bias[i]));     // the bias has landed
        if (CONSUME) {
            // the fold is applied to the accumulators in place, in a pass of its own: nothing of it lives through the activation /
            // packing code below
            if (FOLD_REGS) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const float2 st = rstat[FOLD_REGS ? mi : 0];
                    const float nmr = -st.x * st.y;                        // -mean * rstd
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[mi][ni][r] = fmaf(st.y, acc[mi][ni][r], fmaf(nmr, csum[4 * ni + r], bias[4 * ni + r]));
                }
            }                                                      // (MT = 8: the caller has already run fold_accumulators)
            __builtin_amdgcn_sched_barrier(0);
        }
        after_loads();
    }
    // residual / pos_embed rows of ALL the lane's rows are requested up front: one exposed HBM latency per tile instead
    // of one per row (the per-row form serialised 8 round trips and cost the proj GEMM 25 %).  With 8 row groups that is 64
    // registers: the 256x256 kernel's BIAS_RESIDUAL tiles take gemm_epilogue_residual8 instead and only ragged edges come here,
    // row by row.
    constexpr bool RES = (EPI == EPI_BIAS_RESIDUAL || EPI == EPI_PATCH_EMBED);
    constexpr bool PREFETCH_R = RES && !(EPI == EPI_BIAS_RESIDUAL && MT > 4);
    uint4 rr[PREFETCH_R ? MT : 1][2];
    if (PREFETCH_R) {
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
    if (RES) after_loads();                                        // (ragged / small-tile paths: no ordering benefit sought)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int m = mrow0 + 16 * mi + li;
        if (!span_full && m >= Mend) continue;                    // (the transposing path keeps every lane of the row alive)
        float v[16];
        if (CONSUME) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * ni + r] = acc[mi][ni][r];
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * ni + r] = acc[mi][ni][r] + bias[4 * ni + r];
        }

        int out_row = m;
        uint32_t pk[8];                                                    // packed result words (table path only)
        bool packed = false;
        if (EPI == EPI_BIAS_GELU && lut && GELU_BY_TABLE) {                // GELU of the half fc1 output, by table
            bool bad = false;
            uint32_t hw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                hw[i] = pack2bf(v[2 * i], v[2 * i + 1]);
                pk[i] = gelu_lut_pair(lut, hw[i], bad);
            }
            if (__builtin_amdgcn_ballot_w64(bad) != 0) {                  // wave-uniform, rare: |x| < 2^-16 or >= 16 somewhere
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t lo = hw[i] & 0xffffu, hi = hw[i] >> 16;
                    const uint32_t fl = f2bf(gelu_erf(lo_bf(hw[i]))), fh = f2bf(gelu_erf(hi_bf(hw[i])));
                    const uint32_t rl = gelu_in_table(lo) ? (pk[i] & 0xffffu) : fl;
                    const uint32_t rh = gelu_in_table(hi) ? (pk[i] >> 16) : fh;
                    pk[i] = rl | (rh << 16);
                }
            }
            packed = true;
        } else if (EPI == EPI_BIAS_GELU) {
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
                const uint4 r0 = PREFETCH_R ? rr[PREFETCH_R ? mi : 0][0] : *(const uint4*)rp;
                const uint4 r1 = PREFETCH_R ? rr[PREFETCH_R ? mi : 0][1] : *(const uint4*)(rp + 8);
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

        if (EPI == EPI_BIAS_RESIDUAL && LNF) {
            uint32_t spk[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) spk[i] = pack2bf(v[2 * i], v[2 * i + 1]);
            if (m < Mend) store_row_stats(p, row_stats(spk, 16), m, nb);
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
            // (element offsets fit 32 bits: the entry point refuses matrices of 2^31 elements or more)
            bf16_t* cp = p.C + ((uint32_t)out_row * (uint32_t)p.ldc + (uint32_t)nb);
            if (full) {
                uint4 s0, s1;
                if (EPI == EPI_BIAS_GELU && packed) {
                    s0 = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    s1 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                } else {
                    s0.x = pack2bf(v[0], v[1]); s0.y = pack2bf(v[2], v[3]); s0.z = pack2bf(v[4], v[5]); s0.w = pack2bf(v[6], v[7]);
                    s1.x = pack2bf(v[8], v[9]); s1.y = pack2bf(v[10], v[11]); s1.z = pack2bf(v[12], v[13]); s1.w = pack2bf(v[14], v[15]);
                }
                if (span_full) {
                    // rows of 16 lanes = column groups g: s0 = pieces [0,2,4,6], s1 = [1,3,5,7] of the row's eight 16-byte pieces;
                    // permlane16_swap -> [0,1,4,5] / [2,3,6,7]; permlane32_swap -> [0,1,2,3] / [4,5,6,7]
                    // ("gemm_store" = 128: the full 128-byte span of 8 rows per instruction instead - rows_to_full_lines)
                    const unsigned a[4] = {s0.x, s0.y, s0.z, s0.w}, c[4] = {s1.x, s1.y, s1.z, s1.w};
                    seedmi_u32x4 oa, oc;
                    // (the patch embedding maps rows - out_row != m - and keeps the lane's own row)
                    const bool full_lines = EPI != EPI_PATCH_EMBED && SEEDMI_FULL_LINES(p);
                    if (full_lines) rows_to_full_lines(a, c, oa, oc);
                    else rows_to_half_lines(a, c, oa, oc);
                    const SpanStoreLane sl(full_lines, nb, li);
                    const int d1 = sl.r1 - li, d2 = sl.r2 - li;              // 0 / 0 for the half-line layout
                    if (m + d1 < Mend) __builtin_nontemporal_store(oa, (seedmi_u32x4*)(p.C + ((uint32_t)(out_row + d1) * (uint32_t)p.ldc + sl.c1)));
                    if (m + d2 < Mend) __builtin_nontemporal_store(oc, (seedmi_u32x4*)(p.C + ((uint32_t)(out_row + d2) * (uint32_t)p.ldc + sl.c2)));
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
                for (int i = 0; i < 16; ++i)
                    if (nb + i < p.N) cp[i] = (EPI == EPI_BIAS_GELU && packed) ? (bf16_t)((pk[i >> 1] >> (16 * (i & 1))) & 0xffffu) : f2bf(v[i]);
            }
        }
        // (the fold variants have no ragged branches: without this the scheduler overlaps the rows' table reads across iterations
        // until the straight-line code needs more than 256 registers)
        if (CONSUME) __builtin_amdgcn_sched_barrier(0);
    }
}

// BIAS_RESIDUAL epilogue of the 256x256 kernel (8 row groups per lane) for a wave whose 64-column span lies inside N.
// Same arithmetic as gemm_epilogue<EPI_BIAS_RESIDUAL, 8>, ordered for the memory pipe: vmcnt retires in order and hipcc waits
// vmcnt(0) for every ordinary load while LDS-DMA is in flight, so a load issued AFTER the tile's first stores (a second batch of
// residual rows, a register reloaded from scratch) waits for those stores to reach memory - microseconds each.  The generic
// form needed 64 registers for the residual rows, spilled, and paid exactly that: +87 us on the ViT proj GEMM, +68 us on fc2 at
// B = 256 over the bias-only epilogue.  Here the residual rows are fetched in two halves of 32 registers and the first half's
// finished rows are HELD (packed, 32 registers, while their accumulators die) until the second half's loads have been issued: every
// load of the epilogue precedes every store, nothing spills.
template <bool STATS, typename Hook, bool EARLY = false>
SEEDMI_DEVINL void gemm_epilogue_residual8(const GemmParams& p, f32x4 (&acc)[8][4], int mrow0, int nb, int li, Hook after_loads,
                                           char* stat_lds, int row_end = -1) {
    const int Mend = row_end < 0 ? p.M : row_end;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    float bias[16];
    {
        const uint4 b0 = p.bias ? *(const uint4*)(p.bias + nb) : make_uint4(0, 0, 0, 0);
        const uint4 b1 = p.bias ? *(const uint4*)(p.bias + nb + 8) : make_uint4(0, 0, 0, 0);
        const uint32_t bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) { bias[2 * i] = lo_bf(bw[i]); bias[2 * i + 1] = hi_bf(bw[i]); }
    }
    uint4 rr[4][2];
    auto load_rows = [&](int h) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const bf16_t* rp = p.R + (size_t)min(mrow0 + 16 * (4 * h + mi) + li, p.M - 1) * p.ldr + nb;
            rr[mi][0] = *(const uint4*)rp;
            rr[mi][1] = *(const uint4*)(rp + 8);
        }
    };
    // EARLY: rows 4..7 in their own registers, requested once rows 0 and 1 are finished (16 registers of residual + 32 accumulators freed,
    // 16 taken by the two held rows: room for 32), i.e. a whole two rows of arithmetic before they are needed
    uint4 r2[EARLY ? 4 : 1][2];
    auto load_rows2 = [&]() {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const bf16_t* rp = p.R + (size_t)min(mrow0 + 16 * (4 + mi) + li, p.M - 1) * p.ldr + nb;
            r2[EARLY ? mi : 0][0] = *(const uint4*)rp;
            r2[EARLY ? mi : 0][1] = *(const uint4*)(rp + 8);
        }
    };
    // finished row (packed, lanes exchanged so that a store instruction writes the full 128-byte span of 8 rows)
    auto finish_row = [&](int mi_abs, int mi_rr, u32x4_t& oa, u32x4_t& oc) {
        const uint4 r0 = (EARLY && mi_abs >= 4) ? r2[EARLY ? mi_rr : 0][0] : rr[mi_rr][0];
        const uint4 r1 = (EARLY && mi_abs >= 4) ? r2[EARLY ? mi_rr : 0][1] : rr[mi_rr][1];
        const uint32_t rw[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        float v[16];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * ni + r] = acc[mi_abs][ni][r] + bias[4 * ni + r];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[2 * i] = rbf(v[2 * i]) + lo_bf(rw[i]);               // half GEMM output + half residual
            v[2 * i + 1] = rbf(v[2 * i + 1]) + hi_bf(rw[i]);
        }
        unsigned a[4] = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
        unsigned c[4] = {pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15])};
        if (STATS) {                                               // LayerNorm fold, producer side (before the lane transposition)
            const uint32_t spk[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
            const float2 st = row_stats(spk, 16);                  // parked in the wave's LDS slice: no register held, no store yet
            if ((nb & 48) == 0) *(float2*)(stat_lds + 8 * (16 * mi_abs + li)) = st;
        }
        if (SEEDMI_FULL_LINES(p)) rows_to_full_lines(a, c, oa, oc);
        else rows_to_half_lines(a, c, oa, oc);
    };
    const SpanStoreLane sl(SEEDMI_FULL_LINES(p), nb, li);
    auto store_row = [&](int mi_abs, const u32x4_t& oa, const u32x4_t& oc) {
        const int m1 = mrow0 + 16 * mi_abs + sl.r1, m2 = mrow0 + 16 * mi_abs + sl.r2;
        u32x4_t* w1 = (u32x4_t*)(p.C + ((uint32_t)m1 * (uint32_t)p.ldc + sl.c1));
        u32x4_t* w2 = (u32x4_t*)(p.C + ((uint32_t)m2 * (uint32_t)p.ldc + sl.c2));
        if (p.residual_nt) {
            if (m1 < Mend) __builtin_nontemporal_store(oa, w1);
            if (m2 < Mend) __builtin_nontemporal_store(oc, w2);
        } else {                                                  // the residual stream is re-read soon: let it allocate in the caches
            if (m1 < Mend) *w1 = oa;
            if (m2 < Mend) *w2 = oc;
        }
    };
    u32x4_t ha[4], hc[4];
    load_rows(0);
    if (EARLY) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) finish_row(mi, mi, ha[mi], hc[mi]);
        __builtin_amdgcn_sched_barrier(0);
        load_rows2();                                             // the last loads of the epilogue, two rows of arithmetic ahead of their use
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 2; mi < 4; ++mi) finish_row(mi, mi, ha[mi], hc[mi]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
            asm volatile("" : "+v"(r2[EARLY ? mi : 0][0].x), "+v"(r2[EARLY ? mi : 0][0].y), "+v"(r2[EARLY ? mi : 0][0].z), "+v"(r2[EARLY ? mi : 0][0].w),
                         "+v"(r2[EARLY ? mi : 0][1].x), "+v"(r2[EARLY ? mi : 0][1].y), "+v"(r2[EARLY ? mi : 0][1].z), "+v"(r2[EARLY ? mi : 0][1].w));
    } else {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) finish_row(mi, mi, ha[mi], hc[mi]);
        __builtin_amdgcn_sched_barrier(0);
        load_rows(1);                                                 // the last loads of the epilogue ...
        // ... must have LANDED before the first store is issued (a later wait for them would also wait for the stores in front of it):
        // an empty asm that "reads" the loaded registers makes the compiler place its vmcnt wait here
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
            asm volatile("" : "+v"(rr[mi][0].x), "+v"(rr[mi][0].y), "+v"(rr[mi][0].z), "+v"(rr[mi][0].w), "+v"(rr[mi][1].x), "+v"(rr[mi][1].y),
                         "+v"(rr[mi][1].z), "+v"(rr[mi][1].w));
    }
    __builtin_amdgcn_sched_barrier(0);
    after_loads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) store_row(mi, ha[mi], hc[mi]);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        u32x4_t oa, oc;
        finish_row(4 + mi, mi, oa, oc);
        store_row(4 + mi, oa, oc);
    }
    if (STATS && !p.stats_by_tile) {                  // (by tile: the workgroup sums its four spans after the tile - combine_tile_stats)
        // the wave's 128 (sum, sum of squares) pairs of this span: 1 KiB contiguous in the span's plane, two 512-byte store instructions
        // (lane id taken here, by volatile asm: derived from threadIdx up front it is one more value held across the whole tile, and
        // what hipcc then spills is reloaded right here, behind the tile's stores)
        int lane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = mrow0 + 64 * j + lane;
            const float2 st = *(const float2*)(stat_lds + 8 * (64 * j + lane));
            if (m < Mend) *(float2*)(p.stats_out + ((size_t)(nb >> 6) * p.stats_ld + m) * 2) = st;
        }
    }
}

// BIAS / BIAS_GELU epilogue of the 256x256 kernel's LayerNorm-fold consumers (QKV, fc1) for the schedules that wait for the NEXT tile's
// first K-tile inside the epilogue (SCHED bit 0): the accumulators are finished sums already (fold_accumulators ran), there is no
// global load, N % 64 == 0 (no ragged span).  Same arithmetic as gemm_epilogue<EPI, 8, true, ., true>.  The first four rows are finished
// and HELD (packed, 32 registers, while their 64 accumulators die) so that the next tile's LDS-DMA - issued right before the fold pass -
// has the whole of that work to land in; then the hook waits for it (vmcnt retires in order: waiting AFTER the stores would also wait
// for the stores, which is what the tile's opening wait used to do), and only then the tile's 16 stores go out.
template <int EPI, typename Hook>
SEEDMI_DEVINL void gemm_epilogue_fold8(const GemmParams& p, f32x4 (&acc)[8][4], int mrow0, int nb, int li, const char* lut, Hook after_loads,
                                       int row_end = -1) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const int Mend = row_end < 0 ? p.M : row_end;
    if (nb >= p.N) {
        after_loads();
        return;
    }
    auto finish_row = [&](int mi, u32x4_t& oa, u32x4_t& oc) {
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = pack2bf(acc[mi][i >> 1][2 * (i & 1)], acc[mi][i >> 1][2 * (i & 1) + 1]);
        if (EPI == EPI_BIAS_GELU && !GELU_BY_TABLE) {                 // fp16 build: arithmetic GELU of the half fc1 output
#pragma unroll
            for (int i = 0; i < 8; ++i) pk[i] = pack2bf(gelu_erf(lo_bf(pk[i])), gelu_erf(hi_bf(pk[i])));
        } else if (EPI == EPI_BIAS_GELU) {                            // GELU of the half fc1 output, by table (see gemm_epilogue)
            bool bad = false;
            uint32_t hw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                hw[i] = pk[i];
                pk[i] = gelu_lut_pair(lut, hw[i], bad);
            }
            if (__builtin_amdgcn_ballot_w64(bad) != 0) {              // wave-uniform, rare: |x| < 2^-16 or >= 16 somewhere
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t lo = hw[i] & 0xffffu, hi = hw[i] >> 16;
                    const uint32_t fl = f2bf(gelu_erf(lo_bf(hw[i]))), fh = f2bf(gelu_erf(hi_bf(hw[i])));
                    const uint32_t rl = gelu_in_table(lo) ? (pk[i] & 0xffffu) : fl;
                    const uint32_t rh = gelu_in_table(hi) ? (pk[i] >> 16) : fh;
                    pk[i] = rl | (rh << 16);
                }
            }
        }
        unsigned a[4] = {pk[0], pk[1], pk[2], pk[3]}, c[4] = {pk[4], pk[5], pk[6], pk[7]};
        if (SEEDMI_FULL_LINES(p)) rows_to_full_lines(a, c, oa, oc);
        else rows_to_half_lines(a, c, oa, oc);
    };
    const SpanStoreLane sl(SEEDMI_FULL_LINES(p), nb, li);
    auto store_row = [&](int mi, const u32x4_t& oa, const u32x4_t& oc) {
        const int m1 = mrow0 + 16 * mi + sl.r1, m2 = mrow0 + 16 * mi + sl.r2;
        if (m1 < Mend) __builtin_nontemporal_store(oa, (u32x4_t*)(p.C + ((uint32_t)m1 * (uint32_t)p.ldc + sl.c1)));
        if (m2 < Mend) __builtin_nontemporal_store(oc, (u32x4_t*)(p.C + ((uint32_t)m2 * (uint32_t)p.ldc + sl.c2)));
    };
    u32x4_t ha[4], hc[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        finish_row(mi, ha[mi], hc[mi]);
        __builtin_amdgcn_sched_barrier(0);
    }
    after_loads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) store_row(mi, ha[mi], hc[mi]);
#pragma unroll
    for (int mi = 4; mi < 8; ++mi) {
        u32x4_t oa, oc;
        finish_row(mi, oa, oc);
        store_row(mi, oa, oc);
        __builtin_amdgcn_sched_barrier(0);
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

template <int EPI, bool LNF = false>
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

    if (EPI == EPI_BIAS_GELU && GELU_BY_TABLE) load_gelu_lut(smem + 2 * STAGE_BYTES, tid, 256);     // (the K loop's barriers order it before the epilogue)
    const char* lut = (EPI == EPI_BIAS_GELU) ? smem + 2 * STAGE_BYTES : nullptr;
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
                    acc[mi][ni] = seedmi_mfma_16x16x32(w[ni], a[mi], acc[mi][ni]);
        }
        __syncthreads();
    }

    gemm_epilogue<EPI, 4, true, NoHook, LNF>(p, acc, m0 + 64 * wm, n0 + 64 * wn + 16 * g, li, lut);
}


// ======================================================================================================
// Kernel "gemm64": 64x64x64 block tile for SMALL M - one image through the ViT is M = 257, the reference scripts' own call pattern
// (scripts/seed_tokenizer_inference.py:26-29); the 128x128 kernel then has 33..144 workgroups for 256 CUs and fc2 (K = 6144) walks
// 96 K-tiles in each of 33 of them behind a two-deep buffer: 90 us for 17 MB of weights.  The k-ordered fp32 chain of every output
// element is what keeps a batch of one bit-identical to the same image inside a batch of 256 (the DP sharding property), so K cannot
// be split; what can be raised is the number of tiles (4x) and the bytes each of them keeps in flight:
//  * 4 waves stacked along M (16 rows x 64 columns each: the shared epilogue's lane layout with MT = 1), 8 MFMAs per wave and K-tile;
//  * an LDS ring of NS stages of 16 KiB (A tile | W tile, the 128x128 kernel's swizzled 128-byte rows), filled NS - 1 stages ahead by
//    LDS-DMA with ONE counted wait and ONE barrier per K-tile: stage kt must have landed (the NS - 2 younger stages stay in flight), the
//    barrier publishes it and retires every read of stage kt - 1, whose slot takes the request for stage kt + NS - 1;
//  * position-free body (round 4's lesson): requests beyond the last K-tile re-fetch it, so the wait count never changes;
//  * NS = 8 (128 KiB, one workgroup per CU) when the tiles do not fill the CUs once - Little's law at ~1.5 us of latency wants ~128 KiB
//    in flight per CU for ~85 GB/s - and NS = 4 with two workgroups per CU otherwise.
// Same MFMA orientation, fragment layout and epilogues as the 128x128 kernel: bit-identical results (tests run every GEMM case on it).
template <int EPI, bool LNF, int NS, bool PROD>
__global__ __launch_bounds__(PROD ? 512 : 256, NS <= 4 ? 2 : 1) void gemm64_kernel(GemmParams p) {
    // PROD: four PRODUCER waves (4..7) issue every LDS-DMA request and wait for their landing; the four consumer waves (0..3) never touch
    // the vector-memory pipe inside the K loop - a request costs its issuing wave 60-185 cycles, which was most of a consumer's K-tile
    // (8 MFMAs + 10 fragment reads) in the four-wave form.  Same ring, same single barrier per K-tile.
    constexpr int T64 = 64 * BK * 2;                  // 8 KiB per operand tile
    constexpr int ST64 = 2 * T64;                     // stage: A tile | W tile
    constexpr int NT = PROD ? 512 : 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = PROD && wave >= 4;
    const int sw = PROD ? (wave & 3) : wave;          // staging role: rows [16 sw, 16 sw + 16) of both tiles
    const int li = lane & 15, g = lane >> 4;
    // m-tiles fastest, XCD-contiguous: workgroup b runs on XCD b % 8 (observed placement; a speed choice only), so XCD x takes the x-th
    // eighth of the (n-tile, m-tile) order - the (few) m-tiles that share a W panel then run next to each other ON ONE XCD and the panel
    // crosses the fabric once instead of once per m-tile (at M = 257 a plain b -> tile map put the five m-tiles of a panel on five XCDs:
    // 5 x |W| through the fabric per GEMM of a one-image pass, whose floor is 1 x |W|)
    int tm, tn;
    {
        const int nt = p.tiles_m * p.tiles_n, bid = (int)blockIdx.x;
        const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
        const int cs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int t = p.xcd_map ? cs + idx : bid;
        tm = t % p.tiles_m;
        tn = t / p.tiles_m;
    }
    const int m0 = tm * 64, n0 = tn * 64;
    uint32_t offA[2], offW[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 16 * sw + 8 * j + (lane >> 3);
        const int cs = lane & 7;
        offA[j] = 2u * ((uint32_t)min(m0 + row, p.M - 1) * (uint32_t)p.lda + 8u * (uint32_t)(cs ^ swzA(row)));
        offW[j] = 2u * ((uint32_t)min(n0 + row, p.N - 1) * (uint32_t)p.ldw + 8u * (uint32_t)(cs ^ swzW(row)));
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, -1, 0x00020000);
    if (EPI == EPI_BIAS_GELU && GELU_BY_TABLE) load_gelu_lut(smem + NS * ST64, tid, NT);     // (the K loop's barriers order it before the epilogue)
    const char* lut = (EPI == EPI_BIAS_GELU) ? smem + NS * ST64 : nullptr;
    const int nk = p.K / BK;
    auto stage = [&](int kt) {                        // K-tile min(kt, nk - 1) into ring slot kt % NS
        char* base = smem + (kt % NS) * ST64 + sw * 2048;
        const uint32_t koff = 2u * (uint32_t)(min(kt, nk - 1) * BK);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16_buf(rsA, offA[j], koff, base + j * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16_buf(rsW, offW[j], koff, base + T64 + j * 1024);
    };
    auto wait_stage = [&]() {                         // the oldest stage in flight has landed: the NS - 2 younger ones (4 requests each) stay
        if (NS == 8) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (NS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (producer) {
#pragma unroll
        for (int kt = 0; kt < NS - 1; ++kt) stage(kt);
        for (int kt = 0; kt < nk; ++kt) {
            wait_stage();
            __builtin_amdgcn_s_barrier();             // stage kt published; the consumers have retired their reads of stage kt - 1
            stage(kt + NS - 1);                       // into the slot of stage kt - 1
        }
        return;                                       // (the epilogue has no barrier; requests still in flight target ring slots only)
    }
    // fragment read addresses inside a stage (k-step 0; k-step 1 = ^64)
    const int ra = 16 * wave + li;
    const int rdA = ra * 128 + ((g ^ swzA(ra)) << 4);
    int rdW[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rw = 16 * (li >> 2) + 4 * t + (li & 3);
        rdW[t] = T64 + rw * 128 + ((g ^ swzW(rw)) << 4);
    }
    f32x4 acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!PROD) {
#pragma unroll
        for (int kt = 0; kt < NS - 1; ++kt) stage(kt);
    }
    for (int kt = 0; kt < nk; ++kt) {
        if (!PROD) wait_stage();
        __builtin_amdgcn_s_barrier();
        if (!PROD) stage(kt + NS - 1);
        const char* sb = smem + (kt % NS) * ST64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a, w[4];
            a = *(const bf16x8*)(sb + (rdA ^ (ks << 6)));
#pragma unroll
            for (int t = 0; t < 4; ++t) w[t] = *(const bf16x8*)(sb + (rdW[t] ^ (ks << 6)));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[0][ni] = seedmi_mfma_16x16x32(w[ni], a, acc[0][ni]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the fragments are in registers before the next barrier lets the slot be restaged)
    }
    gemm_epilogue<EPI, 1, true, NoHook, LNF>(p, acc, m0 + 16 * wave, n0 + 16 * g, li, lut);
}

// ======================================================================================================
// Kernel "gemms": the small-M kernel in four tile shapes (round 6).  One image is M = 257: 64x64 tiles put 330 (QKV) / 480 (fc1) tiles
// on 256 CUs - the CUs that get two decide the launch - and 110 (proj, fc2) leave 146 CUs idle.  So: one workgroup per CU and, per GEMM, the
// shape whose busiest CU streams the fewest operand bytes - 64x64 (CM = 4, CN = 1), 128x64 (8, 1: QKV 198 tiles), 64x128 (4, 2: fc1 240
// tiles) or 32x64 (2, 1: proj / fc2 198 tiles; the Q-Former's M = 32 GEMMs stop staging 32 rows of padding).  Dedicated producer waves
// (8, or 4 for 32x64) issue every LDS-DMA request; a ring slot holds two K-tiles.
// What was learnt building it (LABNOTES round 6): a launch costs 5.6 us + ~150-190 ns per K-tile at one image, and the per-K-tile time did
// not move with the number of producers (4 / 8), the number of barriers (one or two K-tiles per slot), double-buffered fragment reads or
// L2 touch prefetch (slower) - memory latency is not it (average L2 round trip 420 cycles, profiles/r06_pmc_stalls_b1_*.json); what moved
// the pass was tile COUNT and bytes per CU.
// Consumer wave = 16 rows x 64 columns exactly as in gemm64 (same fragment layout, same k-ordered chain, same epilogue): bit-identical.
template <int EPI, bool LNF, int CM, int CN, int PW>
__global__ __launch_bounds__(64 * (CM * CN + PW), 1) void gemms_kernel(GemmParams p) {
    constexpr int TM = 16 * CM, TN = 64 * CN, NC = CM * CN;
    constexpr int A_BYTES = TM * BK * 2, W_BYTES = TN * BK * 2, ST = A_BYTES + W_BYTES;     // one ring stage: A tile | W tile, 128-byte rows
    constexpr int PA = TM / 8, PN = TN / 8, PP = (PA + PN) / PW;                            // 1 KiB pieces (8 rows) per stage / per producer
    static_assert((PA + PN) % PW == 0 && PA % PW == 0, "pieces deal evenly over the producers");
    // A ring slot holds KPS = 2 consecutive K-tiles and there is ONE barrier per slot (half the barrier round trips per K-tile).  Measured
    // neutral against one K-tile per slot, as were eight producers against four and double-buffered fragments: ~190 ns per K-tile at one
    // image whatever is varied inside the workgroup (LABNOTES round 6) - kept because it costs nothing.
    constexpr int KPS = 2, SLOT = KPS * ST;
    constexpr int NS = SLOT <= 24576 ? 5 : SLOT <= 32768 ? 4 : 3;                             // 120 / 128 / 144 KiB of ring
    constexpr int NT = 64 * (NC + PW);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= NC;
    const int li = lane & 15, g = lane >> 4;
    int tm, tn;
    {   // m-tiles fastest, XCD-contiguous (see gemm64_kernel)
        const int nt = p.tiles_m * p.tiles_n, bid = (int)blockIdx.x;
        const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tm = t % p.tiles_m;
        tn = t / p.tiles_m;
    }
    const int m0 = tm * TM, n0 = tn * TN;
    if (EPI == EPI_BIAS_GELU && GELU_BY_TABLE) load_gelu_lut(smem + NS * SLOT, tid, NT);    // (the K loop's barriers order it before the epilogue)
    const char* lut = (EPI == EPI_BIAS_GELU) ? smem + NS * SLOT : nullptr;
    const int nk = p.K / BK, nslots = (nk + KPS - 1) / KPS;
    if (producer) {
        const int pw = wave - NC;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, -1, 0x00020000);
        uint32_t off[PP];
#pragma unroll
        for (int j = 0; j < PP; ++j) {                 // piece pw + 8 j of [A pieces | W pieces]: rows 8 i .. 8 i + 7 of that operand's tile
            const bool isA = PW * j < PA;
            const int i = isA ? pw + PW * j : pw + PW * j - PA;
            const int row = 8 * i + (lane >> 3), cs = lane & 7;
            off[j] = isA ? 2u * ((uint32_t)min(m0 + row, p.M - 1) * (uint32_t)p.lda + 8u * (uint32_t)(cs ^ swzA(row)))
                         : 2u * ((uint32_t)min(n0 + row, p.N - 1) * (uint32_t)p.ldw + 8u * (uint32_t)(cs ^ swzW(row)));
        }
        auto stage = [&](int sl) {                     // K-tiles min(KPS sl + u, nk - 1) into ring slot sl % NS (position-free: the tail re-fetches the last one)
            char* base = smem + (sl % NS) * SLOT;
#pragma unroll
            for (int u = 0; u < KPS; ++u) {
                const uint32_t koff = 2u * (uint32_t)(min(KPS * sl + u, nk - 1) * BK);
#pragma unroll
                for (int j = 0; j < PP; ++j) {
                    const bool isA = PW * j < PA;
                    glds16_buf(isA ? rsA : rsW, off[j], koff, base + u * ST + (isA ? (pw + PW * j) * 1024 : A_BYTES + (pw + PW * j - PA) * 1024));
                }
            }
        };
#pragma unroll
        for (int sl = 0; sl < NS - 1; ++sl) stage(sl);
        for (int sl = 0; sl < nslots; ++sl) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KPS * PP * (NS - 2)) : "memory");     // the oldest slot in flight has landed
            __builtin_amdgcn_s_barrier();             // slot sl published; the consumers have retired their reads of slot sl - 1
            stage(sl + NS - 1);                       // into the ring position of slot sl - 1
        }
        return;                                       // (the epilogue has no barrier; requests still in flight target ring slots only)
    }
    const int wm = wave % CM, wn = wave / CM;
    // fragment read addresses inside a stage (k-step 0; k-step 1 = ^64)
    const int ra = 16 * wm + li;
    const int rdA = ra * 128 + ((g ^ swzA(ra)) << 4);
    int rdW[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rw = 64 * wn + 16 * (li >> 2) + 4 * t + (li & 3);
        rdW[t] = A_BYTES + rw * 128 + ((g ^ swzW(rw)) << 4);
    }
    f32x4 acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < nslots; ++sl) {
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int u = 0; u < KPS; ++u) {
            if (u > 0 && KPS * sl + u >= nk) break;                // odd K-tile count: the last slot's second half is a re-fetch, not a K-tile (uniform)
            const char* sb = smem + (sl % NS) * SLOT + u * ST;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a, w[4];
                a = *(const bf16x8*)(sb + (rdA ^ (ks << 6)));
#pragma unroll
                for (int t = 0; t < 4; ++t) w[t] = *(const bf16x8*)(sb + (rdW[t] ^ (ks << 6)));
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[0][ni] = seedmi_mfma_16x16x32(w[ni], a, acc[0][ni]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the fragments are in registers before the next barrier lets the slot be restaged)
    }
    gemm_epilogue<EPI, 1, true, NoHook, LNF>(p, acc, m0 + 16 * wm, n0 + 64 * wn + 16 * g, li, lut);
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
constexpr int B2 = 256;
constexpr int SK_FLAGS_WORDS = 1024;               // stream-K flag area: one word per workgroup, the last one a sticky error word
constexpr int MAX_SEGS = 128;                     // segment list of one workgroup in LDS (1.5 KiB; the launcher keeps tiles / workgroup below it)
// LayerNorm fold, consumer side: LDS behind the scratch rows = two sets of (column sums 1 KiB | folded bias 1 KiB) for this tile / the next
// one, then FIN = (mean, rstd) of the tile's 256 rows (2 KiB), then X (12 KiB) = the second FIN buffer when the statistics arrive finished
// (they are requested while slower waves may still read the current tile's), or the tile's up to six partial planes (2 KiB each) when the
// tile finalizes them itself.  Producer side: eight 1 KiB wave-private rows of (sum, sum of squares) pairs.
constexpr int FOLD_SET_BYTES = 2048, FOLD_FIN_OFF = 2 * FOLD_SET_BYTES, FOLD_X_OFF = FOLD_FIN_OFF + 2048, FOLD_MAX_PLANES = 6;
constexpr int FOLD_LDS_BYTES = FOLD_X_OFF + FOLD_MAX_PLANES * 2048;     // 18 KiB
constexpr int STAT_PARK_BYTES = 8 * 1024;
constexpr int SEG_BYTES = MAX_SEGS * 3 * 4;
constexpr int SCRATCH_BYTES = 8 * 256;            // one 256-byte LDS-DMA landing row per wave (residual prefetch touches)
constexpr int HALF_BYTES = 128 * BK * 2;          // 16 KiB
constexpr int KT_BYTES = 4 * HALF_BYTES;          // 64 KiB per K-tile
// The 256x256 kernel's operand ring (two K-tiles): [A slot 0 | A slot 1 | W slot 0 | W slot 1], 32 KiB each.  A slot is 32 KiB from its
// twin, so with the slot known at compile time (the K loop is unrolled by two and segments start on even K-tiles) a fragment read is
// ds_read_b128 v_base offset:(slot * 32768 + piece) - the 16-bit offset field holds it, and the per-K-tile v_add_u32 of a slot base to
// each of the four lane bases (A / W x k-step 0 / 1: the last VALU instructions of the K loop's LOAD sections) disappears.
constexpr int RING_SLOT = 2 * HALF_BYTES;         // slot stride inside the A and inside the W region
constexpr int RING_W = 4 * HALF_BYTES;            // the W region starts behind both A slots

#define SEEDMI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)


// Segment list (tile, first K-iteration, end K-iteration) of this workgroup, written to LDS (see gemm256_kernel): the data-parallel
// tiles of its XCD chunk, then - with a stream-K workspace - its share of the chunk's last 2..3 rounds, walked backwards.
// (PACKED: the tile entry is (m-tile << 12) | n-tile instead of the tile's index in the grouped order - no division between two tiles)
template <bool PACKED = false>
SEEDMI_DEVINL int build_segments(const GemmParams& p, int nk, int* segs, int tid) {
    auto coords = [&](int t) {
        if (!PACKED) return t;
        const int gsize = p.group_m * p.tiles_n;
        const int gid = t / gsize;
        const int first_m = gid * p.group_m;
        const int gm = min(p.tiles_m - first_m, p.group_m);
        const int in_g = t - gid * gsize;
        return ((first_m + in_g % gm) << 12) | (in_g / gm);
    };
    const int nt = p.tiles_m * p.tiles_n;
    int n_seg = 0;
    const int bid = blockIdx.x, q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    const int cs = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q);
    const int n_x = q + (xcd < r ? 1 : 0);
    const int G = ((int)gridDim.x + 7 - xcd) >> 3;
    int n_dp = (n_x - idx + G - 1) / G;             // data-parallel tiles cs + idx + j G < cs + n_x
    if (n_dp < 0) n_dp = 0;
    int sk_it = 0, sk_hi = 0, sk_tile0 = 0;         // stream-K range in K iterations over tiles sk_tile0 + i / nk
    if (p.sk_slabs && n_x >= G) {
        n_dp = max(n_x / G - 2, 0);                 // whole data-parallel rounds
        sk_tile0 = cs + n_dp * G;
        const int I = (n_x - n_dp * G) * nk;
        // cuts fall on EVEN K-tiles of their tile (the K loop's ring slots are compile-time constants of an unrolled pair that starts even)
        auto cut = [&](int i) { const int c = (int)(((long long)i * I) / G); return c - ((c % nk) & 1); };
        sk_it = cut(idx);
        sk_hi = idx + 1 == G ? I : cut(idx + 1);
    }
    if (n_dp > MAX_SEGS - 5) n_dp = MAX_SEGS - 5;   // (the launcher keeps nt / grid below this)
    for (int j = tid; j < n_dp; j += 512) { segs[3 * j] = coords(cs + idx + j * G); segs[3 * j + 1] = 0; segs[3 * j + 2] = nk; }
    n_seg = n_dp;
    while (sk_it < sk_hi) {                         // <= 4 segments, from the END of the range (uniform)
        const int tl = (sk_hi - 1) / nk;
        const int ke = sk_hi - tl * nk;
        const int kb = max(0, ke - (sk_hi - sk_it));
        if (tid == 0) { segs[3 * n_seg] = coords(sk_tile0 + tl); segs[3 * n_seg + 1] = kb; segs[3 * n_seg + 2] = ke; }
        sk_hi -= ke - kb;
        ++n_seg;
    }
    return n_seg;
}

// SCHED (seedmi_set_option("gemm_sched", ...)): variants of the schedule that compute bit-identical results.
//   bit 0: the next tile's first K-tile is waited for INSIDE this tile's epilogue, after its loads and before its first store, and the tile's
//          opening wait is dropped: vmcnt retires in order, so the opening wait (issued after the epilogue) also waited for 12 of the
//          epilogue's 16 stores to reach memory.  LayerNorm-fold consumers take gemm_epilogue_fold8 (first four rows held back).
//   bit 1: W(nh0) of K-tile kt+1 is read in P4 of K-tile kt (into the registers W(nh1) of kt just vacated; the two fragment sets swap
//          roles every K-tile, so the loop is unrolled by two) instead of in P1 of kt+1: P1's LOAD section - 12 fragment reads + 4 LDS-DMA
//          requests against 16 MFMAs of the partner wave - becomes 8 + 4, P4's 0 + 4 becomes 4 + 4.  Needs W(kt+1) retired one phase
//          earlier: a counted vmcnt(4) in P3 (the four A(kt+1) requests of P1 stay in flight).
//   bit 2: the A(kt+1) requests are split between P1 (half-tile 0) and P2 (half-tile 1) instead of all four in P1.
//   bit 3: the W(kt+2) requests are split between P3 and P4.  A wave's two pieces of a W half-tile are 8 rows each, and the MFMA row
//          permutation makes piece 0 all nh0 rows (last read in P1 of kt, or P4 of kt-1 with bit 1) and piece 1 all nh1 rows (last read
//          in P2 of kt): piece 0 may be restaged from P3 on, piece 1 from P4.  With bit 2 every phase then issues two requests.  P4's wait
//          becomes a counted vmcnt(2): the two requests of P3 stay in flight.
//   bit 4: residual epilogue: the second half's residual rows are requested after TWO rows of the first half are finished (their
//          registers are free by then) instead of after four, so that their HBM round trip runs under the other two rows' arithmetic.
//   bit 5: the second-dispatched wave group (waves 4..7) runs at s_setprio 1 throughout (the CDNA guide's static form of T5).
//   bit 7: ragged last n-tile re-divided.  N = 1408 (proj, fc2) is 5.5 n-tiles, N = 4224 (qkv) 16.5: in the last one only the column groups
//          wn = 0, 1 have columns and the waves wn = 2, 3 stand by (8.3 % / 3 % of those GEMMs' MFMA time spent on a half-empty tile).
//          Where the tile has <= 128 valid columns every wave takes 64 rows x 64 columns instead: wave (wm, wn) computes rows
//          128 wm + 64 (wn >> 1) .. + 63 of columns 64 (wn & 1) .. + 63, i.e. phases P1 / P2 on its own A rows and the W fragments of column
//          group wn & 1, nothing in P3 / P4 (same requests, waits and barriers).  The K loop of such a tile is its own instantiation of the
//          K-tile body (no branch in the ordinary one); the epilogue is the ordinary one, told to keep rows below mrow0 + 64 only - no
//          second epilogue instantiation (what made round 2's three attempts spill).  Each element's K chain is unchanged: bit-identical.
//   bit 6: TWO phases of 32 MFMAs per K-tile instead of four of 16 (not combined with bits 1-3).  The phase stamps of the four-phase loop
//          (tools/gemm_phase_times.py, profiles/r03_call2_*.log) show every phase costing the partner's MFMA section (300-330 cycles for 16
//          MFMAs) plus ~90 cycles of hand-over (barrier release + the fragment reads' tail): 8 x ~400 = 3200 cycles per K-tile against 2048
//          of MFMA issue.  Phase a = rows mh0 x all 64 columns, phase b = rows mh1 x the same W fragments: half as many hand-overs, the same 64
//          fragment registers.  The fragment reads are waited for BEFORE the phase's first barrier (the partner's 32-MFMA section is long enough
//          to cover them), which (i) lets the MFMA section start the moment the barrier opens and (ii) allows a slot to be restaged one phase
//          after its last read (CDNA guide, WAR rule): every LDS-DMA request gets two phases (~1300 cycles each) of flight,
//              a(kt):  vmcnt(6) | reads A(mh0), W | request A-mh1(kt+1) (2)           b(kt):  vmcnt(2) | reads A(mh1) | request W(kt+2) (4), A-mh0(kt+2) (2)
//          with A staged so that every wave owns one mh0 piece and one mh1 piece of each half-tile (rows 64 j + 8 w + ..).
template <int EPI, bool LNF = false, int SCHED = 0>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmParams p) {
    constexpr bool PRIO = SEEDMI_GEMM_PRIO;
    constexpr bool PREWAIT = (SCHED & 1) != 0, WPRE = (SCHED & 2) != 0, ASPLIT = (SCHED & 4) != 0, WSPLIT = (SCHED & 8) != 0;
    constexpr bool RES_EARLY = (SCHED & 16) != 0, STATIC_PRIO = (SCHED & 32) != 0, TWOPH = (SCHED & 64) != 0;
    // bit 8: the same re-division as bit 7, but ONE K-tile body for every tile with wave-uniform branches around the P3 reads / MFMAs and
    // the P4 MFMAs (two separate K loops made hipcc spill ~125 registers per lane into both of them)
    constexpr bool RAGBR = (SCHED & 256) != 0;
    constexpr bool RAGSPLIT = (SCHED & 128) != 0 || RAGBR;
    // bit 9 (A/B only): LDS-DMA requests in the FLAT form of rounds 1-2 (a 64-bit lane address per request: v_add_u32 + v_lshl_add_u64 each)
    // instead of the buffer form (lane byte offset in one VGPR per tile, K advance in the scalar offset: no VALU per request)
    constexpr bool FLATDMA = (SCHED & 512) != 0;
    // bit 10 (A/B only): the ring slot of a K-tile always taken from kt & 1 at run time (one v_add_u32 per lane base per K-tile), as before
    // the ring's slots were laid 32 KiB apart
    // (not in the BIAS_RESIDUAL variants: at 255 VGPRs the compile-time slots cost 32-56 B of scratch inside the loop)
    constexpr bool STATIC_SLOT = (SCHED & 1024) == 0 && EPI != EPI_BIAS_RESIDUAL;
    // bit 11 (two-phase K-tile): the fragment reads of a phase are waited for BEHIND the phase's barrier instead of in front of it
    constexpr bool LATEWAIT = (SCHED & 2048) != 0;
    // (round 4, measured and removed - profiles/r04_call6_store_tolerant_waits.log: "store-tolerant" waits.  vmcnt retires in order and
    // counts stores, so the two-phase loop's vmcnt(2) in phase b of a tile's FIRST K-tile also waits for the previous epilogue's 16 output
    // stores, a fraction of a microsecond after they were issued.  Variant: the prologue requests both K-tiles of the ring in full (16
    // requests, all ahead of the stores) and the two waits that concern only prologue requests allow a known lower bound of 16 younger
    // stores to stay in flight, so that the first request behind the stores is not waited for before phase b of the SECOND K-tile.
    // Bit-identical, no scratch - and 2.5-3.6 % SLOWER on all four ViT shapes, 125.5 vs 122.3 ms per tokenize pass: a request issued
    // behind the stores queues behind them in the memory pipeline whatever the counter says, and the early wait was also what kept the
    // store burst from overlapping the next tile's operand stream.)
    // bit 13 (two-phase K-tile, round 4): UNIFORM loop body - no K-tile looks at its position inside the segment.  Calls 9 / 10 of round 4
    // showed what wave-uniform branches cost this loop (the seam-request variant lost 5-7 % with its branches compiled in and never taken), and
    // peeling first / last K-tiles into bodies of their own makes hipcc spill 300+ B per lane.  Here every K-tile issues its eight requests
    // and waits with the steady-state counts; the requests that would reach beyond the segment re-fetch its LAST K-tile (L2-hot) into ring slots
    // nobody reads any more - 14 wasted requests per tile - and the next segment's prologue, issued by the same waves behind them, lands last.
    // Bit-identical on all four ViT shapes (race screen included), no scratch; measured on two boxes (profiles/r04_call11_/r04_call12_
    // uniform_ktile_body.log): QKV +0.2..0.3 %, proj +0.2..0.4 %, fc1 +0.2..0.4 %, fc2 +1.0..1.4 %, the tokenize pass +0.5 % both times
    // (122.09 vs 122.73 ms, 120.11 vs 120.73 ms).  Small, and the only schedule change of round 4 that did not lose: the default (8273 = 81 + 8192).
    constexpr bool UNIFORM = TWOPH && (SCHED & 8192) != 0;
    // bit 14 (uniform two-phase K-tile, round 5): SEAM - the K loop never stops requesting at a tile boundary.  The 14 requests per tile that the
    // uniform body issues beyond its segment (bit 13 lets them re-fetch the last K-tile into dead ring slots) are exactly the next tile's prologue
    // (its first K-tile in full, W and the mh0 rows of its second), so they are pointed at the NEXT tile instead: the tile of a request is a
    // wave-uniform buffer descriptor (base = the tile's first row, num_records = what is left of the matrix: rows beyond M / N are out of range
    // and never fetched), the lane offsets inside a tile never change, and the switch from this tile's descriptor to the next one's is two
    // scalar selects per K-tile - no branch in the body, no VALU, no per-tile address set-up, no prologue between the K loop and the epilogue.
    // Applies to whole-tile segments with an even number of K-tiles (ring slot parity carries over) and finished LayerNorm statistics; anything
    // else takes the prologue path of bit 13 inside the same kernel.  Same k-ordered accumulation chain per element: bit-identical.
    constexpr bool SEAM = UNIFORM && (SCHED & 16384) != 0;
    // bit 15 (on top of the seam): PEEL - a tile's output stores drain UNDER the next tile's first two K-tiles.  vmcnt retires in order and counts
    // stores, so the steady-state waits of a tile's first K-tile also wait for 10 of the 16 output stores the wave issued a moment ago (the K loop
    // of a QKV tile takes ~5 k cycles longer than 22 steady-state K-tiles; with the stores ablated the GEMM is 9 % faster).  With the seam every
    // request of the next tile's first TWO K-tiles is issued before the stores (the last two mh1 pieces right behind the K loop), so their waits can
    // let the 16 stores stay in flight - but a wait count is an immediate: the two K-tiles are peeled into bodies of their own (mode 1 / 2 of
    // ktile2: straight-line copies that differ in three immediates and one absent request - no branch in any body), entered only through a seamed
    // boundary.  Round 4's "store-tolerant waits" tried the counts with branches inside the one body and lost 3 %; the same round later measured
    // that such a branch alone costs the loop 2-7 %.  A wave that does not issue exactly its 16 stores (ragged tile edge) drains its queue instead.
    constexpr bool PEEL = SEAM && (SCHED & 32768) != 0 && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESIDUAL);
    // (round 4 call 32, timing-only probe, removed: W read as if packed request-major at load time - every W request 1 KiB contiguous instead of 8 rows x 128
    //  bytes at a stride of 2 K bytes: QKV / proj / fc1 / fc2 1245.8 / 967.9 / 1197.4 / 1241.0 against 1242.2 / 964.0 / 1194.1 / 1235.2 TF, +0.3..0.5 %: not worth a
    //  weight format - profiles/r04_call32_gemm_w_packed_probe.log)
    // (round 4 call 28, measured and removed: s_setprio 1 during a wave's LOAD sections - fragment reads + LDS-DMA requests - and 0 during its MFMA
    //  sections, the reverse of SEEDMI_GEMM_PRIO: QKV / proj / fc1 / fc2 1194.8 / 945.4 / 1156.7 / 1199.8 against 1194.3 / 944.7 / 1157.4 / 1202.5 TF, the
    //  pass 126.01 vs 125.90 ms - nothing either way: profiles/r04_call28_gemm_load_section_priority.log)
    bool tile_ragged = false;
    static_assert(!TWOPH || (SCHED & 14) == 0, "the two-phase schedule has its own request placement");
    // (round 4: the branch form of the split inside the two-phase K-tile - phase a on every wave's own 64 x 64 block, phase b empty in a
    // ragged tile - compiled without scratch (252 VGPRs) and measured -1.7 .. -2.3 % on every shape, fc1 - which has no ragged tile - included:
    // the wave-uniform branches around phase b's reads and MFMAs cost every tile more than the ragged tiles return; 122.7 vs 120.9 ms per
    // tokenize pass, profiles/r04_call3_ragged_split_two_phase.log.  Removed again.)
    static_assert(!(TWOPH && RAGSPLIT), "the ragged-tile split lives in the four-phase K-tile body");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, g = lane >> 4;

    // ---- persistent walk: the launch holds one workgroup per CU; workgroup b (XCD b % 8) takes every
    //      (workgroups-on-that-XCD)-th tile of its XCD's contiguous chunk of the grouped tile order, so the tiles
    //      resident on an XCD at any time are neighbours sharing A / W panels in its L2.
    //      With a stream-K workspace the last 2..3 rounds of the chunk are not walked tile by tile: their K-tile
    //      iterations (R tiles x nk, 2G <= R < 3G for G workgroups) are cut into G equal contiguous ranges, so every
    //      workgroup ends at the same time instead of 256 CUs waiting for the few that drew a tile of the partial round
    //      (257 x 6 tiles of the N = 1408 GEMMs are 6.02 rounds: 7 data-parallel).  A range is >= nk long, so a tile is
    //      shared by at most two workgroups.  Each workgroup walks its range BACKWARDS: the K head [0, k) of its last tile
    //      comes first and is published as an fp32 accumulator image; the K tail [k, nk) of its first tile comes last and
    //      STARTS from the image the previous workgroup published two tiles' time earlier (carry-in instead of zero), then
    //      runs the ordinary epilogue.  The accumulation chain of a shared tile is therefore the same k-ordered chain as
    //      an unshared one: results are bit-identical to the data-parallel walk.
    const int nk = p.K / BK;
    // The segment list (tile, first K-tile, end K-tile) of this workgroup is written to LDS once (behind the operand ring and the
    // activation table) and read back one entry per tile: the walk then costs two SGPRs of state instead of a dozen.
    int* const segs = (int*)(smem + 2 * KT_BYTES + GELU_LUT_BYTES);
    constexpr int SCRATCH_OFF = 2 * KT_BYTES + GELU_LUT_BYTES + SEG_BYTES;
    // LayerNorm fold: consumers keep two 4 KiB operand sets here (this tile's and the next one's: column sums [256] f32 | folded bias
    // [256] f32 | (mean, rstd) [256]); the producer (BIAS_RESIDUAL) parks 8 waves x 128 (sum, sum of squares) pairs
    constexpr int STAT_OFF = SCRATCH_OFF + SCRATCH_BYTES;
    constexpr bool FOLD_IN = LNF && EPI != EPI_BIAS_RESIDUAL;
    int fold_par = 0;                                               // operand set the NEXT prologue fills
    const int n_seg = build_segments<SEAM>(p, nk, segs, tid);
    if (n_seg == 0) return;                                            // uniform for the whole workgroup
    if (EPI == EPI_BIAS_GELU && GELU_BY_TABLE) load_gelu_lut(smem + 2 * KT_BYTES, tid, 512);   // activation table behind the operand ring
    __syncthreads();
    const char* lut = (EPI == EPI_BIAS_GELU) ? smem + 2 * KT_BYTES : nullptr;
    if (STATIC_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);      // (wave is a readfirstlane value: a scalar branch around one s_setprio)
    int i_seg = 0;
    int s_tile = 0, s_kb = 0, s_ke = 0;             // current segment: K-tiles [s_kb, s_ke) of tile s_tile
    auto next_seg = [&](int& tile, int& kb, int& ke) -> bool {
        if (i_seg >= n_seg) return false;
        tile = __builtin_amdgcn_readfirstlane(segs[3 * i_seg]);
        kb = __builtin_amdgcn_readfirstlane(segs[3 * i_seg + 1]);
        ke = __builtin_amdgcn_readfirstlane(segs[3 * i_seg + 2]);
        ++i_seg;
        return true;
    };
    next_seg(s_tile, s_kb, s_ke);

    int m0 = 0, n0 = 0;
    uint32_t offA[2][2], offW[2][2];                    // BYTE offsets of this lane's requests inside A / W (elements < 2^31, checked at the entry point)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, -1, 0x00020000);      // (raw: no stride, 4 GiB range)
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, -1, 0x00020000);
    // LDS-DMA sources of a tile: every wave copies rows [16w, 16w+16) of each half-tile (2 pieces of 8 rows)
    // SEAM: the tile of a request lives in a wave-uniform descriptor; cur = the tile being computed, nxt = where the requests beyond its
    // last K-tile go (the next tile; the same tile again - its last K-tile, as in bit 13 - where the boundary is not seamed)
    __amdgpu_buffer_rsrc_t curA = rsA, curW = rsW, nxtA = rsA, nxtW = rsW;
    int nkb = 0, nke = 1;                               // K-tiles [nkb, nke) of the tile behind nxtA / nxtW
    auto tile_desc = [&](int tm0, int tn0, __amdgpu_buffer_rsrc_t& dA, __amdgpu_buffer_rsrc_t& dW) {
        // (bytes left below the tile's first row: < 2^32, the entry point keeps every matrix below 2^31 elements)
        dA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (size_t)tm0 * (size_t)p.lda), 0,
                                               (int)((uint32_t)(p.M - tm0) * (uint32_t)p.lda * 2u), 0x00020000);
        dW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (size_t)tn0 * (size_t)p.ldw), 0,
                                               (int)((uint32_t)(p.N - tn0) * (uint32_t)p.ldw * 2u), 0x00020000);
    };
    if (SEAM) {                                         // lane offsets inside ANY tile: formed once
        const int ln = fresh_lane();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 128 * h + 16 * wave + 8 * j + (ln >> 3);
                const int rowA = 128 * h + 64 * j + 8 * wave + (ln >> 3);
                const int cs = ln & 7;
                offA[h][j] = 2u * ((uint32_t)rowA * (uint32_t)p.lda + 8u * (uint32_t)(cs ^ swzA(rowA)));
                offW[h][j] = 2u * ((uint32_t)row * (uint32_t)p.ldw + 8u * (uint32_t)(cs ^ swzW(row)));
            }
    }
    auto set_tile = [&](int t) {
        if (SEAM) {                                     // t = (m-tile << 12) | n-tile (build_segments<true>)
            m0 = (t >> 12) * B2;
            n0 = (t & 4095) * B2;
            tile_desc(m0, n0, curA, curW);
            return;
        }
        const int gsize = p.group_m * p.tiles_n;
        const int gid = t / gsize;
        const int first_m = gid * p.group_m;
        const int gm = min(p.tiles_m - first_m, p.group_m);
        const int in_g = t - gid * gsize;
        m0 = (first_m + in_g % gm) * B2;
        n0 = (in_g / gm) * B2;
        const int ln = SCHED != 0 ? fresh_lane() : lane;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 128 * h + 16 * wave + 8 * j + (ln >> 3);       // row inside the 256-row tile
                // TWOPH: a wave's A piece j is 8 rows of the mh0 (j = 0) / mh1 (j = 1) half of the half-tile: rows 64 j + 8 w + ..
                const int rowA = TWOPH ? 128 * h + 64 * j + 8 * wave + (ln >> 3) : row;
                const int cs = ln & 7;
                offA[h][j] = 2u * ((uint32_t)min(m0 + rowA, p.M - 1) * (uint32_t)p.lda + 8u * (uint32_t)(cs ^ swzA(rowA)));
                offW[h][j] = 2u * ((uint32_t)min(n0 + row, p.N - 1) * (uint32_t)p.ldw + 8u * (uint32_t)(cs ^ swzW(row)));
            }
    };
    // ---- fragment read bases (byte offsets inside a K-tile buffer); tile index adds an immediate
    //   A: row = 16*mi + li inside half wm   (swizzle depends on li only)
    //   W: row = 64*wn + 16*(li>>2) + 4*ni + (li&3) inside the 256-row tile, half wn>>1 (swizzle independent of ni)
    const int rowW0 = 64 * wn + 16 * (li >> 2) + (li & 3);
    const int rdA0 = wm * HALF_BYTES + li * 128 + ((g ^ swzA(li)) << 4);
    const int rdW0 = (wn >> 1) * HALF_BYTES + (rowW0 & 127) * 128 + ((g ^ swzW(rowW0)) << 4);      // (inside a W slot)
    // (RAGSPLIT: the bases are formed per tile, from a lane id taken there - tile_rdA / tile_rdW below - so that neither K loop's address
    // arithmetic is hoisted over the whole kernel next to the other's)
    int tile_rdA = rdA0, tile_rdW = rdW0;

    f32x4 acc[8][4];
    // one 16-byte-per-lane request of A / W: lane byte offset `off` (set_tile), K origin k0 (elements), LDS destination of the wave
    auto dmaA = [&](uint32_t off, int k0, char* dst) {
        if (FLATDMA) glds16(p.A + ((off >> 1) + (uint32_t)k0), dst);
        else glds16_buf(SEAM ? curA : rsA, off, 2u * (uint32_t)k0, dst);
    };
    auto dmaW = [&](uint32_t off, int k0, char* dst) {
        if (FLATDMA) glds16(p.W + ((off >> 1) + (uint32_t)k0), dst);
        else glds16_buf(SEAM ? curW : rsW, off, 2u * (uint32_t)k0, dst);
    };
    // SEAM: the requests of K-tile index q of the current walk: inside the segment -> this tile, beyond it -> K-tile nkb + (q - ke) of the tile
    // behind nxtA / nxtW.  Scalar selects only.
    auto seam_A_rows = [&](int q, int ke, int j) {      // this wave's mh0 (j = 0) / mh1 (j = 1) piece of both A half-tiles into ring slot q & 1
        const bool over = q >= ke;
        const uint32_t koff = (uint32_t)(over ? min(nkb + (q - ke), nke - 1) : q) * (uint32_t)(2 * BK);
        const __amdgpu_buffer_rsrc_t d = over ? nxtA : curA;
        char* base = smem + (q & 1) * RING_SLOT + wave * 1024 + j * 8192;
#pragma unroll
        for (int h = 0; h < 2; ++h) glds16_buf(d, offA[h][j], koff, base + h * HALF_BYTES);
    };
    auto seam_W = [&](int q, int ke) {
        const bool over = q >= ke;
        const uint32_t koff = (uint32_t)(over ? min(nkb + (q - ke), nke - 1) : q) * (uint32_t)(2 * BK);
        const __amdgpu_buffer_rsrc_t d = over ? nxtW : curW;
        char* base = smem + RING_W + (q & 1) * RING_SLOT + wave * 2048;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16_buf(d, offW[h][j], koff, base + h * HALF_BYTES + j * 1024);
    };
    auto stageA = [&](int kt) {            // both A half-tiles of K-tile kt
        char* base = smem + (kt & 1) * RING_SLOT + (TWOPH ? wave * 1024 : wave * 2048);
        const int k0 = kt * BK;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) dmaA(offA[h][j], k0, base + h * HALF_BYTES + (TWOPH ? j * 8192 : j * 1024));
    };
    // (kidx >= 0: ring slot of K-tile kt, data of K-tile kidx)
    auto stageA_rows = [&](int kt, int j, int kidx = -1) {  // TWOPH: this wave's mh0 (j = 0) or mh1 (j = 1) piece of both A half-tiles
        char* base = smem + (kt & 1) * RING_SLOT + wave * 1024 + j * 8192;
        const int k0 = (kidx < 0 ? kt : kidx) * BK;
#pragma unroll
        for (int h = 0; h < 2; ++h) dmaA(offA[h][j], k0, base + h * HALF_BYTES);
    };
    auto stageA_half = [&](int kt, int h) {  // one A half-tile of K-tile kt (ASPLIT)
        char* base = smem + (kt & 1) * RING_SLOT + wave * 2048;
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) dmaA(offA[h][j], k0, base + h * HALF_BYTES + j * 1024);
    };
    auto stageW = [&](int kt, int kidx = -1) {
        char* base = smem + RING_W + (kt & 1) * RING_SLOT + wave * 2048;
        const int k0 = (kidx < 0 ? kt : kidx) * BK;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) dmaW(offW[h][j], k0, base + h * HALF_BYTES + j * 1024);
    };

    auto stageW_piece = [&](int kt, int j) {   // piece j (8 rows) of both W half-tiles of K-tile kt (WSPLIT): j = 0 nh0 rows, j = 1 nh1 rows
        char* base = smem + RING_W + (kt & 1) * RING_SLOT + wave * 2048;
        const int k0 = kt * BK;
#pragma unroll
        for (int h = 0; h < 2; ++h) dmaW(offW[h][j], k0, base + h * HALF_BYTES + j * 1024);
    };
    // prologue loads of a segment: its first K-tile (A and W) and, already in flight behind it, W of the second
    auto issue_fold_pieces = [&]() {
        if (FOLD_IN) {
            // the tile's fold operands travel with its first K-tile, as its OLDEST LDS-DMA requests (1 KiB pieces dealt round-robin: wave w
            // takes pieces w, w + 8, ...; with finished statistics that is one piece for waves 0-3 and none for waves 4-7): complete, and
            // published by the K loop's barriers, long before the epilogue reads them
            // pieces: 0 column sums, 1 folded bias, then the statistics of the tile's 256 rows in 1 KiB halves (128 rows x 8 B): finished
            // (mean, rstd) pairs (2 pieces) or ln_planes partial planes (2 pieces each); wave w takes pieces w, w + 8
            const int ln = SCHED != 0 ? fresh_lane() : lane;
            const int npieces = 2 + 2 * (p.ln_planes > 0 ? p.ln_planes : 1);
            for (int pc = wave; pc < npieces; pc += 8) {
                const char* src;
                char* dst;
                if (pc < 2) {
                    src = (const char*)(pc == 0 ? p.ln_colsum : p.bias_f32) + 4u * (uint32_t)min(n0 + 4 * ln, p.N - 4);
                    dst = smem + STAT_OFF + fold_par * FOLD_SET_BYTES + pc * 1024;
                } else {
                    const int q = pc - 2, half = q & 1;
                    const uint32_t row = (uint32_t)min(m0 + 128 * half + 2 * ln, (p.M - 1) & ~1);
                    if (p.ln_planes > 0) {
                        src = (const char*)p.ln_stats + 8 * ((size_t)(q >> 1) * (size_t)p.ln_ld + row);
                        dst = smem + STAT_OFF + FOLD_X_OFF + q * 1024;
                    } else {
                        src = (const char*)p.ln_stats + 8u * row;
                        dst = smem + STAT_OFF + (fold_par ? FOLD_X_OFF : FOLD_FIN_OFF) + half * 1024;
                    }
                }
                glds16((const bf16_t*)src, dst);
            }
            fold_par ^= 1;
        }
    };
    auto issue_prologue = [&](int kb, int ke) {
        issue_fold_pieces();
        stageA(kb);
        stageW(kb);
        if (kb + 1 < ke) {
            stageW(kb + 1);
            if (TWOPH) stageA_rows(kb + 1, 0);
        }
    };
    set_tile(s_tile);
    issue_prologue(s_kb, s_ke);

    bf16x8 fa[8], fw0[4], fw1[4];                       // A(mh) k0/k1 x 4 tiles ; W(nh0), W(nh1): k0/k1 x 2 tiles
    bool pre_waited = false;                            // PREWAIT: the previous epilogue already waited for this segment's first K-tile
#ifdef SEEDMI_DEVTOOLS
    // phase clock stamps (tools/gemm_phase_times.py): waves 0 and 4 of workgroup 0 record s_memtime around every barrier of every other
    // K-tile of a window of their second tile, plus the tile-level events.  s_memtime answers through the scalar cache after ~1000 cycles:
    // a stamp is ISSUED into its own SGPR pair and nothing waits for it inside the stamped K-tile; the sixteen stamps of a K-tile are
    // written to LDS (no VM operation: the vmcnt pipeline is untouched) in the NEXT, unstamped K-tile, behind that K-tile's own first
    // lgkmcnt(0).  Dumped to global memory at the kernel's end.
    // (consumers: inside the X area, behind the second FIN buffer - the stamped runs pass finished statistics; producers: behind the parks)
    constexpr int DBG_OFF = STAT_OFF + (FOLD_IN ? FOLD_X_OFF + 2048 : (LNF ? STAT_PARK_BYTES : 0));
    const bool dbg_wave = p.dbg != nullptr && blockIdx.x == 0 && (wave & 3) == 0;
    char* const dbg_lds = smem + DBG_OFF + (wave >> 2) * 2048;
    int dbg_n = 0, dbg_tile = 0;
    unsigned long long dbg_t[17];                       // 0..15: the phases' four stamps each; 16: K-tile entered
    bool dbg_full = false;
#define GSTAMP_ON (dbg_wave && dbg_tile == 1)
#define GSTAMP(slot_) do { if (GSTAMP_ON) asm volatile("s_memtime %0" : "=s"(dbg_t[slot_])); } while (0)
    // behind an lgkmcnt(0) that is old enough: (first_, count_) of the slots -> LDS with their codes
#define GSTAMP_STORE(first_, count_, code0_)                                                   \
    do {                                                                                       \
        SEEDMI_SCHED_FENCE();                                                                  \
        _Pragma("unroll") for (int q_ = 0; q_ < (count_); ++q_) {                              \
            if (lane == 0 && dbg_n < 250)                                                      \
                *(unsigned long long*)(dbg_lds + 8 * dbg_n) = (dbg_t[(first_) + q_] & 0x00ffffffffffffffull) | ((unsigned long long)((code0_) + q_) << 56); \
            ++dbg_n;                                                                           \
        }                                                                                      \
        SEEDMI_SCHED_FENCE();                                                                  \
    } while (0)
#else
#define GSTAMP(slot_) do {} while (0)
#endif

    // ---- one K-tile, four phases.  fx holds / receives W(nh0), fy W(nh1).  WPRE: on entry fx already holds this K-tile's W(nh0) (read in the
    //      previous K-tile's P4 or ahead of the loop); in P4 fy - dead after P3 - receives W(nh0) of K-tile kt + 1, so the caller swaps roles.
    auto ktile = [&](auto rg_tag, auto slot_tag, const int kt, const int ke, bf16x8 (&fx)[4], bf16x8 (&fy)[4], const bool stamp) {
        constexpr bool RGT = decltype(rg_tag)::value;   // ragged-tile split: this wave's 64 rows x 64 columns, phases P1 / P2 only
        const bool RG = RGT || (RAGBR && tile_ragged);  // (RAGBR: decided per tile at run time)
        constexpr int SLOT = decltype(slot_tag)::value; // 0 / 1: the ring slot of K-tile kt, known at compile time; -1: kt & 1
        const int slot = SLOT >= 0 ? SLOT : (kt & 1);
        const int bA = RAGSPLIT ? tile_rdA : rdA0, bW = RAGSPLIT ? tile_rdW : rdW0;
        const char* pa0 = smem + bA + slot * RING_SLOT;                      // k-step 0
        const char* pa1 = smem + (bA ^ 64) + slot * RING_SLOT;               // k-step 1
        const char* pw0 = smem + bW + (RING_W + slot * RING_SLOT);
        const char* pw1 = smem + (bW ^ 64) + (RING_W + slot * RING_SLOT);
        (void)stamp;
        if (stamp) GSTAMP(16);

        // ================= P1: (mh0, nh0) =================
        if (!WPRE) {
#pragma unroll
            for (int t = 0; t < 2; ++t) { fx[t] = *(const bf16x8*)(pw0 + t * 512); fx[2 + t] = *(const bf16x8*)(pw1 + t * 512); }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) { fa[t] = *(const bf16x8*)(pa0 + t * 2048); fa[4 + t] = *(const bf16x8*)(pa1 + t * 2048); }
        if (kt + 1 < ke) {
            if (ASPLIT) stageA_half(kt + 1, 0);
            else stageA(kt + 1);
        }
        if (EPI == EPI_BIAS_RESIDUAL && p.prefetch_residual) {
            // The epilogue reads this tile's 128 KB of residual.  Left to the epilogue, all CUs ask HBM for their tiles in the same
            // few microseconds (32 MB per round) while C goes the other way: measured +81 us on the proj GEMM, +68 us on fc2 (B = 256)
            // over the bias-only epilogue.  Touch the wave's 128 x 128-byte residual block in four pieces spread over the last twelve
            // K-tiles instead (one dword per 64-byte line by LDS-DMA into a scratch row: no VGPR destination, retired by this
            // K-tile's own vmcnt(0) three phases later), so the lines wait in the Infinity Cache / L2 when the epilogue asks.
            // (measured neutral to negative: off by default; not combined with the counted waits of SCHED bit 1)
            const int rem = ke - 1 - kt;
            if (!WPRE && (rem == 12 || rem == 9 || rem == 6 || rem == 3)) {
                const int j = (12 - rem) / 3;
                const int col = n0 + 64 * wn + 32 * (j & 1);
                if (col < p.N) {
                    const int row = min(m0 + 128 * wm + 64 * (j >> 1) + lane, p.M - 1);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.R + (size_t)row * p.ldr + col),
                                                     (__attribute__((address_space(3))) void*)(smem + SCRATCH_OFF + wave * 256), 4, 0, 0);
                }
            }
        }
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
#ifdef SEEDMI_DEVTOOLS
        if (GSTAMP_ON && dbg_full && !stamp) {          // the previous (stamped) K-tile's sixteen stamps: long landed
            GSTAMP_STORE(0, 17, 0);
            dbg_full = false;
        }
#endif
        if (stamp) GSTAMP(1);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = seedmi_mfma_16x16x32(fx[2 * ks + ni], fa[4 * ks + mi], acc[mi][ni]);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(2);
        __builtin_amdgcn_s_barrier();
        if (stamp) GSTAMP(3);

        // ================= P2: (mh0, nh1) =================
#pragma unroll
        for (int t = 0; t < 2; ++t) { fy[t] = *(const bf16x8*)(pw0 + (2 + t) * 512); fy[2 + t] = *(const bf16x8*)(pw1 + (2 + t) * 512); }
        if (ASPLIT && kt + 1 < ke) stageA_half(kt + 1, 1);
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(4);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(5);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][2 + ni] = seedmi_mfma_16x16x32(fy[2 * ks + ni], fa[4 * ks + mi], acc[mi][2 + ni]);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(6);
        __builtin_amdgcn_s_barrier();
        if (stamp) GSTAMP(7);

        // ================= P3: (mh1, nh1) =================
        if (!RG) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { fa[t] = *(const bf16x8*)(pa0 + (4 + t) * 2048); fa[4 + t] = *(const bf16x8*)(pa1 + (4 + t) * 2048); }
        }
        // WPRE: W(kt+1) - requested in P4 of kt-1 (or by the prologue), i.e. older than the four A(kt+1) requests of this K-tile - must be
        // complete one phase before P4 reads it
        if (WPRE) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (WSPLIT && kt + 2 < ke) stageW_piece(kt + 2, 0);           // nh0 rows of this parity: last read in P1 (P4 of kt-1 with WPRE)
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(8);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(9);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        if (!RG) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[4 + mi][2 + ni] = seedmi_mfma_16x16x32(fy[2 * ks + ni], fa[4 * ks + mi], acc[4 + mi][2 + ni]);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(10);
        __builtin_amdgcn_s_barrier();
        if (stamp) GSTAMP(11);

        // ================= P4: (mh1, nh0) =================
        // K-tile kt+1 must be complete before anyone reads it in the next P1; its loads are 3-4 phases old.
        if (WSPLIT) {
            // everything but P3's two requests (W(kt+2), piece 0) must have landed: A(kt+1) for the next P1 (and W(kt+1) without WPRE)
            if (kt + 2 < ke) {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                stageW_piece(kt + 2, 1);                   // nh1 rows: last read in P2
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (kt + 2 < ke) stageW(kt + 2);               // W slots of this parity were last read in P2
        }
        if (WPRE && kt + 1 < ke) {                     // next K-tile's W(nh0): other parity, retired by P3's counted wait + two barriers
            const char* qw0 = smem + bW + (RING_W + (slot ^ 1) * RING_SLOT);
            const char* qw1 = smem + (bW ^ 64) + (RING_W + (slot ^ 1) * RING_SLOT);
#pragma unroll
            for (int t = 0; t < 2; ++t) { fy[t] = *(const bf16x8*)(qw0 + t * 512); fy[2 + t] = *(const bf16x8*)(qw1 + t * 512); }
        }
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(12);
        __builtin_amdgcn_s_barrier();
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(13);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
        if (!RG) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[4 + mi][ni] = seedmi_mfma_16x16x32(fx[2 * ks + ni], fa[4 * ks + mi], acc[4 + mi][ni]);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        SEEDMI_SCHED_FENCE();
        if (stamp) GSTAMP(14);
        __builtin_amdgcn_s_barrier();
#ifdef SEEDMI_DEVTOOLS
        if (stamp) { GSTAMP(15); if (GSTAMP_ON) dbg_full = true; }
#endif
    };

    // ---- statistics by tile.  Producer: the tile's eight waves have parked (sum, sum of squares) of their 128 rows x 64 columns in LDS;
    //      once every wave has passed a barrier behind its epilogue, wave w sums the (up to four live) spans of rows 32 w .. 32 w + 31 in span
    //      order and writes ONE pair per row into the tile's plane.  Done at the opening barrier of the NEXT tile (or behind the last one).
    constexpr bool STATS_OUT = LNF && EPI == EPI_BIAS_RESIDUAL;
    bool stats_pending = false;
    int sp_m0 = 0, sp_n0 = 0;
    auto combine_tile_stats = [&]() {
        const int ln = fresh_lane();
        if (ln < 32) {
            const int r = 32 * wave + ln;                                   // row of the tile
            const char* park = smem + STAT_OFF + (r >> 7) * 4096 + 8 * (r & 127);   // wave 4 (r >> 7) + wn parks at + 1024 wn
            const int nlive = min(4, (p.N - sp_n0 + 63) >> 6);
            float s1 = 0.f, s2 = 0.f;
            for (int wq = 0; wq < nlive; ++wq) { const float2 t = *(const float2*)(park + 1024 * wq); s1 += t.x; s2 += t.y; }
            const int m = sp_m0 + r;
            if (m < p.M) *(float2*)(p.stats_out + ((size_t)(sp_n0 >> 8) * p.stats_ld + m) * 2) = make_float2(s1, s2);
        }
        stats_pending = false;
    };
    //      Consumer: the tile's partial planes have landed with its first K-tile; wave w turns rows 32 w .. of them into (mean, rstd) for the
    //      fold pass at the end of the K loop (same arithmetic as seedmi_layernorm_stats_finalize: planes summed in order)
    auto finalize_tile_stats = [&]() {
        const int ln = fresh_lane();
        if (ln < 32) {
            const int r = 32 * wave + ln;
            const char* src = smem + STAT_OFF + FOLD_X_OFF + (r >> 7) * 1024 + 8 * (r & 127);
            float s1 = 0.f, s2 = 0.f;
            for (int pl = 0; pl < p.ln_planes; ++pl) { const float2 t = *(const float2*)(src + 2048 * pl); s1 += t.x; s2 += t.y; }
            *(float2*)(smem + STAT_OFF + FOLD_FIN_OFF + 8 * r) = seedmi_ln_finish(s1, s2, p.ln_inv_cols, p.ln_eps);
        }
    };

    // ---- TWOPH: one K-tile in two phases of 32 MFMAs (see the SCHED notes above the kernel)
    auto ktile2 = [&](auto slot_tag, auto mode_tag, const int kt, const int kb, const int ke) {
        // MODE (PEEL): 0 = steady state; 1 / 2 = first / second K-tile of a tile entered through a seamed boundary: every request of these two
        // K-tiles is older than the previous tile's 16 output stores, which may still be in flight
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int SLOT = decltype(slot_tag)::value; // 0 / 1: the ring slot of K-tile kt, known at compile time; -1: kt & 1
        const int slot = SLOT >= 0 ? SLOT : (kt & 1);
        const char* pa0 = smem + rdA0 + slot * RING_SLOT;                    // k-step 0
        const char* pa1 = smem + (rdA0 ^ 64) + slot * RING_SLOT;             // k-step 1
        const char* pw0 = smem + rdW0 + (RING_W + slot * RING_SLOT);
        const char* pw1 = smem + (rdW0 ^ 64) + (RING_W + slot * RING_SLOT);
        // ================= phase a: rows mh0 x (nh0, nh1) =================
        // A-mh1(kt), requested in phase a of kt-1, is read in phase b: everything but the six requests of phase b of kt-1 must have landed
        // (the segment's first K-tile came with the prologue and was waited for at the tile's opening)
        if (MODE == 1) {
            // (this K-tile was complete before the stores went out: the epilogue's hook waited for it)
        } else if (MODE == 2) {
            asm volatile("s_waitcnt vmcnt(22)" ::: "memory");         // A-mh1 of this K-tile landed; behind it: 16 stores + the six requests of phase b of K-tile 0
        } else if (UNIFORM) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // (first K-tile: the opening wait left at most six in flight already)
        } else if (kt > kb) {
            if (kt + 1 < ke) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) { fw0[t] = *(const bf16x8*)(pw0 + t * 512); fw0[2 + t] = *(const bf16x8*)(pw1 + t * 512); }
#pragma unroll
        for (int t = 0; t < 2; ++t) { fw1[t] = *(const bf16x8*)(pw0 + (2 + t) * 512); fw1[2 + t] = *(const bf16x8*)(pw1 + (2 + t) * 512); }
#pragma unroll
        for (int t = 0; t < 4; ++t) { fa[t] = *(const bf16x8*)(pa0 + t * 2048); fa[4 + t] = *(const bf16x8*)(pa1 + t * 2048); }
        // mh1 rows of the other parity: last read in phase b of kt-1, retired before its barrier
        if (MODE == 1) {}                               // (A-mh1 of the second K-tile was requested right behind the previous K loop)
        else if (SEAM) seam_A_rows(kt + 1, ke, 1);
        else if (UNIFORM) stageA_rows(kt + 1, 1, min(kt + 1, ke - 1));
        else if (kt + 1 < ke) stageA_rows(kt + 1, 1);
        if (!LATEWAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        if (LATEWAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the fragments land while the wave waits at the barrier)
        SEEDMI_SCHED_FENCE();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = seedmi_mfma_16x16x32(fw0[2 * ks + ni], fa[4 * ks + mi], acc[mi][ni]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][2 + ni] = seedmi_mfma_16x16x32(fw1[2 * ks + ni], fa[4 * ks + mi], acc[mi][2 + ni]);
            }
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        // ================= phase b: rows mh1 x the same W fragments =================
        // W(kt+1) and A-mh0(kt+1) (phase b of kt-1, or the prologue) are read in phase a of kt+1: only this K-tile's two requests stay in flight
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");     // W and A-mh0 of the second K-tile landed; behind them: its A-mh1 pieces (2) + 16 stores
        else if (UNIFORM || kt + 1 < ke) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t) { fa[t] = *(const bf16x8*)(pa0 + (4 + t) * 2048); fa[4 + t] = *(const bf16x8*)(pa1 + (4 + t) * 2048); }
        if (SEAM) {
            seam_W(kt + 2, ke);
            seam_A_rows(kt + 2, ke, 0);
        } else if (UNIFORM) {                           // this parity's W and mh0 rows were last read in phase a, retired before its barrier
            stageW(kt + 2, min(kt + 2, ke - 1));
            stageA_rows(kt + 2, 0, min(kt + 2, ke - 1));
        } else if (kt + 2 < ke) {
            stageW(kt + 2);
            stageA_rows(kt + 2, 0);
        }
        if (!LATEWAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
        if (LATEWAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the fragments land while the wave waits at the barrier)
        SEEDMI_SCHED_FENCE();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[4 + mi][ni] = seedmi_mfma_16x16x32(fw0[2 * ks + ni], fa[4 * ks + mi], acc[4 + mi][ni]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[4 + mi][2 + ni] = seedmi_mfma_16x16x32(fw1[2 * ks + ni], fa[4 * ks + mi], acc[4 + mi][2 + ni]);
            }
        SEEDMI_SCHED_FENCE();
        __builtin_amdgcn_s_barrier();
    };

    // SEAM applies to launches of whole tiles with an even number of K-tiles (the ring slot of a K-tile is its index's parity: an even tile
    // hands the next one the right slots) and finished statistics (partial planes are finalized at a tile's opening, from requests that would
    // then be the youngest of the seam); everything else keeps the prologue path
    const bool seam_ok = SEAM && (nk & 1) == 0 && nk >= 2 && p.sk_slabs == nullptr && p.ln_planes == 0;
    bool peel_now = false;                              // PEEL: this tile was entered through a seamed boundary (peeled first two K-tiles)
    for (;;) {
    const int em0 = m0, en0 = n0;                       // this tile's output origin (m0/n0 move on to the next tile early)
    const int fold_cur = fold_par ^ 1;                  // ... and its fold operand set
    const int kb = s_kb, ke = s_ke;                     // (s_* move on to the next segment before the epilogue)
    bool seam_more = false, seamed = false;
    if (SEAM) {
        // the next segment is looked up BEFORE the K loop: its tile is where this K loop's last requests go
        seam_more = next_seg(s_tile, s_kb, s_ke);
        seamed = seam_ok && seam_more && ke - kb >= 2 && s_ke - s_kb >= 2;
        if (seamed) {
            tile_desc((s_tile >> 12) * B2, (s_tile & 4095) * B2, nxtA, nxtW);
            nkb = s_kb; nke = s_ke;
        } else {                                        // (as bit 13: the last K-tile again, into ring slots nobody reads any more)
            nxtA = curA; nxtW = curW;
            nkb = ke - 1; nke = ke;
        }
    }
    if (kb > 0) {
        // ---- K tail of a shared tile (always this workgroup's last segment): continue the accumulation of the workgroup in
        //      front of it on this XCD, which published the head at the START of its stream-K range, two tiles' time ago.
        //      One relaxed poll loop, one agent-scope acquire, barrier, then the image lands straight in the accumulators
        //      (bounded spin: a lost partner leaves a wrong tile, not a hang).
        const int partner = blockIdx.x - 8;
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(p.sk_flags + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch &&
                   ++spins < (1 << 24))
                __builtin_amdgcn_s_sleep(4);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            // The flag is consumed: clear it, so that the flag area is all-zero again when the launch ends.  The epoch is a host
            // counter frozen into the kernel arguments; a hipGraph REPLAY of this launch reuses it, and without the reset the
            // previous replay's flag would already match (the wait skipped, a stale or half-written image read).  A partner that never
            // showed up leaves a wrong tile: recorded in the sticky error word (last word of the flag area).
            if (spins < (1 << 24)) __hip_atomic_store(p.sk_flags + partner, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(p.sk_flags + (SK_FLAGS_WORDS - 1), 1u + (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.sk_slabs + (size_t)partner * (B2 * B2), 0,
                                                                             B2 * B2 * 4, 0x00020000);
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const int voff = (wave * 32 * 64 + lane) * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (i * 4 + j) * 1024, 0));
        // The image must have landed HERE, inside this branch: left to the first use, hipcc merges the "loads pending" state of this (rare)
        // path into the common one and puts an unconditional s_waitcnt vmcnt(0) in front of EVERY tile's K loop - which also waits for the
        // previous tile's 16 epilogue stores and for the second W's requests (found in the ISA of rounds 1-2's kernel).
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
        pre_waited = false;                             // (keep the opening wait on this path)
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // first K-tile complete (the up-to-4 youngest VM ops are the second W's LDS-DMA or the previous tile's epilogue stores).
    // PREWAIT: the previous tile's epilogue has waited for it already, ahead of its stores, which may still be on their way.
    if (!(PREWAIT && pre_waited)) {
        if (ke - kb > 1) {
            if (TWOPH) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // (+ the two A-mh0 requests of the second K-tile)
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    pre_waited = false;
    if (STATS_OUT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the previous tile's parked statistics are in LDS)
    __builtin_amdgcn_s_barrier();
    if (STATS_OUT && stats_pending) combine_tile_stats();
    if (FOLD_IN && p.ln_planes > 0) finalize_tile_stats();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // stagger the second wave group by one barrier
    SEEDMI_SCHED_FENCE();
#ifdef SEEDMI_DEVTOOLS
    if (GSTAMP_ON) { GSTAMP(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); GSTAMP_STORE(0, 1, 100); }
#endif

    // A wave whose 64-column span lies beyond N (the half-empty last n-tile of N = 1408: the column groups wn = 2, 3) has nothing to
    // compute.  It walks the same barriers and issues its share of the LDS-DMA, but reads no fragments and issues no MFMA: the tile's
    // live waves keep the LDS bandwidth and the power budget to themselves (proj -2 %, fc2 -2.9 % at B = 256).
    const bool live = (en0 + 64 * wn) < p.N;
    // ragged split (SCHED bit 7): every wave computes in a tile with <= 128 valid columns (not with statistics by tile: their combine
    // assumes the ordinary wave -> row map)
    const bool ragged = RAGSPLIT && (p.N - en0) <= 128 && !p.stats_by_tile;
    using rg_no = std::false_type;
    using rg_yes = std::true_type;
    using slot_dyn = std::integral_constant<int, -1>;
    if (RAGSPLIT) {
        // this tile's fragment read bases: ordinary = A rows 16 mi + li of half-tile wm, W rows of column group wn; ragged split = A rows
        // 64 (wn >> 1) + 16 mi + li, W rows of column group wn & 1 (W half-tile 0)
        const int kl = fresh_lane(), kli = kl & 15, kg = kl >> 4;
        const int wcol = ragged ? (wn & 1) : wn;
        const int rw = 64 * wcol + 16 * (kli >> 2) + (kli & 3);
        tile_rdW = (wcol >> 1) * HALF_BYTES + (rw & 127) * 128 + ((kg ^ swzW(rw)) << 4);
        tile_rdA = wm * HALF_BYTES + (ragged ? (wn >> 1) * (64 * 128) : 0) + kli * 128 + ((kg ^ swzA(kli)) << 4);
    }
    tile_ragged = ragged;
    if (ragged && !RAGBR) {
        if (WPRE) {
            const char* sb = smem + RING_W + (kb & 1) * RING_SLOT;
#pragma unroll
            for (int t = 0; t < 2; ++t) { fw0[t] = *(const bf16x8*)(sb + tile_rdW + t * 512); fw0[2 + t] = *(const bf16x8*)(sb + (tile_rdW ^ 64) + t * 512); }
            int kt = kb;
            for (; kt + 1 < ke; kt += 2) {
                ktile(rg_yes(), slot_dyn(), kt, ke, fw0, fw1, false);
                ktile(rg_yes(), slot_dyn(), kt + 1, ke, fw1, fw0, false);
            }
            if (kt < ke) ktile(rg_yes(), slot_dyn(), kt, ke, fw0, fw1, false);
        } else {
            for (int kt = kb; kt < ke; ++kt) ktile(rg_yes(), slot_dyn(), kt, ke, fw0, fw1, false);
        }
    } else if ((live || ragged) && TWOPH) {
        // (a pair-unrolled form with compile-time ring slots - as in the four-phase loop - costs 64 B of scratch here: run-time slots)
        // (measured and removed, round 4 call 12: the same loop pair-unrolled with compile-time ring slots - segments start on even
        //  K-tiles - compiles with 12-52 B of scratch and runs proj -4.3 %, fc2 -4.6 %, QKV/fc1 equal, the pass -1.9 %:
        //  profiles/r04_call12_uniform_ktile_body.log.  The four v_add per K-tile of the run-time slots are cheaper than two bodies.)
        using mode_steady = std::integral_constant<int, 0>;
        int kt = kb;
        if (PEEL && peel_now) {                        // (wave-uniform, outside the bodies)
            ktile2(slot_dyn(), std::integral_constant<int, 1>(), kb, kb, ke);
            ktile2(slot_dyn(), std::integral_constant<int, 2>(), kb + 1, kb, ke);
            kt = kb + 2;
        }
        for (; kt < ke; ++kt) ktile2(slot_dyn(), mode_steady(), kt, kb, ke);
    } else if (!live && TWOPH) {
        // same requests, waits and barriers as ktile2, no fragment reads, no MFMA
        for (int kt = kb; kt < ke; ++kt) {
            if (UNIFORM) {
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else if (kt > kb) {
                if (kt + 1 < ke) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (PEEL && peel_now && kt == kb) {}       // (requested behind the previous K loop; this wave stored nothing and drained its queue)
            else if (SEAM) seam_A_rows(kt + 1, ke, 1);
            else if (UNIFORM) stageA_rows(kt + 1, 1, min(kt + 1, ke - 1));
            else if (kt + 1 < ke) stageA_rows(kt + 1, 1);
            SEEDMI_SCHED_FENCE();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
            if (UNIFORM || kt + 1 < ke) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (SEAM) {
                seam_W(kt + 2, ke);
                seam_A_rows(kt + 2, ke, 0);
            } else if (UNIFORM) {
                stageW(kt + 2, min(kt + 2, ke - 1));
                stageA_rows(kt + 2, 0, min(kt + 2, ke - 1));
            } else if (kt + 2 < ke) {
                stageW(kt + 2);
                stageA_rows(kt + 2, 0);
            }
            SEEDMI_SCHED_FENCE();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
    } else if (live || ragged) {
        if (WPRE) {
            // W(nh0) of the segment's first K-tile (every later one is read in the P4 in front of it); the two fragment sets swap roles per K-tile
            const char* sb = smem + RING_W + (kb & 1) * RING_SLOT;
            const int bW = RAGSPLIT ? tile_rdW : rdW0;
#pragma unroll
            for (int t = 0; t < 2; ++t) { fw0[t] = *(const bf16x8*)(sb + bW + t * 512); fw0[2 + t] = *(const bf16x8*)(sb + (bW ^ 64) + t * 512); }
            int kt = kb;
            // segments start on even K-tiles (build_segments): the ring slot of each K-tile of the unrolled pair is a compile-time constant and
            // travels in the fragment reads' offset field.  (ONE loop: a run-time-slot twin beside it made hipcc spill ~480 B per lane.)
            using slot_a = std::integral_constant<int, STATIC_SLOT ? 0 : -1>;
            using slot_b = std::integral_constant<int, STATIC_SLOT ? 1 : -1>;
            for (; kt + 1 < ke; kt += 2) {
                ktile(rg_no(), slot_a(), kt, ke, fw0, fw1, kt - kb >= 4 && kt - kb < 16 && ((kt - kb) & 1) == 0);
                ktile(rg_no(), slot_b(), kt + 1, ke, fw1, fw0, kt - kb >= 4 && kt - kb < 16 && ((kt - kb) & 1) == 0);
            }
            if (kt < ke) ktile(rg_no(), slot_a(), kt, ke, fw0, fw1, false);
        } else {
            for (int kt = kb; kt < ke; ++kt) ktile(rg_no(), slot_dyn(), kt, ke, fw0, fw1, kt - kb >= 4 && kt - kb < 16 && ((kt - kb) & 1) == 0);
        }
    } else {
        for (int kt = kb; kt < ke; ++kt) {
            if (kt + 1 < ke) stageA(kt + 1);
            SEEDMI_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_barrier();      // P1, P2, P3: two barriers each
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (kt + 2 < ke) stageW(kt + 2);
            SEEDMI_SCHED_FENCE();
            __builtin_amdgcn_s_barrier();                                  // P4
            __builtin_amdgcn_s_barrier();
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();          // matches the extra barrier the other group took up front
    SEEDMI_SCHED_FENCE();
#ifdef SEEDMI_DEVTOOLS
    if (GSTAMP_ON) {
        if (dbg_full) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); GSTAMP_STORE(0, 17, 0); dbg_full = false; }
        GSTAMP(0);
    }
#endif

    if (FOLD_IN && ke == nk) {
        // LayerNorm fold, applied while nothing but the accumulators is live (ahead of the next tile's address set-up: placed inside
        // the epilogue it cost 110-290 B of scratch per lane, reloaded behind the tile's stores)
        FoldLds fold;
        const int ln = SCHED != 0 ? fresh_lane() : lane;
        fold.cs = smem + STAT_OFF + fold_cur * FOLD_SET_BYTES + 4 * (64 * (ragged ? (wn & 1) : wn) + 16 * (ln >> 4));
        fold.st = smem + STAT_OFF + ((p.ln_planes > 0 || fold_cur == 0) ? FOLD_FIN_OFF : FOLD_X_OFF) +
                  8 * (128 * wm + (ragged ? 64 * (wn >> 1) : 0) + (ln & 15));
        fold_accumulators(acc, fold);
        SEEDMI_SCHED_FENCE();
    }
    // every LDS read of this segment is done: start the next segment's prologue loads now so that their latency (and the
    // epilogue's own loads and stores) overlap instead of opening the next tile with an empty pipeline
    const bool more = SEAM ? seam_more : next_seg(s_tile, s_kb, s_ke);
    constexpr bool late = SEEDMI_LATE_PROLOGUE != 0;
    auto start_next = [&]() {
        if (more) {
            if (SEAM && seamed) {
                // the next tile's first K-tile (and most of its second) was requested by this tile's last two K-tiles: only the tile
                // origin moves on and the fold operands (needed by the next tile's END) are requested
                // PEEL: the last two pieces of the next tile's SECOND K-tile (its mh1 rows; their ring rows were last read in phase b of this
                // tile's last K-tile, two barriers ago for either wave group) - everything its first two K-tiles read is now older than the stores
                if (PEEL) seam_A_rows(ke + 1, ke, 1);
                m0 = (s_tile >> 12) * B2;
                n0 = (s_tile & 4095) * B2;
                curA = nxtA; curW = nxtW;
                issue_fold_pieces();
            } else {
                set_tile(s_tile);
                issue_prologue(s_kb, s_ke);
            }
        }
    };
    // the epilogues call this once: after their own loads have landed, before their first store
    auto hook = [&]() {
        if (late) start_next();
        if (PREWAIT && more) {
            if (PEEL && seamed) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               // (the second K-tile's eight requests stay in flight)
            } else if (s_ke - s_kb > 1) {
                if (TWOPH) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // (the second K-tile's W and A-mh0 requests stay in flight)
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // (the second W's four requests stay in flight)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            pre_waited = true;
        }
    };
    if (!late || ke < nk) start_next();
#ifdef SEEDMI_DEVTOOLS
    if (GSTAMP_ON) { GSTAMP(1); }
#endif
    if (ke < nk) {
        // ---- K head of a shared tile (the first stream-K segment): publish the accumulator image ([wave][4-register group][lane]
        //      x 16 B: every store instruction writes 1 KiB contiguous) write-through, then the flag.  Protocol of the CDNA guide
        //      (G16 R1): sc1 stores -> every wave drains vmcnt -> workgroup barrier -> one relaxed agent-scope flag store.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.sk_slabs + (size_t)blockIdx.x * (B2 * B2), 0,
                                                                             B2 * B2 * 4, 0x00020000);
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const int voff = (wave * 32 * 64 + lane) * 16;  // one address register; the piece index travels in the scalar offset
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rs, voff, (i * 4 + j) * 1024, /*sc1*/ 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.sk_flags + blockIdx.x, p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const int eln = SCHED != 0 ? fresh_lane() : lane;
        const int eli = eln & 15;
        // (ragged split: the wave's accumulators 0..3 are rows 64 (wn >> 1) .. + 63 of column group wn & 1; row groups 4..7 hold nothing)
        const int enb = en0 + 64 * (ragged ? (wn & 1) : wn) + 16 * (eln >> 4);
        const int emrow0 = em0 + 128 * wm + (ragged ? 64 * (wn >> 1) : 0);
        const int erow_end = ragged ? min(p.M, emrow0 + 64) : -1;
        if (EPI == EPI_BIAS_RESIDUAL && p.skip_epilogue == 0 && (enb & ~63) + 64 <= p.N) {
            gemm_epilogue_residual8<LNF, decltype(hook), RES_EARLY>(p, acc, emrow0, enb, eli, hook, smem + STAT_OFF + wave * 1024, erow_end);
        } else if (PREWAIT && FOLD_IN && p.skip_epilogue == 0) {
            gemm_epilogue_fold8<EPI>(p, acc, emrow0, enb, eli, lut, hook, erow_end);
        } else if (p.skip_epilogue != 1) {
            gemm_epilogue<EPI, 8, true, decltype(hook), LNF>(p, acc, emrow0, enb, eli, lut, hook, erow_end);
        } else {
            hook();
            if (acc[0][0][0] == 123.456f) p.C[0] = 0;       // keep the accumulators alive
        }
        // (every wave of the workgroup, also those whose span lies beyond N and parked nothing: the combine is a workgroup job)
        if (STATS_OUT && p.stats_by_tile) { stats_pending = true; sp_m0 = em0; sp_n0 = en0; }
        if (PEEL && seamed) {
            // the peeled K-tiles' wait counts assume this wave's 16 output stores behind the seam's requests.  A wave at a ragged edge of the
            // matrix (rows beyond M or a span not wholly inside N: some or all of its stores are skipped) drains its queue instead - then the
            // counts hold trivially
            const bool full16 = (en0 + 64 * wn + 64 <= p.N) && (em0 + 128 * wm + 128 <= p.M) && p.skip_epilogue == 0;
            if (!full16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    peel_now = PEEL && seamed && more;
#ifdef SEEDMI_DEVTOOLS
    if (GSTAMP_ON) { GSTAMP(2); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); GSTAMP_STORE(0, 3, 101); }
    ++dbg_tile;
#endif
    if (!more) break;
    }
    if (STATS_OUT && stats_pending) {                   // the last tile's statistics
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        combine_tile_stats();
    }
#ifdef SEEDMI_DEVTOOLS
    if (dbg_wave) {                                     // dump: [2 stamped waves][256] x u64
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < 256; i += 64)
            p.dbg[(wave >> 2) * 256 + i] = i < dbg_n ? *(const unsigned long long*)(dbg_lds + 8 * i) : 0ull;
    }
#endif
#undef GSTAMP
#ifdef SEEDMI_DEVTOOLS
#undef GSTAMP_STORE
#undef GSTAMP_ON
#endif
}


#ifdef SEEDMI_DEVTOOLS
#include "gemm_devtools.inc"          // tools/devtools_csrc/ (lab build only: -DSEEDMI_DEVTOOLS adds that directory to the include path)
#endif

// per-device launch state (function attributes are per device; so is the CU count): seedmi_internal.h
constexpr int MAX_DEVICES = SEEDMI_MAX_DEVICES;
int current_device() { return seedmi_current_device(); }
int device_cus(int dev) { return seedmi_device_cus(dev); }

constexpr size_t SK_SLAB_BYTES = (size_t)B2 * B2 * 4;       // one fp32 accumulator image per workgroup
constexpr size_t SK_FLAGS_BYTES = 4 * SK_FLAGS_WORDS;        // one flag word per workgroup (< 1024 CUs) + the error word
std::atomic<unsigned> g_sk_epoch{0};

template <int EPI, bool LNF, int SCHED>
int launch_gemm256_sched(GemmParams p, hipStream_t stream, void* sk_ws, size_t sk_ws_bytes) {
#ifdef SEEDMI_DEVTOOLS
    constexpr bool fold_in = LNF && EPI != EPI_BIAS_RESIDUAL;
    constexpr int lds = 2 * KT_BYTES + GELU_LUT_BYTES + SEG_BYTES + SCRATCH_BYTES + (fold_in ? FOLD_LDS_BYTES : (LNF ? STAT_PARK_BYTES + 4096 : 4096));   // (+ phase stamps)
    p.dbg = g_gemm_dbg;
#else
    constexpr bool fold_in = LNF && EPI != EPI_BIAS_RESIDUAL;
    constexpr int lds = 2 * KT_BYTES + GELU_LUT_BYTES + SEG_BYTES + SCRATCH_BYTES + (fold_in ? FOLD_LDS_BYTES : (LNF ? STAT_PARK_BYTES : 0));
#endif
    static_assert(lds <= 160 * 1024, "LDS budget of the 256x256 kernel");
    static bool attr_set[MAX_DEVICES] = {};
    const int dev = current_device();
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<EPI, LNF, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set[dev] = true;
    }
    p.tiles_m = (p.M + B2 - 1) / B2;
    p.tiles_n = (p.N + B2 - 1) / B2;
    const int n_cu = device_cus(dev);
    const int nt = p.tiles_m * p.tiles_n;
    const int grid = (g_gemm_persist && nt > n_cu && nt / n_cu < MAX_SEGS - 4) ? n_cu : nt;
    // stream-K tail: only a persistent launch whose every XCD holds at least one full round of tiles, and only when the last
    // round is partial (otherwise the data-parallel walk is already balanced and needs no exchange)
    p.sk_slabs = nullptr; p.sk_flags = nullptr; p.sk_epoch = 0;
    if (sk_ws && g_gemm_streamk && grid == n_cu && nt >= grid && (nt % grid) != 0 && (grid % 8) == 0 && grid < SK_FLAGS_WORDS &&
        sk_ws_bytes >= SK_FLAGS_BYTES + (size_t)grid * SK_SLAB_BYTES) {
        p.sk_flags = (unsigned*)sk_ws;
        p.sk_slabs = (float*)((char*)sk_ws + SK_FLAGS_BYTES);
        unsigned e = ++g_sk_epoch;
        if (e == 0) e = ++g_sk_epoch;                      // 0 is what a cleared flag area holds
        p.sk_epoch = e;
    }
    hipLaunchKernelGGL((gemm256_kernel<EPI, LNF, SCHED>), dim3(grid), dim3(512), lds, stream, p);
    return seedmi_check_launch("gemm256");
}

// schedule variants (bit-identical results) exist for the three epilogues of the ViT GEMMs; everything else runs schedule 0
template <int EPI, bool LNF = false>
int launch_gemm256(const GemmParams& p, hipStream_t stream, void* sk_ws, size_t sk_ws_bytes) {
    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESIDUAL) {
        switch (g_gemm_sched.load()) {
#ifdef SEEDMI_SCHED_ONLY                               // (register-pressure experiments: one variant, short compile)
            case SEEDMI_SCHED_ONLY: return launch_gemm256_sched<EPI, LNF, SEEDMI_SCHED_ONLY>(p, stream, sk_ws, sk_ws_bytes);
#else
            case 57425: return launch_gemm256_sched<EPI, LNF, 57425>(p, stream, sk_ws, sk_ws_bytes);  // + peeled first two K-tiles (bit 15): the stores drain under them
            case 24657: return launch_gemm256_sched<EPI, LNF, 24657>(p, stream, sk_ws, sk_ws_bytes);  // 8273 + the seam (bit 14): the K loop's overshoot requests fetch the next tile
            case 8273: return launch_gemm256_sched<EPI, LNF, 8273>(p, stream, sk_ws, sk_ws_bytes);  // two-phase K-tile with the position-free body: the default
#ifdef SEEDMI_DEVTOOLS                                 // the schedule's history and measured steps (tools/gemm_sched_ab.py): devtools build only
            case 81: return launch_gemm256_sched<EPI, LNF, 81>(p, stream, sk_ws, sk_ws_bytes);      // two-phase K-tile, requests guarded by position (the default of round 3)
            case 31: return launch_gemm256_sched<EPI, LNF, 31>(p, stream, sk_ws, sk_ws_bytes);      // four-phase K-tile (the default before the buffer-form requests)
            case 7: return launch_gemm256_sched<EPI, LNF, 7>(p, stream, sk_ws, sk_ws_bytes);
            case 15: return launch_gemm256_sched<EPI, LNF, 15>(p, stream, sk_ws, sk_ws_bytes);
            case 63: return launch_gemm256_sched<EPI, LNF, 63>(p, stream, sk_ws, sk_ws_bytes);
            case 593: return launch_gemm256_sched<EPI, LNF, 593>(p, stream, sk_ws, sk_ws_bytes);     // two-phase K-tile with the flat requests of rounds 1-2
            case 113: return launch_gemm256_sched<EPI, LNF, 113>(p, stream, sk_ws, sk_ws_bytes);     // two-phase K-tile + static wave priority
            case 2129: return launch_gemm256_sched<EPI, LNF, 2129>(p, stream, sk_ws, sk_ws_bytes);   // two-phase K-tile, fragment waits behind the barriers
            case 65: return launch_gemm256_sched<EPI, LNF, 65>(p, stream, sk_ws, sk_ws_bytes);       // two-phase K-tile without the early residual rows
            case 287: return launch_gemm256_sched<EPI, LNF, 287>(p, stream, sk_ws, sk_ws_bytes);     // ragged n-tile re-divided: -4 % on EVERY tile
            case 543: return launch_gemm256_sched<EPI, LNF, 543>(p, stream, sk_ws, sk_ws_bytes);     // schedule 31 with the flat LDS-DMA requests of rounds 1-2
            case 512: return launch_gemm256_sched<EPI, LNF, 512>(p, stream, sk_ws, sk_ws_bytes);     // schedule 0 with them
            case 1055: return launch_gemm256_sched<EPI, LNF, 1055>(p, stream, sk_ws, sk_ws_bytes);   // schedule 31 with run-time ring slots
#endif
#endif
            default: break;
        }
    }
    // the other epilogues (plain, SwiGLU, tanh, ReLU: LLaMA prefill, the head MLPs) take the two-phase K-tile too - one variant each
    if constexpr (!LNF && (EPI == EPI_NONE || EPI == EPI_SWIGLU || EPI == EPI_BIAS_TANH || EPI == EPI_RELU)) {
#ifndef SEEDMI_SCHED_ONLY
        if (g_gemm_sched.load() == 24657 || g_gemm_sched.load() == 57425) return launch_gemm256_sched<EPI, LNF, 24641>(p, stream, sk_ws, sk_ws_bytes);
        if (g_gemm_sched.load() == 8273) return launch_gemm256_sched<EPI, LNF, 8257>(p, stream, sk_ws, sk_ws_bytes);
#ifdef SEEDMI_DEVTOOLS
        if (g_gemm_sched.load() == 81) return launch_gemm256_sched<EPI, LNF, 65>(p, stream, sk_ws, sk_ws_bytes);
#endif
#endif
    }
    return launch_gemm256_sched<EPI, LNF, 0>(p, stream, sk_ws, sk_ws_bytes);
}

#ifdef SEEDMI_DEVTOOLS
#include "gemm_devtools_k.inc"        // tools/devtools_csrc/
#endif

template <int EPI, bool LNF = false>
int launch_gemm128(const GemmParams& p, hipStream_t stream) {
    constexpr int lds = 2 * STAGE_BYTES + (EPI == EPI_BIAS_GELU ? GELU_LUT_BYTES : 0);
    static bool attr_set[MAX_DEVICES] = {};
    const int dev = current_device();
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm128_kernel<EPI, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set[dev] = true;
    }
    const int grid = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm128_kernel<EPI, LNF>), dim3(grid), dim3(256), lds, stream, p);
    return seedmi_check_launch("gemm128");
}


std::atomic<int> g_gemm_small{1};    // "gemm_small": the automatic selection may take the 64x64 kernel (1) or stays with 128x128 / 256x256 (0: A/B)
std::atomic<int> g_gemm64_xcd{1};    // "gemm64_xcd": XCD-contiguous tile order of the 64x64 kernel (1, default) or workgroup b = tile b (0: A/B); same bits
std::atomic<int> g_gemm64_prod{1};   // "gemm" = 65 selects the four-wave form of the 64x64 kernel (no producer waves) for A/B; 64 / automatic: producer waves

template <int EPI, bool LNF, int NS, bool PROD>
int launch_gemm64_ns(GemmParams p, hipStream_t stream) {
    constexpr int lds = NS * 2 * 64 * BK * 2 + (EPI == EPI_BIAS_GELU ? GELU_LUT_BYTES : 0);
    static_assert(lds <= 160 * 1024, "LDS budget of the 64x64 kernel");
    static bool attr_set[MAX_DEVICES] = {};
    const int dev = current_device();
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm64_kernel<EPI, LNF, NS, PROD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm64_kernel<EPI, LNF, NS, PROD>), dim3(p.tiles_m * p.tiles_n), dim3(PROD ? 512 : 256), lds, stream, p);
    return seedmi_check_launch("gemm64");
}
template <int EPI, bool LNF, int CM, int CN, int PW = 8>
int launch_gemms_shape(GemmParams p, hipStream_t stream) {
    constexpr int TM = 16 * CM, TN = 64 * CN, SLOT = 2 * (TM + TN) * BK * 2, NS = SLOT <= 24576 ? 5 : SLOT <= 32768 ? 4 : 3;
    constexpr int lds = NS * SLOT + (EPI == EPI_BIAS_GELU ? GELU_LUT_BYTES : 0);
    static_assert(lds <= 160 * 1024, "LDS budget of the shaped small-M kernel");
    p.tiles_m = (p.M + TM - 1) / TM;
    p.tiles_n = (p.N + TN - 1) / TN;
    static bool attr_set[MAX_DEVICES] = {};
    const int dev = current_device();
    if (!attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)gemms_kernel<EPI, LNF, CM, CN, PW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemms_kernel<EPI, LNF, CM, CN, PW>), dim3(p.tiles_m * p.tiles_n), dim3(64 * (CM * CN + PW)), lds, stream, p);
    return seedmi_check_launch("gemms");
}
// The shape whose busiest CU streams the fewest operand bytes (what bounds a small-M launch): ceil(tiles / CUs) x (TM + TN) rows of K.
// `force`: 66 / 67 / 68 / 70 = 64x64 / 128x64 / 64x128 / 32x64 ("gemm" option, tests and A/B); 69 = by cost
template <int EPI, bool LNF = false>
int launch_gemms(const GemmParams& p, hipStream_t stream, int force) {
    const int cus = device_cus(current_device());
    auto cost = [&](int TM, int TN) {
        const long long tiles = (long long)((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN);
        return ((tiles + cus - 1) / cus) * (TM + TN);
    };
    int pick = force;
    if (pick != 66 && pick != 67 && pick != 68 && pick != 70) {
        const long long c[4] = {cost(64, 64), cost(128, 64), cost(64, 128), cost(32, 64)};
        const int id[4] = {66, 67, 68, 70};
        int best = 0;
        for (int i = 1; i < 4; ++i)
            if (c[i] < c[best]) best = i;
        pick = id[best];
    }
    if (pick == 67) return launch_gemms_shape<EPI, LNF, 8, 1>(p, stream);
    if (pick == 68) return launch_gemms_shape<EPI, LNF, 4, 2>(p, stream);
    if (pick == 70) return launch_gemms_shape<EPI, LNF, 2, 1, 4>(p, stream);
    return launch_gemms_shape<EPI, LNF, 4, 1>(p, stream);
}

template <int EPI, bool LNF = false>
int launch_gemm64(GemmParams p, hipStream_t stream) {
    p.tiles_m = (p.M + 63) / 64;
    p.tiles_n = (p.N + 63) / 64;
    p.xcd_map = g_gemm64_xcd;
    // one workgroup per CU with a deep ring when the tiles do not fill the CUs once; two per CU with half the ring otherwise
    const bool deep = (long long)p.tiles_m * p.tiles_n <= device_cus(current_device());
    if (g_gemm64_prod) return deep ? launch_gemm64_ns<EPI, LNF, 8, true>(p, stream) : launch_gemm64_ns<EPI, LNF, 4, true>(p, stream);
    return deep ? launch_gemm64_ns<EPI, LNF, 8, false>(p, stream) : launch_gemm64_ns<EPI, LNF, 4, false>(p, stream);
}

template <int EPI, bool LNF = false>
int launch_gemm(const GemmParams& p, hipStream_t s, void* sk_ws, size_t sk_ws_bytes) {
    // the 256x256 kernel wants at least g_gemm_min_tiles tiles (one per CU is 256): below that the 128x128 kernel's four times
    // finer tiling fills the chip better (Q-Former GEMMs at M = B*32)
    const long long tiles256 = (long long)((p.M + B2 - 1) / B2) * ((p.N + B2 - 1) / B2);
    const bool big = p.M >= 1024 && p.N >= 256 && tiles256 >= g_gemm_min_tiles;
    const int variant = g_gemm_variant;
#ifdef SEEDMI_DEVTOOLS
    if (LNF && (variant == 128 || variant == 256)) return (variant == 128) ? launch_gemm128<EPI, LNF>(p, s) : launch_gemm256<EPI, LNF>(p, s, sk_ws, sk_ws_bytes);
    if (variant == 257 && EPI != EPI_SWIGLU && EPI != EPI_PATCH_EMBED) return launch_gemm256k<EPI>(p, s, sk_ws, sk_ws_bytes);
    if (variant == 233 && EPI != EPI_SWIGLU && EPI != EPI_PATCH_EMBED) return launch_gemm256t<EPI>(p, s, sk_ws, sk_ws_bytes);   // two-phase K-tile on 32x32x16
    if (variant == 232) return launch_gemm256x<EPI>(p, s);          // 256x256 tile on v_mfma_f32_32x32x16_bf16
    if (variant == 129 && (EPI == EPI_BIAS || EPI == EPI_NONE)) return launch_gemm128x256<EPI>(p, s);   // 128x256x32, two 4-wave workgroups per CU (round 4 experiment)
    if constexpr (EPI == EPI_BIAS) {                                 // timing-only ablations of it
        if (variant == 130) return launch_gemm128x256<EPI, 1>(p, s);
        if (variant == 131) return launch_gemm128x256<EPI, 3>(p, s);
        if (variant == 132) return launch_gemm128x256<EPI, 7>(p, s);
        if (variant == 133) return launch_gemm128x256<EPI, 8>(p, s);
    }
    if (variant == 255) return launch_gemm256f<EPI>(p, s);          // 256x256, one barrier per K-tile, free-running waves
#endif
    const bool use256 = variant == 256 || (variant == 0 && big);
    if (use256) return launch_gemm256<EPI, LNF>(p, s, sk_ws, sk_ws_bytes);
    // small M (one image: M = 257; a short prompt): when 128x128 tiles cannot give every CU a workgroup, the 64x64 kernel's four times
    // finer tiling and deep ring do (same bits)
    const long long tiles128 = (long long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const bool small = variant == 64 || variant == 65 || (variant >= 66 && variant <= 70) ||
                       (variant == 0 && g_gemm_small && tiles128 < device_cus(current_device()));
    // the shaped kernel (one 12- or 16-wave workgroup per CU) where a launch is a round or two of tiles - measured faster up to four images
    // (M = 1028: 6.22 vs 6.40 ms per pass), slower at eight (10.1 vs 9.7: its rounds run one after the other, gemm64's two 8-wave workgroups
    // per CU overlap each other's prologue and epilogue) - and gemm64 beyond
    constexpr int SHAPED_MAX_M = 1100;
    if (small && variant != 64 && variant != 65 && ((g_gemm_small == 1 && p.M <= SHAPED_MAX_M) || variant >= 66)) return launch_gemms<EPI, LNF>(p, s, variant);
    return small ? launch_gemm64<EPI, LNF>(p, s) : launch_gemm128<EPI, LNF>(p, s);
}

}  // namespace

// m-tiles per tile group of the XCD-contiguous order.  An XCD works on 32 neighbouring tiles at a time, gm m-tiles x 32 / gm n-tiles: gm + 32
// / gm operand panels stream through its L2 (least at gm ~ 6).  Measured back to back at M = 65 792 (tools/gemm_group_sweep.py,
// profiles/r02_group_sweep.txt): 6 is best where there are many n-tiles (QKV +1.6 %, fc1 +0.9 % over 4); with the six n-tiles of N = 1408 a
// group of 3 is best at K = 1408 (proj +2.9 %) and a group of 2 at K = 6144, where one A panel alone is 3 MB (fc2 +1.1 %).
static int auto_group_m(int N, int K) {
    const int tn = (N + B2 - 1) / B2;
    if (tn >= 8) return 6;
    return K > 2048 ? 2 : 3;
}

extern "C" int seedmi_set_option(const char* key, int value) {
#ifdef SEEDMI_DEVTOOLS
    const bool dev_variant = (value >= 129 && value <= 133) || value == 232 || value == 233 || value == 255 || value == 257;
#else
    const bool dev_variant = false;
#endif
    if (key && !strcmp(key, "gemm") && (value == 0 || (value >= 64 && value <= 70) || value == 128 || value == 256 || dev_variant)) {
        g_gemm_variant = value;
        g_gemm64_prod = value != 65;
        return SEEDMI_OK;
    }
#ifdef SEEDMI_DEVTOOLS
    const bool sched_ok = value >= -1 && value <= 65535;                           // (values without a compiled variant run schedule 0)
#else
    const bool sched_ok = value == -1 || value == 0 || value == 8273 || value == 24657 || value == 57425;
#endif
    if (key && !strcmp(key, "gemm_sched") && sched_ok) {                            // (-1 = the default)
        if (value < 0) value = GEMM_SCHED_DEFAULT;
        g_gemm_sched = value;
        return SEEDMI_OK;
    }
#ifdef SEEDMI_DEVTOOLS
    if (key && !strcmp(key, "gemm_store") && (value == 64 || value == 128)) {
        g_gemm_store = value;
        return SEEDMI_OK;
    }
#endif
    if (key && !strcmp(key, "gemm_small") && (value == 0 || value == 1 || value == 2)) {       // 2: the round-5 64x64 kernel (gemm64) instead of the shaped one
        g_gemm_small = value;
        return SEEDMI_OK;
    }
    if (key && !strcmp(key, "gemm64_xcd") && (value == 0 || value == 1)) {
        g_gemm64_xcd = value;
        return SEEDMI_OK;
    }

    if (key && !strcmp(key, "gemm_group_m") && value >= 0 && value <= 64) {
        g_group_m = value;
        return SEEDMI_OK;
    }
    if (key && !strcmp(key, "gemm_min_tiles") && value >= 1 && value <= 4096) {
        g_gemm_min_tiles = value;
        return SEEDMI_OK;
    }
#ifdef SEEDMI_DEVTOOLS
    if (key && !strcmp(key, "gemm_ablate") && value >= 0 && value <= 35) {
        g_gemm_ablate = value;
        return SEEDMI_OK;
    }
#endif
    if (key && !strcmp(key, "gemm_persist") && (value == 0 || value == 1)) {
        g_gemm_persist = value;
        return SEEDMI_OK;
    }
    if (key && !strcmp(key, "gemm_streamk") && (value == 0 || value == 1)) {
        g_gemm_streamk = value;
        return SEEDMI_OK;
    }
#ifdef SEEDMI_DEVTOOLS
    if (key && !strcmp(key, "gemm_prefetch_residual") && (value == 0 || value == 1)) {
        g_gemm_prefetch_r = value;
        return SEEDMI_OK;
    }
    if (key && !strcmp(key, "gemm_residual_nt") && (value == 0 || value == 1)) {
        g_gemm_residual_nt = value;
        return SEEDMI_OK;
    }
    // (statistics by tile associate the row sums differently - other bits: reachable through seedmi_gemm_bf16_ext's own arguments, the
    //  tokenizer-wide switch is a measurement arm)
    if (key && !strcmp(key, "tokenize_tile_stats") && seedmi_tokenizer_set_tilestats(value) == SEEDMI_OK) return SEEDMI_OK;
#endif
    if (key && !strcmp(key, "tokenize_streams") && seedmi_tokenizer_set_streams(value) == SEEDMI_OK) return SEEDMI_OK;
    if (key && !strcmp(key, "tokenize_streamk") && seedmi_tokenizer_set_streamk(value) == SEEDMI_OK) return SEEDMI_OK;
    if (key && !strcmp(key, "tokenize_lnfold") && seedmi_tokenizer_set_lnfold(value) == SEEDMI_OK) return SEEDMI_OK;
    if (key && !strcmp(key, "tokenize_split_rounds") && seedmi_tokenizer_set_split(value) == SEEDMI_OK) return SEEDMI_OK;
    if (key && !strcmp(key, "tokenize_vq_head") && seedmi_tokenizer_set_vqhead(value) == SEEDMI_OK) return SEEDMI_OK;
    if (key && seedmi_llama_set_option(key, value) == SEEDMI_OK) return SEEDMI_OK;
    if (key && seedmi_attn_set_option(key, value) == SEEDMI_OK) return SEEDMI_OK;
    seedmi_set_error("seedmi_set_option: unknown option/value %s=%d", key ? key : "(null)", value);
    return SEEDMI_E_SHAPE;
}

#ifdef SEEDMI_DEVTOOLS
// devtools: device buffer of 2 x 256 uint64 that workgroup 0 of the next 256x256 GEMM launches fills with phase clock stamps
// ((code << 56) | s_memtime; waves 0 and 4), or null to stop
extern "C" int seedmi_gemm_phase_timing(void* buf) { g_gemm_dbg = (unsigned long long*)buf; return SEEDMI_OK; }
#endif

// does seedmi_gemm_bf16_ext(M, N, ...) take the persistent 256x256 kernel under the current options (launch_gemm's rule)?
static bool gemm_uses_256(int M, int N) {
    const long long tiles256 = (long long)((M + B2 - 1) / B2) * ((N + B2 - 1) / B2);
    const bool big = M >= 1024 && N >= 256 && tiles256 >= g_gemm_min_tiles;
    const int variant = g_gemm_variant;
    return variant == 256 || (variant == 0 && big);
}
extern "C" int seedmi_gemm_tile_stats_supported(int M, int N) { return (M > 0 && N > 0 && (N % 64) == 0 && gemm_uses_256(M, N)) ? 1 : 0; }

extern "C" size_t seedmi_gemm_workspace_bytes(void) {
    // stream-K tail of the persistent 256x256 kernel: a flag word and one fp32 accumulator image (256 KiB) per workgroup
    return SK_FLAGS_BYTES + (size_t)device_cus(current_device()) * SK_SLAB_BYTES;
}

extern "C" int seedmi_gemm_bf16_ws(int M, int N, int K, const void* A, int lda, const void* W, int ldw, const void* bias,
                                   const void* residual, int ldr, int epilogue, void* C, int ldc, int row_group,
                                   int row_extra, void* workspace, size_t workspace_bytes, void* stream) {
    return seedmi_gemm_bf16_ext(M, N, K, A, lda, W, ldw, bias, residual, ldr, epilogue, C, ldc, row_group, row_extra, nullptr, workspace,
                                workspace_bytes, stream);
}

extern "C" int seedmi_gemm_bf16_ext(int M, int N, int K, const void* A, int lda, const void* W, int ldw, const void* bias,
                                    const void* residual, int ldr, int epilogue, void* C, int ldc, int row_group, int row_extra,
                                    const seedmi_gemm_ext_t* ext, void* workspace, size_t workspace_bytes, void* stream) {
    const bool lnf = ext && ext->ln_stats;
    if (lnf && (!ext->ln_colsum || !ext->bias_f32 || (epilogue != EPI_BIAS && epilogue != EPI_BIAS_GELU) || (N % 64) ||
                (((uintptr_t)ext->ln_colsum | (uintptr_t)ext->bias_f32) & 15) || ((uintptr_t)ext->ln_stats & 7))) {
        seedmi_set_error("seedmi_gemm_bf16_ext: the LayerNorm fold needs ln_stats, ln_colsum, bias_f32 (16-byte aligned), N %% 64 == 0 and the BIAS or BIAS_GELU epilogue");
        return SEEDMI_E_SHAPE;
    }
    if (ext && (ext->stats_by_tile || ext->ln_planes > 0)) {
        const bool prod_ok = !ext->stats_by_tile || (ext->stats_out && (ext->stats_ld % 2) == 0);
        // consumer planes: the 256x256 kernel takes up to FOLD_MAX_PLANES tile planes through LDS; the small-M kernels sum up to 64 planes (span
        // planes of the 64x64 / 128x128 producers) in their epilogue
        const bool big = gemm_uses_256(M, N);
        const bool cons_ok = ext->ln_planes <= 0 || (lnf && ext->ln_planes <= (big ? FOLD_MAX_PLANES : 64) && ext->ln_ld >= M && ext->ln_cols > 0 &&
                                                     (!big || (ext->ln_ld % 2) == 0));
        if (!prod_ok || !cons_ok || (!big && ext->stats_by_tile)) {
            seedmi_set_error("seedmi_gemm_bf16_ext: stats_by_tile (producer) needs the 256x256 kernel for this shape (seedmi_gemm_tile_stats_supported) and an even "
                             "stats_ld; ln_planes (consumer) needs ln_stats + ln_colsum + bias_f32, ln_ld >= M, ln_cols > 0 and <= %d tile planes with an even "
                             "ln_ld on the 256x256 kernel, <= 64 span planes on the small-M kernels", FOLD_MAX_PLANES);
            return SEEDMI_E_SHAPE;
        }
    }
    if (ext && ext->stats_out && (epilogue != EPI_BIAS_RESIDUAL || (N % 64) || ext->stats_ld < M)) {
        seedmi_set_error("seedmi_gemm_bf16_ext: stats_out belongs to the BIAS_RESIDUAL epilogue and needs N %% 64 == 0 and stats_ld >= M");
        return SEEDMI_E_SHAPE;
    }
    if (workspace && (((uintptr_t)workspace & 255) || workspace_bytes < SK_FLAGS_BYTES)) {
        seedmi_set_error("seedmi_gemm_bf16_ws: workspace must be 256-byte aligned and hold seedmi_gemm_workspace_bytes()");
        return SEEDMI_E_ALIGN;
    }
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
    GemmParams p = {};
    p.M = M; p.N = N; p.K = K;
    p.A = (const bf16_t*)A; p.lda = lda;
    p.W = (const bf16_t*)W; p.ldw = ldw;
    p.bias = (const bf16_t*)bias;
    p.R = (const bf16_t*)residual; p.ldr = ldr;
    p.C = (bf16_t*)C; p.ldc = ldc;
    p.tiles_m = (M + BM - 1) / BM;
    p.tiles_n = (N + BN - 1) / BN;
    p.group_m = g_group_m > 0 ? (int)g_group_m : auto_group_m(N, K);
#ifdef SEEDMI_DEVTOOLS
    const int abl = g_gemm_ablate;
    p.skip_epilogue = (abl == 32) ? 1 : (abl == 33 ? 2 : (abl == 34 ? 3 : (abl == 35 ? 4 : 0)));
#endif
    p.sk_slabs = nullptr; p.sk_flags = nullptr; p.sk_epoch = 0;
    p.ln_stats = lnf ? ext->ln_stats : nullptr;
    p.ln_colsum = lnf ? ext->ln_colsum : nullptr;
    p.bias_f32 = lnf ? ext->bias_f32 : nullptr;
    p.stats_out = ext ? ext->stats_out : nullptr;
    p.stats_ld = ext ? ext->stats_ld : 0;
    p.stats_by_tile = (ext && ext->stats_out) ? ext->stats_by_tile : 0;
    p.ln_planes = (lnf && ext->ln_planes > 0) ? ext->ln_planes : 0;
    p.ln_ld = p.ln_planes ? ext->ln_ld : 0;
    p.ln_inv_cols = p.ln_planes ? 1.0f / (float)ext->ln_cols : 0.f;
    p.ln_eps = p.ln_planes ? ext->ln_eps : 0.f;
    p.prefetch_residual = g_gemm_prefetch_r;
    p.residual_nt = g_gemm_residual_nt;
    p.store128 = g_gemm_store == 128;
    p.row_group = row_group > 0 ? row_group : 1;
    p.row_extra = row_extra;
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case EPI_NONE: return launch_gemm<EPI_NONE>(p, s, workspace, workspace_bytes);
        case EPI_BIAS: return lnf ? launch_gemm<EPI_BIAS, true>(p, s, workspace, workspace_bytes) : launch_gemm<EPI_BIAS>(p, s, workspace, workspace_bytes);
        case EPI_BIAS_GELU:
            return lnf ? launch_gemm<EPI_BIAS_GELU, true>(p, s, workspace, workspace_bytes) : launch_gemm<EPI_BIAS_GELU>(p, s, workspace, workspace_bytes);
        case EPI_BIAS_RESIDUAL:
            return p.stats_out ? launch_gemm<EPI_BIAS_RESIDUAL, true>(p, s, workspace, workspace_bytes)
                               : launch_gemm<EPI_BIAS_RESIDUAL>(p, s, workspace, workspace_bytes);
        case EPI_BIAS_TANH: return launch_gemm<EPI_BIAS_TANH>(p, s, workspace, workspace_bytes);
        case EPI_SWIGLU: return launch_gemm<EPI_SWIGLU>(p, s, workspace, workspace_bytes);
        case EPI_PATCH_EMBED: return launch_gemm<EPI_PATCH_EMBED>(p, s, workspace, workspace_bytes);
        case EPI_RELU: return launch_gemm<EPI_RELU>(p, s, workspace, workspace_bytes);
        default:
            seedmi_set_error("seedmi_gemm_bf16: unknown epilogue %d", epilogue);
            return SEEDMI_E_SHAPE;
    }
}

extern "C" int seedmi_gemm_bf16(int M, int N, int K, const void* A, int lda, const void* W, int ldw, const void* bias,
                                const void* residual, int ldr, int epilogue, void* C, int ldc, int row_group,
                                int row_extra, void* stream) {
    return seedmi_gemm_bf16_ws(M, N, K, A, lda, W, ldw, bias, residual, ldr, epilogue, C, ldc, row_group, row_extra, nullptr, 0,
                               stream);
}
