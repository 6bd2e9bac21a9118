// ViT attention (eva_vit.py:139-156) as a persistent, fully staged pipeline on gfx950: head_dim 88, <= 288 keys.
//
// One workgroup (12 waves) per CU walks (image, head) items.  Per item:
//     wait K | QK^T + softmax (P kept packed in registers) | wait V | PV | store
// with every HBM->LDS transfer issued by LDS-DMA one stage ahead, so that nothing is ever waited for cold:
//     K(i+1) is requested right before PV(i)   (the K image is free once every wave has finished QK^T(i)),
//     V(i+1) right after PV(i)                 (lands during QK^T(i+1)),
//     Q(i+1) travels with K(i+1) into its own LDS image (no ordinary global loads inside the loop: vmcnt retires in
//     order, and a register load younger than V(i+1)'s DMA would make its consumer wait for V as well).
// K, Q and V live in LDS row-major with 192-byte rows (12 x 16 B: head dim padded to 96) whose 16-byte chunks are XOR-swizzled
// inside each aligned group of four: chunk c of row r sits at position c ^ T[(r >> 2) & 3], T = {0,2,3,1}.  With that every
// ds_read_b128 of a K / Q fragment (16 rows x one chunk column per 16-lane group) and every ds_read_b64_tr_b16 of a V fragment
// (8 rows x two chunks per half wave) touches each bank once: the round-1 image (unpadded 176-byte rows) was 2-way on every one of
// them (SQ_LDS_BANK_CONFLICT = 49 % of SQ_LDS_IDX_ACTIVE, profiles/r02_pmc_attention_head.json).  LDS-DMA writes lane-linear images, so
// the permutation is applied to the per-lane SOURCE chunk.  The 12th chunk (columns 88..95) is a copy of the 11th: the Q
// fragment's 12th chunk is zeroed in registers, so it is multiplied by 0, and V columns 88..95 are never stored.
// Same arithmetic and rounding points as attn_fullrow.hip (q*scale -> half, S -> half, P normalised -> half).
#include <string.h>
#include <atomic>
#include "common.h"
#include "seedmi_internal.h"

namespace {

constexpr int VHD = 88, VCH = 11, VNKP = 288, VNT = 17, VKK = 9, VHT = 6;
constexpr int VLD = 96, VCHL = 12;                      // LDS row: 96 elements = 12 chunks of 16 B
constexpr int VWAVES = 12;
constexpr int VMAXT = 2;                                // q-tiles per wave: tiles w, w+12 (n <= 272 -> 17 tiles)
constexpr int VKQ_ROWS = 16 * VNT;                      // 272 rows of K and of Q are ever read
constexpr int VKQ_BYTES = VKQ_ROWS * VLD * 2;           // 52,224 B
constexpr int VV_BYTES = VNKP * VLD * 2;                // 55,296 B: PV walks 9 x 32 keys
constexpr int VLDS_BYTES = 2 * VKQ_BYTES + VV_BYTES;    // 159,744 B
SEEDMI_DEVINL int vswz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

struct VitAttnParams {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
    int ldq, ldk, ldv, ldo;
    int n, heads, items;
    float scale;
    int store_wait;               // 16-wave kernel: 1 = the K / Q wait tolerates the previous item's output stores (seedmi_set_option "attn_store_wait")
    int xcd_map;                  // staggered kernel: 1 = all heads of an image on ONE XCD (see attn_vit16s_kernel; "attn_xcd")
    int qsplit;                   // lock-step 16-wave kernel: an (image, head) item is shared by qsplit workgroups (1, 2, 4, 8 or 16), each running 16 / qsplit of
                                  // its query tiles (part 0 also the side row): small launches (one image = 16 items for 256 CUs)
#ifdef SEEDMI_DEVTOOLS
    unsigned long long* dbg;      // phase clock stamps of workgroup 0 (tools/attn_phase_times.py): [wave][item][6]
#endif
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

SEEDMI_DEVINL void glds16v(const bf16_t* gptr, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

SEEDMI_DEVINL void wait_vm(int leave) {                 // wave-uniform count of youngest VM ops allowed to stay in flight
    switch (leave) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// NFIX > 0: the token count is a compile-time constant (257 for EVA-ViT-g at 224x224), so the dead-key handling of the
// last key tiles costs nothing; NFIX == 0 keeps it a runtime value (the compiler then materialises 72 lane masks in
// SGPRs, spills them and pays ~280 VALU per q-tile for it — measured; only odd test shapes take that path).
template <bool ROUND_S, int NFIX>
__global__ __launch_bounds__(64 * VWAVES) void attn_vit_kernel(VitAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ksm = (bf16_t*)smem;
    bf16_t* Qsm = (bf16_t*)(smem + VKQ_BYTES);
    bf16_t* Vsm = (bf16_t*)(smem + 2 * VKQ_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = NFIX > 0 ? NFIX : p.n;
    const int total_chunks = n * VCHL;
    const int npieces = (total_chunks + 63) >> 6;                     // 1 KiB LDS-DMA pieces per matrix (<= 48)
    const int my_pieces = (npieces - wave + VWAVES - 1) / VWAVES;     // pieces wave, wave+8, ... (wave-uniform, <= 6)
    const int nqt = (n + 15) >> 4;
    const int my_tiles = (nqt - wave + VWAVES - 1) / VWAVES;          // q-tiles wave, wave+8, wave+16

    // V rows >= n must hold finite values (they meet P == 0) and K / Q rows >= n are read too (their scores are overwritten, their
    // output rows never stored): clear all of LDS once, the DMA only ever writes rows < n
    for (int i = tid; i < VLDS_BYTES / 16; i += 64 * VWAVES) *(uint4*)(smem + 16 * i) = make_uint4(0, 0, 0, 0);
    __syncthreads();

    auto stage = [&](const bf16_t* base, int ld, bf16_t* dst, int item) {
        const int b = item / p.heads, h = item - b * p.heads;
        const bf16_t* src = base + (size_t)b * n * ld + h * VHD;
        for (int j = 0; j < my_pieces; ++j) {
            const int piece = wave + VWAVES * j;
            const int q = min(64 * piece + lane, total_chunks - 1);  // LDS chunk position (clamped lanes rewrite the last one)
            const int row = q / VCHL;
            const int c = min((q - row * VCHL) ^ vswz(row), VCH - 1);   // source chunk of that position; the pad chunk copies the 11th
            glds16v(src + (size_t)row * ld + 8 * c, (char*)dst + piece * 1024);
        }
    };

    int item = blockIdx.x;
    if (item >= p.items) return;
    stage(p.K, p.ldk, Ksm, item);
    stage(p.Q, p.ldq, Qsm, item);
    stage(p.V, p.ldv, Vsm, item);
    const float L2E = 1.4426950408889634f;

#ifdef SEEDMI_DEVTOOLS
    int dbg_it = 0;
#define VSTAMP(k_)                                                                                                     \
    do {                                                                                                               \
        if (p.dbg && blockIdx.x == 0 && lane == 0 && dbg_it < 8) p.dbg[(wave * 8 + dbg_it) * 6 + (k_)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define VSTAMP(k_) do {} while (0)
#endif
    for (;;) {
        const int b = item / p.heads, h = item - b * p.heads;
        VSTAMP(0);
        // ---- K(item), Q(item) landed everywhere (only this wave's V pieces may still be in flight)
        wait_vm(my_pieces);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        VSTAMP(1);

        // ---- S^T = K Q^T and softmax per q-tile; P packed to bf16 MFMA operands and kept in registers
        bf16x8 pf[VMAXT][VKK];
#pragma unroll
        for (int t = 0; t < VMAXT; ++t) {
            if (t < my_tiles) {
                const int qt = wave + VWAVES * t;
                bf16x8 qf[3];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {            // q * scale rounded to half; the 12th chunk (cols 88..95) is zero
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (32 * ks + 8 * g < VHD) v = *(const uint4*)(Qsm + (16 * qt + li) * VLD + 8 * ((4 * ks + g) ^ vswz(li)));
                    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = pack2bf(lo_bf(w[i]) * p.scale, hi_bf(w[i]) * p.scale);
                    qf[ks] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
                }
                f32x4 s[VNT];
                // software-pipelined K fragment reads: group gi+1 is requested before group gi's MFMAs are issued
                bf16x8 fk0[6], fk1[6];
                auto ldk = [&](bf16x8 (&f)[6], int grp) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int ks = 0; ks < 3; ++ks)
                            if (2 * grp + u < VNT)
                                f[3 * u + ks] = *(const bf16x8*)(Ksm + (16 * (2 * grp + u) + li) * VLD + 8 * ((4 * ks + g) ^ vswz(li)));
                };
                auto mmk = [&](bf16x8 (&f)[6], int grp) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (2 * grp + u >= VNT) continue;
                        s[2 * grp + u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < 3; ++ks)
                            s[2 * grp + u] = seedmi_mfma_16x16x32(f[3 * u + ks], qf[ks], s[2 * grp + u]);
                    }
                };
                ldk(fk0, 0);
#pragma unroll
                for (int grp = 0; grp < (VNT + 1) / 2; ++grp) {
                    if (grp & 1) {
                        if (grp + 1 < (VNT + 1) / 2) ldk(fk0, grp + 1);
                        mmk(fk1, grp);
                    } else {
                        if (grp + 1 < (VNT + 1) / 2) ldk(fk1, grp + 1);
                        mmk(fk0, grp);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // softmax over keys {16*kt + 4g + r}: S rounded to half like the reference's matmul output
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < VNT; ++kt) {
                    float v0 = s[kt][0], v1 = s[kt][1], v2 = s[kt][2], v3 = s[kt][3];
                    if (ROUND_S) {
                        const uint32_t w0 = pack2bf(v0, v1), w1 = pack2bf(v2, v3);
                        v0 = lo_bf(w0); v1 = hi_bf(w0); v2 = lo_bf(w1); v3 = hi_bf(w1);
                    }
                    const int live = n - 16 * kt;            // keys of this tile that exist (compile-time when NFIX > 0)
                    if (live <= 0) {
                        v0 = v1 = v2 = v3 = -INFINITY;
                    } else if (live < 16) {
                        const int k0 = 4 * g;
                        v0 = (k0 + 0 >= live) ? -INFINITY : v0;
                        v1 = (k0 + 1 >= live) ? -INFINITY : v1;
                        v2 = (k0 + 2 >= live) ? -INFINITY : v2;
                        v3 = (k0 + 3 >= live) ? -INFINITY : v3;
                    }
                    s[kt][0] = v0; s[kt][1] = v1; s[kt][2] = v2; s[kt][3] = v3;
                    mx = fmaxf(fmaxf(mx, v0), fmaxf(v1, fmaxf(v2, v3)));
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float nmx = -mx * L2E;
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < VNT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][r], L2E, nmx));
                        s[kt][r] = e;
                        sum += e;
                    }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int kk = 0; kk < VKK; ++kk) {
                    uint4 pw;
                    pw.x = pack2bf(s[2 * kk][0] * inv, s[2 * kk][1] * inv);
                    pw.y = pack2bf(s[2 * kk][2] * inv, s[2 * kk][3] * inv);
                    pw.z = pw.w = 0u;                                  // (keys 272..287 do not exist: P = 0)
                    if (2 * kk + 1 < VNT) {
                        pw.z = pack2bf(s[2 * kk + 1][0] * inv, s[2 * kk + 1][1] * inv);
                        pw.w = pack2bf(s[2 * kk + 1][2] * inv, s[2 * kk + 1][3] * inv);
                    }
                    pf[t][kk] = __builtin_bit_cast(bf16x8, pw);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        VSTAMP(2);
        // ---- V(item) landed; every wave is done with K(item) and Q(item)
        wait_vm(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        VSTAMP(3);
        const int next = item + gridDim.x;
        const bool more = next < p.items;
        if (more) {
            stage(p.K, p.ldk, Ksm, next);
            stage(p.Q, p.ldq, Qsm, next);
        }

        // ---- O^T = V^T P^T (hardware transpose read of the row-major V image), store
#pragma unroll
        for (int t = 0; t < VMAXT; ++t) {
            if (t < my_tiles) {
                f32x4 o[VHT];
#pragma unroll
                for (int nn = 0; nn < VHT; ++nn) o[nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
                bf16x8 fv0[VHT], fv1[VHT];
                auto ldv = [&](bf16x8 (&f)[VHT], int kk) {
#pragma unroll
                    for (int nn = 0; nn < VHT; ++nn) {
                        // rows 32kk + 4g + (li>>2) and + 16: both have (row >> 2) & 3 == g, i.e. the same chunk permutation
                        const bf16_t* vp = Vsm + (32 * kk + 4 * g + (li >> 2)) * VLD + 8 * ((2 * nn + ((li & 3) >> 1)) ^ vswz(4 * g)) + 4 * (li & 1);
                        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)vp);
                        const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 16 * VLD));
                        const uint2 lo = __builtin_bit_cast(uint2, a), hi = __builtin_bit_cast(uint2, c);
                        f[nn] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                    }
                };
                ldv(fv0, 0);
#pragma unroll
                for (int kk = 0; kk < VKK; ++kk) {
                    if (kk & 1) {
                        if (kk + 1 < VKK) ldv(fv0, kk + 1);
#pragma unroll
                        for (int nn = 0; nn < VHT; ++nn) o[nn] = seedmi_mfma_16x16x32(fv1[nn], pf[t][kk], o[nn]);
                    } else {
                        if (kk + 1 < VKK) ldv(fv1, kk + 1);
#pragma unroll
                        for (int nn = 0; nn < VHT; ++nn) o[nn] = seedmi_mfma_16x16x32(fv0[nn], pf[t][kk], o[nn]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int qrow = 16 * (wave + VWAVES * t) + li;
                if (qrow < n) {
                    bf16_t* op = p.O + ((size_t)b * n + qrow) * p.ldo + h * VHD;
#pragma unroll
                    for (int nn = 0; nn < VHT; ++nn) {
                        const int c0 = 16 * nn + 4 * g;
                        if (c0 + 4 <= VHD) {
                            uint2 w;
                            w.x = pack2bf(o[nn][0], o[nn][1]);
                            w.y = pack2bf(o[nn][2], o[nn][3]);
                            *(uint2*)(op + c0) = w;
                        }
                    }
                }
            }
        }
        VSTAMP(4);
        if (!more) break;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                              // every wave is done with V(item)
        VSTAMP(5);
#ifdef SEEDMI_DEVTOOLS
        ++dbg_it;
#endif
        stage(p.V, p.ldv, Vsm, next);
        item = next;
    }
}

// ======================================================================================================
// 16-wave form for n == 257 (EVA-ViT-g at 224x224): ONE 16-query tile per wave, the 257th query row on a side path.
//
// The 12-wave kernel above gives waves 0..4 two query tiles and waves 5..11 one (17 tiles), so every phase lasts as long as a two-tile
// wave needs for its tiles one after the other - softmax is ~450 VALU instructions per tile and a single wave issues one every 4-5
// cycles - while the SIMDs hold 5 / 4 / 4 / 4 tiles (profiles/r02_attn_phase_times.txt: QK^T + softmax 9-14 k cycles, PV 8-11.5 k,
// waits 6-10 k of a 31.6 k cycle item).  Here every wave owns exactly one tile (4 waves per SIMD, 128 registers each), so a phase lasts
// one tile's time with four waves interleaving on each SIMD, and the 17th tile - a single valid row - costs one wave a fraction of a tile:
//   * its scores are formed with the operands SWAPPED (S = Q K^T instead of S^T = K Q^T): the accumulator then holds row 256 as ONE value
//     per key tile in lanes 0..15 (key = 16 kt + lane), i.e. 17 values per lane instead of 68, and the softmax of that row is ~110
//     VALU instructions on 16 lanes instead of ~450 on 64 (same arithmetic and rounding points; the reductions run over the 16 lanes);
//   * the normalised row goes to LDS as 272 halves; after the phase barrier ANOTHER wave (another SIMD) reads it back in the PV operand
//     layout (every lane group g reads keys 32 kk + 4 g .. + 3 and + 16: a broadcast) and runs the ordinary PV for it.
// The two side jobs rotate over the waves item by item.  Same arithmetic and rounding points as attn_vit_kernel / attn_fullrow.hip.
constexpr int V16_WAVES = 16;
constexpr int V16_N = 257;
constexpr int V16_P_OFF = VLDS_BYTES;                   // the side row: 288 halves (keys 272..287 stay zero)
constexpr int V16_LDS_BYTES = VLDS_BYTES + 1024;

// MODE bit 0: the output tile leaves as three 16-byte stores per lane instead of six 8-byte ones (one v_permlane16_swap per dword joins the
//             8-byte pieces of two lane groups: every store instruction then writes 64 contiguous bytes of each of its 16 rows).  The 96 / 51
//             store instructions of an item all end up queued in the CU's one address unit at the end of the PV phase.
//      bit 1: softmax normalisation moved behind PV.  P leaves the exponential UN-normalised (e = 2^((s - max) log2 e) <= 1, rounded to half)
//             and the row sum comes out of the PV MFMAs themselves: the V image's 89th column (first pad column, never loaded: the staging
//             skips the pad chunk) holds 1.0, so O^T row 88 = sum_k half(e_k) - exactly the sum of the values that were multiplied.
//             O = O' / that sum: 24 multiplies instead of a 68-term sum, 68 multiplies and the reduction shuffles per tile; the exponent
//             arguments are formed two at a time (v_pk_fma_f32).  ~270 instead of ~450 VALU instructions per tile, and the phase that
//             holds them is VALU-bound.  This MOVES a rounding point: the reference rounds the normalised probabilities to half before PV
//             (eva_vit.py:153-156 under autocast), here the un-normalised ones are rounded and the quotient is formed in fp32 - the same
//             relative rounding error per term, normalised consistently; results differ from the other kernels in the last half ulp.
template <bool ROUND_S, int MODE>
__global__ __launch_bounds__(64 * V16_WAVES) void attn_vit16_kernel(VitAttnParams p) {
    constexpr bool WIDE = (MODE & 1) != 0, FLASH = (MODE & 2) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ksm = (bf16_t*)smem;
    bf16_t* Qsm = (bf16_t*)(smem + VKQ_BYTES);
    bf16_t* Vsm = (bf16_t*)(smem + 2 * VKQ_BYTES);
    bf16_t* Psm = (bf16_t*)(smem + V16_P_OFF);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (128 registers per wave: every phase takes its lane id where it needs it - fresh_lane() - so that no lane-derived address lives across
    // phases; what hipcc would otherwise spill is reloaded by scratch loads, VM operations in the middle of the LDS-DMA pipeline)
    constexpr int n = V16_N;
    constexpr int total_chunks = n * VCHL;
    constexpr int npieces = (total_chunks + 63) >> 6;                 // 49 pieces of 1 KiB per matrix
    const int my_pieces = (npieces - wave + V16_WAVES - 1) / V16_WAVES;   // pieces wave, wave + 16, ... (wave-uniform, 3 or 4)

    for (int i = tid; i < V16_LDS_BYTES / 16; i += 64 * V16_WAVES) *(uint4*)(smem + 16 * i) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (FLASH) {
        // the V image's pad chunk (columns 88..95) of every row: {1, 0, ..}.  It sits at chunk position 11 ^ swizzle(row); the V staging
        // below never writes it, so it is set once.
        for (int r = tid; r < VNKP; r += 64 * V16_WAVES)
            *(uint4*)((char*)Vsm + r * (VLD * 2) + 16 * ((VCHL - 1) ^ vswz(r))) = make_uint4((unsigned)f2bf(1.0f), 0, 0, 0);      // 1.0 in the build's 16-bit element
        __syncthreads();
    }

    // (Round 3, measured and NOT adopted - profiles/r03_call21_attention_buffer_staging.log, r03_call22_attention_stride3_staging.log: these
    // requests in the buffer form of gemm_bf16.hip - lane offset formed once per call and shared by K and Q - 152.7 vs 152.2 us at B = 128;
    // a wave's pieces dealt as p0, p0 + 3, p0 + 6 (the same chunk column 16 rows further down: ONE lane offset per call, the rest scalar)
    // 160.7 us: the ~90 VALU instructions saved per item cost more than they bring once a wave's requests sit 3 KiB apart instead of
    // adjacent to its neighbours'.  Lane offsets kept in registers across the item: 240 B of scratch in this 128-register kernel.)
    auto stage = [&](const bf16_t* base, int ld, bf16_t* dst, int item, bool skip_pad = false) {
        const int b = item / p.heads, h = item - b * p.heads;
        const bf16_t* src = base + (size_t)b * n * ld + h * VHD;
        const int lane = fresh_lane();
        for (int j = 0; j < my_pieces; ++j) {
            const int piece = wave + V16_WAVES * j;
            const int q0 = 64 * piece + lane;
            const int q = min(q0, total_chunks - 1);                 // LDS chunk position (clamped lanes rewrite the last one)
            const int row = q / VCHL;
            const int c0 = (q - row * VCHL) ^ vswz(row);
            const int c = min(c0, VCH - 1);                          // source chunk of that position; the pad chunk copies the 11th
            // (skip_pad: lanes that would write a pad chunk - or, clamped, re-write the last position - stay out: EXEC-masked LDS-DMA)
            if (!skip_pad || (c0 < VCH && q0 < total_chunks)) glds16v(src + (size_t)row * ld + 8 * c, (char*)dst + piece * 1024);
        }
    };

    // Small launches: p.qsplit workgroups share an item.  A query tile is ONE wave's private work from its Q fragment to its output rows
    // (K, V and the side row's LDS copy are only read), so which workgroup's wave runs it changes nothing in its arithmetic: bit-identical to
    // the unsplit launch and to the staggered kernel (tests).  Part s runs the tiles of waves [s 16 / S, (s + 1) 16 / S) - one per SIMD
    // first - and part 0 the side row; every part stages the whole K / Q / V images (L2-resident after the first part's pass).
    const int S = p.qsplit, part = S > 1 ? (int)blockIdx.x % S : 0;
    const bool main_on = S <= 1 || wave / (V16_WAVES / S) == part, side_on = part == 0;
    int item = S > 1 ? (int)blockIdx.x / S : (int)blockIdx.x;
    const int item_step = S > 1 ? (int)gridDim.x / S : (int)gridDim.x;
    if (item >= p.items) return;
    stage(p.K, p.ldk, Ksm, item);
    stage(p.Q, p.ldq, Qsm, item);
    stage(p.V, p.ldv, Vsm, item, FLASH);
    const float L2E = 1.4426950408889634f;
#ifdef SEEDMI_DEVTOOLS
    // phase clock stamps of workgroup 0 (tools/attn_phase_times.py): s_memtime into SGPR pairs, read back only at the item's end
    unsigned long long vt[9];
#define V16STAMP(k_) do { if (p.dbg && blockIdx.x == 0) asm volatile("s_memtime %0" : "=s"(vt[k_])); } while (0)
#else
#define V16STAMP(k_) do {} while (0)
#endif

    int stores_behind = 0;
    for (int it = 0;; ++it) {
        const int b = item / p.heads, h = item - b * p.heads;
        const int side_a = it & 15, side_b = (it + 6) & 15;           // waves that take row 256: scores + softmax / PV (different SIMDs)
        V16STAMP(0);
        // ---- K(item), Q(item) landed everywhere (only this wave's V pieces may still be in flight).  vmcnt retires in order and counts
        // stores: behind the K / Q requests of this item sit the previous item's output stores (issued at the end of its PV phase) and
        // then the V pieces - counting only the V pieces would also wait for those stores to reach memory, a few hundred cycles after
        // they were issued.  `stores_behind` is a lower bound of the store instructions this wave issued there (round 4; env
        // SEEDMI_ATTN_STORE_WAIT=0 restores the old count for A/B).
        wait_vm(my_pieces + stores_behind);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        V16STAMP(1);

        // ---- S^T = K Q^T for query tile `wave`, softmax, P packed to bf16 MFMA operands
        bf16x8 pf[VKK];
        if (main_on) {
            // fragment addresses (bytes): row li of a 16-row tile, chunk (4 ks + g) ^ swizzle(li); + 16 * 192 bytes per tile (immediate)
            const int lane = fresh_lane(), li = lane & 15, g = lane >> 4;
            int koff[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) koff[ks] = li * (VLD * 2) + 16 * ((4 * ks + g) ^ vswz(li));
            bf16x8 qf[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {                // q * scale rounded to half; the 12th chunk (cols 88..95) is zero
                uint4 v = make_uint4(0, 0, 0, 0);
                if (32 * ks + 8 * g < VHD) v = *(const uint4*)((const char*)Qsm + 16 * wave * (VLD * 2) + koff[ks]);
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = pack2bf(lo_bf(w[i]) * p.scale, hi_bf(w[i]) * p.scale);
                qf[ks] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            }
            f32x4 s[VNT];
            bf16x8 fk[2][3];                                 // K fragments of one key tile, double buffered
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) fk[0][ks] = *(const bf16x8*)((const char*)Ksm + koff[ks]);
#pragma unroll
            for (int kt = 0; kt < VNT; ++kt) {
                if (kt + 1 < VNT) {
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) fk[(kt + 1) & 1][ks] = *(const bf16x8*)((const char*)Ksm + (kt + 1) * 16 * (VLD * 2) + koff[ks]);
                }
                s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) s[kt] = seedmi_mfma_16x16x32(fk[kt & 1][ks], qf[ks], s[kt]);
                __builtin_amdgcn_sched_barrier(0);         // (keeps the fragment reads one tile ahead, not ten: 128 registers per wave)
            }
            V16STAMP(2);
            // softmax over keys {16 kt + 4 g + r}: S rounded to half like the reference's matmul output.  Four running maxima keep the
            // dependent chain short (max is exact in any order).
            float mx4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int kt = 0; kt < VNT; ++kt) {
                float v0 = s[kt][0], v1 = s[kt][1], v2 = s[kt][2], v3 = s[kt][3];
                if (ROUND_S) {
                    const uint32_t w0 = pack2bf(v0, v1), w1 = pack2bf(v2, v3);
                    v0 = lo_bf(w0); v1 = hi_bf(w0); v2 = lo_bf(w1); v3 = hi_bf(w1);
                }
                if (kt == VNT - 1) {                         // keys 256 .. 271: only key 256 (g == 0, r == 0) exists
                    v0 = (g == 0) ? v0 : -INFINITY;
                    v1 = v2 = v3 = -INFINITY;
                }
                s[kt][0] = v0; s[kt][1] = v1; s[kt][2] = v2; s[kt][3] = v3;
                mx4[kt & 3] = fmaxf(mx4[kt & 3], fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)));
            }
            float mx = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float nmx = -mx * L2E;
            if (FLASH) {
                // un-normalised exponentials, arguments two at a time; the row sum is formed by the PV MFMAs (ones column of V)
                const f32x2 l2 = {L2E, L2E}, nm2 = {nmx, nmx};
#pragma unroll
                for (int kk = 0; kk < VKK; ++kk) {
                    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int kt = 2 * kk + u;
                        if (kt < VNT) {
                            const f32x2 a0 = __builtin_elementwise_fma((f32x2){s[kt][0], s[kt][1]}, l2, nm2);
                            const f32x2 a1 = __builtin_elementwise_fma((f32x2){s[kt][2], s[kt][3]}, l2, nm2);
                            w[2 * u] = pack2bf(__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1]));
                            w[2 * u + 1] = pack2bf(__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1]));
                        }
                    }
                    pf[kk] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
                }
            } else {
            float sum = 0.f;                                 // (one chain in (kt, r) order, like the 12-wave kernel: bit-identical rows)
#pragma unroll
            for (int kt = 0; kt < VNT; ++kt) {
                // (exponent arguments two at a time - v_pk_fma_f32 - the same fma per element)
                const f32x2 a0 = __builtin_elementwise_fma((f32x2){s[kt][0], s[kt][1]}, (f32x2){L2E, L2E}, (f32x2){nmx, nmx});
                const f32x2 a1 = __builtin_elementwise_fma((f32x2){s[kt][2], s[kt][3]}, (f32x2){L2E, L2E}, (f32x2){nmx, nmx});
                const float arg[4] = {a0[0], a0[1], a1[0], a1[1]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(arg[r]);
                    s[kt][r] = e;
                    sum += e;
                }
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int kk = 0; kk < VKK; ++kk) {
                uint4 pw;
                pw.x = pack2bf(s[2 * kk][0] * inv, s[2 * kk][1] * inv);
                pw.y = pack2bf(s[2 * kk][2] * inv, s[2 * kk][3] * inv);
                pw.z = pw.w = 0u;                                  // (keys 272..287 do not exist: P = 0)
                if (2 * kk + 1 < VNT) {
                    pw.z = pack2bf(s[2 * kk + 1][0] * inv, s[2 * kk + 1][1] * inv);
                    pw.w = pack2bf(s[2 * kk + 1][2] * inv, s[2 * kk + 1][3] * inv);
                }
                pf[kk] = __builtin_bit_cast(bf16x8, pw);
            }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        V16STAMP(3);
        if (side_on && wave == side_a) {
            // ---- row 256 (query tile 16, rows 257..271 of the Q image are zero): operands swapped, S[q = 4 g + r][key = 16 kt + li].
            //      Only q == 0 exists: lanes 0..15, accumulator register 0.
            const int lane = fresh_lane(), li = lane & 15, g = lane >> 4;
            int koff[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) koff[ks] = li * (VLD * 2) + 16 * ((4 * ks + g) ^ vswz(li));
            bf16x8 qf[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (32 * ks + 8 * g < VHD) v = *(const uint4*)((const char*)Qsm + 16 * 16 * (VLD * 2) + koff[ks]);
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = pack2bf(lo_bf(w[i]) * p.scale, hi_bf(w[i]) * p.scale);
                qf[ks] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            }
            float t[VNT];
            bf16x8 fk[2][3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) fk[0][ks] = *(const bf16x8*)((const char*)Ksm + koff[ks]);
#pragma unroll
            for (int kt = 0; kt < VNT; ++kt) {
                if (kt + 1 < VNT) {
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) fk[(kt + 1) & 1][ks] = *(const bf16x8*)((const char*)Ksm + (kt + 1) * 16 * (VLD * 2) + koff[ks]);
                }
                f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) a = seedmi_mfma_16x16x32(qf[ks], fk[kt & 1][ks], a);
                float v = a[0];
                if (ROUND_S) v = rbf(v);
                if (kt == VNT - 1) v = (li == 0) ? v : -INFINITY;   // keys 257..271 do not exist
                t[kt] = v;
                __builtin_amdgcn_sched_barrier(0);
            }
            float mx = t[0];
#pragma unroll
            for (int kt = 1; kt < VNT; ++kt) mx = fmaxf(mx, t[kt]);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            const float nmx = -mx * L2E;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < VNT; ++kt) {
                t[kt] = __builtin_amdgcn_exp2f(fmaf(t[kt], L2E, nmx));
                sum += t[kt];
            }
            float inv = 1.0f;                               // FLASH: the row leaves un-normalised, like the tiles
            if (!FLASH) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
                inv = __builtin_amdgcn_rcpf(sum);
            }
            if (g == 0) {
#pragma unroll
                for (int kt = 0; kt < VNT; ++kt) Psm[16 * kt + li] = f2bf(t[kt] * inv);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        V16STAMP(4);
        // ---- V(item) landed; every wave is done with K(item) and Q(item); the side row is in LDS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        V16STAMP(5);
        const int next = item + item_step;
        const bool more = next < p.items;
        if (more) {
            stage(p.K, p.ldk, Ksm, next);
            stage(p.Q, p.ldq, Qsm, next);
        }

        // ---- O^T = V^T P^T (hardware transpose read of the row-major V image), store
        const int lane = fresh_lane(), li = lane & 15, g = lane >> 4;
        const bf16_t* vlane = Vsm + (4 * g + (li >> 2)) * VLD + 4 * (li & 1);
        auto ldv = [&](bf16x8 (&f)[VHT], int kk) {
#pragma unroll
            for (int nn = 0; nn < VHT; ++nn) {
                // rows 32kk + 4g + (li>>2) and + 16: both have (row >> 2) & 3 == g, i.e. the same chunk permutation
                const bf16_t* vp = vlane + 32 * kk * VLD + 8 * ((2 * nn + ((li & 3) >> 1)) ^ vswz(4 * g));
                const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)vp);
                const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 16 * VLD));
                const uint2 lo = __builtin_bit_cast(uint2, a), hi = __builtin_bit_cast(uint2, c);
                f[nn] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }
        };
        auto store_o = [&](const f32x4 (&o)[VHT], int qtile, bool first_row_only) {
            const int sl = fresh_lane(), g = sl >> 4;
            float inv = 1.0f;
            if (FLASH) {
                // O^T row 88 (= 16 * 5 + 4 * 2 + 0: tile nn = 5, lane group 2, register 0) is the row sum of the query in column li
                const float rs = __shfl(o[VHT - 1][0], (sl & 15) + 32, 64);
                inv = __builtin_amdgcn_rcpf(rs);
            }
            uint2 w[VHT];
#pragma unroll
            for (int nn = 0; nn < VHT; ++nn) {
                w[nn].x = pack2bf(o[nn][0] * inv, o[nn][1] * inv);
                w[nn].y = pack2bf(o[nn][2] * inv, o[nn][3] * inv);
            }
            bf16_t* op = p.O + ((size_t)b * n + 16 * qtile + (sl & 15)) * p.ldo + h * VHD;
            if (WIDE) {
                // tiles nn, nn + 1: after the swaps lane group g holds columns 16 (nn + (g & 1)) + 8 (g >> 1) .. + 7 as (x, y | x', y')
#pragma unroll
                for (int nn = 0; nn < VHT; nn += 2) {
                    const auto tx = __builtin_amdgcn_permlane16_swap(w[nn].x, w[nn + 1].x, false, false);
                    const auto ty = __builtin_amdgcn_permlane16_swap(w[nn].y, w[nn + 1].y, false, false);
                    const int c0 = 16 * (nn + (g & 1)) + 8 * (g >> 1);
                    if (c0 + 8 <= VHD && !(first_row_only && (sl & 15) != 0)) *(uint4*)(op + c0) = make_uint4(tx[0], ty[0], tx[1], ty[1]);
                }
            } else {
                if (first_row_only && (sl & 15) != 0) return;
#pragma unroll
                for (int nn = 0; nn < VHT; ++nn) {
                    const int c0 = 16 * nn + 4 * g;
                    if (c0 + 4 <= VHD) *(uint2*)(op + c0) = w[nn];
                }
            }
        };
        if (main_on) {
            f32x4 o[VHT];
#pragma unroll
            for (int nn = 0; nn < VHT; ++nn) o[nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16x8 fv0[VHT], fv1[VHT];
            ldv(fv0, 0);
#pragma unroll
            for (int kk = 0; kk < VKK; ++kk) {
                if (kk & 1) {
                    if (kk + 1 < VKK) ldv(fv0, kk + 1);
#pragma unroll
                    for (int nn = 0; nn < VHT; ++nn) o[nn] = seedmi_mfma_16x16x32(fv1[nn], pf[kk], o[nn]);
                } else {
                    if (kk + 1 < VKK) ldv(fv1, kk + 1);
#pragma unroll
                    for (int nn = 0; nn < VHT; ++nn) o[nn] = seedmi_mfma_16x16x32(fv0[nn], pf[kk], o[nn]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            V16STAMP(6);
            store_o(o, wave, false);
            // the tile's rows all exist (16 wave + li < 257): every store instruction of store_o has active lanes
            stores_behind = p.store_wait ? (WIDE ? VHT / 2 : VHT) : 0;
        }
        V16STAMP(7);
        if (side_on && wave == side_b) {
            // row 256: P^T operand straight from the side row (lane group g takes keys 32 kk + 4 g .. + 3 and + 16 .. : every column of
            // the operand is the same row, only column li == 0 is stored)
            f32x4 o[VHT];
#pragma unroll
            for (int nn = 0; nn < VHT; ++nn) o[nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16x8 fv0[VHT], fv1[VHT];
            const bf16_t* prow = Psm + 4 * (fresh_lane() >> 4);
            ldv(fv0, 0);
#pragma unroll
            for (int kk = 0; kk < VKK; ++kk) {
                const uint2 plo = *(const uint2*)(prow + 32 * kk), phi = *(const uint2*)(prow + 32 * kk + 16);
                const bf16x8 pr = __builtin_bit_cast(bf16x8, make_uint4(plo.x, plo.y, phi.x, phi.y));
                if (kk & 1) {
                    if (kk + 1 < VKK) ldv(fv0, kk + 1);
#pragma unroll
                    for (int nn = 0; nn < VHT; ++nn) o[nn] = seedmi_mfma_16x16x32(fv1[nn], pr, o[nn]);
                } else {
                    if (kk + 1 < VKK) ldv(fv1, kk + 1);
#pragma unroll
                    for (int nn = 0; nn < VHT; ++nn) o[nn] = seedmi_mfma_16x16x32(fv0[nn], pr, o[nn]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            store_o(o, 16, true);
        }
        V16STAMP(8);
#ifdef SEEDMI_DEVTOOLS
        if (p.dbg && blockIdx.x == 0 && it < 8) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (fresh_lane() == 0)
                for (int q = 0; q < 9; ++q) p.dbg[(wave * 8 + it) * 9 + q] = vt[q];
        }
#endif
        if (!more) break;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // every wave is done with V(item) and the side row
        stage(p.V, p.ldv, Vsm, next, FLASH);
        item = next;
    }
}

// ======================================================================================================
// Staggered form of the 16-wave kernel (round 4): the same per-wave work - one 16-query tile, the same fragment reads, MFMAs, softmax and
// rounding points: bit-identical outputs - but the two halves of the workgroup run ONE PHASE APART.
//
// Why: in attn_vit16_kernel all 16 waves are in the same phase (barriers between QK^T + softmax and PV), and the phases load different units:
// QK^T and PV are LDS-read + MFMA work (every wave reads the whole K image / the whole V image: 0.84 + 0.89 MB of LDS reads per item against
// 0.1 MB of output), the softmax is ~450 VALU instructions + 68 quarter-rate exponentials per wave during which LDS and the matrix pipes idle.
// The phase stamps (profiles/r03_call8_attention_phase_stamps.log) show the four waves of a SIMD leaving QK^T after 2.4 / 3.8 / 7.8 / 10.1 k
// cycles and PV after 5.3 .. 10.4 k: a phase lasts four waves' worth of one kind of work while the other units wait for it.
// Here waves 0..7 (group A, two per SIMD) and waves 8..15 (group B, two per SIMD) walk the same items, B one slot behind A:
//     slot 3k     A: QK^T(k)        B: PV(k-1)
//     slot 3k+1   A: softmax(k)     B: QK^T(k)          V(k) requested at the slot's start      (V(k-1) was last read in slot 3k)
//     slot 3k+2   A: PV(k)          B: softmax(k)       K(k+1), Q(k+1) requested at its start   (K(k), Q(k) were last read in slot 3k+1)
// so in two slots of three a VALU-bound phase runs next to an LDS / MFMA-bound one, with two waves of each kind on every SIMD.  One s_barrier
// per slot; every request has one slot (~5 k cycles) of flight before the barrier that publishes it.  The three LDS images (their request addresses are formed more cheaply: stage2 below),
// the side path of row 256 (scores by an A wave in slot 3k+1, its PV by a B wave in slot 3k+3) are those of attn_vit16_kernel.
// Measured at B = 128, bit-identical to the lock-step kernel on every element (first launch and 20-launch bursts, three batch sizes;
// profiles/r04_call14_ .. r04_call23_*attention*.log), us per launch, lock-step kernel (attn_vit = 3) on the same box in brackets:
//   the stagger alone                                             129.7  [138.5]
//   + cheaper staging addresses, max3 chain                       127.2  [134.2]   (the VALU saved - a quarter of a wave's instructions - bought 1 %)
//   + XCD-aware item walk (all heads of an image on one XCD)      119.7  [133.3]   (plain walk on the same box: 128.8)
//   + wave priorities softmax > PV > QK^T                         114.6  [133.6]   the tokenize pass 120.80 vs 121.64 ms (+0.7 %)
//   + the last key tile's three non-existent keys skipped         112.7  [132.6]   = -15 %
// What did NOT help: K fragments two key tiles ahead; fencing the fragment requests in front of the MFMAs (hipcc sinks them to 2-3 MFMAs
// of distance: restoring the written order is 3 % slower); dealing the side row's PV over six waves (kept: no slower, and no wave runs a
// second PV pass); the normalisation behind PV (attn_vit = 6: 118.6 - slower than the exact form now).  Slot stamps (devtools build):
// 6.2 / 9.1 / 10.4 k cycles; an LDS fragment read costs a wave ~45 cycles in either MFMA phase (54 x ds_read_b128 in 2.4 k, 108 x
// ds_read_b64_tr_b16 in 5.3 k), i.e. the CU's LDS runs at about a third of its 256 B/clk while MFMA issue is at a quarter: the
// remaining bound is the per-wave LDS request rate, which only larger per-wave query tiles (fewer waves: measured worse) would cut.
// The two groups are two straight-line loops (not one loop with a phase switch): the scores (68 registers) live from QK^T to softmax and the
// packed probabilities (36) from softmax to PV, and only a loop per group lets the compiler see that they are never live together.
template <int MODE>
__global__ __launch_bounds__(64 * V16_WAVES) void attn_vit16s_kernel(VitAttnParams p) {
    constexpr bool WIDE = (MODE & 1) != 0, FLASH = (MODE & 2) != 0;
    // Wave priorities per phase (s_setprio; MODE bit 2 = none, for A/B): the softmax - pure VALU issue, nothing to overlap it with inside a
    // wave - above PV above QK^T.  Without them the hardware's oldest-first arbitration lets group A win every slot (its QK^T 3.0 k cycles
    // against B's 6.1 k, softmax 4.2 k against 8.4 k).  Measured at B = 128 (profiles/r04_call20_attention_priorities.log), {QK^T, softmax,
    // PV} = none 118.4 us | {3,2,1} 116.9 | {2,3,1} 116.7 | {3,1,2} 119.9 | {1,3,2} 114.3 | {1,2,3} 117.7.
    // Second sweep around the winner (r04_call27_attention_priorities_2.log): {1,3,2} 113.6 | {0,3,1} 113.7 | {1,3,3} 115.0 | {1,2,2} 115.4 | {2,3,3} 115.4 | {1,3,1} 116.4.
    constexpr bool PRS = (MODE & 4) == 0;
    // (measured and removed, profiles/r04_call24_attention_last_tile.log: the side row's scores in group A's QK^T slot - the shortest phase of
    // the three - with two side rows in LDS taken in turn: 121.2 us against 112.7)
    constexpr int PR_QK = 1, PR_SM = 3, PR_PV = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ksm = (bf16_t*)smem;
    bf16_t* Qsm = (bf16_t*)(smem + VKQ_BYTES);
    bf16_t* Vsm = (bf16_t*)(smem + 2 * VKQ_BYTES);
    bf16_t* Psm = (bf16_t*)(smem + V16_P_OFF);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int n = V16_N;
    constexpr int total_chunks = n * VCHL;
    constexpr int npieces = (total_chunks + 63) >> 6;
    const int my_pieces = (npieces - wave + V16_WAVES - 1) / V16_WAVES;

    for (int i = tid; i < V16_LDS_BYTES / 16; i += 64 * V16_WAVES) *(uint4*)(smem + 16 * i) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (FLASH) {
        for (int r = tid; r < VNKP; r += 64 * V16_WAVES)
            *(uint4*)((char*)Vsm + r * (VLD * 2) + 16 * ((VCHL - 1) ^ vswz(r))) = make_uint4((unsigned)f2bf(1.0f), 0, 0, 0);      // 1.0 in the build's 16-bit element
        __syncthreads();
    }
    if ((int)blockIdx.x >= p.items) return;
    // The items of this workgroup.  Plain walk: blockIdx.x + k gridDim.x.  XCD-aware walk (p.xcd_map): the 16 heads of an image are adjacent
    // 176-byte slices of the same 8448-byte QKV rows, i.e. a 128-byte line of Q, K or V belongs to two heads (a head's slice touches 2.4
    // lines on average: 1.75 x its bytes).  In the plain walk neighbouring heads run at the same time on DIFFERENT XCDs (workgroup i sits on
    // XCD i % 8), so every shared line passes the fabric twice; here image b goes to XCD b % 8 and that XCD's workgroups take its heads
    // side by side (32 CUs = two images at a time, 4.3 MB of rows against 4 MB of L2), so the second head finds the line in its own L2.
    // PMC at 128 images (profiles/r04_pmc_attention_traffic.json): 556 MB -> 386 MB past the L2s (1.50 x -> 1.04 x the algorithmic bytes), L2 hits 24 % -> 53 %.
    int wg_first = blockIdx.x, wg_step = gridDim.x, wg_items = p.items;
    const int xcd = blockIdx.x & 7;
    if (p.xcd_map) {
        const int images = p.items / p.heads;
        wg_first = blockIdx.x >> 3;                                     // position among the XCD's workgroups
        wg_step = ((int)gridDim.x - xcd + 7) >> 3;                      // workgroups on this XCD
        wg_items = ((images - xcd + 7) >> 3) * p.heads;                 // items of this XCD: images xcd, xcd + 8, ...
        if (wg_first >= wg_items) return;
    }
    const int nit = (wg_items - wg_first + wg_step - 1) / wg_step;
    auto item_of = [&](int k) {
        const int l = wg_first + k * wg_step;
        if (!p.xcd_map) return l;
        const int il = l / p.heads;
        return (8 * il + xcd) * p.heads + (l - il * p.heads);
    };
    const float L2E = 1.4426950408889634f;

    // LDS-DMA requests of one matrix image - or of two images that share their row stride (K and Q of an item: one set of lane offsets for
    // both).  Piece `wave + 16 j` is LDS chunks 64 piece .. + 63; chunk position q = 12 row + cpos takes source chunk cpos ^ swizzle(row) (the pad
    // chunk copies the 11th).  Lane offsets are formed incrementally - the next piece is 1024 chunks = 85 rows + 4 chunks further - as 32-bit
    // byte offsets on a wave-uniform base: ~13 VALU instructions per piece where the form of attn_vit16_kernel (a division by 12 and a 64-bit
    // address per piece and matrix) takes ~27, and this kernel is bound by instruction issue (the three stagings were a quarter of a
    // wave's VALU instructions per item).
    auto stage2 = [&](const bf16_t* baseA, bf16_t* dstA, const bf16_t* baseB, bf16_t* dstB, int ld, int k, bool skip_pad) {
        const int item = item_of(k);
        const int b = item / p.heads, h = item - b * p.heads;
        const size_t ioff = (size_t)b * n * ld + h * VHD;
        const char* gA = (const char*)(baseA + ioff);
        const char* gB = (const char*)(baseB + ioff);      // (only used where baseB != nullptr)
        const uint32_t ld2 = 2u * (uint32_t)ld;
        const int lane = fresh_lane();
        const int q = 64 * wave + lane;
        int row = q / VCHL, cpos = q - row * VCHL;
        uint32_t rowoff = (uint32_t)__umul24(row, ld2);     // row * ld2, carried along by addition
        auto request = [&](int piece, int r, uint32_t roff, int cp, bool live) {
            const int c0 = cp ^ vswz(r);
            const int c = min(c0, VCH - 1);
            const uint32_t off = roff + 16u * (uint32_t)c;
            // (skip_pad: lanes that would write a pad chunk - or, clamped, re-write the last position - stay out: EXEC-masked LDS-DMA)
            if (!skip_pad || (c0 < VCH && live)) {
                glds16v((const bf16_t*)(gA + off), (char*)dstA + piece * 1024);
                if (baseB) glds16v((const bf16_t*)(gB + off), (char*)dstB + piece * 1024);
            }
        };
        // the image's last piece (piece 48 = wave 0's fourth) runs past the image: it is taken out of the loop, its lanes beyond the image
        // rewrite the last chunk
        const bool has_last = wave == (npieces - 1) % V16_WAVES;
        const int full = has_last ? my_pieces - 1 : my_pieces;
        for (int j = 0; j < full; ++j) {
            request(wave + V16_WAVES * j, row, rowoff, cpos, true);
            cpos += 4;                                       // the next piece: 1024 chunks = 85 rows + 4 chunks further
            row += 85;
            rowoff += 85u * ld2;
            if (cpos >= VCHL) { cpos -= VCHL; row += 1; rowoff += ld2; }
        }
        if (has_last) {
            const bool live = row < n;
            request(npieces - 1, live ? row : n - 1, live ? rowoff : (uint32_t)(n - 1) * ld2, live ? cpos : VCHL - 1, live);
        }
    };
    auto stage_kq = [&](int k) {
        if (p.ldk == p.ldq) stage2(p.K, Ksm, p.Q, Qsm, p.ldk, k, false);
        else { stage2(p.K, Ksm, nullptr, nullptr, p.ldk, k, false); stage2(p.Q, Qsm, nullptr, nullptr, p.ldq, k, false); }
    };
    auto stage_v = [&](int k) { stage2(p.V, Vsm, nullptr, nullptr, p.ldv, k, FLASH); };
    // the end of a slot: this wave's LDS traffic retired, at most `leave_vm` of its youngest VM operations in flight, then the workgroup barrier
    auto slot_end = [&](int leave_vm) {
        __builtin_amdgcn_sched_barrier(0);
        if (leave_vm >= 0) wait_vm(leave_vm);                // (< 0: nothing of this slot's requests is needed next slot)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- S^T = K Q^T for query tile `wave`
    auto scores = [&](f32x4 (&s)[VNT]) {
        const int lane = fresh_lane(), li = lane & 15, g = lane >> 4;
        int koff[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) koff[ks] = li * (VLD * 2) + 16 * ((4 * ks + g) ^ vswz(li));
        bf16x8 qf[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {                    // q * scale rounded to half; the 12th chunk (cols 88..95) is zero
            uint4 v = make_uint4(0, 0, 0, 0);
            if (32 * ks + 8 * g < VHD) v = *(const uint4*)((const char*)Qsm + 16 * wave * (VLD * 2) + koff[ks]);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = pack2bf(lo_bf(w[i]) * p.scale, hi_bf(w[i]) * p.scale);
            qf[ks] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
        }
        constexpr int DEPTH = 1;                            // key tiles requested ahead of the one being multiplied (2: measured equal, below)
        bf16x8 fk[DEPTH + 1][3];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) fk[d][ks] = *(const bf16x8*)((const char*)Ksm + d * 16 * (VLD * 2) + koff[ks]);
#pragma unroll
        for (int kt = 0; kt < VNT; ++kt) {
            if (kt + DEPTH < VNT) {
#pragma unroll
                for (int ks = 0; ks < 3; ++ks)
                    fk[(kt + DEPTH) % (DEPTH + 1)][ks] = *(const bf16x8*)((const char*)Ksm + (kt + DEPTH) * 16 * (VLD * 2) + koff[ks]);
            }
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) s[kt] = seedmi_mfma_16x16x32(fk[kt % (DEPTH + 1)][ks], qf[ks], s[kt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- softmax over keys {16 kt + 4 g + r} (S rounded to half like the reference's matmul output), P packed to bf16 MFMA operands
    auto softmax = [&](f32x4 (&s)[VNT], bf16x8 (&pf)[VKK]) {
        const int g = fresh_lane() >> 4;
        float mx4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kt = 0; kt < VNT; ++kt) {
            if (kt == VNT - 1) {
                // keys 256 .. 271: only key 256 (g == 0, r == 0) exists.  The three other registers of this tile are never touched again:
                // their exponentials are 0 and adding 0 changes no sum (the lock-step kernel computes exp2(-inf) for them: the same bits)
                const float v0 = (g == 0) ? rbf(s[kt][0]) : -INFINITY;
                s[kt][0] = v0;
                mx4[kt & 3] = fmaxf(mx4[kt & 3], v0);
                continue;
            }
            const uint32_t w0 = pack2bf(s[kt][0], s[kt][1]), w1 = pack2bf(s[kt][2], s[kt][3]);
            const float v0 = lo_bf(w0), v1 = hi_bf(w0), v2 = lo_bf(w1), v3 = hi_bf(w1);
            s[kt][0] = v0; s[kt][1] = v1; s[kt][2] = v2; s[kt][3] = v3;
            mx4[kt & 3] = fmaxf(fmaxf(fmaxf(fmaxf(mx4[kt & 3], v0), v1), v2), v3);   // (two v_max3_f32 per key tile; max is exact in any order)
        }
        float mx = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nmx = -mx * L2E;
        if (FLASH) {
            const f32x2 l2 = {L2E, L2E}, nm2 = {nmx, nmx};
#pragma unroll
            for (int kk = 0; kk < VKK; ++kk) {
                uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kt = 2 * kk + u;
                    if (kt == VNT - 1) {
                        w[2 * u] = pack2bf(__builtin_amdgcn_exp2f(fmaf(s[kt][0], L2E, nmx)), 0.f);
                    } else if (kt < VNT) {
                        const f32x2 a0 = __builtin_elementwise_fma((f32x2){s[kt][0], s[kt][1]}, l2, nm2);
                        const f32x2 a1 = __builtin_elementwise_fma((f32x2){s[kt][2], s[kt][3]}, l2, nm2);
                        w[2 * u] = pack2bf(__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1]));
                        w[2 * u + 1] = pack2bf(__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1]));
                    }
                }
                pf[kk] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
            }
        } else {
            float sum = 0.f;                                 // (one chain in (kt, r) order, like the other kernels: bit-identical rows)
#pragma unroll
            for (int kt = 0; kt < VNT; ++kt) {
                if (kt == VNT - 1) {                         // (one element: see above)
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][0], L2E, nmx));
                    s[kt][0] = e;
                    sum += e;
                    continue;
                }
                const f32x2 a0 = __builtin_elementwise_fma((f32x2){s[kt][0], s[kt][1]}, (f32x2){L2E, L2E}, (f32x2){nmx, nmx});
                const f32x2 a1 = __builtin_elementwise_fma((f32x2){s[kt][2], s[kt][3]}, (f32x2){L2E, L2E}, (f32x2){nmx, nmx});
                const float arg[4] = {a0[0], a0[1], a1[0], a1[1]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(arg[r]);
                    s[kt][r] = e;
                    sum += e;
                }
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int kk = 0; kk < VKK; ++kk) {
                uint4 pw;
                if (2 * kk == VNT - 1) {                     // keys 256 (this lane group's first) .. 287
                    pw.x = pack2bf(s[2 * kk][0] * inv, 0.f);
                    pw.y = pw.z = pw.w = 0u;
                    pf[kk] = __builtin_bit_cast(bf16x8, pw);
                    continue;
                }
                pw.x = pack2bf(s[2 * kk][0] * inv, s[2 * kk][1] * inv);
                pw.y = pack2bf(s[2 * kk][2] * inv, s[2 * kk][3] * inv);
                pw.z = pw.w = 0u;                            // (keys 272..287 do not exist: P = 0)
                if (2 * kk + 1 < VNT) {
                    pw.z = pack2bf(s[2 * kk + 1][0] * inv, s[2 * kk + 1][1] * inv);
                    pw.w = pack2bf(s[2 * kk + 1][2] * inv, s[2 * kk + 1][3] * inv);
                }
                pf[kk] = __builtin_bit_cast(bf16x8, pw);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- row 256 (query tile 16): operands swapped, S[q = 4 g + r][key = 16 kt + li]; only q == 0 exists (lanes 0..15, register 0).
    //      The normalised row (FLASH: un-normalised) goes to the side row in LDS.
    auto side_scores = [&]() {
        const int lane = fresh_lane(), li = lane & 15, g = lane >> 4;
        int koff[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) koff[ks] = li * (VLD * 2) + 16 * ((4 * ks + g) ^ vswz(li));
        bf16x8 qf[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (32 * ks + 8 * g < VHD) v = *(const uint4*)((const char*)Qsm + 16 * 16 * (VLD * 2) + koff[ks]);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = pack2bf(lo_bf(w[i]) * p.scale, hi_bf(w[i]) * p.scale);
            qf[ks] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
        }
        float t[VNT];
        bf16x8 fk[2][3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) fk[0][ks] = *(const bf16x8*)((const char*)Ksm + koff[ks]);
#pragma unroll
        for (int kt = 0; kt < VNT; ++kt) {
            if (kt + 1 < VNT) {
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) fk[(kt + 1) & 1][ks] = *(const bf16x8*)((const char*)Ksm + (kt + 1) * 16 * (VLD * 2) + koff[ks]);
            }
            f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) a = seedmi_mfma_16x16x32(qf[ks], fk[kt & 1][ks], a);
            float v = rbf(a[0]);
            if (kt == VNT - 1) v = (li == 0) ? v : -INFINITY;   // keys 257..271 do not exist
            t[kt] = v;
            __builtin_amdgcn_sched_barrier(0);
        }
        float mx = t[0];
#pragma unroll
        for (int kt = 1; kt < VNT; ++kt) mx = fmaxf(mx, t[kt]);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float nmx = -mx * L2E;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < VNT; ++kt) {
            t[kt] = __builtin_amdgcn_exp2f(fmaf(t[kt], L2E, nmx));
            sum += t[kt];
        }
        float inv = 1.0f;
        if (!FLASH) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
            inv = __builtin_amdgcn_rcpf(sum);
        }
        if (g == 0) {
#pragma unroll
            for (int kt = 0; kt < VNT; ++kt) Psm[16 * kt + li] = f2bf(t[kt] * inv);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- O^T = V^T P^T (hardware transpose read of the row-major V image) and the output stores of one query tile.
    //      side: the P^T operand comes from the side row (every column the same row; only column li == 0 is stored) as query tile 16.
    auto pv = [&](const bf16x8 (&pf)[VKK], int k, bool side) {
        const int item = item_of(k);
        const int b = item / p.heads, h = item - b * p.heads;
        const int lane = fresh_lane(), li = lane & 15, g = lane >> 4;
        const bf16_t* vlane = Vsm + (4 * g + (li >> 2)) * VLD + 4 * (li & 1);
        const bf16_t* prow = Psm + 4 * g;
        auto ldv = [&](bf16x8 (&f)[VHT], int kk) {
#pragma unroll
            for (int nn = 0; nn < VHT; ++nn) {
                const bf16_t* vp = vlane + 32 * kk * VLD + 8 * ((2 * nn + ((li & 3) >> 1)) ^ vswz(4 * g));
                const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)vp);
                const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 16 * VLD));
                const uint2 lo = __builtin_bit_cast(uint2, a), hi = __builtin_bit_cast(uint2, c);
                f[nn] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            }
        };
        f32x4 o[VHT];
#pragma unroll
        for (int nn = 0; nn < VHT; ++nn) o[nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 fv0[VHT], fv1[VHT];
        ldv(fv0, 0);
#pragma unroll
        for (int kk = 0; kk < VKK; ++kk) {
            bf16x8 pr = pf[kk];
            if (side) {
                const uint2 plo = *(const uint2*)(prow + 32 * kk), phi = *(const uint2*)(prow + 32 * kk + 16);
                pr = __builtin_bit_cast(bf16x8, make_uint4(plo.x, plo.y, phi.x, phi.y));
            }
            // (hipcc sinks the reads of key block kk + 1 in between the MFMAs of kk - two or three MFMAs of distance instead of six.  A
            //  sched_barrier between the requests and the MFMAs, and the same for K fragments two key tiles ahead, restores the written order
            //  and measures 3 % SLOWER - profiles/r04_call22_staggered_attention_fences.log: the phase is not bound by that distance.)
            if (kk & 1) {
                if (kk + 1 < VKK) ldv(fv0, kk + 1);
#pragma unroll
                for (int nn = 0; nn < VHT; ++nn) o[nn] = seedmi_mfma_16x16x32(fv1[nn], pr, o[nn]);
            } else {
                if (kk + 1 < VKK) ldv(fv1, kk + 1);
#pragma unroll
                for (int nn = 0; nn < VHT; ++nn) o[nn] = seedmi_mfma_16x16x32(fv0[nn], pr, o[nn]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- stores
        const int sl = fresh_lane(), sg = sl >> 4;
        float inv = 1.0f;
        if (FLASH) {
            const float rs = __shfl(o[VHT - 1][0], (sl & 15) + 32, 64);      // O^T row 88 = the row sum of the query in column li
            inv = __builtin_amdgcn_rcpf(rs);
        }
        uint2 w[VHT];
#pragma unroll
        for (int nn = 0; nn < VHT; ++nn) {
            w[nn].x = pack2bf(o[nn][0] * inv, o[nn][1] * inv);
            w[nn].y = pack2bf(o[nn][2] * inv, o[nn][3] * inv);
        }
        const int qtile = side ? 16 : wave;
        bf16_t* op = p.O + ((size_t)b * n + 16 * qtile + (sl & 15)) * p.ldo + h * VHD;
        if (WIDE) {
#pragma unroll
            for (int nn = 0; nn < VHT; nn += 2) {
                const auto tx = __builtin_amdgcn_permlane16_swap(w[nn].x, w[nn + 1].x, false, false);
                const auto ty = __builtin_amdgcn_permlane16_swap(w[nn].y, w[nn + 1].y, false, false);
                const int c0 = 16 * (nn + (sg & 1)) + 8 * (sg >> 1);
                if (c0 + 8 <= VHD && !(side && (sl & 15) != 0)) *(uint4*)(op + c0) = make_uint4(tx[0], ty[0], tx[1], ty[1]);
            }
        } else {
            if (!(side && (sl & 15) != 0)) {
#pragma unroll
                for (int nn = 0; nn < VHT; ++nn) {
                    const int c0 = 16 * nn + 4 * sg;
                    if (c0 + 4 <= VHD) *(uint2*)(op + c0) = w[nn];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- row 256 of O, ONE 16-column tile `nn` of it (wave-uniform, run-time): the side row's PV dealt over six waves of group B instead of a
    //      whole second PV pass by one of them (9 MFMAs + 18 fragment reads per wave against 54 + 108 for one: the slot ends when its
    //      slowest wave does).  Same k-ordered accumulation chain per element as pv(.., side = true): the same bits.
    auto side_pv_part = [&](int k, int nn) {
        const int item = item_of(k);
        const int b = item / p.heads, h = item - b * p.heads;
        const int lane = fresh_lane(), li = lane & 15, g = lane >> 4;
        const bf16_t* vp0 = Vsm + (4 * g + (li >> 2)) * VLD + 4 * (li & 1) + 8 * ((2 * nn + ((li & 3) >> 1)) ^ vswz(4 * g));
        const bf16_t* prow = Psm + 4 * g;
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < VKK; ++kk) {
            const bf16_t* vp = vp0 + 32 * kk * VLD;
            const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)vp);
            const s16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vp + 16 * VLD));
            const uint2 lo = __builtin_bit_cast(uint2, a), hi = __builtin_bit_cast(uint2, c);
            const bf16x8 fv = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            const uint2 plo = *(const uint2*)(prow + 32 * kk), phi = *(const uint2*)(prow + 32 * kk + 16);
            const bf16x8 pr = __builtin_bit_cast(bf16x8, make_uint4(plo.x, plo.y, phi.x, phi.y));
            o = seedmi_mfma_16x16x32(fv, pr, o);
        }
        const int c0 = 16 * nn + 4 * g;                       // O^T rows 16 nn + 4 g + r of column li: only li == 0 (row 256) exists
        if (li == 0 && c0 + 4 <= VHD) {
            bf16_t* op = p.O + ((size_t)b * n + 256) * p.ldo + h * VHD + c0;
            *(uint2*)op = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int NSTORES = WIDE ? VHT / 2 : VHT;          // store instructions of one tile (each has active lanes: 16 wave + li < 257)
#ifdef SEEDMI_DEVTOOLS
    // phase clock stamps of workgroup 0 (tools/attn16s_phase_times.py): per wave and item 0 = slot start, 1 = QK^T done, 2 = barrier passed,
    // 3 = softmax (+ side scores) done, 4 = barrier passed, 5 = PV (+ side PV) done, 6 = barrier passed
    unsigned long long vt[9];
#define V16DUMP(k_) do { if (p.dbg && blockIdx.x == 0 && (k_) < 8) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if (fresh_lane() == 0) for (int q_ = 0; q_ < 7; ++q_) p.dbg[(wave * 8 + (k_)) * 9 + q_] = vt[q_]; } } while (0)
#else
#define V16DUMP(k_) do {} while (0)
#endif

    stage_kq(0);
    stage_v(0);
    slot_end(0);                                            // K(0), Q(0), V(0) landed everywhere

    if (wave < 8) {
        // ---------------- group A: slots 3k (QK^T), 3k+1 (softmax, side scores), 3k+2 (PV)
        for (int k = 0; k < nit; ++k) {
            f32x4 s[VNT];
            bf16x8 pf[VKK];
            V16STAMP(0);
            if (PRS) __builtin_amdgcn_s_setprio(PR_QK);
            scores(s);
            V16STAMP(1);
            slot_end(-1);                                   // (nothing requested in this slot)
            V16STAMP(2);
            if (PRS) __builtin_amdgcn_s_setprio(PR_SM);
            if (k >= 1) stage_v(k);                         // slot 3k+1: V(k-1) was last read in slot 3k
            softmax(s, pf);
            if (wave == (k & 7)) side_scores();
            V16STAMP(3);
            slot_end(0);                                    // V(k) landed
            V16STAMP(4);
            if (PRS) __builtin_amdgcn_s_setprio(PR_PV);
            if (k + 1 < nit) stage_kq(k + 1);               // slot 3k+2: K(k), Q(k) were last read in slot 3k+1
            pv(pf, k, false);
            V16STAMP(5);
            slot_end(p.store_wait ? NSTORES : 0);           // K(k+1), Q(k+1) landed; this tile's stores stay in flight
            V16STAMP(6);
            V16DUMP(k);
        }
    } else {
        // ---------------- group B: one slot behind - slots 3k+1 (QK^T), 3k+2 (softmax), 3k+3 (PV, side PV)
        slot_end(0);                                        // slot 0
        for (int k = 0; k < nit; ++k) {
            f32x4 s[VNT];
            bf16x8 pf[VKK];
            V16STAMP(0);
            if (PRS) __builtin_amdgcn_s_setprio(PR_QK);
            if (k >= 1) stage_v(k);                         // slot 3k+1
            scores(s);
            V16STAMP(1);
            slot_end(0);                                    // V(k) landed (and this wave's stores of PV(k-1), a slot old)
            V16STAMP(2);
            if (PRS) __builtin_amdgcn_s_setprio(PR_SM);
            if (k + 1 < nit) stage_kq(k + 1);               // slot 3k+2
            softmax(s, pf);
            V16STAMP(3);
            slot_end(0);                                    // K(k+1), Q(k+1) landed
            V16STAMP(4);
            if (PRS) __builtin_amdgcn_s_setprio(PR_PV);
            pv(pf, k, false);                               // slot 3k+3
            if (FLASH) {                                    // (the row sum sits in tile 5: one wave takes the whole side row)
                if (wave == 8 + ((k + 3) & 7)) pv(pf, k, true);
            } else if (wave < 8 + VHT) {
                side_pv_part(k, wave - 8);
            }
            V16STAMP(5);
            if (k + 1 < nit) slot_end(-1);                  // (the last slot needs no barrier: group A has left)
            V16STAMP(6);
            V16DUMP(k);
        }
    }
}

// 0 = off (attn_fullrow), 1 = 12-wave kernel; where n == 257: 2 = 16-wave kernel, 3 = + 16-byte output stores (default), 4 = + normalisation
// behind PV; 5 / 6 = 3 / 4 with the two halves of the workgroup one phase apart (attn_vit16s_kernel).  Measured at B = 128 (profiles/r03_call5_attention_modes.log): 159.2 / 159.1 / 152.2 / 137.5 us for 1 / 2 / 3 / 4; end to end
// 121.9 / - / 121.5 / 121.0 ms per 256 images.  4 moves a rounding point away from the reference's (the normalised probabilities are no
// longer what is rounded to half): its outputs sit ~0.8 bf16 ulp (rms) from the other kernels', at the same distance from fp32 - left
// selectable, not the default.  (An 8-wave form with two query tiles per wave on shared K / V fragments - half the LDS fragment reads per
// item - was written and measured: bit-identical rows, 162.0 us against 150.6 for mode 3 and 143.3 against 143.4 for mode 4
// (profiles/r03_call9_attention_8wave.log): two waves per SIMD hide less latency than the halved LDS traffic buys.  Removed.)
std::atomic<int> g_attn_vit{5};
std::atomic<int> g_attn_store_wait{1};
std::atomic<int> g_attn_xcd{1};
std::atomic<int> g_attn_small{1};      // "attn_small": 1 = launches with fewer items than half the CUs split every item's query tiles over several workgroups
                                       // (automatic factor), 2 / 4 / 8 / 16 = that factor, 0 = never (A/B); same bits
#undef V16STAMP
#undef V16DUMP
#ifdef SEEDMI_DEVTOOLS
unsigned long long* g_attn_dbg = nullptr;
#endif

}  // namespace
#ifdef SEEDMI_DEVTOOLS
// devtools: device buffer (>= 16 x 8 x 9 uint64) that workgroup 0 of the next ViT attention launches fills with phase clock stamps
// (12-wave kernel: [wave][item][6]; 16-wave kernel: [wave][item][9])
extern "C" int seedmi_attn_vit_timing(void* buf) { g_attn_dbg = (unsigned long long*)buf; return SEEDMI_OK; }
#endif

int seedmi_attn_vit_small(int v) { g_attn_small = v; return SEEDMI_OK; }
int seedmi_attn_vit_enabled() { return g_attn_vit; }
int seedmi_attn_vit_set(int v) { g_attn_vit = v; return SEEDMI_OK; }
int seedmi_attn_vit_store_wait(int v) { g_attn_store_wait = v; return SEEDMI_OK; }
int seedmi_attn_vit_xcd(int v) { g_attn_xcd = v; return SEEDMI_OK; }

// returns SEEDMI_OK after launching, or 1 if the shape is not handled by this kernel (caller falls back to attn_fullrow)
int seedmi_attention_vit_try(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                             int batch, int heads, int head_dim, int nq, int nk, float scale, int causal, int round_scores,
                             void* stream) {
    if (!g_attn_vit || head_dim != VHD || causal || nq != nk || nk > 16 * VNT || nk < 64) return 1;
    VitAttnParams p;
    p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (bf16_t*)O;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.n = nq; p.heads = heads; p.items = batch * heads; p.scale = scale;
    p.store_wait = g_attn_store_wait.load(std::memory_order_relaxed);
    p.xcd_map = 0;
    p.qsplit = 1;
#ifdef SEEDMI_DEVTOOLS
    p.dbg = g_attn_dbg;
#endif
    const int dev = seedmi_current_device();
    const int n_cu = seedmi_device_cus(dev);
    const int grid = p.items < n_cu ? p.items : n_cu;
    constexpr int lds = VLDS_BYTES;
    static bool attr_set_dev[SEEDMI_MAX_DEVICES] = {};
    bool& attr_set = attr_set_dev[dev];
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_vit_kernel<true, 257>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)attn_vit_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)attn_vit_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    // Small launches: fewer items than half the CUs (one image: 16 items; the reference scripts tokenize ONE image) -> the lock-step 16-wave
    // kernel with every item's query tiles split over `qsplit` workgroups (17.6 us per ViT block at one image, 16 CUs busy, before).
    const int small = g_attn_small.load(std::memory_order_relaxed);
    if (small && g_attn_vit >= 2 && nq == V16_N && round_scores && 2 * p.items <= n_cu) {
        int S = small > 1 ? small : (4 * p.items <= n_cu ? 4 : 2);
        S = (S >= 16) ? 16 : (S >= 8 ? 8 : (S >= 4 ? 4 : 2));
        while (S > 2 && p.items * S > n_cu) S >>= 1;
        p.qsplit = S;
        static bool attr16q_dev[SEEDMI_MAX_DEVICES] = {};
        if (!attr16q_dev[dev]) {
            (void)hipFuncSetAttribute((const void*)attn_vit16_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, V16_LDS_BYTES);
            attr16q_dev[dev] = true;
        }
        hipLaunchKernelGGL((attn_vit16_kernel<true, 1>), dim3(p.items * S), dim3(64 * V16_WAVES), V16_LDS_BYTES, (hipStream_t)stream, p);
        return seedmi_check_launch("attn_vit16 (split query tiles)");
    }
    if (g_attn_vit >= 2 && nq == V16_N && round_scores) {        // (unrounded scores: only tests ask for them; the 12-wave kernel serves those)
        static bool attr16_dev[SEEDMI_MAX_DEVICES] = {};
        if (!attr16_dev[dev]) {
            (void)hipFuncSetAttribute((const void*)attn_vit16_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, V16_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)attn_vit16_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, V16_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)attn_vit16_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, V16_LDS_BYTES);
            attr16_dev[dev] = true;
        }
        const dim3 blk16(64 * V16_WAVES);
        if (g_attn_vit >= 5) {
            // XCD-aware item walk on a full launch (one workgroup per CU, CUs a multiple of 8); measured from 16 images up: 20.1 vs 24.7 us at 16,
            // 30.7 vs 35.9 at 32, 52.7 vs 60.7 at 63 (profiles/r04_call26_attention_xcd_walk_small_batches.log)
            p.xcd_map = (g_attn_xcd.load(std::memory_order_relaxed) && grid == n_cu && n_cu % 8 == 0 && batch >= 16) ? 1 : 0;
            static bool attr16s_dev[SEEDMI_MAX_DEVICES] = {};
            if (!attr16s_dev[dev]) {
                (void)hipFuncSetAttribute((const void*)attn_vit16s_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, V16_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)attn_vit16s_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, V16_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)attn_vit16s_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, V16_LDS_BYTES);
                attr16s_dev[dev] = true;
            }
            if (g_attn_vit == 5) hipLaunchKernelGGL((attn_vit16s_kernel<1>), dim3(grid), blk16, V16_LDS_BYTES, (hipStream_t)stream, p);
            else if (g_attn_vit == 6) hipLaunchKernelGGL((attn_vit16s_kernel<3>), dim3(grid), blk16, V16_LDS_BYTES, (hipStream_t)stream, p);
            else hipLaunchKernelGGL((attn_vit16s_kernel<5>), dim3(grid), blk16, V16_LDS_BYTES, (hipStream_t)stream, p);   // 7: 5 without wave priorities (A/B)
            return seedmi_check_launch("attn_vit16s");
        }
        if (g_attn_vit == 2) hipLaunchKernelGGL((attn_vit16_kernel<true, 0>), dim3(grid), blk16, V16_LDS_BYTES, (hipStream_t)stream, p);
        else if (g_attn_vit == 3) hipLaunchKernelGGL((attn_vit16_kernel<true, 1>), dim3(grid), blk16, V16_LDS_BYTES, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((attn_vit16_kernel<true, 3>), dim3(grid), blk16, V16_LDS_BYTES, (hipStream_t)stream, p);
        return seedmi_check_launch("attn_vit16");
    }
    const dim3 blk(64 * VWAVES);
    if (round_scores && nq == 257)
        hipLaunchKernelGGL((attn_vit_kernel<true, 257>), dim3(grid), blk, lds, (hipStream_t)stream, p);
    else if (round_scores)
        hipLaunchKernelGGL((attn_vit_kernel<true, 0>), dim3(grid), blk, lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((attn_vit_kernel<false, 0>), dim3(grid), blk, lds, (hipStream_t)stream, p);
    return seedmi_check_launch("attn_vit");
}
