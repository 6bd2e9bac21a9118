// Probe (tools/probes): do two waves of one SIMD overlap DIFFERENT kinds of work?  A 512-thread workgroup puts waves w and w + 4 on the same SIMD.
// Waves 0..3 run role A, waves 4..7 role B, each role alone and both together; if the units overlap, "together" takes max(A, B), if the SIMD
// issues one kind at a time it takes A + B.  Roles: MFMA (4 independent v_mfma_f32_16x16x32_bf16 chains), VALU (16 independent v_fma_f32),
// TRANS (8 independent v_exp_f32), LDS (8 ds_read_b128 per wait from a conflict-free image).  One workgroup per CU, 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/issue_overlap_probe.bin tools/probes/issue_overlap_probe.hip && tools/probes/issue_overlap_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
enum { NONE = 0, MFMA = 1, VALU = 2, TRANS = 3, LDS = 4 };

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, const char* smem) {
    float keep = 0.f;
    const unsigned t = threadIdx.x * 2654435761u;
    if (ROLE == MFMA) {
        bf16x8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + ((t >> i) & 0x3ff)); b[i] = (short)(0x3c80 + ((t >> (i + 7)) & 0x3ff)); }
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {                      // 4 x 16 cycles of the matrix pipe per iteration
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, c3, 0, 0, 0);
        }
        keep = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (ROLE == VALU) {
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = 1.0f + 1e-3f * (float)((t >> j) & 7);
        const float m = 0.999f, c = 1e-4f;
        for (int i = 0; i < iters; ++i) {                      // 16 x 4 cycles of the VALU per iteration
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(m), "v"(c));
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) keep += x[j];
    } else if (ROLE == TRANS) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = -0.5f - 1e-3f * (float)((t >> j) & 7);
        for (int i = 0; i < iters; ++i) {                      // 8 quarter-rate instructions per iteration (2^x of a value in (-1, 0) stays in (0.5, 1))
#pragma unroll
            for (int j = 0; j < 8; ++j) { asm volatile("v_exp_f32 %0, %0" : "+v"(x[j])); }
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = -x[j];           // (8 cheap VALU ops: keeps the argument in range)
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) keep += x[j];
    } else if (ROLE == LDS) {
        const char* base = smem + 16 * (threadIdx.x & 63) + 1024 * (threadIdx.x >> 6);
        u32x4 v[8];
        unsigned acc = 0;
        for (int i = 0; i < iters; ++i) {                      // 8 x ds_read_b128 (1 KiB per wave each) per iteration
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const volatile u32x4*)(base + 8192 * j);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j].x;
        }
        keep = (float)acc;
    }
    return keep;
}

// LDS read rate against the number of reads a wave keeps in flight: NRD x ds_read_b128 (or, TR, ds_read_b64_tr_b16) per s_waitcnt, WAVES waves per CU
template <int NRD, bool TR>
__global__ __launch_bounds__(1024) void lds_depth_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((unsigned*)smem)[i] = i;
    __syncthreads();
    const char* base = smem + (TR ? 8 : 16) * (threadIdx.x & 63) + 1024 * ((threadIdx.x >> 6) & 3);
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        u32x4 v[NRD];
#pragma unroll
        for (int j = 0; j < NRD; ++j) {
            if (TR) {
                typedef __attribute__((ext_vector_type(4))) short s16x4;
                const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4096 * j));
                v[j].x = (unsigned)a[0] | ((unsigned)a[1] << 16);
            } else {
                asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(v[j]) : "v"((unsigned)(size_t)(base + 4096 * j) & 0xffffu));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NRD; ++j) acc ^= v[j].x;
    }
    if (acc == 0x12345u) out[threadIdx.x] = (float)acc;
}
template <int NRD, bool TR>
static float lds_rate(float* out, int waves, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)lds_depth_kernel<NRD, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((lds_depth_kernel<NRD, TR>), dim3(256), dim3(64 * waves), 65536, 0, out, iters);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((lds_depth_kernel<NRD, TR>), dim3(256), dim3(64 * waves), 65536, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double bytes = 256.0 * waves * (double)iters * NRD * (TR ? 512.0 : 1024.0);
    return (float)(bytes / (best * 1e-3) / 1e12);
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void probe_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) ((unsigned*)smem)[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float keep = 0.f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    if (wave < 4) keep = run_role<RA>(iters, smem);
    else keep = run_role<RB>(iters, smem);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(keep));
    if (keep == 123.456f) out[threadIdx.x] = keep;
    // shader-clock cycles of wave 0 (role A) and wave 4 (role B, same SIMD) of workgroup 0: independent of the clock the box settles at
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) ((unsigned long long*)(out + 512))[threadIdx.x >> 8] = t1 - t0;
}

// FOUR waves per SIMD (a 1024-thread workgroup: waves w, w + 4, w + 8, w + 12 share a SIMD): two of role A (waves 0..7), two of role B (waves 8..15),
// at s_setprio PA / PB - the situation of the staggered attention kernel's slots (two softmax waves next to two MFMA-phase waves per SIMD)
template <int RA, int RB, int PA, int PB>
__global__ __launch_bounds__(1024) void probe4_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 1024) ((unsigned*)smem)[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float keep = 0.f;
    unsigned long long t0, t1;
    if (wave < 8) __builtin_amdgcn_s_setprio(PA); else __builtin_amdgcn_s_setprio(PB);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    if (wave < 8) keep = run_role<RA>(iters, smem);
    else keep = run_role<RB>(iters, smem);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(keep));
    if (keep == 123.456f) out[threadIdx.x] = keep;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ((unsigned long long*)(out + 512))[wave] = t1 - t0;
}
template <int RA, int RB, int PA, int PB>
static void run4(float* out, int iters, const char* name) {
    hipFuncSetAttribute((const void*)probe4_kernel<RA, RB, PA, PB>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe4_kernel<RA, RB, PA, PB>), dim3(256), dim3(1024), 65536, 0, out, iters);
    hipDeviceSynchronize();
    unsigned long long cyc[16];
    hipMemcpy(cyc, out + 512, 128, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int w = 0; w < 8; ++w) { a = cyc[w] > a ? cyc[w] : a; b = cyc[8 + w] > b ? cyc[8 + w] : b; }
    printf("  4 waves/SIMD %-34s slowest A %.1f  slowest B %.1f ticks/iteration\n", name, a / iters, b / iters);
    fflush(stdout);
}
static double g_cyc[2];
template <int RA, int RB>
static float time_ms(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)probe_kernel<RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe_kernel<RA, RB>), dim3(256), dim3(512), 65536, 0, out, iters);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe_kernel<RA, RB>), dim3(256), dim3(512), 65536, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    unsigned long long cyc[2];
    hipMemcpy(cyc, out + 512, 16, hipMemcpyDeviceToHost);
    g_cyc[0] = (double)cyc[0]; g_cyc[1] = (double)cyc[1];
    return best;
}

int main() {
    float* out; hipMalloc(&out, 8192);
    const int iters = 20000;
    const float m = time_ms<MFMA, NONE>(out, iters), v = time_ms<NONE, VALU>(out, iters), tr = time_ms<NONE, TRANS>(out, iters), l = time_ms<NONE, LDS>(out, iters);
    printf("alone (ms, %d iterations, one wave per SIMD):  MFMA %.3f   VALU %.3f   TRANS %.3f   LDS %.3f\n", iters, m, v, tr, l);
    printf("MFMA + MFMA  (two waves per SIMD) %.3f   (2 x alone = %.3f)\n", time_ms<MFMA, MFMA>(out, iters), 2 * m);
    printf("VALU + VALU  %.3f   (2 x alone = %.3f)\n", time_ms<VALU, VALU>(out, iters), 2 * v);
    const float mv = time_ms<MFMA, VALU>(out, iters), mt = time_ms<MFMA, TRANS>(out, iters), ml = time_ms<MFMA, LDS>(out, iters);
    const float vl = time_ms<VALU, LDS>(out, iters), vt = time_ms<VALU, TRANS>(out, iters), ll = time_ms<LDS, LDS>(out, iters);
    {   // the same in shader-clock cycles per iteration (wave 0 = role A, wave 4 = role B on the same SIMD)
        auto cyc = [&](auto fn, const char* name) { fn(); printf("  cycles/iteration %-14s A %.1f  B %.1f\n", name, g_cyc[0] / iters, g_cyc[1] / iters); };
        cyc([&] { time_ms<MFMA, NONE>(out, iters); }, "MFMA | -");
        cyc([&] { time_ms<NONE, VALU>(out, iters); }, "- | VALU");
        cyc([&] { time_ms<NONE, TRANS>(out, iters); }, "- | TRANS");
        cyc([&] { time_ms<MFMA, MFMA>(out, iters); }, "MFMA | MFMA");
        cyc([&] { time_ms<VALU, VALU>(out, iters); }, "VALU | VALU");
        cyc([&] { time_ms<MFMA, VALU>(out, iters); }, "MFMA | VALU");
        cyc([&] { time_ms<MFMA, TRANS>(out, iters); }, "MFMA | TRANS");
        cyc([&] { time_ms<VALU, TRANS>(out, iters); }, "VALU | TRANS");
        fflush(stdout);
    }
    run4<MFMA, NONE, 0, 0>(out, iters, "2 MFMA | -");
    run4<NONE, VALU, 0, 0>(out, iters, "- | 2 VALU");
    run4<NONE, TRANS, 0, 0>(out, iters, "- | 2 TRANS");
    run4<MFMA, VALU, 0, 0>(out, iters, "2 MFMA | 2 VALU, no priorities");
    run4<MFMA, VALU, 2, 0>(out, iters, "2 MFMA (prio 2) | 2 VALU (0)");
    run4<MFMA, VALU, 0, 2>(out, iters, "2 MFMA (0) | 2 VALU (prio 2)");
    run4<VALU, MFMA, 0, 0>(out, iters, "2 VALU | 2 MFMA, no priorities");
    run4<MFMA, TRANS, 0, 0>(out, iters, "2 MFMA | 2 TRANS, no priorities");
    run4<MFMA, TRANS, 0, 2>(out, iters, "2 MFMA (0) | 2 TRANS (prio 2)");
    run4<MFMA, TRANS, 2, 0>(out, iters, "2 MFMA (prio 2) | 2 TRANS (0)");
    printf("MFMA + VALU  %.3f   (max %.3f, sum %.3f)\n", mv, m > v ? m : v, m + v);
    printf("MFMA + TRANS %.3f   (max %.3f, sum %.3f)\n", mt, m > tr ? m : tr, m + tr);
    printf("MFMA + LDS   %.3f   (max %.3f, sum %.3f)\n", ml, m > l ? m : l, m + l);
    printf("VALU + LDS   %.3f   (max %.3f, sum %.3f)\n", vl, v > l ? v : l, v + l);
    printf("VALU + TRANS %.3f   (max %.3f, sum %.3f)\n", vt, v > tr ? v : tr, v + tr);
    printf("LDS + LDS    %.3f   (2 x alone = %.3f; all 8 waves of 256 CUs: %.1f TB/s of ds_read_b128)\n", ll, 2 * l,
           256.0 * 8 * iters * 8 * 1024.0 / (ll * 1e-3) / 1e12);
    fflush(stdout);
    printf("LDS read rate, TB/s over 256 CUs (256 B/clk/CU at 2.1 GHz = 137.6); reads per s_waitcnt x waves per CU:\n");
    fflush(stdout);
    for (int waves = 4; waves <= 16; waves *= 2) {
        printf("  ds_read_b128       %2d waves: x4 %.1f", waves, lds_rate<4, false>(out, waves, 8000)); fflush(stdout);
        printf("  x8 %.1f", lds_rate<8, false>(out, waves, 8000)); fflush(stdout);
        printf("  x12 %.1f", lds_rate<12, false>(out, waves, 8000)); fflush(stdout);
        printf("  x15 %.1f\n", lds_rate<15, false>(out, waves, 8000)); fflush(stdout);
        printf("  ds_read_b64_tr_b16 %2d waves: x4 %.1f", waves, lds_rate<4, true>(out, waves, 8000)); fflush(stdout);
        printf("  x8 %.1f", lds_rate<8, true>(out, waves, 8000)); fflush(stdout);
        printf("  x12 %.1f", lds_rate<12, true>(out, waves, 8000)); fflush(stdout);
        printf("  x15 %.1f\n", lds_rate<15, true>(out, waves, 8000)); fflush(stdout);
    }
    return 0;
}
