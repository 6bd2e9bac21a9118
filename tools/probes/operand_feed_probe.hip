// Probe (tools/probes): how many bytes per second can ONE CU pull from its XCD's L2 into LDS (buffer_load ... lds, 16 bytes per lane) and into
// registers (buffer_load_dwordx4), by number of issuing waves and requests in flight per wave?  The K loops of the tile GEMMs are bound by
// their operand feed (LABNOTES rounds 1-5: ~43-50 GB/s per CU in the 256x256 kernel and in the 64x64 small-M kernel alike); this measures the
// ceiling of that feed on an L2-resident source, with every CU streaming (one workgroup per CU), against HBM-resident sources.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/operand_feed_probe.bin tools/probes/operand_feed_probe.hip && tools/probes/operand_feed_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// MODE 0: LDS-DMA 16 B / lane; MODE 1: loads to VGPRs (xor-folded into a sink); DEPTH requests in flight per wave
template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void feed(const char* src, size_t span_bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    // each workgroup walks its own span (span_bytes per workgroup, L2-resident when small), wave w takes 1 KiB pieces w, w + nw, ...
    const char* base = src + (size_t)blockIdx.x * span_bytes;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const unsigned pieces = (unsigned)(span_bytes >> 10);
    unsigned acc = 0;
    unsigned piece = wave;
    char* dst = sm + wave * (DEPTH * 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned off = (piece % pieces) * 1024u;
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + d * 1024), 16, lane * 16, off, 0, 0);
            } else {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, off, 0);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
            piece += nw;
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 0) acc = ((unsigned*)sm)[threadIdx.x];
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int MODE, int DEPTH>
double run(const char* src, size_t span, int waves, int iters, unsigned* sink, int blocks) {
    hipFuncSetAttribute((const void*)feed<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = MODE == 0 ? (size_t)waves * DEPTH * 1024 : 1024;
    hipLaunchKernelGGL((feed<MODE, DEPTH>), dim3(blocks), dim3(waves * 64), lds, 0, src, span, iters, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((feed<MODE, DEPTH>), dim3(blocks), dim3(waves * 64), lds, 0, src, span, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * waves * iters * DEPTH * 1024.0;
    return bytes / (ms * 1e-3) / 1e9 / blocks;               // GB/s per CU
}
int main() {
    const int blocks = 256;
    const size_t total = (size_t)4 << 30;
    char* src; unsigned* sink;
    hipMalloc(&src, total); hipMalloc(&sink, 64);
    hipMemset(src, 1, total);
    printf("GB/s per CU, 256 workgroups (one per CU); span = bytes each workgroup cycles through\n");
    for (size_t span : {(size_t)64 << 10, (size_t)16 << 20}) {          // 64 KiB per workgroup (L2-resident: 2 MiB per XCD) | 16 MiB (HBM stream)
        const int iters = span <= ((size_t)64 << 10) ? 400 : 200;
        printf("span %zu KiB\n", span >> 10);
        for (int waves : {1, 2, 4, 8, 16}) {
            printf("  waves %2d: LDS-DMA depth 4: %6.1f  depth 8: %6.1f   to-VGPR depth 4: %6.1f  depth 8: %6.1f\n", waves,
                   run<0, 4>(src, span, waves, iters, sink, blocks), run<0, 8>(src, span, waves, iters, sink, blocks),
                   run<1, 4>(src, span, waves, iters, sink, blocks), run<1, 8>(src, span, waves, iters, sink, blocks));
        }
    }
    return 0;
}
