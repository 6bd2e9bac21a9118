// Probe (tools/probes): does `buffer_load_dwordx4 ... lds` (__builtin_amdgcn_raw_ptr_buffer_load_lds, 16 bytes per lane) on gfx950 (a) place lane l's
// 16 bytes at M0-base + 16 l like global_load_lds_dwordx4 does, and (b) accept LDS destinations above 64 KiB (M0 wider than 16 bits)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/buffer_lds_probe.bin tools/probes/buffer_lds_probe.hip && tools/probes/buffer_lds_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(const unsigned* src, unsigned* out, int lds_off) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4096; i += 256) ((unsigned*)(sm + lds_off))[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (lane ^ 5) * 16;                       // a lane permutation on the SOURCE side
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sm + lds_off + wave * 1024), 16, voff, wave * 4096, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = ((unsigned*)(sm + lds_off))[i];
}
int main() {
    std::vector<unsigned> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = i;
    unsigned *src, *out;
    hipMalloc(&src, 8192 * 4); hipMalloc(&out, 1024 * 4);
    hipMemcpy(src, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lds_off : {0, 32768, 65536, 100 * 1024, 140 * 1024}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), lds_off + 16384, 0, src, out, lds_off);
        std::vector<unsigned> r(1024);
        hipError_t e = hipMemcpy(r.data(), out, 1024 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 4; ++w)
            for (int l = 0; l < 64; ++l)
                for (int d = 0; d < 4; ++d) {
                    const unsigned want = w * 1024 + (l ^ 5) * 4 + d;         // source dword index: (soffset 4096 w + voffset) / 4 + d
                    if (r[w * 256 + l * 4 + d] != want) ++bad;
                }
        printf("lds_off %6d: %s (%d of 1024 dwords differ; first %u %u %u %u) %s\n", lds_off, bad ? "MISMATCH" : "ok", bad, r[0], r[1], r[2], r[3],
               hipGetErrorString(e));
    }
    return 0;
}
