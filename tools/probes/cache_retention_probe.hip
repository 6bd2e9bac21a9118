// Probe (tools/probes): do lines stay in the XCDs' L2s / in the Infinity Cache (MALL) ACROSS kernel boundaries?  A buffer is read by the same
// grid (block b always reads slice b, so a line returns to the XCD that cached it) in back-to-back launches; a launch that finds its lines in
// L2 / MALL is faster than the first touch.  Plain (temporal) and non-temporal loads, sizes below the 32 MiB of L2, below the 256 MiB of MALL,
// and above both; a 1 GiB sweep in between = the cold case.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/cache_retention_probe.bin tools/probes/cache_retention_probe.hip && tools/probes/cache_retention_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(512) void read_kernel(const u32x4_t* __restrict__ p, size_t n16, unsigned* __restrict__ out) {
    const size_t per_block = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per_block, b1 = b0 + per_block < n16 ? b0 + per_block : n16;
    unsigned acc = 0;
    for (size_t i = b0 + threadIdx.x; i < b1; i += 512 * 4) {
        u32x4_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t j = i + (size_t)u * 512 < b1 ? i + (size_t)u * 512 : b1 - 1;
            v[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9e3779b9u) out[0] = acc;
}
int main() {
    const size_t big = (size_t)1 << 30;
    char *buf, *thrash; unsigned* out;
    hipMalloc(&buf, big); hipMalloc(&thrash, big); hipMalloc(&out, 64);
    hipMemset(buf, 1, big); hipMemset(thrash, 2, big);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](bool nt, const char* b, size_t bytes) {
        hipEventRecord(e0);
        if (nt) hipLaunchKernelGGL(read_kernel<true>, dim3(256), dim3(512), 0, 0, (const u32x4_t*)b, bytes / 16, out);
        else hipLaunchKernelGGL(read_kernel<false>, dim3(256), dim3(512), 0, 0, (const u32x4_t*)b, bytes / 16, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
    };
    for (int nt = 0; nt < 2; ++nt)
        for (size_t mb : {4, 16, 28, 64, 128, 200, 512}) {
            const size_t bytes = mb << 20;
            std::vector<float> cold, warm;
            for (int rep = 0; rep < 5; ++rep) {
                run(true, thrash, big);                                  // evict: 1 GiB through every cache
                cold.push_back(run(nt, buf, bytes));                     // first touch
                for (int k = 0; k < 3; ++k) warm.push_back(run(nt, buf, bytes));   // the same lines again, new launches
            }
            std::sort(cold.begin(), cold.end()); std::sort(warm.begin(), warm.end());
            printf("%s %4zu MiB: first touch %7.1f us (%5.2f TB/s)   repeated launch %7.1f us (%5.2f TB/s)\n", nt ? "non-temporal" : "temporal    ", mb,
                   cold[2], bytes / cold[2] / 1e6, warm[warm.size() / 2], bytes / warm[warm.size() / 2] / 1e6);
        }
    return 0;
}
