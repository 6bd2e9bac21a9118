// 8192-way VQ codebook nearest neighbour as a wavefront-level argmin reduce (VALU only, no MFMA).
//
// Replaces VectorQuantizer2.forward's distance + argmin (qformer_quantizer.py:94-98):
//     d = (sum(z**2, 1, keepdim) + sum(e**2, 1)) - 2 * (z @ e.T);  ids = argmin(d, 1)      [first index on ties]
// evaluated in the model dtype (bf16): every sub-expression is a materialised half tensor.  The kernel
// reproduces exactly those rounding points and fixes the one thing torch leaves unspecified — the order of the
// fp32 accumulation — to a sequential k = 0..31 chain (products of two bf16 values are exact in fp32, so the
// fmaf chain equals separately rounded mul+add).  oracle/seed_oracle.py::vq_distances_fixed_order and
// oracle/vq_oracle.c state the same arithmetic; ids are bit-identical to them by construction.
//
// Work split: a workgroup owns VQ_ROWS rows of z (staged as fp32 in LDS, read back as broadcasts) and sweeps the
// whole codebook; lane l of wave w visits codes {256*it + 64*w + l} in increasing order keeping a running
// (min, first index) per row in registers, then a 6-step cross-lane and a 4-way cross-wave reduce pick the
// winner.  The codebook (512 KiB) and its norms stay L2-resident; HBM traffic is z + ids only.
// The decode-side work the reference also does here (z_q gather, MSE loss, straight-through estimator,
// decode_task_layer: qformer_quantizer.py:99-114,305) is dead for encode_image and is not computed.
#include "common.h"
#include "seedmi_internal.h"

namespace {

constexpr int VQ_D = 32;
constexpr int VQ_ROWS = 8;
constexpr int VQ_SMALL_ROWS = 256;               // up to this many rows (8 images) the one-row, sixteen-wave shape of the sweep is launched

__global__ __launch_bounds__(256) void vq_code_sqnorm_kernel(const bf16_t* __restrict__ cb, float* __restrict__ ee, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < VQ_D; ++k) {
        const float e = bf2f(cb[(size_t)i * VQ_D + k]);
        acc = acc + rbf(e * e);                  // e**2 is a half tensor; the row sum accumulates in fp32
    }
    ee[i] = rbf(acc);                            // torch.sum(...) output in the model dtype
}

// torch.argmin's order on (distance, index): NaN before every number, then the smaller distance, then the smaller index
SEEDMI_DEVINL bool vq_before(float d2, int i2, float d, int i) {
    const bool n2 = d2 != d2, n1 = d != d;
    if (n2 || n1) return n2 && (!n1 || i2 < i);
    return d2 < d || (d2 == d && i2 < i);
}

// the sweep: zs holds the workgroup's ROWS rows of z (fp32 copies of half values), filled by the caller.  Two shapes of the same sweep:
// ROWS = 8 rows x 4 waves (batches: thousands of rows, the codebook re-streamed once per 8 rows) and ROWS = 1 row x 16 waves (ONE image is
// 32 rows: four 8-row workgroups swept 8192 codes x 8 rows each on four CUs, 105 us; thirty-two 1024-thread workgroups take ~a tenth).
// A lane still visits its codes in increasing order and every reduction picks by (NaN first, distance, index): the same winner.
template <int ROWS, int NW>
SEEDMI_DEVINL void vq_sweep(const float (&zs)[ROWS][VQ_D], const bf16_t* __restrict__ cb, const float* __restrict__ ee,
                            long long* __restrict__ out, int rows, int n_embed) {
    constexpr int VQ_ROWS = ROWS;                                    // (shadows the batch shape's constant inside the sweep)
    __shared__ float zz_s[VQ_ROWS];
    __shared__ float red_d[NW][VQ_ROWS];
    __shared__ int red_i[NW][VQ_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * VQ_ROWS;
    __syncthreads();
    if (tid < VQ_ROWS) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < VQ_D; ++k) acc = acc + rbf(zs[tid][k] * zs[tid][k]);
        zz_s[tid] = rbf(acc);
    }
    __syncthreads();

    float best_d[VQ_ROWS];
    int best_i[VQ_ROWS];
    // torch.argmin (qformer_quantizer.py:98) on non-finite rows: a NaN distance is the minimum (the FIRST NaN wins), and a row whose
    // distances are all +inf (fp16 overflow of |z|^2) returns index 0.  So the running minimum starts at (+inf, the lane's first code)
    // - not at an index no code has - and a NaN replaces any non-NaN minimum and is never replaced.
    const int n_first = wave * 64 + lane;
#pragma unroll
    for (int r = 0; r < VQ_ROWS; ++r) { best_d[r] = INFINITY; best_i[r] = n_first < n_embed ? n_first : 0x7fffffff; }

    for (int n = wave * 64 + lane; n < n_embed; n += 64 * NW) {
        float e[VQ_D];
        const uint4* ep = (const uint4*)(cb + (size_t)n * VQ_D);
#pragma unroll
        for (int c = 0; c < VQ_D / 8; ++c) {
            const uint4 u = ep[c];
            e[8 * c + 0] = lo_bf(u.x); e[8 * c + 1] = hi_bf(u.x);
            e[8 * c + 2] = lo_bf(u.y); e[8 * c + 3] = hi_bf(u.y);
            e[8 * c + 4] = lo_bf(u.z); e[8 * c + 5] = hi_bf(u.z);
            e[8 * c + 6] = lo_bf(u.w); e[8 * c + 7] = hi_bf(u.w);
        }
        const float een = ee[n];
#pragma unroll
        for (int r = 0; r < VQ_ROWS; ++r) {
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < VQ_D; ++k) dot = __builtin_fmaf(zs[r][k], e[k], dot);
            const float s = rbf(zz_s[r] + een);                  // [rows,1] + [n_embed] broadcast add -> half
            const float d = rbf(s - 2.0f * rbf(dot));            // einsum output -> half, *2 exact, subtract -> half
            if (d < best_d[r] || (d != d && best_d[r] == best_d[r])) { best_d[r] = d; best_i[r] = n; }   // strict <: first index wins inside a lane
            __builtin_amdgcn_sched_barrier(0);                   // one row at a time: keeps z broadcasts out of the live set
        }
    }
    // cross-lane: lexicographic (d, index) minimum
#pragma unroll
    for (int r = 0; r < VQ_ROWS; ++r) {
        float d = best_d[r];
        int i = best_i[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float d2 = __shfl_xor(d, o, 64);
            const int i2 = __shfl_xor(i, o, 64);
            if (vq_before(d2, i2, d, i)) { d = d2; i = i2; }
        }
        if (lane == 0) { red_d[wave][r] = d; red_i[wave][r] = i; }
    }
    __syncthreads();
    if (tid < VQ_ROWS && row0 + tid < rows) {
        float d = red_d[0][tid];
        int i = red_i[0][tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float d2 = red_d[w][tid];
            const int i2 = red_i[w][tid];
            if (vq_before(d2, i2, d, i)) { d = d2; i = i2; }
        }
        out[row0 + tid] = (long long)i;
    }
}

template <int ROWS, int NW>
__global__ __launch_bounds__(64 * NW) void vq_argmin_kernel(const bf16_t* __restrict__ z, int ldz, const bf16_t* __restrict__ cb,
                                                            const float* __restrict__ ee, long long* __restrict__ out,
                                                            int rows, int n_embed) {
    __shared__ __attribute__((aligned(16))) float zs[ROWS][VQ_D];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * ROWS;
    for (int idx = tid; idx < ROWS * VQ_D; idx += 64 * NW) {
        const int r = idx / VQ_D, k = idx - r * VQ_D;
        const int row = min(row0 + r, rows - 1);
        zs[r][k] = bf2f(z[(size_t)row * ldz + k]);
    }
    vq_sweep<ROWS, NW>(zs, cb, ee, out, rows, n_embed);
}

// encode_task_layer's second Linear (qformer_quantizer.py:219-223: Linear(768, 32) after the Tanh) fused in front of the sweep
// (SURVEY 8a a13 -> a14): the workgroup's VQ_ROWS rows of the tanh output are staged in LDS, thread (r, k) = (tid / 32, tid % 32) forms
// z[r][k] = half(sum_j t[r][j] w[k][j] + b[k]) with a sequential fp32 chain, z goes straight into the sweep's LDS copy and, when asked
// for (taps), to memory.  No [rows, 32] round trip, no 32-column launch of a 128-column GEMM tile.
constexpr int VQ_HMAX = 1024;                    // widest hidden size staged (Q-Former: 768)
template <int ROWS, int NW>
__global__ __launch_bounds__(64 * NW) void vq_head_argmin_kernel(const bf16_t* __restrict__ t, int ldt, int hidden, const bf16_t* __restrict__ w1,
                                                             int ldw, const bf16_t* __restrict__ b1, const bf16_t* __restrict__ cb,
                                                             const float* __restrict__ ee, long long* __restrict__ out,
                                                             bf16_t* __restrict__ z_out, int ldz, int rows, int n_embed) {
    __shared__ __attribute__((aligned(16))) float zs[ROWS][VQ_D];
    __shared__ __attribute__((aligned(16))) bf16_t ts[ROWS][VQ_HMAX];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * ROWS;
    const int chunks = hidden >> 3;
    for (int idx = tid; idx < ROWS * chunks; idx += 64 * NW) {
        const int r = idx / chunks, c = idx - r * chunks;
        const int row = min(row0 + r, rows - 1);
        *(uint4*)&ts[r][8 * c] = *(const uint4*)(t + (size_t)row * ldt + 8 * c);
    }
    __syncthreads();
    if (tid < ROWS * VQ_D) {                                        // (z[r][k]: one sequential fp32 chain per element, whatever the workgroup's shape)
        const int r = tid >> 5, k = tid & 31;
        const bf16_t* wr = w1 + (size_t)k * ldw;
        float acc = 0.f;
        for (int c = 0; c < chunks; ++c) {
            const uint4 wv = *(const uint4*)(wr + 8 * c), tv = *(const uint4*)&ts[r][8 * c];
            const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w}, tw[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc = __builtin_fmaf(lo_bf(tw[i]), lo_bf(ww[i]), acc);
                acc = __builtin_fmaf(hi_bf(tw[i]), hi_bf(ww[i]), acc);
            }
        }
        const bf16_t zh = f2bf(acc + (b1 ? bf2f(b1[k]) : 0.f));
        zs[r][k] = bf2f(zh);
        if (z_out && row0 + r < rows) z_out[(size_t)(row0 + r) * ldz + k] = zh;
    }
    vq_sweep<ROWS, NW>(zs, cb, ee, out, rows, n_embed);
}

}  // namespace

extern "C" int seedmi_vq_code_sqnorm(const void* codebook, void* ee_f32, int n_embed, int dim, void* stream) {
    if (dim != VQ_D || n_embed <= 0) {
        seedmi_set_error("seedmi_vq_code_sqnorm: dim=%d (only %d supported) n_embed=%d", dim, VQ_D, n_embed);
        return SEEDMI_E_SHAPE;
    }
    hipLaunchKernelGGL(vq_code_sqnorm_kernel, dim3((n_embed + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)codebook, (float*)ee_f32, n_embed);
    return seedmi_check_launch("vq_code_sqnorm");
}

extern "C" int seedmi_vq_argmin_bf16(const void* z, int ldz, const void* codebook, const void* ee_f32, void* ids_i64,
                                     int rows, int n_embed, int dim, void* stream) {
    if (dim != VQ_D || rows <= 0 || n_embed <= 0) {
        seedmi_set_error("seedmi_vq_argmin_bf16: rows=%d n_embed=%d dim=%d (dim must be %d)", rows, n_embed, dim, VQ_D);
        return SEEDMI_E_SHAPE;
    }
    if (((uintptr_t)codebook & 15)) {
        seedmi_set_error("seedmi_vq_argmin_bf16: codebook must be 16-byte aligned");
        return SEEDMI_E_ALIGN;
    }
    if (rows <= VQ_SMALL_ROWS)
        hipLaunchKernelGGL((vq_argmin_kernel<1, 16>), dim3(rows), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)z, ldz, (const bf16_t*)codebook,
                           (const float*)ee_f32, (long long*)ids_i64, rows, n_embed);
    else
        hipLaunchKernelGGL((vq_argmin_kernel<VQ_ROWS, 4>), dim3((rows + VQ_ROWS - 1) / VQ_ROWS), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)z, ldz, (const bf16_t*)codebook, (const float*)ee_f32, (long long*)ids_i64, rows, n_embed);
    return seedmi_check_launch("vq_argmin");
}

extern "C" int seedmi_vq_head_argmin_bf16(const void* t, int ldt, int hidden, const void* w, int ldw, const void* bias, const void* codebook,
                                          const void* ee_f32, void* ids_i64, void* z_out, int ldz, int rows, int n_embed, int dim,
                                          void* stream) {
    if (dim != VQ_D || rows <= 0 || n_embed <= 0 || hidden <= 0 || hidden > VQ_HMAX || (hidden % 8) || (ldt % 8) || (ldw % 8)) {
        seedmi_set_error("seedmi_vq_head_argmin_bf16: rows=%d n_embed=%d dim=%d (must be %d) hidden=%d (multiple of 8, <= %d)", rows, n_embed, dim,
                         VQ_D, hidden, VQ_HMAX);
        return SEEDMI_E_SHAPE;
    }
    if ((((uintptr_t)codebook | (uintptr_t)t | (uintptr_t)w) & 15)) {
        seedmi_set_error("seedmi_vq_head_argmin_bf16: t, w and the codebook must be 16-byte aligned");
        return SEEDMI_E_ALIGN;
    }
    if (rows <= VQ_SMALL_ROWS)
        hipLaunchKernelGGL((vq_head_argmin_kernel<1, 16>), dim3(rows), dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)t, ldt, hidden, (const bf16_t*)w,
                           ldw, (const bf16_t*)bias, (const bf16_t*)codebook, (const float*)ee_f32, (long long*)ids_i64, (bf16_t*)z_out, ldz, rows, n_embed);
    else
        hipLaunchKernelGGL((vq_head_argmin_kernel<VQ_ROWS, 4>), dim3((rows + VQ_ROWS - 1) / VQ_ROWS), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)t,
                           ldt, hidden, (const bf16_t*)w, ldw, (const bf16_t*)bias, (const bf16_t*)codebook, (const float*)ee_f32, (long long*)ids_i64,
                           (bf16_t*)z_out, ldz, rows, n_embed);
    return seedmi_check_launch("vq_head_argmin");
}
