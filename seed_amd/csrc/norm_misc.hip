// Row-wise normalisations and small data-movement kernels (HBM-bound; 16-byte accesses, one wave per row).
//
//  seedmi_layernorm_bf16  nn.LayerNorm with fp32 statistics: eva_vit.py:199-202 (eps 1e-6, autocast fp32),
//                         blip2.py:179-184 (ln_vision, eps 1e-5), qformer_causual.py:96,254,336 (eps 1e-12)
//  seedmi_rmsnorm_bf16    LlamaRMSNorm.forward, llama_xformer.py:105-113
//  seedmi_im2col_patch    PatchEmbed's Conv2d(k=s=patch) unfolded to GEMM rows in (c,kh,kw) order, eva_vit.py:222-229
//  seedmi_fill_rows       cls rows (eva_vit.py:373-377) / query-token expand (qformer_quantizer.py:293)
//  seedmi_rope_kv_append  apply_rotary_pos_emb + KV-cache append, llama_xformer.py:160-168,234-239
//  seedmi_embed_rows      nn.Embedding gather, llama_xformer.py:544
#include "common.h"
#include "seedmi_internal.h"

namespace {

constexpr int NORM_WAVES = 4;

// one wave per row; row held in registers as MAXC chunks of 8 bf16 per lane
// PACK: write the row in the decode GEMM's fragment-major activation layout (16-row tile x 32-k step = 1 KiB block in MFMA
// lane order) so the consumer's operand loads are contiguous: chunk c of row m goes to
// (((m>>4)*(cols>>5) + (c>>2))*64 + (c&3)*16 + (m&15)) * 8 elements.
template <int MAXC, bool RMS, bool PACK>
__global__ __launch_bounds__(64 * NORM_WAVES) void norm_kernel(const bf16_t* __restrict__ x, int ldx,
                                                               const bf16_t* __restrict__ gamma,
                                                               const bf16_t* __restrict__ beta, float eps,
                                                               bf16_t* __restrict__ out, int ldo, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * NORM_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunks = cols >> 3;
    const bf16_t* xr = x + (size_t)row * ldx;
    float v[MAXC][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (c < nchunks) u = *(const uint4*)(xr + 8 * c);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[i][2 * j] = lo_bf(w[j]);
            v[i][2 * j + 1] = hi_bf(w[j]);
            sum += RMS ? (v[i][2 * j] * v[i][2 * j] + v[i][2 * j + 1] * v[i][2 * j + 1]) : (v[i][2 * j] + v[i][2 * j + 1]);
        }
    }
    sum = wave_sum(sum);
    const float inv_n = 1.0f / (float)cols;
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(sum * inv_n + eps);
    } else {
        mean = sum * inv_n;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
            }
        }
        sq = wave_sum(sq);
        rstd = rsqrtf(sq * inv_n + eps);
    }
    bf16_t* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
            const uint4 gu = *(const uint4*)(gamma + 8 * c);
            const uint32_t gw[4] = {gu.x, gu.y, gu.z, gu.w};
            uint32_t bw[4] = {0, 0, 0, 0};
            if (!RMS && beta) { const uint4 bu = *(const uint4*)(beta + 8 * c); bw[0] = bu.x; bw[1] = bu.y; bw[2] = bu.z; bw[3] = bu.w; }
            uint32_t ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a, b2;
                if (RMS) {
                    // x * rsqrt(var) rounded to the weight dtype BEFORE the multiply by weight (llama_xformer.py:107-113)
                    a = lo_bf(gw[j]) * rbf(v[i][2 * j] * rstd);
                    b2 = hi_bf(gw[j]) * rbf(v[i][2 * j + 1] * rstd);
                } else {
                    a = (v[i][2 * j] - mean) * rstd * lo_bf(gw[j]) + lo_bf(bw[j]);
                    b2 = (v[i][2 * j + 1] - mean) * rstd * hi_bf(gw[j]) + hi_bf(bw[j]);
                }
                ow[j] = pack2bf(a, b2);
            }
            if (PACK) {
                const size_t off = ((size_t)((row >> 4) * (cols >> 5) + (c >> 2)) * 64 + (c & 3) * 16 + (row & 15)) * 8;
                *(uint4*)(out + off) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            } else {
                *(uint4*)(orow + 8 * c) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
        }
    }
}

// LayerNorm statistics only (the normalisation itself is folded into the consuming GEMM, seedmi_gemm_bf16_ext): one wave per row,
// two-pass variance like norm_kernel
template <int MAXC>
__global__ __launch_bounds__(64 * NORM_WAVES) void row_stats_kernel(const bf16_t* __restrict__ x, int ldx, float eps,
                                                                    float2* __restrict__ out, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * NORM_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunks = cols >> 3;
    const bf16_t* xr = x + (size_t)row * ldx;
    float v[MAXC][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (c < nchunks) u = *(const uint4*)(xr + 8 * c);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[i][2 * j] = lo_bf(w[j]);
            v[i][2 * j + 1] = hi_bf(w[j]);
            sum += v[i][2 * j] + v[i][2 * j + 1];
        }
    }
    sum = wave_sum(sum);
    const float inv_n = 1.0f / (float)cols;
    const float mean = sum * inv_n;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
        }
    }
    sq = wave_sum(sq);
    if (lane == 0) out[row] = make_float2(mean, rsqrtf(sq * inv_n + eps));
}

// (sum, sum of squares) partials per 64-column span, span-major planes of ld rows -> (mean, rstd); one thread per row (every plane is
// read coalesced), spans summed in index order
__global__ void stats_finalize_kernel(const float2* __restrict__ part, int spans, int ld, int rows, float inv_n, float eps,
                                      float2* __restrict__ out) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const float2* pr = part + row;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < spans; ++i) { const float2 t = pr[(size_t)i * ld]; s1 += t.x; s2 += t.y; }
    out[row] = seedmi_ln_finish(s1, s2, inv_n, eps);
}

template <bool RMS, bool PACK = false>
int launch_norm(const void* x, int ldx, const void* gamma, const void* beta, float eps, void* out, int ldo, int rows,
                int cols, hipStream_t s) {
    const int grid = (rows + NORM_WAVES - 1) / NORM_WAVES;
    const int chunks = cols / 8;
#define SEEDMI_NORM_CASE(MAXC_)                                                                                   \
    if (chunks <= 64 * MAXC_) {                                                                                   \
        hipLaunchKernelGGL((norm_kernel<MAXC_, RMS, PACK>), dim3(grid), dim3(64 * NORM_WAVES), 0, s, (const bf16_t*)x, ldx, \
                           (const bf16_t*)gamma, (const bf16_t*)beta, eps, (bf16_t*)out, ldo, rows, cols);        \
        return seedmi_check_launch("norm");                                                                       \
    }
    SEEDMI_NORM_CASE(1)
    SEEDMI_NORM_CASE(2)
    SEEDMI_NORM_CASE(3)
    SEEDMI_NORM_CASE(4)
    SEEDMI_NORM_CASE(8)
    SEEDMI_NORM_CASE(16)
#undef SEEDMI_NORM_CASE
    seedmi_set_error("norm: cols=%d too large (max 8192)", cols);
    return SEEDMI_E_SHAPE;
}

// ---- im2col for a stride==kernel patch conv; one thread per (row, 8-wide k chunk)
template <typename TIn>
__global__ void im2col_patch_kernel(const TIn* __restrict__ img, bf16_t* __restrict__ col, int batch, int chans, int hw,
                                    int patch, int kpad) {
    const int grid_w = hw / patch;
    const int ppi = grid_w * grid_w;
    const int kchunks = kpad >> 3;
    const long long total = (long long)batch * ppi * kchunks;
    const int kreal = chans * patch * patch;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int kc = (int)(idx % kchunks);
        const long long m = idx / kchunks;
        const int pidx = (int)(m % ppi);
        const int b = (int)(m / ppi);
        const int py = pidx / grid_w, px = pidx - py * grid_w;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kc * 8 + j;
            float val = 0.f;
            if (k < kreal) {
                const int c = k / (patch * patch);
                const int rem = k - c * patch * patch;
                const int kh = rem / patch, kw = rem - kh * patch;
                const size_t off = (((size_t)b * chans + c) * hw + (py * patch + kh)) * hw + (px * patch + kw);
                if constexpr (sizeof(TIn) == 2) val = bf2f(((const bf16_t*)img)[off]);
                else val = ((const float*)img)[off];
            }
            v[j] = val;
        }
        uint4 o;
        o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
        *(uint4*)(col + m * kpad + kc * 8) = o;
    }
}

// dst[(g*group_rows + r0 + i) * ld + :] = src[i * lds + :]  for g < ngroups, i < nsrc  (cols multiple of 8)
__global__ void fill_rows_kernel(bf16_t* __restrict__ dst, int ld, int group_rows, int r0, int ngroups,
                                 const bf16_t* __restrict__ src, int lds_, int nsrc, int cols) {
    const int chunks = cols >> 3;
    const long long total = (long long)ngroups * nsrc * chunks;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % chunks);
        const long long t = idx / chunks;
        const int i = (int)(t % nsrc);
        const long long gi = t / nsrc;
        *(uint4*)(dst + ((size_t)gi * group_rows + r0 + i) * ld + 8 * c) = *(const uint4*)(src + (size_t)i * lds_ + 8 * c);
    }
}

__global__ void embed_rows_kernel(const long long* __restrict__ ids, const bf16_t* __restrict__ table, int ldt,
                                  bf16_t* __restrict__ out, int ldo, int n, int cols, int vocab) {
    const int chunks = cols >> 3;
    const long long total = (long long)n * chunks;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % chunks);
        const long long r = idx / chunks;
        long long id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        *(uint4*)(out + r * ldo + 8 * c) = *(const uint4*)(table + id * ldt + 8 * c);
    }
}

// The first launch of a decode step: embedding rows -> the residual stream (row-major) AND its fragment-major copy (the layout of
// pack_rows_kernel below: what the first folded-RMSNorm GEMM reads), and the split-K GEMMs' flag words cleared - one launch where the
// step used to issue three (embed_rows, a memset node, pack_rows).
__global__ void embed_rows_decode_kernel(const long long* __restrict__ ids, const bf16_t* __restrict__ table, int ldt,
                                         bf16_t* __restrict__ out, int ldo, bf16_t* __restrict__ out_packed, int n, int cols, int vocab,
                                         unsigned* __restrict__ zero_words, int n_zero) {
    const int chunks = cols >> 3;
    const long long total = (long long)n * chunks;
    const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    for (long long i = t0; i < n_zero; i += step) zero_words[i] = 0u;
    for (long long idx = t0; idx < total; idx += step) {
        const int c = (int)(idx % chunks);
        const int m = (int)(idx / chunks);
        long long id = ids[m];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const uint4 v = *(const uint4*)(table + id * ldt + 8 * c);
        *(uint4*)(out + (size_t)m * ldo + 8 * c) = v;
        *(uint4*)(out_packed + ((size_t)((m >> 4) * (cols >> 5) + (c >> 2)) * 64 + (c & 3) * 16 + (m & 15)) * 8) = v;
    }
}

// qkv [B*T, 3*H*hd] (q|k|v) -> q_out [B*T, H*hd] rotated; K/V caches [B][H][Tmax][hd] appended at pos_base+t.
// RoPE arithmetic in the activation dtype with the reference's rounding points:
//   half(half(x*cos) + half(rotate_half(x)*sin)), cos/sin tables pre-rounded to half  (llama_xformer.py:147-168)
__global__ void rope_kv_append_kernel(const bf16_t* __restrict__ qkv, int ldqkv, const long long* __restrict__ pos_ids,
                                      const bf16_t* __restrict__ cos_t, const bf16_t* __restrict__ sin_t,
                                      bf16_t* __restrict__ q_out, int ldq, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                      int B, int T, int H, int hd, int tmax, int past_len_arg,
                                      const int* __restrict__ past_dev, int max_pos) {
    // past_dev: the cache length lives in device memory so that a captured hipGraph of the decode step can be replayed
    // for every position (a kernel argument would be frozen at capture time)
    // a device-resident length can run past what the launch was sized for (a graph replayed too often): the append then stays
    // on the last cache row instead of leaving the allocation
    const int past_len = past_dev ? min(*past_dev, tmax - T) : past_len_arg;
    const int half = hd >> 1;
    const int pairs = half >> 2;                       // 4 element pairs (x[i], x[i+half]) per thread
    const long long total = (long long)B * T * H * pairs;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int pc = (int)(idx % pairs);
        long long r = idx / pairs;
        const int h = (int)(r % H);
        r /= H;
        const int t = (int)(r % T);
        const int b = (int)(r / T);
        const long long row = (long long)b * T + t;
        long long pos = pos_ids ? pos_ids[row] : (long long)(past_len + t);
        pos = pos < 0 ? 0 : (pos >= max_pos ? max_pos - 1 : pos);       // rows of the cos/sin tables ([max_pos, hd]); the reference device-asserts
        const int i0 = pc * 4;
        const bf16_t* cs = cos_t + pos * hd;
        const bf16_t* sn = sin_t + pos * hd;
        const bf16_t* qp = qkv + row * ldqkv + h * hd;
        const bf16_t* kp = qp + H * hd;
        const bf16_t* vp = kp + H * hd;
        bf16_t* qo = q_out + row * ldq + h * hd;
        const size_t coff = (((size_t)b * H + h) * tmax + (past_len + t)) * hd;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j;
            const float c1 = bf2f(cs[i]), s1 = bf2f(sn[i]);
            const float c2 = bf2f(cs[i + half]), s2 = bf2f(sn[i + half]);
            {
                const float x1 = bf2f(qp[i]), x2 = bf2f(qp[i + half]);
                qo[i] = f2bf(rbf(x1 * c1) + rbf(-x2 * s1));
                qo[i + half] = f2bf(rbf(x2 * c2) + rbf(x1 * s2));
            }
            {
                const float x1 = bf2f(kp[i]), x2 = bf2f(kp[i + half]);
                kc[coff + i] = f2bf(rbf(x1 * c1) + rbf(-x2 * s1));
                kc[coff + i + half] = f2bf(rbf(x2 * c2) + rbf(x1 * s2));
            }
            vc[coff + i] = vp[i];
            vc[coff + i + half] = vp[i + half];
        }
    }
}

__global__ void add_i32_kernel(int* p, int delta) { *p += delta; }
__global__ void add_i32_vec_kernel(int* __restrict__ p, const int* __restrict__ inc, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += inc[i];
}

int grid_for(long long total, int block) {
    long long g = (total + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int seedmi_layernorm_bf16(const void* x, int ldx, const void* gamma, const void* beta, float eps, void* out,
                                     int ldo, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0 || (cols % 8) || (ldx % 8) || (ldo % 8)) {
        seedmi_set_error("seedmi_layernorm_bf16: rows=%d cols=%d ldx=%d ldo=%d (cols and strides must be multiples of 8)", rows, cols, ldx, ldo);
        return SEEDMI_E_SHAPE;
    }
    return launch_norm<false>(x, ldx, gamma, beta, eps, out, ldo, rows, cols, (hipStream_t)stream);
}

extern "C" int seedmi_layernorm_stats_bf16(const void* x, int ldx, int rows, int cols, float eps, void* stats, void* stream) {
    if (!x || !stats || rows <= 0 || cols <= 0 || (cols % 8) || (ldx % 8) || cols > 8192) {
        seedmi_set_error("seedmi_layernorm_stats_bf16: rows=%d cols=%d ldx=%d", rows, cols, ldx);
        return SEEDMI_E_SHAPE;
    }
    const int grid = (rows + NORM_WAVES - 1) / NORM_WAVES;
    const int chunks = cols / 8;
    hipStream_t s = (hipStream_t)stream;
#define SEEDMI_STATS_CASE(MAXC_)                                                                                          \
    if (chunks <= 64 * MAXC_) {                                                                                           \
        hipLaunchKernelGGL((row_stats_kernel<MAXC_>), dim3(grid), dim3(64 * NORM_WAVES), 0, s, (const bf16_t*)x, ldx, eps,  \
                           (float2*)stats, rows, cols);                                                                   \
        return seedmi_check_launch("row_stats");                                                                          \
    }
    SEEDMI_STATS_CASE(1)
    SEEDMI_STATS_CASE(2)
    SEEDMI_STATS_CASE(3)
    SEEDMI_STATS_CASE(4)
    SEEDMI_STATS_CASE(8)
    SEEDMI_STATS_CASE(16)
#undef SEEDMI_STATS_CASE
    return SEEDMI_E_SHAPE;
}

extern "C" int seedmi_layernorm_stats_finalize(const void* partial, int spans, int stats_ld, int rows, int cols, float eps, void* stats,
                                               void* stream) {
    if (!partial || !stats || spans <= 0 || stats_ld < rows || rows <= 0 || cols <= 0) {
        seedmi_set_error("seedmi_layernorm_stats_finalize: bad arguments");
        return SEEDMI_E_SHAPE;
    }
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float2*)partial, spans,
                       stats_ld, rows, 1.0f / (float)cols, eps, (float2*)stats);
    return seedmi_check_launch("stats_finalize");
}

extern "C" int seedmi_rmsnorm_bf16(const void* x, int ldx, const void* gamma, float eps, void* out, int ldo, int rows,
                                   int cols, void* stream) {
    if (rows <= 0 || cols <= 0 || (cols % 8) || (ldx % 8) || (ldo % 8)) {
        seedmi_set_error("seedmi_rmsnorm_bf16: rows=%d cols=%d ldx=%d ldo=%d (cols and strides must be multiples of 8)", rows, cols, ldx, ldo);
        return SEEDMI_E_SHAPE;
    }
    return launch_norm<true>(x, ldx, gamma, nullptr, eps, out, ldo, rows, cols, (hipStream_t)stream);
}

extern "C" int seedmi_rmsnorm_packed_bf16(const void* x, int ldx, const void* gamma, float eps, void* out_packed, int rows,
                                          int cols, void* stream) {
    if (rows <= 0 || cols <= 0 || (cols % 32) || (ldx % 8)) {
        seedmi_set_error("seedmi_rmsnorm_packed_bf16: rows=%d cols=%d ldx=%d (cols multiple of 32)", rows, cols, ldx);
        return SEEDMI_E_SHAPE;
    }
    return launch_norm<true, true>(x, ldx, gamma, nullptr, eps, out_packed, cols, rows, cols, (hipStream_t)stream);
}

// rows [rows, cols] (row stride ldx) -> the decode GEMM's fragment-major activation layout, no arithmetic (the folded-RMSNorm
// decode chain feeds UN-normalised rows to its GEMMs): 16-byte chunk c of row m goes to
// (((m>>4)*(cols>>5) + (c>>2))*64 + (c&3)*16 + (m&15)) * 8 elements.
static __global__ void pack_rows_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ out, int rows, int cols) {
    const int nchunks = cols >> 3;
    const long long total = (long long)rows * nchunks;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / nchunks), c = (int)(idx - (long long)m * nchunks);
        const uint4 v = *(const uint4*)(x + (size_t)m * ldx + 8 * c);
        *(uint4*)(out + ((size_t)((m >> 4) * (cols >> 5) + (c >> 2)) * 64 + (c & 3) * 16 + (m & 15)) * 8) = v;
    }
}

extern "C" int seedmi_pack_activations_bf16(const void* x, int ldx, void* out_packed, int rows, int cols, void* stream) {
    if (rows <= 0 || cols <= 0 || (cols % 32) || (ldx % 8) || (((uintptr_t)x | (uintptr_t)out_packed) & 15)) {
        seedmi_set_error("seedmi_pack_activations_bf16: rows=%d cols=%d ldx=%d (cols multiple of 32, 16-byte aligned)", rows, cols, ldx);
        return SEEDMI_E_SHAPE;
    }
    const long long total = (long long)rows * (cols / 8);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                       (bf16_t*)out_packed, rows, cols);
    return seedmi_check_launch("pack_activations");
}

extern "C" int seedmi_im2col_patch(const void* img, int img_is_fp32, void* col, int batch, int chans, int hw, int patch,
                                   int kpad, void* stream) {
    if (batch <= 0 || hw % patch || kpad % 8 || kpad < chans * patch * patch) {
        seedmi_set_error("seedmi_im2col_patch: bad shape batch=%d hw=%d patch=%d kpad=%d", batch, hw, patch, kpad);
        return SEEDMI_E_SHAPE;
    }
    const long long total = (long long)batch * (hw / patch) * (hw / patch) * (kpad / 8);
    const int grid = grid_for(total, 256);
    if (img_is_fp32)
        hipLaunchKernelGGL(im2col_patch_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)img,
                           (bf16_t*)col, batch, chans, hw, patch, kpad);
    else
        hipLaunchKernelGGL(im2col_patch_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)img,
                           (bf16_t*)col, batch, chans, hw, patch, kpad);
    return seedmi_check_launch("im2col_patch");
}

extern "C" int seedmi_fill_rows(void* dst, int ld, int group_rows, int r0, int ngroups, const void* src, int lds_,
                                int nsrc, int cols, void* stream) {
    if (ngroups <= 0 || nsrc <= 0 || cols % 8 || ld % 8 || lds_ % 8) {
        seedmi_set_error("seedmi_fill_rows: bad shape");
        return SEEDMI_E_SHAPE;
    }
    const long long total = (long long)ngroups * nsrc * (cols / 8);
    hipLaunchKernelGGL(fill_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dst, ld,
                       group_rows, r0, ngroups, (const bf16_t*)src, lds_, nsrc, cols);
    return seedmi_check_launch("fill_rows");
}

extern "C" int seedmi_add_i32(void* counter_dev, int delta, void* stream) {
    hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)counter_dev, delta);
    return seedmi_check_launch("add_i32");
}

extern "C" int seedmi_add_i32_vec(void* dst_i32, const void* inc_i32, int n, void* stream) {
    if (!dst_i32 || !inc_i32 || n <= 0) {
        seedmi_set_error("seedmi_add_i32_vec: bad arguments");
        return SEEDMI_E_SHAPE;
    }
    hipLaunchKernelGGL(add_i32_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int*)dst_i32, (const int*)inc_i32, n);
    return seedmi_check_launch("add_i32_vec");
}

extern "C" int seedmi_embed_rows(const void* ids_i64, const void* table, int ldt, void* out, int ldo, int n, int cols,
                                 int vocab, void* stream) {
    if (n <= 0 || cols % 8 || ldt % 8 || ldo % 8) {
        seedmi_set_error("seedmi_embed_rows: bad shape");
        return SEEDMI_E_SHAPE;
    }
    const long long total = (long long)n * (cols / 8);
    hipLaunchKernelGGL(embed_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)ids_i64, (const bf16_t*)table, ldt, (bf16_t*)out, ldo, n, cols, vocab);
    return seedmi_check_launch("embed_rows");
}

int seedmi_embed_rows_decode(const void* ids_i64, const void* table, int ldt, void* out, int ldo, void* out_packed, int n, int cols,
                             int vocab, void* zero_words, int n_zero, void* stream) {
    if (n <= 0 || cols % 32 || ldt % 8 || ldo % 8 || n_zero < 0 || (n_zero > 0 && !zero_words)) {
        seedmi_set_error("seedmi_embed_rows_decode: bad shape");
        return SEEDMI_E_SHAPE;
    }
    const long long total = (long long)n * (cols / 8);
    hipLaunchKernelGGL(embed_rows_decode_kernel, dim3(grid_for(total > n_zero ? total : n_zero, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)ids_i64, (const bf16_t*)table, ldt, (bf16_t*)out, ldo, (bf16_t*)out_packed, n, cols, vocab,
                       (unsigned*)zero_words, n_zero);
    return seedmi_check_launch("embed_rows_decode");
}

extern "C" int seedmi_rope_kv_append(const void* qkv, int ldqkv, const void* pos_ids_i64, const void* cos_t,
                                     const void* sin_t, void* q_out, int ldq, void* k_cache, void* v_cache, int B, int T,
                                     int H, int hd, int tmax, int past_len, const void* past_len_dev, int max_pos, void* stream) {
    if (B <= 0 || T <= 0 || H <= 0 || (hd % 8) || past_len + T > tmax || max_pos <= 0) {
        seedmi_set_error("seedmi_rope_kv_append: bad shape B=%d T=%d H=%d hd=%d past=%d tmax=%d", B, T, H, hd, past_len, tmax);
        return SEEDMI_E_SHAPE;
    }
    const long long total = (long long)B * T * H * (hd / 8);
    hipLaunchKernelGGL(rope_kv_append_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qkv, ldqkv, (const long long*)pos_ids_i64, (const bf16_t*)cos_t, (const bf16_t*)sin_t,
                       (bf16_t*)q_out, ldq, (bf16_t*)k_cache, (bf16_t*)v_cache, B, T, H, hd, tmax, past_len,
                       (const int*)past_len_dev, max_pos);
    return seedmi_check_launch("rope_kv_append");
}
