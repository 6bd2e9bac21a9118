// Next-token selection on the device: greedy argmax or temperature / top-p (nucleus) sampling, one launch per decode step, token
// written where the next step and the caller read it — so a captured decode graph needs no host round trip per token.
//
// Replaces, for the scripts' generation_config (do_sample, top_p 0.5, temperature 1.0, scripts/seed_llama_inference_8B.py:81-87),
// the logits-processor + multinomial part of transformers.GenerationMixin (third-party, transformers == 4.30.2):
//   TemperatureLogitsWarper  scores / temperature
//   TopPLogitsWarper         sort ascending, softmax, cumsum; drop tokens whose cumulative probability <= 1 - top_p (keep >= 1)
//   multinomial(softmax(filtered))
// In exact arithmetic a token survives TopP iff the probability mass of the tokens ranked before it (descending) is < top_p.
// The rank order is made total here: descending probability, ties by ascending token id (torch.sort leaves ties unspecified).
// The draw is an inverse-CDF lookup over the kept tokens in that order with a caller-provided uniform u in [0, 1), which makes
// the step reproducible and testable (torch.multinomial's internal RNG stream is not).
//
// One workgroup per row; the row's weights exp(l - max) live in registers (<= 48 per thread).  The nucleus boundary and the
// drawn token are found by two binary searches on the float bit pattern of the weight (monotonic for positive floats), each step
// a block-wide masked sum — no sort.  Ties at a boundary are resolved by token id with a block-wide prefix count.
#include "common.h"
#include "seedmi_internal.h"

namespace {

constexpr int ST = 1024;          // threads per row
constexpr int SW = ST / 64;       // waves

struct BlockRed {
    float f[SW];
    int i[SW];
};

SEEDMI_DEVINL float block_sum(float v, float* sh, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SW; ++w) t += sh[w];
    return t;
}

SEEDMI_DEVINL int block_sum_i(int v, int* sh, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < SW; ++w) t += sh[w];
    return t;
}

template <int EPT>
__global__ __launch_bounds__(ST) void sample_kernel(const bf16_t* __restrict__ logits, int ldl, int vocab, float inv_temp,
                                                    float top_p, const float* __restrict__ u, const int* __restrict__ step_dev,
                                                    int step_off, int batch, long long* __restrict__ tok_out,
                                                    long long* __restrict__ hist, int hist_ld, int n_steps) {
    __shared__ BlockRed red;
    __shared__ int scan[SW];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int base = tid * EPT;
    const bf16_t* row = logits + (size_t)b * ldl;
    // step indexes the uniforms [n_steps, batch] and the history columns: a graph replayed more often than it was sized for keeps
    // drawing from the last row and stops recording instead of leaving either buffer
    const int step_raw = (step_dev ? *step_dev : 0) + step_off;
    const bool step_ok = step_raw >= 0 && (n_steps <= 0 || step_raw < n_steps);
    const int step = step_raw < 0 ? 0 : ((n_steps > 0 && step_raw >= n_steps) ? n_steps - 1 : step_raw);

    // ---- scaled logits, row max with first-index tie-break (torch.argmax / greedy search)
    float p[EPT];
    float lmax = -INFINITY;
    int amax = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int idx = base + e;
        const float v = idx < vocab ? bf2f(row[idx]) * inv_temp : -INFINITY;
        p[e] = v;
        if (v > lmax) { lmax = v; amax = idx; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(lmax, o, 64);
        const int oi = __shfl_xor(amax, o, 64);
        if (ov > lmax || (ov == lmax && oi < amax)) { lmax = ov; amax = oi; }
    }
    if ((tid & 63) == 0) { red.f[tid >> 6] = lmax; red.i[tid >> 6] = amax; }
    __syncthreads();
    float gmax = red.f[0];
    int gidx = red.i[0];
#pragma unroll
    for (int w = 1; w < SW; ++w)
        if (red.f[w] > gmax || (red.f[w] == gmax && red.i[w] < gidx)) { gmax = red.f[w]; gidx = red.i[w]; }

    int choice = gidx;
    if (u != nullptr && top_p > 0.f) {
        // ---- weights and their total
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            p[e] = (base + e < vocab) ? expf(p[e] - gmax) : 0.f;
            part += p[e];
        }
        const float Z = block_sum(part, red.f, tid);
        auto mass_gt = [&](unsigned t) {
            float m = 0.f;
#pragma unroll
            for (int e = 0; e < EPT; ++e) m += (__float_as_uint(p[e]) > t) ? p[e] : 0.f;
            return block_sum(m, red.f, tid);
        };
        auto count_eq = [&](unsigned t) {
            int c = 0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) c += (__float_as_uint(p[e]) == t) ? 1 : 0;
            return c;
        };
        const unsigned ONE = 0x3f800000u;                                   // the largest weight is exp(0) = 1
        // ---- nucleus boundary: smallest bit pattern tau with mass(p > tau) < top_p * Z
        const float target = top_p * Z;
        unsigned tau = 0;                                                    // top_p >= 1: every positive weight is kept
        float m_gt_tau = Z;
        int keep_ties = 0x7fffffff;
        if (top_p < 1.f) {
            unsigned lo = 0, hi = ONE;                                       // mass_gt(hi) = 0 < target <= mass_gt(lo)
            while (hi - lo > 1) {
                const unsigned mid = lo + ((hi - lo) >> 1);
                if (mass_gt(mid) < target) hi = mid; else lo = mid;
            }
            tau = hi;
            m_gt_tau = mass_gt(tau);
            const float tv = __uint_as_float(tau);
            const int n_ties = block_sum_i(count_eq(tau), red.i, tid);
            // the j-th tie (by token id) has mass m_gt_tau + j * tv in front of it and is kept while that is < target
            int k = (int)ceilf((target - m_gt_tau) / tv);
            keep_ties = k < 1 ? 1 : (k > n_ties ? n_ties : k);
        }
        const float tauv = __uint_as_float(tau);
        const float kept = (top_p < 1.f) ? m_gt_tau + keep_ties * tauv : Z;
        // ---- inverse CDF over the kept tokens (descending weight, ties by id): smallest t2 >= tau with mass(p > t2) <= r
        const float r = u[(size_t)step * batch + b] * kept;
        unsigned lo = (top_p < 1.f) ? tau - 1 : 0, hi = ONE;                 // mass_gt(lo) > r >= mass_gt(hi) = 0
        if (top_p >= 1.f && !(mass_gt(0) > r)) lo = hi - 1;                  // (degenerate guard)
        while (hi - lo > 1) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            if (mass_gt(mid) <= r) hi = mid; else lo = mid;
        }
        const unsigned t2 = hi;
        const float v2 = __uint_as_float(t2);
        const float m2 = mass_gt(t2);
        const int mine = count_eq(t2);
        const int n2 = block_sum_i(mine, red.i, tid);
        int limit = (t2 == tau && top_p < 1.f) ? keep_ties : n2;
        if (limit > n2) limit = n2;
        int j = (int)floorf((r - m2) / v2);
        j = j < 0 ? 0 : (j >= limit ? limit - 1 : j);
        // ---- the j-th token (ascending id) whose weight equals v2: block-wide exclusive prefix count (threads own contiguous ids)
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if ((tid & 63) >= o) incl += t;
        }
        __syncthreads();
        if ((tid & 63) == 63) scan[tid >> 6] = incl;
        __syncthreads();
        int before = incl - mine;
        for (int w = 0; w < (tid >> 6); ++w) before += scan[w];
        if (n2 > 0 && j >= before && j < before + mine) {
            int k = j - before;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                if (__float_as_uint(p[e]) == t2) {
                    if (k == 0) red.i[0] = base + e;
                    --k;
                }
            }
        }
        __syncthreads();
        if (n2 > 0) choice = red.i[0];
    }
    if (tid == 0) {
        tok_out[b] = choice;
        if (hist && step_ok) hist[(size_t)b * hist_ld + step] = choice;
    }
}

}  // namespace

extern "C" int seedmi_sample_token_bf16(const void* logits, int ldl, int batch, int vocab, float temperature, float top_p,
                                        const void* uniforms_f32, const void* step_dev, int step_offset, void* tok_out_i64,
                                        void* history_i64, int history_ld, int n_steps, void* stream) {
    if (!logits || !tok_out_i64 || batch <= 0 || vocab <= 0 || vocab > 48 * ST || ldl < vocab || !(temperature > 0.f) ||
        top_p < 0.f) {
        seedmi_set_error("seedmi_sample_token_bf16: bad arguments (batch %d vocab %d ldl %d temperature %g top_p %g)", batch, vocab,
                         ldl, (double)temperature, (double)top_p);
        return SEEDMI_E_SHAPE;
    }
    const int ept = (vocab + ST - 1) / ST;
    const float inv_temp = 1.0f / temperature;
#define SEEDMI_SAMPLE_CASE(E)                                                                                              \
    if (ept <= E) {                                                                                                        \
        hipLaunchKernelGGL(sample_kernel<E>, dim3(batch), dim3(ST), 0, (hipStream_t)stream, (const bf16_t*)logits, ldl, vocab, \
                           inv_temp, top_p, (const float*)uniforms_f32, (const int*)step_dev, step_offset, batch,           \
                           (long long*)tok_out_i64, (long long*)history_i64, history_ld, n_steps);                          \
        return seedmi_check_launch("sample_token");                                                                        \
    }
    SEEDMI_SAMPLE_CASE(8)
    SEEDMI_SAMPLE_CASE(16)
    SEEDMI_SAMPLE_CASE(32)
    SEEDMI_SAMPLE_CASE(40)
    SEEDMI_SAMPLE_CASE(48)
#undef SEEDMI_SAMPLE_CASE
    return SEEDMI_E_SHAPE;
}
