/* CPU restatement (plain C) of VectorQuantizer2.forward's distance + argmin
 * (reference: models/seed_qformer/qformer_quantizer.py:94-98) in the bf16 model dtype.
 *
 * TEST INFRASTRUCTURE: linked only by tests/ and __graft_entry__.smoke() as a checker.
 * Same arithmetic as oracle/seed_oracle.py::vq_distances_fixed_order and as the HIP kernel
 * seed_amd/csrc/vq_argmin.hip: every sub-expression rounded to bf16 where the reference materialises a
 * half tensor, fp32 accumulation in the fixed order k = 0..D-1, torch.argmin's first-index tie-break.
 *
 *   gcc -O2 -shared -fPIC -ffp-contract=off -o oracle/_build/libvq_oracle.so oracle/vq_oracle.c
 */
#include <stdint.h>
#include <string.h>

static float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u = (u + (0x7FFFu + ((u >> 16) & 1u))) & 0xFFFF0000u;   /* round to nearest even */
    memcpy(&f, &u, 4);
    return f;
}

/* z: [rows, dim] fp32 holding bf16 values; e: [n_embed, dim]; ids: int64 [rows]; gap (optional): d2 - d1 */
void vq_argmin_bf16_oracle(const float* z, const float* e, int rows, int n_embed, int dim, int64_t* ids, float* gap) {
    for (int r = 0; r < rows; ++r) {
        const float* zr = z + (size_t)r * dim;
        float zz = 0.f;
        for (int k = 0; k < dim; ++k) zz = zz + bf16_round(zr[k] * zr[k]);
        zz = bf16_round(zz);
        float best = 0.f, second = 0.f;
        int64_t bi = -1;
        for (int n = 0; n < n_embed; ++n) {
            const float* en = e + (size_t)n * dim;
            float ee = 0.f, dot = 0.f;
            for (int k = 0; k < dim; ++k) ee = ee + bf16_round(en[k] * en[k]);
            ee = bf16_round(ee);
            for (int k = 0; k < dim; ++k) dot = dot + zr[k] * en[k];   /* product of two bf16 values is exact */
            const float s = bf16_round(zz + ee);
            const float d = bf16_round(s - 2.0f * bf16_round(dot));
            if (bi < 0) { best = d; second = d; bi = n; if (n_embed > 1) second = 3.0e38f; }
            else if (d < best || (d != d && best == best)) { second = best; best = d; bi = n; }   /* torch.argmin: the first NaN is the minimum */
            else if (d < second) { second = d; }
        }
        ids[r] = bi;
        if (gap) gap[r] = second - best;
    }
}
