/*
 * seedcal — calibration micro-benchmarks of the MI355X box (libseedcal.so).  NOT part of the product library and NOT part of the
 * drop-in boundary (include/seedmi.h): measurement infrastructure for bench.py's `extra.measured_ceilings` and the tools/ scripts
 * ("re-measure at build time with a pure-MFMA loop and a streaming-copy microbench").
 */
#ifndef SEEDCAL_H
#define SEEDCAL_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Streams `bytes` of device memory with 16-byte non-temporal loads from 256*blocks_per_cu workgroups (the decode GEMMs' access
 * pattern) and discards them: the HBM read rate this box actually delivers.  0 on success, negative on bad arguments / launch failure. */
int seedcal_stream_read(const void* p, size_t bytes, int blocks_per_cu, void* scratch4, void* stream);
/* MFMA-only loop from registers (shape 0: v_mfma_f32_16x16x32_bf16, 1: v_mfma_f32_32x32x16_bf16; 8 waves per workgroup, four
 * accumulator chains each, varied non-zero operands): the matrix-pipe rate this box sustains at the clock its power budget allows.
 * *flops_out = floating-point operations of the launch. */
int seedcal_mfma_bf16(int shape, int iters, int workgroups, void* scratch4, double* flops_out, void* stream);
#ifdef __cplusplus
}
#endif
#endif
