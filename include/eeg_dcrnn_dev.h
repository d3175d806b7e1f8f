/*
 * eeg_dcrnn_dev.h — DEVELOPMENT entry points, exported only by the dev build of the library
 * (`make -C eeg_gnn_ssl_amd/csrc dev` -> libeeg_dcrnn_hip_dev.so, compiled with -DEEG_DEV) and by the
 * test emulator.  The product library libeeg_dcrnn_hip.so does not export them: its kernel-variant
 * choices are compile-time constants and it carries no process-global mutable tuning state.
 * Users: tools/ (A/B timing, cycle probe) and `bench.py --tune`.
 */
#ifndef EEG_DCRNN_DEV_H
#define EEG_DCRNN_DEV_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* When set to a device buffer of B*4*32 int64, the recurrent kernels store the shader-clock cycles each
 * wave spent per phase (slots 0-7 forward, 8-15 backward); NULL disables.  Only the H=64, M=3
 * instantiations carry the probe. */
int eeg_dcrnn_set_seq_probe(int64_t* probe);
/* Integer knobs selecting kernel variants for A/B timing.  key 0 = 1: register-staged NN GEMM;
 * key 1 = 1: register-staged TN GEMM; key 4 = 1: NO XCD-aware placement of the TN k-blocks in the LDS-DMA kernel (2: placement in the register-staged one);
 * key 9 = 1: LDS/MFMA adjoint diffusion; key 12 = 1: single-wave-per-SIMD forward recurrent kernel also
 * where the two-wave one exists (64 units, M <= 3); key 13 = 1: the same for the BPTT kernel; key 11 = 1: per-step
 * launches in the decoder forward instead of the persistent kernel, key 10 = 1: the same for the decoder backward;
 * key 8: unused (round 1-3: k-steps per weight group of the layer-0 input part in the persistent decoder forward); key 3: the
 * streamed-weight BPTT kernel with two workgroups per CU (1 = wherever it exists, 2 = never; default: batches beyond
 * 1.5 clips per CU at M >= 4); keys 14 / 15: 8-wave TN GEMM from this dY width up / its workgroup target;
 * key 18 = 1: the round-4
 * adjoint diffusion (hop planes consumed one at a time) instead of the row-streaming one;
 * key 19 = 1: the two h-part weight-gradient GEMMs of a cell as two launches instead of the paired one;
 * key 17 = 1: the input gradient of a spectral layer as grouped GEMM + node-mix pass instead of gemm_dxf_kernel;
 * key 20 = 1: the round-5 grouped NN GEMM for the spectral x-part instead of gemm_nnf_kernel (kernels_gemm_f.h);
 * key 21 = 1 / 22 = 1: the general-path BPTT / forward recurrent kernels under the spectral form; key 23 = 1: the three grouped
 * weight-gradient launches instead of the fused one (kernels_gemm_f.h);
 * keys 5 / 6 / 7: target workgroup counts of the streaming diffusion (forward / adjoint) and the correlation-Gram launches.
 * Defaults (all 0) = the product configuration. */
int eeg_dcrnn_set_tuning(int key, int value);

#ifdef __cplusplus
}
#endif
#endif /* EEG_DCRNN_DEV_H */
