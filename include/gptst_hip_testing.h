/* gptst_hip_testing.h — hooks for the test-suite and the micro-benchmarks.  NOT part of the drop-in C ABI (include/gptst_hip.h): nothing a
 * consumer of the pretraining path needs; they select alternative kernels / launch geometries of the same entry points so that A/B runs and
 * parity tests of superseded kernels can use one build of the library.  Thread-local state. */
#ifndef GPTST_HIP_TESTING_H
#define GPTST_HIP_TESTING_H
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
/* launch-geometry knobs for benchmarking, not needed for correctness.  1: rows per block of the poolgen forward; 2 / 5: forced split
 * count of the NODE / TIME weight gradient; 3: 1 = first-generation (LDS-staged) apply for C = 64; 4: tiles per wave of apply64 / apply128;
 * 6: workgroups of the loss-head kernels; 7 / 8: 1 = first-generation weight gradient / apply for C = 128; 10: 0 = VALU forward of the
 * pool jobs instead of the (bit-identical) MFMA one; 20: 1 = second-generation cap routing forward (cap_route_fwd2_kernel); 21: cap routing
 * forward variant (0 = cap_route_fwd4_kernel, 1 / 2 = cap_route_fwd3_kernel at <= 128 / <= 80 VGPRs); 23: 1 = cross-time backward as a replicated prologue of
 * cap_route_bwd2_kernel instead of a role; 25: node halves of the routing backward (gptst_cap_split_units: 0 = off, the default; -1 = by the CU count; n = the
 * last n (b,t)); 26: 0 = the halves never ride behind the cross-time role workgroups.  The Python binding applies GPTST_TUNE="id=value,..." from the environment. */
int gptst_tune(int id, int value);
/* mask selection: 1 = the multi-launch radix select for every size (a single-workgroup launch serves M <= 8192 cells otherwise);
 * 2 (gptst_mask_*_u24 only) = the one-workgroup lattice kernel for every size up to 65536 cells; either value also keeps the cooperative launch
 * (gptst_mask_cooperative, gptst_hip.h) out.  Process-wide, not thread-local. */
int gptst_mask_force_multi(int on);
/* puts n hand-off expiries on record without poisoning anything (synchronises): what the optimiser's guard and the steppers' recovery see when a
 * bounded in-launch wait ran out (tests/test_gpu_step.py::test_lost_handoff_*). */
int gptst_handoff_inject(int n);
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GPTST_HIP_TESTING_H */
