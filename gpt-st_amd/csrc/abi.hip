// ABI version + trivial probes of the C ABI (include/gptst_hip.h).
#include "common.h"
#include "gptst_hip.h"

extern "C" int gptst_abi_version(void) { return GPTST_ABI_VERSION; }

thread_local int g_deterministic = 0;
// 1: bit-reproducible steps (single-owner reductions in a fixed order where the default path uses float atomics); thread-local.
extern "C" int gptst_set_deterministic(int on) { g_deterministic = on ? 1 : 0; return GPTST_OK; }

// number of bounded in-launch hand-off waits that expired since the library was loaded (cap_route_bwd2_kernel's roles, hypertem_bwd_pair_kernel's
// lower weight-gradient role, the grid barriers of the cooperative mask launch): 0 in a healthy run.  An expiry poisons that launch's output with NaN; this tells such a NaN from numerical trouble.
GPTST_INTERNAL int gptst_handoff_lost_capmfma(unsigned* out);
GPTST_INTERNAL int gptst_handoff_lost_hypertem(unsigned* out);
GPTST_INTERNAL int gptst_handoff_lost_masksel(unsigned* out);
extern "C" int gptst_handoff_timeouts(int* out) {
    if (!out) return GPTST_EARG;
    unsigned a = 0u, b = 0u, c = 0u;
    if (gptst_handoff_lost_capmfma(&a) || gptst_handoff_lost_hypertem(&b) || gptst_handoff_lost_masksel(&c)) return -5;     // (hipMemcpyFromSymbol failed)
    *out = (int)(a + b + c);
    return GPTST_OK;
}

// Clears the expiry counters (synchronises with the device): the host calls it once it has dealt with a lost hand-off — until then the optimiser
// skips every update (gptst_clip_adam's guard), so that a poisoned gradient never reaches the weights.
GPTST_INTERNAL int gptst_handoff_clear_capmfma(unsigned to);
GPTST_INTERNAL int gptst_handoff_clear_hypertem(unsigned to);
GPTST_INTERNAL int gptst_handoff_clear_masksel(unsigned to);
GPTST_INTERNAL int gptst_adam_skipped_clear(void);
extern "C" int gptst_handoff_reset(void) {
    if (hipDeviceSynchronize() != hipSuccess) return -5;
    return (gptst_handoff_clear_capmfma(0u) || gptst_handoff_clear_hypertem(0u) || gptst_handoff_clear_masksel(0u) || gptst_adam_skipped_clear()) ? -5 : GPTST_OK;
}
// (include/gptst_hip_testing.h)
extern "C" int gptst_handoff_inject(int n) {
    if (n < 0) return GPTST_EARG;
    if (hipDeviceSynchronize() != hipSuccess) return -5;
    return gptst_handoff_clear_capmfma((unsigned)n) ? -5 : GPTST_OK;
}
