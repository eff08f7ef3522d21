// ABI version + trivial probes of the C ABI (include/gptst_hip.h).
#include "common.h"
#include "gptst_hip.h"

extern "C" int gptst_abi_version(void) { return GPTST_ABI_VERSION; }

thread_local int g_deterministic = 0;
// 1: bit-reproducible steps (single-owner reductions in a fixed order where the default path uses float atomics); thread-local.
extern "C" int gptst_set_deterministic(int on) { g_deterministic = on ? 1 : 0; return GPTST_OK; }
