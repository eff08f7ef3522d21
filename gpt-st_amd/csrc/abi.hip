// ABI version + trivial probes of the C ABI (include/gptst_hip.h).
#include "common.h"
#include "gptst_hip.h"

extern "C" int gptst_abi_version(void) { return GPTST_ABI_VERSION; }
