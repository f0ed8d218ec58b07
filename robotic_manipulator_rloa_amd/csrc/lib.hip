// Library identity for libnaf_hip.so (gfx950 only: no other code object is built).
#include "common.h"
#include "../../include/naf_hip.h"

extern "C" int naf_hip_abi_version(void) { return NAF_HIP_ABI_VERSION; }
extern "C" const char* naf_hip_arch(void) { return "gfx950"; }
