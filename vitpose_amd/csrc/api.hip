#include "common.h"
#include "vitpose_hip.h"
extern "C" int vp_abi_version(void) { return VP_ABI_VERSION; }
extern "C" const char* vp_build_arch(void) { return "gfx950"; }
