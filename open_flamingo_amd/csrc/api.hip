// ABI bookkeeping entry points of libofhip (include/of_hip.h).
#include "of_platform.h"
#include "../../include/of_hip.h"

extern "C" int of_abi_version(void) { return OF_ABI_VERSION; }
extern "C" int of_build_kind(void) {
#ifdef OF_HOST_EMU
    return 2;
#else
    return 1;
#endif
}
