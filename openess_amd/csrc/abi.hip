// ABI identification for liboess.
#include <hip/hip_runtime.h>
#include "oess.h"

extern "C" {
int oess_abi_version(void) { return OESS_ABI_VERSION; }
const char* oess_build_info(void) {
    return "liboess 0.1 gfx950 hip " __VERSION__;
}
const char* oess_strerror(int code) {
    switch (code) {
        case OESS_OK: return "ok";
        case OESS_EINVAL: return "invalid argument";
        case OESS_ENOMEM: return "workspace too small";
        case OESS_ELAUNCH: return "HIP launch/runtime error";
        default: return "unknown error";
    }
}
}
