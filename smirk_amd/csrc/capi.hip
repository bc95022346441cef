// Error strings / ABI version of libsmirk_hip.so.
#include "common.h"

extern "C" const char* smirk_strerror(int code) {
    switch (code) {
        case SMIRK_OK: return "ok";
        case SMIRK_ERR_BAD_ARG: return "bad argument (null pointer, size or alignment constraint violated)";
        case SMIRK_ERR_WORKSPACE: return "workspace too small (see smirk_*_workspace_bytes)";
        case SMIRK_ERR_LAUNCH: return "HIP kernel launch failed";
        case SMIRK_ERR_UNSUPPORTED: return "configuration not supported by the gfx950 kernels";
        default: return "unknown smirk error";
    }
}

extern "C" int smirk_abi_version(void) { return 3; }
