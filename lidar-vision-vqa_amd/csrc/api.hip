// csrc/api.hip -- version / error strings of the C ABI (include/lvq.h)
#include "common.h"

extern "C" const char *lvq_version(void) { return "lvq-hip 0.1.0 (gfx950)"; }

extern "C" const char *lvq_strerror(int code) {
    switch (code) {
        case LVQ_OK: return "ok";
        case LVQ_EINVAL: return "invalid argument";
        case LVQ_EWORKSPACE: return "workspace too small";
        case LVQ_ELAUNCH: return "HIP launch/runtime error";
        case LVQ_EOVERFLOW: return "key space overflow (batch * grid cells >= 2^31)";
        case LVQ_EUNSUPPORTED: return "shape not supported by the kernels";
        default: return "unknown error";
    }
}
