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

// ---- the tuning record (include/lvq.h): the library's only process-wide state, set explicitly by the caller ----
static lvq_tuning g_tuning = {};
const lvq_tuning &lvq_tune() { return g_tuning; }

extern "C" void lvq_tuning_defaults(lvq_tuning *t) {
    if (t) *t = lvq_tuning{};
}
extern "C" int lvq_set_tuning(const lvq_tuning *t) {
    g_tuning = t ? *t : lvq_tuning{};
    return LVQ_OK;
}
extern "C" int lvq_get_tuning(lvq_tuning *t) {
    if (!t) return LVQ_EINVAL;
    *t = g_tuning;
    return LVQ_OK;
}
