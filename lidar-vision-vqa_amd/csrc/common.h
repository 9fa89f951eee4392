// csrc/common.h -- shared helpers for the gfx950 kernels behind include/lvq.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <initializer_list>
#include "../../include/lvq.h"

#define LVQ_WAVE 64

// the tuning record (api.hip): kernel-family choices set through lvq_set_tuning; the library never reads the environment
const lvq_tuning &lvq_tune();

static inline hipStream_t lvq_s(lvq_stream_t s) { return (hipStream_t)s; }

// every entry point ends with this: launch errors become a return code, never an abort
static inline int lvq_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LVQ_OK : LVQ_ELAUNCH;
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (function, DEVICE): a process-wide "done" flag leaves the kernels of a
// second GPU at the 64 KB default (launch failure there).  One LvqLdsOnce per call site remembers, per device, whether the
// attribute was set; racing threads both set it (idempotent).
struct LvqLdsOnce { std::atomic<uint64_t> tried{0}, ok{0}; };
static inline bool lvq_ensure_lds(LvqLdsOnce &st, std::initializer_list<const void *> funcs, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    const uint64_t bit = 1ull << (dev & 63);
    if (st.tried.load(std::memory_order_acquire) & bit) return (st.ok.load(std::memory_order_acquire) & bit) != 0;
    bool good = true;
    for (const void *f : funcs) good = (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess) && good;
    if (!good) (void)hipGetLastError();
    if (good) st.ok.fetch_or(bit, std::memory_order_release);
    st.tried.fetch_or(bit, std::memory_order_release);
    return good;
}
// compute units of the CURRENT device (cached per device id)
static inline int lvq_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    int v = cache[dev & 63].load(std::memory_order_relaxed);
    if (v > 0) return v;
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cache[dev & 63].store(v, std::memory_order_relaxed);
    return v;
}

static inline int64_t lvq_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t lvq_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct LvqArena {
    char *base;
    size_t off, cap;
    bool ok;
    LvqArena(void *p, size_t bytes) : base((char *)p), off(0), cap(bytes), ok(true) {}
    template <typename T> T *take(size_t n) {
        size_t o = lvq_align(off);
        size_t e = o + n * sizeof(T);
        if (base == nullptr || e > cap) { ok = false; off = e; return nullptr; }
        off = e;
        return (T *)(base + o);
    }
};
// same arithmetic without memory, for *_workspace_bytes
struct LvqSizer {
    size_t off = 0;
    template <typename T> void take(size_t n) { off = lvq_align(off) + n * sizeof(T); }
    size_t total() const { return lvq_align(off) + 256; }
};

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even fp32 -> bf16; NaN stays NaN (MI355X_MICROARCH "Correctness boundaries")
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
