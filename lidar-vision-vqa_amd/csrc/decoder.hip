// csrc/decoder.hip -- one decode step of the Qwen2-architecture stand-in head as ONE native call.
//
// The Python loop of head.StandInHead issues ~14 launches per layer through ctypes at ~9.5 us of host time each: 3.2 ms per token
// for 24 layers although the kernels of a one-token step are microseconds.  This file is the host-side runtime for that loop: the
// same kernels, in the same order, with the same arguments (=> bit-identical logits, tests/test_gpu_head.py), issued from C++.
// inference_engine.py:283-296 -> transformers' generate() with a KV cache is the reference behaviour.
#include "common.h"

namespace {

// rotary embedding at position pos applied in place to the q and k heads of the packed rows (the arithmetic of k_rope in
// elementwise.hip: angle = pos * theta^(-2e/dh), rotate-half pairs), the rotated keys and the values appended to the caches:
// rope + rope + append were three launches of a one-token step
__global__ void __launch_bounds__(256) k_rope_cache(uint16_t *__restrict__ xh, uint16_t *__restrict__ xl, int batch, int n_heads, int n_kv_heads,
                                                    int dh, int pos, int lmax, float theta, uint16_t *__restrict__ kc, uint16_t *__restrict__ kcl,
                                                    uint16_t *__restrict__ vc, uint16_t *__restrict__ vcl) {
    const int half = dh >> 1, d = n_heads * dh, dkv = n_kv_heads * dh;
    const int64_t ld = d + 2 * dkv;
    const int nrope = (n_heads + n_kv_heads) * half;           // rotary pairs per row
    const int per_row = nrope + dkv;                           // + value elements to copy
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= batch * per_row) return;
    const int b = i / per_row, j = i - b * per_row;
    if (j < nrope) {
        const int hd = j / half, e = j - hd * half;
        const float inv = powf(theta, -2.0f * (float)e / (float)dh);
        float sn, cs;
        sincosf((float)pos * inv, &sn, &cs);
        const int64_t o1 = (int64_t)b * ld + (int64_t)hd * dh + e, o2 = o1 + half;
        const float a = bf16_to_f32(xh[o1]) + (xl ? bf16_to_f32(xl[o1]) : 0.f);
        const float bb = bf16_to_f32(xh[o2]) + (xl ? bf16_to_f32(xl[o2]) : 0.f);
        const float ra = a * cs - bb * sn, rb = bb * cs + a * sn;
        const uint16_t ha = f32_to_bf16(ra), hb = f32_to_bf16(rb);
        xh[o1] = ha; xh[o2] = hb;
        uint16_t la = 0, lb = 0;
        if (xl) { la = f32_to_bf16(ra - bf16_to_f32(ha)); lb = f32_to_bf16(rb - bf16_to_f32(hb)); xl[o1] = la; xl[o2] = lb; }
        if (hd >= n_heads) {                                   // a key head: the rotated pair also goes to the cache
            const int c = (hd - n_heads) * dh + e;
            const int64_t dst = ((int64_t)b * lmax + pos) * dkv + c;
            kc[dst] = ha; kc[dst + half] = hb;
            if (xl) { kcl[dst] = la; kcl[dst + half] = lb; }
        }
    } else {
        const int c = j - nrope;
        const int64_t src = (int64_t)b * ld + d + dkv + c, dst = ((int64_t)b * lmax + pos) * dkv + c;
        vc[dst] = xh[src];
        if (xl) vcl[dst] = xl[src];
    }
}

struct StepWs {
    uint16_t *h, *h_lo, *qkv, *qkv_lo, *o, *o_lo, *act, *act_lo;
    float *xb, *gu;
    void *attn;
    size_t attn_bytes;
};
template <typename A> void step_layout(A &a, StepWs &w, int batch, int d, int dkv, int inter, size_t attn_bytes) {
    const size_t ld = (size_t)d + 2 * dkv;
    w.h = a.template take<uint16_t>((size_t)batch * d);
    w.h_lo = a.template take<uint16_t>((size_t)batch * d);
    w.qkv = a.template take<uint16_t>((size_t)batch * ld);
    w.qkv_lo = a.template take<uint16_t>((size_t)batch * ld);
    w.o = a.template take<uint16_t>((size_t)batch * d);
    w.o_lo = a.template take<uint16_t>((size_t)batch * d);
    w.act = a.template take<uint16_t>((size_t)batch * inter);
    w.act_lo = a.template take<uint16_t>((size_t)batch * inter);
    w.xb = a.template take<float>((size_t)batch * d);
    w.gu = a.template take<float>((size_t)batch * 2 * inter);
    w.attn = a.template take<char>(attn_bytes);
    w.attn_bytes = attn_bytes;
}
struct SizerA {
    LvqSizer s;
    template <typename T> T *take(size_t n) { s.template take<T>(n); return nullptr; }
};

}  // namespace

extern "C" size_t lvq_qwen2_decode_workspace_bytes(int batch, int d, int n_heads, int n_kv_heads, int inter, int lmax, int precision) {
    if (batch <= 0 || d <= 0 || n_heads <= 0 || n_kv_heads <= 0 || d % n_heads || inter <= 0 || lmax <= 0) return 0;
    const int dh = d / n_heads;
    const size_t attn = lvq_attention_workspace_bytes(batch, n_heads, 1, lmax, dh, precision);
    SizerA a;
    StepWs w;
    step_layout(a, w, batch, d, dh * n_kv_heads, inter, attn);
    return a.s.total();
}

extern "C" int lvq_qwen2_decode_step(const lvq_qwen2_layer *layers, int n_layers, float *x, int batch, int d, int n_heads, int n_kv_heads,
                                     int inter, int pos, int lmax, float rms_eps, float rope_theta, int precision, void *ws, size_t ws_bytes,
                                     lvq_stream_t stream) {
    if (!layers || n_layers <= 0 || !x || batch <= 0 || d <= 0 || n_heads <= 0 || n_kv_heads <= 0 || d % n_heads || n_heads % n_kv_heads ||
        inter <= 0 || pos < 0 || pos >= lmax || (precision != 1 && precision != 3))
        return LVQ_EINVAL;
    const bool x3 = precision == 3;
    const int dh = d / n_heads, dkv = dh * n_kv_heads;
    const int64_t ld = (int64_t)d + 2 * dkv;
    const size_t attn_bytes = lvq_attention_workspace_bytes(batch, n_heads, 1, lmax, dh, precision);
    LvqArena arena(ws, ws_bytes);
    StepWs w;
    step_layout(arena, w, batch, d, dkv, inter, attn_bytes);
    if (!arena.ok) return LVQ_EWORKSPACE;
    hipStream_t st = lvq_s(stream);
    uint16_t *h_lo = x3 ? w.h_lo : nullptr, *qkv_lo = x3 ? w.qkv_lo : nullptr, *o_lo = x3 ? w.o_lo : nullptr, *act_lo = x3 ? w.act_lo : nullptr;
    const float scale = 1.0f / sqrtf((float)dh);
    float *xa = x, *xb = w.xb;                     // residual stream ping-pong: every layer leaves it in xa again
    int rc;
#define LVQ_TRY(call) do { rc = (call); if (rc != LVQ_OK) return rc; } while (0)
    for (int l = 0; l < n_layers; ++l) {
        const lvq_qwen2_layer &L = layers[l];
        if (x3 && !(L.wqkv_lo && L.wo_lo && L.wgu_lo && L.wdown_lo && L.k_cache_lo && L.v_cache_lo)) return LVQ_EINVAL;
        if (batch <= 8) {                      // RMSNorm fused into the projection (bit-identical to the pair, one launch less)
            LVQ_TRY(lvq_gemv_rmsnorm_bf16(xa, L.ln1, rms_eps, L.wqkv, x3 ? L.wqkv_lo : nullptr, L.bqkv, batch, (int)ld, d, d, ld, nullptr, w.qkv, qkv_lo,
                                          stream));
        } else {
            LVQ_TRY(lvq_rmsnorm(xa, L.ln1, rms_eps, batch, d, nullptr, w.h, h_lo, stream));
            LVQ_TRY(lvq_gemm_bf16(w.h, h_lo, L.wqkv, x3 ? L.wqkv_lo : nullptr, L.bqkv, nullptr, nullptr, 0, 1.0f, 0, batch, (int)ld, d, d, d, ld, 1, 0,
                                  0, 0, nullptr, w.qkv, qkv_lo, stream));
        }
        {
            const int per_row = (n_heads + n_kv_heads) * (dh / 2) + dkv;
            hipLaunchKernelGGL(k_rope_cache, dim3((unsigned)lvq_cdiv((int64_t)batch * per_row, 256)), dim3(256), 0, st, w.qkv, qkv_lo, batch, n_heads,
                               n_kv_heads, dh, pos, lmax, rope_theta, L.k_cache, x3 ? L.k_cache_lo : nullptr, L.v_cache,
                               x3 ? L.v_cache_lo : nullptr);
        }
        LVQ_TRY(lvq_attention_bf16(w.qkv, qkv_lo, L.k_cache, x3 ? L.k_cache_lo : nullptr, L.v_cache, x3 ? L.v_cache_lo : nullptr, nullptr, batch,
                                   n_heads, n_kv_heads, 1, pos + 1, dh, ld, ld, dh, (int64_t)lmax * dkv, dkv, dh, (int64_t)lmax * dkv, dkv, dh,
                                   d, d, dh, scale, 0, w.o, o_lo, w.attn, w.attn_bytes, stream));
        LVQ_TRY(lvq_gemm_bf16(w.o, o_lo, L.wo, x3 ? L.wo_lo : nullptr, nullptr, xa, nullptr, 0, 1.0f, 0, batch, d, d, d, d, d, 1, 0, 0, 0, xb,
                              nullptr, nullptr, stream));
        if (batch <= 8) {
            LVQ_TRY(lvq_gemv_rmsnorm_bf16(xb, L.ln2, rms_eps, L.wgu, x3 ? L.wgu_lo : nullptr, nullptr, batch, 2 * inter, d, d, 2 * (int64_t)inter, w.gu,
                                          nullptr, nullptr, stream));
        } else {
            LVQ_TRY(lvq_rmsnorm(xb, L.ln2, rms_eps, batch, d, nullptr, w.h, h_lo, stream));
            LVQ_TRY(lvq_gemm_bf16(w.h, h_lo, L.wgu, x3 ? L.wgu_lo : nullptr, nullptr, nullptr, nullptr, 0, 1.0f, 0, batch, 2 * inter, d, d, d,
                                  2 * (int64_t)inter, 1, 0, 0, 0, w.gu, nullptr, nullptr, stream));
        }
        // (SiLU(gate) * up produced inside the down projection was tried: every one-row wave re-evaluates 4864 exps and IEEE divisions
        //  and reads the fp32 gate|up row -- 16.1 us against 5.3 + 4.9 us for the separate kernels)
        LVQ_TRY(lvq_swiglu(w.gu, batch, inter, w.act, act_lo, stream));
        LVQ_TRY(lvq_gemm_bf16(w.act, act_lo, L.wdown, x3 ? L.wdown_lo : nullptr, nullptr, xb, nullptr, 0, 1.0f, 0, batch, d, inter, inter, inter, d,
                              1, 0, 0, 0, xa, nullptr, nullptr, stream));
    }
#undef LVQ_TRY
    return lvq_launch_status();
}
