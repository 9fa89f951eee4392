// csrc/attention.hip -- softmax(Q K^T * scale + bias) V on bf16 MFMA tiles with fp32 online softmax (gfx950).
//
// Replaces F.scaled_dot_product_attention inside nn.MultiheadAttention (VATBlock.sa / VATBlock.ca,
// vat_blocks.py:39,42) and deepencoder's sdp_attention (clip_sdpa.py:50-66, sam_vary_sdpa.py:27-42).
//
// Layout trick (no score matrix in HBM, no P through LDS): each wave owns 16 queries and computes the
// TRANSPOSED score tile S^T = K Q^T, so the MFMA C layout puts the QUERY on lane&15 and four KEYS in the
// lane's registers.  Row statistics (max / sum over keys) are then register reductions plus two
// xor-shuffles, and the exponentiated tile is ALREADY the B operand of the second product
// O^T = V^T P^T (keys = MFMA k index, permuted consistently on both operands), whose C layout again has
// the query on lane&15 -- so the online-softmax rescale is a per-lane scalar and nothing is transposed
// except V, once, while it is staged into LDS.
//
// Per 256-thread workgroup: 64 queries (4 waves x 16) of one (batch, head); K/V stream through LDS in
// 64-key tiles.  Head dim is padded to DHP in {32,64,96,128} (dh = 112 = 896/8 runs as 128 with zero
// fill); larger head dims (448, 1024: the reference's 2-head defaults) take the split path in
// lvq_attention_bf16 (scores via lvq_gemm_bf16 + lvq_softmax_rows).  NSPLIT = 3 is the bf16x3 mode
// (hi/lo operands, three MFMA passes) that meets the 1e-3 parity bar; NSPLIT = 1 is plain bf16.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct AttnArgs {
    const uint16_t *q, *ql, *k, *kl, *v, *vl;
    const float *bias;
    int B, H, Hkv, Nq, Nkv, dh;
    int64_t q_bs, ldq, q_hs, k_bs, ldk, k_hs, v_bs, ldv, v_hs, o_bs, ldo, o_hs;
    float scale;
    int causal;
    uint16_t *o, *ol;
};

constexpr int KVB = 64;  // keys per tile

template <int DHP, int NSPLIT>
__global__ void __launch_bounds__(256) k_attn(AttnArgs a) {
    constexpr int NS = (NSPLIT == 3) ? 2 : 1;
    constexpr int KROW = DHP + 8;       // bf16 elements per K row (+16 B pad: odd multiple of 16 B)
    constexpr int VROW = KVB + 4;       // bf16 elements per V^T row
    constexpr int NC = DHP / 32;        // 32-wide k chunks of the head dim
    constexpr int ND = DHP / 16;        // 16-wide output tiles of the head dim
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t *Ks = smem;                          // [NS][KVB][KROW]
    uint16_t *Vt = smem + NS * KVB * KROW;        // [NS][DHP][VROW]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y, hk = h / (a.H / a.Hkv);
    const int q0 = blockIdx.x * 64 + wid * 16;
    const int qi = q0 + l15;
    const float LOG2E = 1.4426950408889634f;

    // Q fragments (B operand of S^T = K Q^T): lane supplies Q[qi][c*32 + 8g .. +7]
    bf16x8 qf[NS][NC];
    {
        const uint16_t *src[2] = {a.q, a.ql};
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int kk = c * 32 + g * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (qi < a.Nq && kk < a.dh)
                    v = *reinterpret_cast<const uint4 *>(src[s] + (int64_t)b * a.q_bs + (int64_t)qi * a.ldq + (int64_t)h * a.q_hs + kk);
                qf[s][c] = *reinterpret_cast<bf16x8 *>(&v);
            }
    }

    f32x4 o[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int n_tiles = (a.Nkv + KVB - 1) / KVB;
    const uint16_t *ksrc[2] = {a.k, a.kl}, *vsrc[2] = {a.v, a.vl};
    for (int t = 0; t < n_tiles; ++t) {
        // ---- stage K (row-major) and V (transposed) into LDS ----
        constexpr int CH = DHP / 8;  // 16-byte chunks per row
        for (int e = tid; e < KVB * CH; e += 256) {
            const int r = e / CH, ch = e - r * CH, kk = ch * 8;
            const int key = t * KVB + r;
            const bool ok = key < a.Nkv && kk < a.dh;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                uint4 kv4 = make_uint4(0, 0, 0, 0), vv4 = make_uint4(0, 0, 0, 0);
                if (ok) {
                    kv4 = *reinterpret_cast<const uint4 *>(ksrc[s] + (int64_t)b * a.k_bs + (int64_t)key * a.ldk + (int64_t)hk * a.k_hs + kk);
                    vv4 = *reinterpret_cast<const uint4 *>(vsrc[s] + (int64_t)b * a.v_bs + (int64_t)key * a.ldv + (int64_t)hk * a.v_hs + kk);
                }
                *reinterpret_cast<uint4 *>(Ks + (s * KVB + r) * KROW + kk) = kv4;
                const uint16_t *ve = reinterpret_cast<const uint16_t *>(&vv4);
#pragma unroll
                for (int i = 0; i < 8; ++i) Vt[(s * DHP + kk + i) * VROW + r] = ve[i];
            }
        }
        __syncthreads();

        // ---- S^T tiles: rows = keys kt*16 + g*4 + r, column = query l15 ----
        f32x4 sc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const bf16x8 kh = *reinterpret_cast<const bf16x8 *>(Ks + (kt * 16 + l15) * KROW + c * 32 + g * 8);
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qf[0][c], sc[kt], 0, 0, 0);
                if (NSPLIT == 3) {
                    const bf16x8 kl = *reinterpret_cast<const bf16x8 *>(Ks + (KVB + kt * 16 + l15) * KROW + c * 32 + g * 8);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qf[NS - 1][c], sc[kt], 0, 0, 0);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qf[0][c], sc[kt], 0, 0, 0);
                }
            }
        }
        // ---- scale, bias, masks; online softmax in the log2 domain ----
        float tmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t * KVB + kt * 16 + g * 4 + r;
                float x = sc[kt][r] * a.scale;
                if (a.bias && key < a.Nkv && qi < a.Nq)
                    x += a.bias[(((int64_t)b * a.H + h) * a.Nq + qi) * a.Nkv + key];
                x *= LOG2E;
                if (key >= a.Nkv || (a.causal && key > qi + a.Nkv - a.Nq)) x = -INFINITY;
                sc[kt][r] = x;
                tmax = fmaxf(tmax, x);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = exp2f(m_run - m_safe);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(sc[kt][r] - m_safe);
                sc[kt][r] = p;
                rs += p;
            }
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int n = 0; n < ND; ++n) o[n] *= alpha;

        // ---- O^T += V^T P^T : A = V^T[d][keys], B = P^T[keys][query] straight from the score registers ----
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = (j < 4) ? sc[2 * s2][j] : sc[2 * s2 + 1][j - 4];
                const uint16_t hh = f32_to_bf16(p);
                ph[j] = (short)hh;
                if (NSPLIT == 3) pl[j] = (short)f32_to_bf16(p - bf16_to_f32(hh));
            }
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                const uint16_t *vrow = Vt + (n * 16 + l15) * VROW + g * 4;
                bf16x4 v0 = *reinterpret_cast<const bf16x4 *>(vrow + (2 * s2) * 16);
                bf16x4 v1 = *reinterpret_cast<const bf16x4 *>(vrow + (2 * s2 + 1) * 16);
                bf16x8 vh = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph, o[n], 0, 0, 0);
                if (NSPLIT == 3) {
                    const uint16_t *vrl = vrow + DHP * VROW;
                    bf16x4 w0 = *reinterpret_cast<const bf16x4 *>(vrl + (2 * s2) * 16);
                    bf16x4 w1 = *reinterpret_cast<const bf16x4 *>(vrl + (2 * s2 + 1) * 16);
                    bf16x8 vlo = __builtin_shufflevector(w0, w1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl, o[n], 0, 0, 0);
                    o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vlo, ph, o[n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- normalise and store: lane holds O[qi][n*16 + g*4 + r] ----
    if (qi < a.Nq) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        uint16_t *dst = a.o + (int64_t)b * a.o_bs + (int64_t)qi * a.ldo + (int64_t)h * a.o_hs;
        uint16_t *dl = a.ol ? a.ol + (int64_t)b * a.o_bs + (int64_t)qi * a.ldo + (int64_t)h * a.o_hs : nullptr;
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            const int d0 = n * 16 + g * 4;
            if (d0 >= a.dh) continue;
            ushort4 hv, lv;
            uint16_t *hp = reinterpret_cast<uint16_t *>(&hv), *lp = reinterpret_cast<uint16_t *>(&lv);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = o[n][r] * inv;
                hp[r] = f32_to_bf16(y);
                lp[r] = f32_to_bf16(y - bf16_to_f32(hp[r]));
            }
            *reinterpret_cast<ushort4 *>(dst + d0) = hv;
            if (dl) *reinterpret_cast<ushort4 *>(dl + d0) = lv;
        }
    }
}

template <int DHP, int NSPLIT> int launch_attn(const AttnArgs &a, hipStream_t st) {
    constexpr int NS = (NSPLIT == 3) ? 2 : 1;
    const size_t lds = (size_t)(NS * KVB * (DHP + 8) + NS * DHP * (KVB + 4)) * sizeof(uint16_t);
    if (lds > 64 * 1024)
        hipFuncSetAttribute((const void *)k_attn<DHP, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((unsigned)lvq_cdiv(a.Nq, 64), (unsigned)a.H, (unsigned)a.B);
    hipLaunchKernelGGL((k_attn<DHP, NSPLIT>), grid, dim3(256), lds, st, a);
    return lvq_launch_status();
}

// row softmax for the split (large head-dim) path: p = softmax(s*scale + bias, causal) -> bf16 hi/lo
__global__ void __launch_bounds__(256) k_softmax_rows(const float *__restrict__ s, const float *__restrict__ bias, int64_t rows,
                                                      int nq, int nkv, float scale, int causal, int64_t ldp,
                                                      uint16_t *__restrict__ ph, uint16_t *__restrict__ pl) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int qi = (int)(row % nq);
    const float *sr = s + row * nkv;
    const float *br = bias ? bias + row * nkv : nullptr;
    const int lim = causal ? qi + nkv - nq : nkv - 1;  // last visible key
    float mx = -INFINITY;
    for (int k = lane; k < nkv; k += 64)
        if (k <= lim) mx = fmaxf(mx, sr[k] * scale + (br ? br[k] : 0.f));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int k = lane; k < nkv; k += 64)
        if (k <= lim) sum += expf(sr[k] * scale + (br ? br[k] : 0.f) - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    for (int k = lane; k < ldp; k += 64) {
        float p = 0.f;
        if (k < nkv && k <= lim) p = expf(sr[k] * scale + (br ? br[k] : 0.f) - mx) * inv;
        const uint16_t hh = f32_to_bf16(p);
        ph[row * ldp + k] = hh;
        if (pl) pl[row * ldp + k] = f32_to_bf16(p - bf16_to_f32(hh));
    }
}

// [rows, cols] (ld) bf16 -> transposed [cols, rows_pad] bf16, zero padded (V^T for the split path)
__global__ void __launch_bounds__(256) k_transpose_bf16(const uint16_t *__restrict__ x, int rows, int cols, int64_t ld,
                                                        int rows_pad, uint16_t *__restrict__ y) {
    __shared__ uint16_t tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? x[(int64_t)r * ld + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows_pad) y[(int64_t)c * rows_pad + r] = tile[tx][i];
    }
}

}  // namespace

extern "C" size_t lvq_attention_workspace_bytes(int batch, int n_heads, int nq, int nkv, int dh, int precision) {
    if (dh <= 128) return 256;
    const int64_t nkp = (nkv + 7) / 8 * 8;
    const int ns = precision == 3 ? 2 : 1;
    size_t s = (size_t)n_heads * nq * nkv * sizeof(float);            // scores of one batch element
    size_t p = (size_t)ns * n_heads * nq * nkp * sizeof(uint16_t);    // probabilities (hi, lo)
    size_t vt = (size_t)ns * n_heads * dh * nkp * sizeof(uint16_t);   // V^T (hi, lo)
    (void)batch;
    return lvq_align(s) + lvq_align(p) + lvq_align(vt) + 1024;
}

extern "C" int lvq_attention_bf16(const lvq_bf16 *q, const lvq_bf16 *q_lo, const lvq_bf16 *k, const lvq_bf16 *k_lo,
                                  const lvq_bf16 *v, const lvq_bf16 *v_lo, const float *bias, int batch, int n_heads,
                                  int n_kv_heads, int nq, int nkv, int dh, int64_t q_bstride, int64_t ldq, int64_t q_hstride,
                                  int64_t k_bstride, int64_t ldk, int64_t k_hstride, int64_t v_bstride, int64_t ldv,
                                  int64_t v_hstride, int64_t o_bstride, int64_t ldo, int64_t o_hstride, float scale, int causal,
                                  lvq_bf16 *o, lvq_bf16 *o_lo, void *ws, size_t ws_bytes, lvq_stream_t stream) {
    if (batch <= 0 || n_heads <= 0 || n_kv_heads <= 0 || n_heads % n_kv_heads || nq < 0 || nkv <= 0 || dh <= 0 || !q || !k ||
        !v || !o)
        return LVQ_EINVAL;
    const bool split = q_lo != nullptr;
    if (split != (k_lo != nullptr) || split != (v_lo != nullptr)) return LVQ_EINVAL;
    if (nq == 0) return LVQ_OK;
    if ((dh & 7) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (q_hstride & 7) || (k_hstride & 7) || (v_hstride & 7) ||
        (q_bstride & 7) || (k_bstride & 7) || (v_bstride & 7) || (ldo & 3) || (o_hstride & 3) || (o_bstride & 3))
        return LVQ_EUNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)q_lo | (uintptr_t)k_lo | (uintptr_t)v_lo) & 15)
        return LVQ_EUNSUPPORTED;
    if (((uintptr_t)o | (uintptr_t)o_lo) & 7) return LVQ_EUNSUPPORTED;
    if (n_heads > 65535 || batch > 65535) return LVQ_EUNSUPPORTED;
    hipStream_t st = lvq_s(stream);
    if (dh <= 128 && (dh & 15) == 0) {
        AttnArgs a;
        a.q = q; a.ql = q_lo; a.k = k; a.kl = k_lo; a.v = v; a.vl = v_lo; a.bias = bias;
        a.B = batch; a.H = n_heads; a.Hkv = n_kv_heads; a.Nq = nq; a.Nkv = nkv; a.dh = dh;
        a.q_bs = q_bstride; a.ldq = ldq; a.q_hs = q_hstride; a.k_bs = k_bstride; a.ldk = ldk; a.k_hs = k_hstride;
        a.v_bs = v_bstride; a.ldv = ldv; a.v_hs = v_hstride; a.o_bs = o_bstride; a.ldo = ldo; a.o_hs = o_hstride;
        a.scale = scale; a.causal = causal; a.o = o; a.ol = o_lo;
        const int dhp = (dh + 31) / 32 * 32;
        if (!split) {
            switch (dhp) {
                case 32: return launch_attn<32, 1>(a, st);
                case 64: return launch_attn<64, 1>(a, st);
                case 96: return launch_attn<96, 1>(a, st);
                default: return launch_attn<128, 1>(a, st);
            }
        } else {
            switch (dhp) {
                case 32: return launch_attn<32, 3>(a, st);
                case 64: return launch_attn<64, 3>(a, st);
                case 96: return launch_attn<96, 3>(a, st);
                default: return launch_attn<128, 3>(a, st);
            }
        }
    }
    // ---- split path for large head dims: S = Q K^T (GEMM) -> row softmax -> O = P V (GEMM on V^T) ----
    if (n_heads != n_kv_heads) return LVQ_EUNSUPPORTED;
    const int64_t nkp = (nkv + 7) / 8 * 8;
    const int ns = split ? 2 : 1;
    LvqArena arena(ws, ws_bytes);
    float *S = arena.take<float>((size_t)n_heads * nq * nkv);
    uint16_t *P = arena.take<uint16_t>((size_t)ns * n_heads * nq * nkp);
    uint16_t *VT = arena.take<uint16_t>((size_t)ns * n_heads * dh * nkp);
    if (!arena.ok) return LVQ_EWORKSPACE;
    uint16_t *Pl = split ? P + (size_t)n_heads * nq * nkp : nullptr;
    uint16_t *VTl = split ? VT + (size_t)n_heads * dh * nkp : nullptr;
    for (int b = 0; b < batch; ++b) {
        const uint16_t *qb = q + b * q_bstride, *kb = k + b * k_bstride, *vb = v + b * v_bstride;
        int rc = lvq_gemm_bf16(qb, split ? q_lo + b * q_bstride : nullptr, kb, split ? k_lo + b * k_bstride : nullptr, nullptr,
                               nullptr, nullptr, 0, 1.0f, 0, nq, nkv, dh, ldq, ldk, nkv, n_heads, q_hstride, k_hstride,
                               (int64_t)nq * nkv, S, nullptr, nullptr, stream);
        if (rc != LVQ_OK) return rc;
        hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)lvq_cdiv((int64_t)n_heads * nq, 4)), dim3(256), 0, st, S,
                           bias ? bias + (int64_t)b * n_heads * nq * nkv : nullptr, (int64_t)n_heads * nq, nq, nkv, scale, causal,
                           nkp, P, Pl);
        for (int hh = 0; hh < n_heads; ++hh) {
            dim3 tg((unsigned)lvq_cdiv(dh, 32), (unsigned)lvq_cdiv(nkp, 32));
            hipLaunchKernelGGL(k_transpose_bf16, tg, dim3(256), 0, st, vb + hh * v_hstride, nkv, dh, ldv, (int)nkp,
                               VT + (size_t)hh * dh * nkp);
            if (split)
                hipLaunchKernelGGL(k_transpose_bf16, tg, dim3(256), 0, st, v_lo + b * v_bstride + hh * v_hstride, nkv, dh, ldv,
                                   (int)nkp, VTl + (size_t)hh * dh * nkp);
        }
        // O[b, :, h, :] = P[h] (nq x nkp) . VT[h]^T (dh x nkp): C row stride ldo, batch (head) stride o_hstride
        rc = lvq_gemm_bf16(P, Pl, VT, VTl, nullptr, nullptr, nullptr, 0, 1.0f, 0, nq, dh, (int)nkp, nkp, nkp, ldo, n_heads,
                           (int64_t)nq * nkp, (int64_t)dh * nkp, o_hstride, nullptr, o + b * o_bstride,
                           o_lo ? o_lo + b * o_bstride : nullptr, stream);
        if (rc != LVQ_OK) return rc;
    }
    return lvq_launch_status();
}
