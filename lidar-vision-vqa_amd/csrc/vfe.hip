// csrc/vfe.hip -- per-voxel feature encoders and the voxel->BEV bridge (fp32, HBM-bound; gfx950).
//
//   lvq_mean_vfe        a5  backbones_3d/vfe/mean_vfe.py:25-29
//   lvq_pillar_vfe      a6  backbones_3d/vfe/pillar_vfe.py:8-49,94-123   (PillarVFE + PFNLayer)
//   lvq_scatter_mean    a7  torch_scatter.scatter_mean as used in dynamic_mean_vfe.py:64, dynamic_pillar_vfe.py:105
//   lvq_dynamic_pfn     a7  dynamic_pillar_vfe.py:105-127,210-227, dynamic_voxel_vfe.py:73-92 + PFNLayerV2 35-46
//   lvq_pillar_scatter  a8  backbones_2d/map_to_bev/pointpillar_scatter.py:14-37
//
// These are skinny (K = 10..192) fp32 linear layers fused with their elementwise prologue
// (feature augmentation, padding mask) and epilogue (folded BatchNorm, ReLU, max-pool), so every voxel
// / point is read once and only the pooled row is written.  One 64-lane wave owns one voxel (hard
// path) or a strip of points (dynamic path); lanes map to OUTPUT CHANNELS, so the pooled row leaves as
// one coalesced 256-B store / one contiguous 256-B atomic-max wave instruction (post-ReLU values are
// >= 0, so float max == signed-int max on the bit patterns: exact and order-independent).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int MAX_LAYERS = 4;

// LDS hand-off between lanes of ONE wave: the LDS queue is in order per wave, so only the compiler must
// be kept from reordering the accesses.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct PfnParams {
    const float *w[MAX_LAYERS];
    const float *scale[MAX_LAYERS];
    const float *shift[MAX_LAYERS];
    int cin[MAX_LAYERS];
    int cout[MAX_LAYERS];
    int n_layers;
    int flags;  // bit0 USE_ABSLOTE_XYZ, bit1 WITH_DISTANCE
    float vs[3];
    float off[3];
};

__global__ void __launch_bounds__(256) k_mean_vfe(const float *__restrict__ voxels, const int32_t *__restrict__ num_pts,
                                                  int64_t m_cap, const int32_t *__restrict__ n_live, int t, int c,
                                                  float *__restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t m = n_live ? (int64_t)*n_live : m_cap;
    if (m > m_cap) m = m_cap;
    if (idx >= m * c) return;
    int64_t v = idx / c;
    int k = (int)(idx - v * c);
    const float *p = voxels + v * t * c + k;
    float s = 0.f;
    for (int j = 0; j < t; ++j) s += p[(int64_t)j * c];
    int np = num_pts[v];
    out[idx] = s / (float)(np < 1 ? 1 : np);
}

// ---------------------------------------------------------------------------------------------
// PillarVFE: one wave per voxel.  LDS per wave: two [T][CMAX] fp32 feature planes (ping-pong).
// ---------------------------------------------------------------------------------------------
template <int TMAX>
__global__ void __launch_bounds__(256) k_pillar_vfe(const float *__restrict__ voxels, const int32_t *__restrict__ num_pts,
                                                    const int32_t *__restrict__ coords, int64_t m_cap,
                                                    const int32_t *__restrict__ n_live, int T, int c, int cmax,
                                                    PfnParams P, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int64_t m = n_live ? (int64_t)*n_live : m_cap;
    if (m > m_cap) m = m_cap;
    const int64_t v = (int64_t)blockIdx.x * 4 + wid;
    if (v >= m) return;  // whole wave exits together; no block barrier is used below
    float *bufA = smem + (size_t)wid * 2 * T * cmax;
    float *bufB = bufA + (size_t)T * cmax;
    float *raw = bufB;  // raw points [T][c] staged in plane B first

    const int np = num_pts[v];
    for (int e = lane; e < T * c; e += 64) raw[e] = voxels[v * T * c + e];
    wave_sync();
    // mean over ALL T slots / num_points (pillar_vfe.py:97: padding is zero, no clamp)
    float mean[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float s = 0.f;
        for (int j = 0; j < T; ++j) s += raw[j * c + k];
        mean[k] = s / (float)np;
    }
    const int4 co = reinterpret_cast<const int4 *>(coords)[v];  // (b, z, y, x)
    const float ctr[3] = {(float)co.w * P.vs[0] + P.off[0], (float)co.z * P.vs[1] + P.off[1],
                          (float)co.y * P.vs[2] + P.off[2]};
    const bool abs_xyz = P.flags & 1, with_dist = P.flags & 2;
    const int cin0 = P.cin[0];
    const int nbase = abs_xyz ? c : c - 3;
    for (int e = lane; e < T * cin0; e += 64) {
        const int j = e / cin0, f = e - j * cin0;
        const float *p = raw + j * c;
        float val;
        if (f < nbase) val = abs_xyz ? p[f] : p[f + 3];
        else if (f < nbase + 3) val = p[f - nbase] - mean[f - nbase];
        else if (f < nbase + 6) val = p[f - nbase - 3] - ctr[f - nbase - 3];
        else val = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        (void)with_dist;
        bufA[j * cmax + f] = (j < np) ? val : 0.f;  // padding mask (pillar_vfe.py:117-120)
    }
    wave_sync();

    float *in = bufA, *outp = bufB;
    for (int l = 0; l < P.n_layers; ++l) {
        const int cin = P.cin[l], cout = P.cout[l];
        const bool last = (l == P.n_layers - 1);
        for (int c0 = 0; c0 < cout; c0 += 64) {
            const int ch = c0 + lane;
            const bool act = ch < cout;
            float acc[TMAX];
#pragma unroll
            for (int j = 0; j < TMAX; ++j) acc[j] = 0.f;
            const float *wrow = P.w[l] + (size_t)(act ? ch : 0) * cin;
            for (int k = 0; k < cin; ++k) {
                const float wk = wrow[k];
#pragma unroll
                for (int j = 0; j < TMAX; ++j)
                    if (j < T) acc[j] = fmaf(in[j * cmax + k], wk, acc[j]);
            }
            const float sc = act ? P.scale[l][ch] : 0.f, sh = act ? P.shift[l][ch] : 0.f;
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < TMAX; ++j)
                if (j < T) {
                    float y = fmaxf(acc[j] * sc + sh, 0.f);
                    acc[j] = y;
                    mx = fmaxf(mx, y);
                }
            if (act) {
                if (last) {
                    out[v * cout + ch] = mx;
                } else {
                    // concat [x, x_max repeated] (pillar_vfe.py:46-49)
#pragma unroll
                    for (int j = 0; j < TMAX; ++j)
                        if (j < T) { outp[j * cmax + ch] = acc[j]; outp[j * cmax + cout + ch] = mx; }
                }
            }
        }
        wave_sync();
        float *tmp = in; in = outp; outp = tmp;
    }
}

// ---------------------------------------------------------------------------------------------
// PillarVFE fast path: ONE PFN layer (the PointPillars default), 4-channel points, T <= 32, cin <= 12, cout <= 64.
// The generic kernel above walks a dependent chain per wave (80-float staging loop, 60 serial LDS adds for the mean,
// a runtime division per feature, one global weight load per k inside the FMA loop): ~350 us for 4 scenes.  Here lane j
// loads point j as one float4, the mean is a 5-step wave reduction, each lane builds its point's feature row in
// registers (no index arithmetic), the lane's weight row is fetched up front, and the FMA loop reads the feature rows
// back as 16-byte LDS broadcasts.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pillar_vfe1(const float4 *__restrict__ voxels, const int32_t *__restrict__ num_pts,
                                                     const int32_t *__restrict__ coords, int64_t m_cap,
                                                     const int32_t *__restrict__ n_live, int T, PfnParams P,
                                                     float *__restrict__ out) {
    constexpr int FP = 12;                                   // padded feature count
    // feature-major staging: feat[w][k][j] = feature k of point j, so one 16-byte broadcast read brings feature k of FOUR points and
    // the products run as packed fp32 FMAs (two points per instruction, the weight broadcast by op_sel).  The kernel is bound by
    // vector issue (64 channels x T points x cin products per pillar): ~260 -> ~160 vector instructions per pillar
    __shared__ __attribute__((aligned(16))) float feat[4][FP][32];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int64_t m = n_live ? (int64_t)*n_live : m_cap;
    if (m > m_cap) m = m_cap;
    constexpr int PPW = 8;                                   // pillars per wave: the lane's weight row, scale and shift are fetched once for all of them
    const int64_t v_first = ((int64_t)blockIdx.x * 4 + wid) * PPW;
    if (v_first >= m) return;                                // whole wave exits together; no block barrier below
    const int cin = P.cin[0], cout = P.cout[0];
    const bool abs_xyz = P.flags & 1;
    // this lane's output channel: weight row + folded BatchNorm, requested before anything depends on them
    const bool act = lane < cout;
    float wreg[FP];
    {
        const float *wrow = P.w[0] + (size_t)(act ? lane : 0) * cin;
#pragma unroll
        for (int k = 0; k < FP; ++k) wreg[k] = (k < cin) ? wrow[k] : 0.f;
    }
    const float sc = act ? P.scale[0][lane] : 0.f, sh = act ? P.shift[0][lane] : 0.f;
    for (int64_t v = v_first; v < v_first + PPW && v < m; ++v) {
    const int np = num_pts[v];
    const int4 co = reinterpret_cast<const int4 *>(coords)[v];   // (b, z, y, x)
    float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < T) pt = voxels[v * T + lane];
    // mean over ALL T slots / num_points (pillar_vfe.py:97: padding is zero, no clamp)
    float sx = pt.x, sy = pt.y, sz = pt.z;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
    const float inv_np = (float)np;
    const float mx_ = sx / inv_np, my_ = sy / inv_np, mz_ = sz / inv_np;
    if (lane < 32) {
        float f[FP];
#pragma unroll
        for (int k = 0; k < FP; ++k) f[k] = 0.f;
        int q = 0;
        if (abs_xyz) { f[0] = pt.x; f[1] = pt.y; f[2] = pt.z; f[3] = pt.w; q = 4; }
        else         { f[0] = pt.w; q = 1; }
        const float cl[3] = {pt.x - mx_, pt.y - my_, pt.z - mz_};
        const float ce[3] = {pt.x - ((float)co.w * P.vs[0] + P.off[0]), pt.y - ((float)co.z * P.vs[1] + P.off[1]),
                             pt.z - ((float)co.y * P.vs[2] + P.off[2])};
        const float dist = sqrtf(pt.x * pt.x + pt.y * pt.y + pt.z * pt.z);
        // q is 4 or 1: both layouts are written with static register indices
        if (abs_xyz) { f[4] = cl[0]; f[5] = cl[1]; f[6] = cl[2]; f[7] = ce[0]; f[8] = ce[1]; f[9] = ce[2]; if (P.flags & 2) f[10] = dist; }
        else         { f[1] = cl[0]; f[2] = cl[1]; f[3] = cl[2]; f[4] = ce[0]; f[5] = ce[1]; f[6] = ce[2]; if (P.flags & 2) f[7] = dist; }
        (void)q;
        const bool live = lane < np && lane < T;              // padding mask (pillar_vfe.py:117-120)
#pragma unroll
        for (int k = 0; k < FP; ++k) feat[wid][k][lane] = live ? f[k] : 0.f;
    }
    wave_sync();
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float mx = -INFINITY;
    // The padding slots (np .. T - 1) carry all-zero feature rows (pillar_vfe.py:117-120), so each of them yields the same value per channel:
    // the accumulator stays +0 and y = (+0) * scale + shift.  It is taken once, with the same two operations, instead of T - np times --
    // a LiDAR pillar holds 3-4 points on average against T = 20 or 32 slots (the loop below was 3 rounds of 8 slots for every pillar).
    const int jmax = np < T ? np : T;                        // (wave-uniform: one pillar per wave)
    if (jmax < T) {
        const float zero = 0.f;
        const float ypad = zero * sc + sh;
        mx = fmaxf(ypad, 0.f);
    }
    // per point: acc = sum_k f[k] * w[k] in ascending k from 0 (the order of the scalar form: bit-identical), then BN + ReLU + max
    auto quads = [&](auto kk_tag) __attribute__((always_inline)) {
        constexpr int KK = decltype(kk_tag)::value;
        const f32x2 sc2 = {sc, sc}, sh2 = {sh, sh};
        for (int j0 = 0; j0 < jmax; j0 += 8) {                // two quads per round: four independent accumulation chains
            f32x2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                const float4 p0 = *reinterpret_cast<const float4 *>(&feat[wid][k][j0]);
                const float4 p1 = *reinterpret_cast<const float4 *>(&feat[wid][k][(j0 + 4) & 31]);
                const f32x2 w2 = {wreg[k], wreg[k]};
                acc[0] = __builtin_elementwise_fma(f32x2{p0.x, p0.y}, w2, acc[0]);
                acc[1] = __builtin_elementwise_fma(f32x2{p0.z, p0.w}, w2, acc[1]);
                acc[2] = __builtin_elementwise_fma(f32x2{p1.x, p1.y}, w2, acc[2]);
                acc[3] = __builtin_elementwise_fma(f32x2{p1.z, p1.w}, w2, acc[3]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 y = acc[q] * sc2 + sh2;               // (mul, then add: -ffp-contract=off, as the scalar form)
                if (j0 + 2 * q < jmax) mx = fmaxf(mx, fmaxf(y[0], 0.f));
                if (j0 + 2 * q + 1 < jmax) mx = fmaxf(mx, fmaxf(y[1], 0.f));
            }
        }
    };
    if (cin <= 10) quads(std::integral_constant<int, 10>{});
    else           quads(std::integral_constant<int, 12>{});
    if (act) out[v * cout + lane] = mx;
    wave_sync();                                             // the next pillar's feature rows overwrite this one's
    }
}

// ---------------------------------------------------------------------------------------------
// scatter_mean
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scatter_add(const float *__restrict__ pts, int64_t n, int c, int col0, int nc,
                                                     const int32_t *__restrict__ inv, float *__restrict__ sums) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * nc) return;
    int64_t i = idx / nc;
    int k = (int)(idx - i * nc);
    int v = inv[i];
    if (v < 0) return;
    atomicAdd(&sums[(int64_t)v * nc + k], pts[i * c + col0 + k]);
}

__global__ void __launch_bounds__(256) k_div_count(const float *__restrict__ sums, const int32_t *__restrict__ cnt,
                                                   int64_t m_cap, int nc, float *__restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= m_cap * nc) return;
    int k = cnt[idx / nc];
    out[idx] = sums[idx] / (float)(k < 1 ? 1 : k);
}

// ---------------------------------------------------------------------------------------------
// dynamic PFN (PFNLayerV2): one wave per strip of 64 points, lanes = output channels.
//   STAGE 0: first layer of a 2-layer net -> atomic max into xmax_tmp
//   STAGE 1: last layer (input = augmented features for 1-layer nets, [x1, xmax1[inv]] for 2-layer)
// ---------------------------------------------------------------------------------------------
constexpr int DYN_MAXF = 16;  // augmented feature count: c-1 (+3 cluster) +3 centre (+1 dist) <= 16

__device__ __forceinline__ void atomic_max_nonneg(float *addr, float v) {
    atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
}

template <int STAGE>
__global__ void __launch_bounds__(256) k_dynamic_pfn(const float *__restrict__ pts, int n, int c,
                                                     const int32_t *__restrict__ inv, const int32_t *__restrict__ pcoord,
                                                     const float *__restrict__ pmean, int kind, PfnParams P,
                                                     float *__restrict__ xmax_tmp, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int cin0 = P.cin[0], cout0 = P.cout[0];
    const int two = (P.n_layers == 2);
    const int cin1 = two ? P.cin[1] : 0, cout1 = two ? P.cout[1] : 0;
    float *feat = smem + (size_t)wid * (64 * DYN_MAXF + (two ? 64 * cin1 : 0));  // [64][DYN_MAXF]
    float *l2in = feat + 64 * DYN_MAXF;                                            // [64][cin1] (2-layer only)
    const int i0 = (blockIdx.x * (blockDim.x >> 6) + wid) * 64;
    if (i0 >= n) return;
    const int i = i0 + lane;
    const bool abs_xyz = P.flags & 1, with_dist = P.flags & 2;
    int myv = -1;
    if (i < n) myv = inv[i];
    // ---- lane = point: build the augmented feature row (dynamic_pillar_vfe.py:105-123) ----
    {
        float f[DYN_MAXF];
#pragma unroll
        for (int k = 0; k < DYN_MAXF; ++k) f[k] = 0.f;
        if (myv >= 0) {
            const float *p = pts + (int64_t)i * c;
            const float x = p[1], y = p[2], z = p[3];
            const int cx = pcoord[i * 3], cy = pcoord[i * 3 + 1], cz = pcoord[i * 3 + 2];
            float fc[3];
            fc[0] = x - ((float)cx * P.vs[0] + P.off[0]);
            fc[1] = y - ((float)cy * P.vs[1] + P.off[1]);
            fc[2] = (kind == 1) ? z - ((float)cz * P.vs[2] + P.off[2]) : z - P.off[2];
            int k = 0;
            if (kind == 2) {  // simple2d: [f_center, points[:,1:] | points[:,4:]] (dynamic_pillar_vfe.py:210-219)
                f[k++] = fc[0]; f[k++] = fc[1]; f[k++] = fc[2];
                for (int q = abs_xyz ? 1 : 4; q < c && k < DYN_MAXF; ++q) f[k++] = p[q];
            } else {
                for (int q = abs_xyz ? 1 : 4; q < c && k < DYN_MAXF; ++q) f[k++] = p[q];
                const float *mu = pmean + (int64_t)myv * 3;
                f[k++] = x - mu[0]; f[k++] = y - mu[1]; f[k++] = z - mu[2];
                f[k++] = fc[0]; f[k++] = fc[1]; f[k++] = fc[2];
            }
            if (with_dist && k < DYN_MAXF) f[k++] = sqrtf(x * x + y * y + z * z);
        }
#pragma unroll
        for (int k = 0; k < DYN_MAXF; ++k) feat[lane * DYN_MAXF + k] = f[k];
    }
    wave_sync();
    const int npts = (n - i0) < 64 ? (n - i0) : 64;

    // ---- lane = channel: first layer for every point of the strip ----
    for (int c0 = 0; c0 < cout0; c0 += 64) {
        const int ch = c0 + lane;
        const bool act = ch < cout0;
        float wreg[DYN_MAXF];
#pragma unroll
        for (int k = 0; k < DYN_MAXF; ++k) wreg[k] = (act && k < cin0) ? P.w[0][(size_t)ch * cin0 + k] : 0.f;
        const float sc = act ? P.scale[0][ch] : 0.f, sh = act ? P.shift[0][ch] : 0.f;
        for (int p = 0; p < npts; ++p) {
            const int v = __shfl(myv, p);
            if (v < 0) continue;  // wave-uniform
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < DYN_MAXF; ++k) acc = fmaf(feat[p * DYN_MAXF + k], wreg[k], acc);
            const float y = fmaxf(acc * sc + sh, 0.f);
            if (!act) continue;
            if (!two) {
                if (STAGE == 1) atomic_max_nonneg(&out[(int64_t)v * cout0 + ch], y);
            } else if (STAGE == 0) {
                atomic_max_nonneg(&xmax_tmp[(int64_t)v * cout0 + ch], y);
            } else {
                l2in[p * cin1 + ch] = y;                                        // x
                l2in[p * cin1 + cout0 + ch] = xmax_tmp[(int64_t)v * cout0 + ch];  // x_max[unq_inv]
            }
        }
    }
    if (!two || STAGE == 0) return;
    wave_sync();
    // ---- second (last) layer: 8 points per register block to amortise the weight loads ----
    for (int c0 = 0; c0 < cout1; c0 += 64) {
        const int ch = c0 + lane;
        const bool act = ch < cout1;
        const float *wrow = P.w[1] + (size_t)(act ? ch : 0) * cin1;
        const float sc = act ? P.scale[1][ch] : 0.f, sh = act ? P.shift[1][ch] : 0.f;
        for (int p0 = 0; p0 < npts; p0 += 8) {
            float acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = 0.f;
            for (int k = 0; k < cin1; ++k) {
                const float wk = wrow[k];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(l2in[((p0 + q) & 63) * cin1 + k], wk, acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int p = p0 + q;
                const int v = __shfl(myv, p & 63);
                if (p < npts && v >= 0 && act) atomic_max_nonneg(&out[(int64_t)v * cout1 + ch], fmaxf(acc[q] * sc + sh, 0.f));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// PointPillarScatter
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pillar_scatter(const float *__restrict__ feat, const int32_t *__restrict__ coords,
                                                        int64_t m_cap, const int32_t *__restrict__ n_live, int ch,
                                                        int batch, int ny, int nx, float *__restrict__ canvas) {
    // lanes run over pillars for a fixed channel block: stores of neighbouring pillars often share lines
    int64_t m = n_live ? (int64_t)*n_live : m_cap;
    if (m > m_cap) m = m_cap;
    int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= m) return;
    const int4 co = reinterpret_cast<const int4 *>(coords)[v];
    if (co.x < 0 || co.x >= batch || co.z < 0 || co.z >= ny || co.w < 0 || co.w >= nx) return;
    // index = z + y*nx + x with nz == 1 (pointpillar_scatter.py:27)
    const int64_t cell = (int64_t)co.y + (int64_t)co.z * nx + co.w;
    if (cell < 0 || cell >= (int64_t)ny * nx) return;
    float *dst = canvas + (int64_t)co.x * ch * ny * nx + cell;
    const float *src = feat + v * ch;
    for (int k = 0; k < ch; ++k) dst[(int64_t)k * ny * nx] = src[k];
}

int fill_params(PfnParams &P, int n_layers, const float *const *w, const float *const *scale, const float *const *shift,
                const int32_t *cin, const int32_t *cout, int flags, const float *vs, const float *off) {
    if (n_layers < 1 || n_layers > MAX_LAYERS || !w || !scale || !shift || !cin || !cout || !vs || !off) return LVQ_EINVAL;
    P.n_layers = n_layers;
    P.flags = flags;
    for (int l = 0; l < n_layers; ++l) {
        if (!w[l] || !scale[l] || !shift[l] || cin[l] <= 0 || cout[l] <= 0) return LVQ_EINVAL;
        if (cin[l] > 512 || cout[l] > 256) return LVQ_EUNSUPPORTED;
        P.w[l] = w[l]; P.scale[l] = scale[l]; P.shift[l] = shift[l]; P.cin[l] = cin[l]; P.cout[l] = cout[l];
    }
    for (int j = 0; j < 3; ++j) { P.vs[j] = vs[j]; P.off[j] = off[j]; }
    return LVQ_OK;
}

}  // namespace

extern "C" int lvq_mean_vfe(const float *voxels, const int32_t *num_pts, int64_t m_cap, const int32_t *n_voxels_dev,
                            int t, int c, float *out, lvq_stream_t stream) {
    if (m_cap < 0 || t <= 0 || c <= 0) return LVQ_EINVAL;
    if (m_cap == 0) return LVQ_OK;
    if (!voxels || !num_pts || !out) return LVQ_EINVAL;
    hipLaunchKernelGGL(k_mean_vfe, dim3((unsigned)lvq_cdiv(m_cap * c, 256)), dim3(256), 0, lvq_s(stream), voxels, num_pts,
                       m_cap, n_voxels_dev, t, c, out);
    return lvq_launch_status();
}

extern "C" int lvq_pillar_vfe(const float *voxels, const int32_t *num_pts, const int32_t *coords_bzyx, int64_t m_cap,
                              const int32_t *n_voxels_dev, int t, int c, int n_layers, const float *const *w_host,
                              const float *const *scale_host, const float *const *shift_host, const int32_t *cin_host,
                              const int32_t *cout_host, int flags, const float *vsize_host, const float *offset_host,
                              float *out, lvq_stream_t stream) {
    if (m_cap < 0 || t <= 0 || c < 3) return LVQ_EINVAL;
    PfnParams P;
    int rc = fill_params(P, n_layers, w_host, scale_host, shift_host, cin_host, cout_host, flags, vsize_host, offset_host);
    if (rc != LVQ_OK) return rc;
    if (t > 64) return LVQ_EUNSUPPORTED;
    const int expect = ((flags & 1) ? c : c - 3) + 6 + ((flags & 2) ? 1 : 0);
    if (P.cin[0] != expect) return LVQ_EINVAL;
    int cmax = c;
    for (int l = 0; l < n_layers; ++l) {
        if (P.cin[l] > cmax) cmax = P.cin[l];
        int wdt = (l == n_layers - 1) ? P.cout[l] : 2 * P.cout[l];
        if (wdt > cmax) cmax = wdt;
        if (l > 0 && P.cin[l] != 2 * P.cout[l - 1]) return LVQ_EINVAL;
    }
    if (m_cap == 0) return LVQ_OK;
    if (!voxels || !num_pts || !coords_bzyx || !out) return LVQ_EINVAL;
    size_t lds = (size_t)4 * 2 * t * cmax * sizeof(float);
    if (lds > 160 * 1024) return LVQ_EUNSUPPORTED;
    dim3 grid((unsigned)lvq_cdiv(m_cap, 4)), block(256);
    if (n_layers == 1 && c == 4 && t <= 32 && P.cin[0] <= 12 && P.cout[0] <= 64 && !(((uintptr_t)voxels) & 15) &&
        !lvq_tune().pillar_vfe_generic) {
        // 4 waves x 8 pillars per workgroup
        hipLaunchKernelGGL(k_pillar_vfe1, dim3((unsigned)lvq_cdiv(m_cap, 32)), block, 0, lvq_s(stream), reinterpret_cast<const float4 *>(voxels), num_pts, coords_bzyx,
                           m_cap, n_voxels_dev, t, P, out);
        return lvq_launch_status();
    }
    if (t <= 32) {
        if (lds > 64 * 1024) hipFuncSetAttribute((const void *)k_pillar_vfe<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_pillar_vfe<32>, grid, block, lds, lvq_s(stream), voxels, num_pts, coords_bzyx, m_cap,
                           n_voxels_dev, t, c, cmax, P, out);
    } else {
        if (lds > 64 * 1024) hipFuncSetAttribute((const void *)k_pillar_vfe<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_pillar_vfe<64>, grid, block, lds, lvq_s(stream), voxels, num_pts, coords_bzyx, m_cap,
                           n_voxels_dev, t, c, cmax, P, out);
    }
    return lvq_launch_status();
}

extern "C" int lvq_scatter_mean(const float *pts, int64_t n, int c, int col0, int nc, const int32_t *unq_inv,
                                const int32_t *unq_cnt, int64_t m_cap, float *sums, float *out, lvq_stream_t stream) {
    if (n < 0 || c <= 0 || col0 < 0 || nc <= 0 || col0 + nc > c || m_cap < 0) return LVQ_EINVAL;
    if (n == 0 || m_cap == 0) return LVQ_OK;
    if (!pts || !unq_inv || !unq_cnt || !sums || !out) return LVQ_EINVAL;
    hipLaunchKernelGGL(k_scatter_add, dim3((unsigned)lvq_cdiv(n * nc, 256)), dim3(256), 0, lvq_s(stream), pts, n, c, col0,
                       nc, unq_inv, sums);
    hipLaunchKernelGGL(k_div_count, dim3((unsigned)lvq_cdiv(m_cap * nc, 256)), dim3(256), 0, lvq_s(stream), sums, unq_cnt,
                       m_cap, nc, out);
    return lvq_launch_status();
}

extern "C" int lvq_dynamic_pfn(const float *pts, int64_t n, int c, const int32_t *unq_inv, const int32_t *pt_coords,
                               const float *points_mean, int kind, int n_layers, const float *const *w_host,
                               const float *const *scale_host, const float *const *shift_host, const int32_t *cin_host,
                               const int32_t *cout_host, int flags, const float *vsize_host, const float *offset_host,
                               float *xmax_tmp, float *out, lvq_stream_t stream) {
    if (n < 0 || c < 4 || kind < 0 || kind > 2) return LVQ_EINVAL;
    PfnParams P;
    int rc = fill_params(P, n_layers, w_host, scale_host, shift_host, cin_host, cout_host, flags, vsize_host, offset_host);
    if (rc != LVQ_OK) return rc;
    if (n_layers > 2) return LVQ_EUNSUPPORTED;
    const int base = (flags & 1) ? c - 1 : c - 4;
    const int expect = base + (kind == 2 ? 3 : 6) + ((flags & 2) ? 1 : 0);
    if (P.cin[0] != expect || expect > DYN_MAXF) return expect > DYN_MAXF ? LVQ_EUNSUPPORTED : LVQ_EINVAL;
    if (n_layers == 2 && P.cin[1] != 2 * P.cout[0]) return LVQ_EINVAL;
    if (n == 0) return LVQ_OK;
    if (!pts || !unq_inv || !pt_coords || !out || (kind != 2 && !points_mean) || (n_layers == 2 && !xmax_tmp))
        return LVQ_EINVAL;
    if (n >= (1ll << 30)) return LVQ_EUNSUPPORTED;
    const size_t per_wave = (size_t)(64 * DYN_MAXF + (n_layers == 2 ? 64 * P.cin[1] : 0)) * sizeof(float);
    int wpb = 4;
    while (wpb > 1 && per_wave * wpb > 64 * 1024) wpb >>= 1;
    const size_t lds = per_wave * wpb;
    if (lds > 160 * 1024) return LVQ_EUNSUPPORTED;
    dim3 grid((unsigned)lvq_cdiv(n, 64 * wpb)), block(64 * wpb);
    if (n_layers == 2) {
        if (lds > 64 * 1024) {
            hipFuncSetAttribute((const void *)k_dynamic_pfn<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void *)k_dynamic_pfn<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        hipLaunchKernelGGL(k_dynamic_pfn<0>, grid, block, lds, lvq_s(stream), pts, (int)n, c, unq_inv, pt_coords, points_mean,
                           kind, P, xmax_tmp, out);
    }
    hipLaunchKernelGGL(k_dynamic_pfn<1>, grid, block, lds, lvq_s(stream), pts, (int)n, c, unq_inv, pt_coords, points_mean,
                       kind, P, xmax_tmp, out);
    return lvq_launch_status();
}

extern "C" int lvq_pillar_scatter(const float *feat, const int32_t *coords_bzyx, int64_t m_cap, const int32_t *n_voxels_dev,
                                  int ch, int batch, int ny, int nx, float *canvas, lvq_stream_t stream) {
    if (m_cap < 0 || ch <= 0 || batch <= 0 || ny <= 0 || nx <= 0 || !canvas) return LVQ_EINVAL;
    hipStream_t st = lvq_s(stream);
    if (hipMemsetAsync(canvas, 0, sizeof(float) * (size_t)batch * ch * ny * nx, st) != hipSuccess) return LVQ_ELAUNCH;
    if (m_cap == 0) return LVQ_OK;
    if (!feat || !coords_bzyx) return LVQ_EINVAL;
    hipLaunchKernelGGL(k_pillar_scatter, dim3((unsigned)lvq_cdiv(m_cap, 256)), dim3(256), 0, st, feat, coords_bzyx, m_cap,
                       n_voxels_dev, ch, batch, ny, nx, canvas);
    return lvq_launch_status();
}
