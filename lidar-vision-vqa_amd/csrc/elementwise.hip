// csrc/elementwise.hip -- the memory-bound glue around the MFMA kernels (fp32 statistics, bf16 hand-off).
//
//   lvq_layernorm       nn.LayerNorm (vat_blocks.py:19,23,27; vat_lidar.py:89,114,116; vision_adapter.py:56,122-125)
//   lvq_dwconv3x3_gelu  VATLiDAR.refine (vat_lidar.py:82-85): depthwise 3x3 + exact GELU, NCHW -> token-major bf16
//   lvq_scale_add_rows  query + view_embed / prefix * prefix_scale (vat_lidar.py:259-270; trainer.py:581,594)
//   lvq_rmsnorm / lvq_rope_inplace / lvq_swiglu / lvq_cross_entropy   stand-in decoder head (validation.py:146-156)
//
// Every kernel writes the bf16 operand (and, for the bf16x3 mode, its lo residual) that the NEXT MFMA
// kernel consumes, so normalised activations never make an fp32 round trip through HBM.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ void put_out(float y, int64_t o, float *y32, uint16_t *y16, uint16_t *ylo) {
    if (y32) y32[o] = y;
    if (y16) {
        const uint16_t h = f32_to_bf16(y);
        y16[o] = h;
        if (ylo) ylo[o] = f32_to_bf16(y - bf16_to_f32(h));
    }
}

// one wave per row; rows up to 2048 floats live in registers (one pass over memory -- the three dependent passes of the first
// version cost 12.5 us per call on a one-row decode step), longer rows re-read from L1/L2
template <bool RMS>
__global__ void __launch_bounds__(256) k_norm(const float *__restrict__ x, const float *__restrict__ add, int add_rows,
                                              int add_group, const float *__restrict__ gamma, const float *__restrict__ beta,
                                              float eps, int64_t rows, int d, const float *__restrict__ post,
                                              int64_t post_rows, float *__restrict__ y32,
                                              uint16_t *__restrict__ y16, uint16_t *__restrict__ ylo) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *xr = x + row * d;
    const float *ar = add ? add + ((row / add_group) % add_rows) * (int64_t)d : nullptr;
    const float *pr = post ? post + (row % post_rows) * (int64_t)d : nullptr;
    if (d <= 2048) {
        float r[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int k = lane + 64 * j;
            r[j] = k < d ? xr[k] + (ar ? ar[k] : 0.f) : 0.f;
        }
        float mean = 0.f;
        if (!RMS) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 32; ++j) s += r[j];                  // padding slots hold 0
            mean = wave_sum(s) / (float)d;
        }
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float t = (lane + 64 * j < d) ? r[j] - mean : 0.f;
            v += t * t;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)d + eps);
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int k = lane + 64 * j;
            if (k < d) {
                const float t = (r[j] - mean) * rstd;
                float y = RMS ? gamma[k] * t : t * gamma[k] + (beta ? beta[k] : 0.f);
                if (pr) y += pr[k];
                put_out(y, row * d + k, y32, y16, ylo);
            }
        }
        return;
    }
    float mean = 0.f;
    if (!RMS) {
        float s = 0.f;
        for (int k = lane; k < d; k += 64) s += xr[k] + (ar ? ar[k] : 0.f);
        mean = wave_sum(s) / (float)d;
    }
    float v = 0.f;
    for (int k = lane; k < d; k += 64) {
        const float t = xr[k] + (ar ? ar[k] : 0.f) - mean;
        v += t * t;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)d + eps);
    for (int k = lane; k < d; k += 64) {
        const float t = (xr[k] + (ar ? ar[k] : 0.f) - mean) * rstd;
        float y = RMS ? gamma[k] * t : t * gamma[k] + (beta ? beta[k] : 0.f);
        if (pr) y += pr[k];
        put_out(y, row * d + k, y32, y16, ylo);
    }
}

// d % 256 == 0 and d <= 2048: the row lives in registers (NV float4 per lane), one pass over HBM, 16-byte accesses
template <bool RMS, int NV>
__global__ void __launch_bounds__(256) k_norm_vec(const float *__restrict__ x, const float *__restrict__ add, int add_rows,
                                                  int add_group, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                  float eps, int64_t rows, int d, const float *__restrict__ post,
                                                  int64_t post_rows, float *__restrict__ y32,
                                                  uint16_t *__restrict__ y16, uint16_t *__restrict__ ylo) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 *xr = reinterpret_cast<const float4 *>(x + row * d);
    const float4 *ar = add ? reinterpret_cast<const float4 *>(add + ((row / add_group) % add_rows) * (int64_t)d) : nullptr;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = xr[i * 64 + lane];
        if (ar) { const float4 t = ar[i * 64 + lane]; v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w; }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = RMS ? 0.f : wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    const float4 *pr = post ? reinterpret_cast<const float4 *>(post + (row % post_rows) * (int64_t)d) : nullptr;
    const float4 *g4 = reinterpret_cast<const float4 *>(gamma), *b4 = reinterpret_cast<const float4 *>(beta);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        const float4 gg = g4[c];
        float4 y = make_float4(v[i].x * rstd * gg.x, v[i].y * rstd * gg.y, v[i].z * rstd * gg.z, v[i].w * rstd * gg.w);
        if (!RMS && beta) { const float4 bb = b4[c]; y.x += bb.x; y.y += bb.y; y.z += bb.z; y.w += bb.w; }
        if (pr) { const float4 pp = pr[c]; y.x += pp.x; y.y += pp.y; y.z += pp.z; y.w += pp.w; }
        const int64_t o = row * d + c * 4;
        if (y32) *reinterpret_cast<float4 *>(y32 + o) = y;
        if (y16) {
            const ushort4 hh = make_ushort4(f32_to_bf16(y.x), f32_to_bf16(y.y), f32_to_bf16(y.z), f32_to_bf16(y.w));
            *reinterpret_cast<ushort4 *>(y16 + o) = hh;
            if (ylo) *reinterpret_cast<ushort4 *>(ylo + o) = make_ushort4(f32_to_bf16(y.x - bf16_to_f32(hh.x)), f32_to_bf16(y.y - bf16_to_f32(hh.y)),
                                                                         f32_to_bf16(y.z - bf16_to_f32(hh.z)), f32_to_bf16(y.w - bf16_to_f32(hh.w)));
        }
    }
}

template <bool RMS>
bool launch_norm_vec(const float *x, const float *add, int add_rows, int add_group, const float *gamma, const float *beta, float eps,
                     int64_t rows, int d, const float *post, int64_t post_rows, float *y32, uint16_t *y16, uint16_t *ylo, hipStream_t st) {
    if (d % 256 || d > 2048) return false;
    if (((uintptr_t)x | (uintptr_t)add | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)post | (uintptr_t)y32) & 15) return false;
    if (((uintptr_t)y16 | (uintptr_t)ylo) & 7) return false;
    dim3 grid((unsigned)lvq_cdiv(rows, 4)), block(256);
#define LVQ_NV(N) case N: hipLaunchKernelGGL((k_norm_vec<RMS, N>), grid, block, 0, st, x, add, add_rows, add_group, gamma, beta, eps, rows, d, post, post_rows, y32, y16, ylo); return true;
    switch (d / 256) { LVQ_NV(1) LVQ_NV(2) LVQ_NV(3) LVQ_NV(4) LVQ_NV(5) LVQ_NV(6) LVQ_NV(7) LVQ_NV(8) }
#undef LVQ_NV
    return false;
}

// depthwise 3x3 (pad 1) + GELU, NCHW fp32 -> token-major bf16.  Block = 64 channels x 4 rows x 64 pixels, one pixel
// column per lane: a wave takes 16 of the channels one after the other and slides a 3-row register window down the
// 6 input rows, reading each row as three coalesced 256-byte loads (x-1, x, x+1: the overlap is an L1 hit, HBM sees the
// halo rows only -- 1.5x instead of the 2.1x of an LDS-staged 4 x 34 tile, and no per-element index arithmetic).  The
// [pixel][channel] outputs leave through LDS as whole 128-byte token rows (16-byte stores); the lo plane only in bf16x3.
constexpr int DW_C = 64, DW_RY = 4, DW_PX = 64;

template <bool LO>
__global__ void __launch_bounds__(256) k_dwconv3x3_gelu(const float *__restrict__ bev, const float *__restrict__ w9,
                                                        const float *__restrict__ bias, int C, int H, int W,
                                                        uint16_t *__restrict__ thi, uint16_t *__restrict__ tlo) {
    __shared__ __attribute__((aligned(16))) uint16_t oh[DW_RY][DW_PX][DW_C + 8];   // +16 B pad per pixel row (36 KB)
    __shared__ __attribute__((aligned(16))) uint16_t ol[LO ? DW_RY : 1][LO ? DW_PX : 1][DW_C + 8];
    const int x0 = blockIdx.x * DW_PX, y0 = blockIdx.y * DW_RY;
    const int cblocks = (C + DW_C - 1) / DW_C;
    const int b = blockIdx.z / cblocks, c0 = (blockIdx.z % cblocks) * DW_C;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gx = x0 + lane;
    const bool in_m = gx < W, in_l = gx - 1 >= 0 && gx - 1 < W, in_r = gx + 1 < W;
#pragma unroll 1
    for (int ci = 0; ci < DW_C / 4; ++ci) {
        const int c = wid * (DW_C / 4) + ci, gc = c0 + c;
        if (gc >= C) break;                                   // wave-uniform
        float wk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = w9[gc * 9 + k];
        const float bs = bias ? bias[gc] : 0.f;
        const float *plane = bev + ((int64_t)b * C + gc) * H * W;
        float win[3][3];
#pragma unroll
        for (int r = 0; r < DW_RY + 2; ++r) {
            const int gy = y0 + r - 1;
            const bool rin = gy >= 0 && gy < H;
            const float *rowp = plane + (int64_t)(rin ? gy : 0) * W + gx;
            float *wr = win[r % 3];
            wr[0] = (rin && in_l) ? rowp[-1] : 0.f;
            wr[1] = (rin && in_m) ? rowp[0] : 0.f;
            wr[2] = (rin && in_r) ? rowp[1] : 0.f;
            if (r >= 2) {
                const int ry = r - 2;
                float acc = bs;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc = fmaf(win[(ry + rr) % 3][k], wk[rr * 3 + k], acc);
                acc = gelu_erf(acc);
                const uint16_t h = f32_to_bf16(acc);
                oh[ry][lane][c] = h;
                if (LO) ol[ry][lane][c] = f32_to_bf16(acc - bf16_to_f32(h));
            }
        }
    }
    __syncthreads();
    const int nc = (C - c0) < DW_C ? (C - c0) : DW_C;            // channels of this block (multiple of 8)
    for (int e = tid; e < DW_RY * DW_PX * (DW_C / 8); e += 256) {
        const int ch = e & 7, px = (e >> 3) & (DW_PX - 1), ry = e / (8 * DW_PX);
        const int ox = x0 + px, gy = y0 + ry;
        if (ox < W && gy < H && ch * 8 < nc) {
            const int64_t o = ((int64_t)b * H * W + (int64_t)gy * W + ox) * C + c0 + ch * 8;
            *reinterpret_cast<uint4 *>(thi + o) = *reinterpret_cast<const uint4 *>(&oh[ry][px][ch * 8]);
            if (LO) *reinterpret_cast<uint4 *>(tlo + o) = *reinterpret_cast<const uint4 *>(&ol[ry][px][ch * 8]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Sparse BEV bridge: PointPillarScatter + depthwise 3x3 + GELU -> tokens WITHOUT the dense canvas.
// The pillar canvas is ~93 % zeros at nuScenes densities, and the dense route writes it (268 MB memset + scatter for 4 scenes),
// reads it back and only then produces the 134 MB of tokens.  Here a dense int32 INDEX map (pillar row or -1; 4 MB) is the
// only thing scattered; the conv kernel keeps each lane's 3 x 6 neighbourhood indices in registers for all 64 channels and
// gathers the few live taps from the [M, C] pillar features (L2-resident).  Tap order and fmaf chain are those of
// k_dwconv3x3_gelu and a zero tap leaves the accumulator unchanged exactly, so the tokens are bit-identical to the dense route.
// Blocks whose whole 6 x 66 neighbourhood is empty write the per-channel constant GELU(bias).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pillar_index(const int32_t *__restrict__ coords, int64_t m_cap, const int32_t *__restrict__ n_live,
                                                      int batch, int ny, int nx, int32_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t m = n_live ? (int64_t)*n_live : m_cap;
    if (m > m_cap) m = m_cap;
    if (i >= m) return;
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];      // (b, z, y, x), nz == 1
    if (c.x < 0 || c.x >= batch || c.z < 0 || c.z >= ny || c.w < 0 || c.w >= nx) return;
    idx[((int64_t)c.x * ny + c.z) * nx + c.w] = (int32_t)i;
}

template <bool LO>
__global__ void __launch_bounds__(256) k_pillar_dwconv_gelu(const float *__restrict__ feat, const int32_t *__restrict__ idx,
                                                            const float *__restrict__ w9, const float *__restrict__ bias,
                                                            int C, int H, int W, uint16_t *__restrict__ thi, uint16_t *__restrict__ tlo) {
    __shared__ __attribute__((aligned(16))) uint16_t oh[DW_RY][DW_PX][DW_C + 8];
    __shared__ __attribute__((aligned(16))) uint16_t ol[LO ? DW_RY : 1][LO ? DW_PX : 1][DW_C + 8];
    __shared__ int any_live;
    const int x0 = blockIdx.x * DW_PX, y0 = blockIdx.y * DW_RY;
    const int cblocks = (C + DW_C - 1) / DW_C;
    const int b = blockIdx.z / cblocks, c0 = (blockIdx.z % cblocks) * DW_C;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gx = x0 + lane;
    if (tid == 0) any_live = 0;
    __syncthreads();
    // this lane's neighbourhood: rows y0-1 .. y0+RY, columns gx-1, gx, gx+1 (-1 outside the image / empty)
    int nb[DW_RY + 2][3];
    bool mine = false;
#pragma unroll
    for (int r = 0; r < DW_RY + 2; ++r) {
        const int gy = y0 + r - 1;
        const bool rin = gy >= 0 && gy < H;
        const int32_t *rowp = idx + ((int64_t)b * H + (rin ? gy : 0)) * W;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int xx = gx + k - 1;
            nb[r][k] = (rin && xx >= 0 && xx < W) ? rowp[xx] : -1;
            mine = mine || nb[r][k] >= 0;
        }
    }
    if (wid == 0 && __any(mine)) any_live = 1;                  // every wave loads the same neighbourhood
    __syncthreads();
    if (!any_live) {
        // constant rows: GELU(bias[c]) for every pixel of the block
        for (int e = tid; e < DW_RY * DW_PX * DW_C; e += 256) {
            const int c = e % DW_C, px = (e / DW_C) % DW_PX, ry = e / (DW_C * DW_PX), gc = c0 + c;
            const float v = gc < C ? gelu_erf(bias ? bias[gc] : 0.f) : 0.f;
            const uint16_t h = f32_to_bf16(v);
            oh[ry][px][c] = h;
            if (LO) ol[ry][px][c] = f32_to_bf16(v - bf16_to_f32(h));
        }
    } else {
#pragma unroll 1
        for (int ci = 0; ci < DW_C / 4; ++ci) {
            const int c = wid * (DW_C / 4) + ci, gc = c0 + c;
            if (gc >= C) break;                                 // wave-uniform
            float wk[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[k] = w9[gc * 9 + k];
            const float bs = bias ? bias[gc] : 0.f;
            float win[3][3];
#pragma unroll
            for (int r = 0; r < DW_RY + 2; ++r) {
                float *wr = win[r % 3];
#pragma unroll
                for (int k = 0; k < 3; ++k) wr[k] = nb[r][k] >= 0 ? feat[(int64_t)nb[r][k] * C + gc] : 0.f;
                if (r >= 2) {
                    const int ry = r - 2;
                    float acc = bs;
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                        for (int k = 0; k < 3; ++k) acc = fmaf(win[(ry + rr) % 3][k], wk[rr * 3 + k], acc);
                    acc = gelu_erf(acc);
                    const uint16_t h = f32_to_bf16(acc);
                    oh[ry][lane][c] = h;
                    if (LO) ol[ry][lane][c] = f32_to_bf16(acc - bf16_to_f32(h));
                }
            }
        }
    }
    __syncthreads();
    const int nc = (C - c0) < DW_C ? (C - c0) : DW_C;
    for (int e = tid; e < DW_RY * DW_PX * (DW_C / 8); e += 256) {
        const int ch = e & 7, px = (e >> 3) & (DW_PX - 1), ry = e / (8 * DW_PX);
        const int ox = x0 + px, gy = y0 + ry;
        if (ox < W && gy < H && ch * 8 < nc) {
            const int64_t o = ((int64_t)b * H * W + (int64_t)gy * W + ox) * C + c0 + ch * 8;
            *reinterpret_cast<uint4 *>(thi + o) = *reinterpret_cast<const uint4 *>(&oh[ry][px][ch * 8]);
            if (LO) *reinterpret_cast<uint4 *>(tlo + o) = *reinterpret_cast<const uint4 *>(&ol[ry][px][ch * 8]);
        }
    }
}

__global__ void __launch_bounds__(256) k_scale_add_rows(const float *__restrict__ x, const float *__restrict__ add,
                                                        int64_t add_rows, float alpha, int64_t rows, int d,
                                                        float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * d) return;
    const int64_t r = i / d;
    const int k = (int)(i - r * d);
    float v = x[i] * alpha;
    if (add) v += add[(r % add_rows) * d + k];
    out[i] = v;
}

// rotate-half rotary embedding (transformers Qwen2): x[..., :h] , x[..., h:] with angle pos * theta^(-2i/dh)
__global__ void __launch_bounds__(256) k_rope(uint16_t *__restrict__ xh, uint16_t *__restrict__ xl, int64_t rows, int seq_len,
                                              int n_heads, int dh, int64_t ld, float theta, int pos0) {
    const int half = dh >> 1;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * n_heads * half) return;
    const int e = (int)(i % half);
    const int hd = (int)((i / half) % n_heads);
    const int64_t row = i / ((int64_t)half * n_heads);
    const int pos = pos0 + (int)(row % seq_len);
    const float inv = powf(theta, -2.0f * (float)e / (float)dh);
    float sn, cs;
    sincosf((float)pos * inv, &sn, &cs);
    const int64_t o1 = row * ld + (int64_t)hd * dh + e, o2 = o1 + half;
    const float a = bf16_to_f32(xh[o1]) + (xl ? bf16_to_f32(xl[o1]) : 0.f);
    const float b = bf16_to_f32(xh[o2]) + (xl ? bf16_to_f32(xl[o2]) : 0.f);
    const float ra = a * cs - b * sn, rb = b * cs + a * sn;
    const uint16_t ha = f32_to_bf16(ra), hb = f32_to_bf16(rb);
    xh[o1] = ha; xh[o2] = hb;
    if (xl) { xl[o1] = f32_to_bf16(ra - bf16_to_f32(ha)); xl[o2] = f32_to_bf16(rb - bf16_to_f32(hb)); }
}

// argmax over wide rows (vocabulary 151 936): one workgroup per row streamed 600 KB through one CU (191 us).  Now a row is split
// into chunks of 8192; every workgroup reduces its chunk and publishes (orderable value bits << 32 | ~index) with one 64-bit
// atomicMax into out[row] (zeroed first), a second tiny kernel unpacks the index.  Larger value wins; equal values: smaller index.
constexpr int ARGMAX_CHUNK = 8192;
__device__ __forceinline__ unsigned long long argmax_pack(float v, int i) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                 // monotone float -> uint
    return ((unsigned long long)u << 32) | (uint32_t)(0xffffffffu - (uint32_t)i);
}
__global__ void __launch_bounds__(256) k_argmax_partial(const float *__restrict__ x, int n, unsigned long long *__restrict__ out) {
    __shared__ unsigned long long sw[4];
    const float *row = x + (int64_t)blockIdx.y * n;
    const int c0 = blockIdx.x * ARGMAX_CHUNK, c1 = c0 + ARGMAX_CHUNK < n ? c0 + ARGMAX_CHUNK : n;
    unsigned long long best = 0ull;
    for (int i = c0 + threadIdx.x; i < c1; i += 256) {
        const unsigned long long p = argmax_pack(row[i], i);
        best = p > best ? p : best;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long q = __shfl_xor(best, o);
        best = q > best ? q : best;
    }
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) best = sw[w] > best ? sw[w] : best;
        atomicMax(&out[blockIdx.y], best);
    }
}
__global__ void __launch_bounds__(256) k_argmax_final(unsigned long long *__restrict__ out, int64_t rows) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const unsigned long long p = out[r];
    reinterpret_cast<int64_t *>(out)[r] = p ? (int64_t)(0xffffffffu - (uint32_t)(p & 0xffffffffull)) : 0;
}

__global__ void __launch_bounds__(256) k_swiglu(const float *__restrict__ gu, int64_t rows, int inter, uint16_t *__restrict__ oh,
                                                uint16_t *__restrict__ ol) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * inter) return;
    const int64_t r = i / inter;
    const int k = (int)(i - r * inter);
    const float g = gu[r * 2 * inter + k], u = gu[r * 2 * inter + inter + k];
    const float v = g / (1.0f + expf(-g)) * u;
    const uint16_t h = f32_to_bf16(v);
    oh[i] = h;
    if (ol) ol[i] = f32_to_bf16(v - bf16_to_f32(h));
}

// one wave per row: loss_sum += logsumexp(row) - row[label]; cnt += 1 (labels == -100 are skipped)
__global__ void __launch_bounds__(256) k_cross_entropy(const float *__restrict__ logits, const int64_t *__restrict__ labels,
                                                       int64_t rows, int vocab, float *__restrict__ acc2) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t lab = labels[row];
    if (lab < 0 || lab >= vocab) return;
    const float *lr = logits + row * vocab;
    float mx = -INFINITY;
    for (int k = lane; k < vocab; k += 64) mx = fmaxf(mx, lr[k]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int k = lane; k < vocab; k += 64) s += expf(lr[k] - mx);
    s = wave_sum(s);
    if (lane == 0) {
        atomicAdd(&acc2[0], logf(s) + mx - lr[lab]);
        atomicAdd(&acc2[1], 1.0f);
    }
}


// ---------------------------------------------------------------------------------------------------------
// Sampling step of `generate(do_sample=True)` as transformers runs it for the reference's default call
// (inference_engine.py:236-240, 283-296: temperature 0.7, top_k 50, top_p 0.9): TemperatureLogitsWarper ->
// TopKLogitsWarper -> TopPLogitsWarper -> softmax -> one multinomial draw per row.
//   * top-k keeps every logit >= the k-th largest (ties stay, as `scores < topk(scores, k)[0][..., -1]` removes only smaller
//     ones): the threshold is found by a 4-pass radix select on the order-preserving uint image of the floats (no sort of the
//     vocabulary), the survivors (<= SAMPLE_CAP; beyond it, ties at the k-th value are kept in ascending index order) are collected into LDS and bitonic-sorted descending (ties: lower id first);
//   * top-p keeps token j iff the probability mass ranked strictly before it is < top_p (the descending-order statement of
//     `cumsum(softmax(sorted ascending)) <= 1 - top_p` is removed; at least one token stays);
//   * the draw is the inverse CDF at the caller's uniform u in [0, 1) over the kept, renormalised probabilities.
// One 1024-thread workgroup per row; the logits row is read 5 times from L2 (608 KB at Qwen2.5's vocabulary).
// ---------------------------------------------------------------------------------------------------------
// column sums of a row-major [rows, d] fp32 matrix in a FIXED order (the payload of the per-step all-reduce):
// workgroup (cx, ry) sums rows ry, ry + gridDim.y, ... of its 64 columns (thread (r, c): every 16th of those rows, the 16 partials
// added in order); with gridDim.y > 1 the per-chunk results go to `part` [gridDim.y, d] and a second launch (rows = gridDim.y) adds
// them.  One workgroup per 64 columns alone streamed 56 MB through 12 CUs: 0.51 ms per step.
__global__ void __launch_bounds__(1024) k_colsum(const float *__restrict__ x, int64_t rows, int d, float *__restrict__ out) {
    __shared__ float part[16][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6, col = blockIdx.x * 64 + c;
    float acc = 0.f;
    if (col < d)
        for (int64_t i = (int64_t)blockIdx.y * 16 + r; i < rows; i += (int64_t)gridDim.y * 16) acc += x[i * d + col];
    part[r][c] = acc;
    __syncthreads();
    if (r == 0 && col < d) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[k][c];
        out[(int64_t)blockIdx.y * d + col] = s;
    }
}

constexpr int SAMPLE_NT = 1024;
constexpr int SAMPLE_CAP = 2048;           // candidates held in LDS (top_k <= 1024 plus ties)

__device__ __forceinline__ uint32_t f32_order_key(float x) {
    const uint32_t b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);      // ascending uint order == ascending float order
}

__global__ void __launch_bounds__(SAMPLE_NT) k_sample_rows(const float *__restrict__ logits, int vocab, float inv_temp, int top_k, float top_p,
                                                          const float *__restrict__ u, int64_t *__restrict__ out) {
    __shared__ uint32_t hist[256];
    __shared__ float c_val[SAMPLE_CAP];
    __shared__ int c_idx[SAMPLE_CAP];
    __shared__ float wsum[SAMPLE_NT / 64];
    __shared__ uint32_t sh_prefix, sh_mask;
    __shared__ int sh_remaining, sh_count, sh_keep;
    __shared__ float sh_total;
    const int tid = threadIdx.x;
    const float *x = logits + (int64_t)blockIdx.x * vocab;
    // ---- radix select: key of the k-th largest element ----
    if (tid == 0) { sh_prefix = 0u; sh_mask = 0u; sh_remaining = top_k; sh_count = 0; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const uint32_t prefix = sh_prefix, mask = sh_mask;
        for (int i = tid; i < vocab; i += SAMPLE_NT) {
            const uint32_t k = f32_order_key(x[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {                                     // walk the digits from the top until `remaining` elements are covered
            int rem = sh_remaining, d = 255;
            for (; d > 0; --d) {
                const int c = (int)hist[d];
                if (c >= rem) break;
                rem -= c;
            }
            sh_remaining = rem;
            sh_prefix = prefix | ((uint32_t)d << shift);
            sh_mask = mask | (255u << shift);
        }
        __syncthreads();
    }
    const uint32_t kth = sh_prefix;                         // every element with key >= kth survives top-k
    // ---- collect the survivors: everything STRICTLY above the k-th key first (at most top_k - 1 <= 1023 entries, order irrelevant: they
    // are sorted below), then the ties at the k-th key.  If strict + ties fit the cap (the common case: one tie), the ties are collected
    // the same way; otherwise (mass ties: a constant or clamped row) the ties are taken in ASCENDING INDEX order up to the cap, so the
    // kept set is a function of the logits alone (never of the atomics' arrival order) and no larger logit is ever displaced by a tie ----
    for (int i = tid; i < SAMPLE_CAP; i += SAMPLE_NT) { c_val[i] = -INFINITY; c_idx[i] = 0x7fffffff; }
    if (tid == 0) sh_keep = 0;                              // (re-used: number of ties)
    __syncthreads();
    for (int i = tid; i < vocab; i += SAMPLE_NT) {
        const float v = x[i];
        const uint32_t k = f32_order_key(v);
        if (k > kth) {
            const int q = atomicAdd(&sh_count, 1);
            c_val[q] = v; c_idx[q] = i;
        } else if (k == kth) {
            atomicAdd(&sh_keep, 1);
        }
    }
    __syncthreads();
    const int n_strict = sh_count, n_ties = sh_keep;
    __syncthreads();
    if (n_strict + n_ties <= SAMPLE_CAP) {
        for (int i = tid; i < vocab; i += SAMPLE_NT) {
            const float v = x[i];
            if (f32_order_key(v) == kth) {
                const int q = atomicAdd(&sh_count, 1);
                c_val[q] = v; c_idx[q] = i;
            }
        }
    } else {
        // ordered pass: chunk by chunk in index order, a tie's slot = ties before it (block-wide exclusive count)
        int placed = n_strict;
        for (int base = 0; base < vocab && placed < SAMPLE_CAP; base += SAMPLE_NT) {
            const int i = base + tid;
            const float v = i < vocab ? x[i] : 0.f;
            const bool tie = i < vocab && f32_order_key(v) == kth;
            const unsigned long long bal = __ballot(tie);
            const int lane_ = tid & 63, w_ = tid >> 6;
            if (lane_ == 0) wsum[w_] = (float)__popcll(bal);
            __syncthreads();
            int before = 0, total_c = 0;
            for (int w = 0; w < SAMPLE_NT / 64; ++w) {
                const int c = (int)wsum[w];
                if (w < w_) before += c;
                total_c += c;
            }
            const int slot = placed + before + __popcll(bal & ((1ull << lane_) - 1ull));
            if (tie && slot < SAMPLE_CAP) { c_val[slot] = v; c_idx[slot] = i; }
            placed += total_c;
            __syncthreads();
        }
        if (tid == 0) sh_count = placed < SAMPLE_CAP ? placed : SAMPLE_CAP;
    }
    __syncthreads();
    const int n = sh_count < SAMPLE_CAP ? sh_count : SAMPLE_CAP;
    // ---- bitonic sort, descending by value, ascending id among equals ----
    for (int size = 2; size <= SAMPLE_CAP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < SAMPLE_CAP / 2; t += SAMPLE_NT) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const float a = c_val[lo], b = c_val[hi];
                const int ia = c_idx[lo], ib = c_idx[hi];
                const bool a_first = a > b || (a == b && ia < ib);      // a belongs before b in the final order
                if (a_first != desc) { c_val[lo] = b; c_val[hi] = a; c_idx[lo] = ib; c_idx[hi] = ia; }
            }
            __syncthreads();
        }
    }
    // ---- softmax numerators (temperature applied here: order is unchanged by a positive scale) and their prefix sums ----
    const float top = c_val[0];
    float p0 = 0.f, p1 = 0.f;
    {
        const int j = 2 * tid;
        if (j < n) p0 = __expf((c_val[j] - top) * inv_temp);
        if (j + 1 < n) p1 = __expf((c_val[j + 1] - top) * inv_temp);
    }
    float incl = p0 + p1;
    const int lane = tid & 63, wid = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    float base = 0.f, total = 0.f;
    for (int w = 0; w < SAMPLE_NT / 64; ++w) {
        const float t = wsum[w];
        if (w < wid) base += t;
        total += t;
    }
    const float ex0 = base + incl - (p0 + p1), ex1 = ex0 + p0;    // mass ranked strictly before elements 2 tid and 2 tid + 1
    __syncthreads();
    // ---- top-p: keep j iff (mass before j) / total < top_p; the kept set is a prefix of the sorted list ----
    if (tid == 0) sh_keep = 1;
    __syncthreads();
    {
        const int j = 2 * tid;
        const float cut = top_p * total;
        int last = -1;
        if (j < n && ex0 < cut) last = j;
        if (j + 1 < n && ex1 < cut) last = j + 1;
        if (last >= 0) atomicMax(&sh_keep, last + 1);
    }
    __syncthreads();
    const int keep = sh_keep;
    // kept mass = exclusive prefix at `keep` (or the total)
    if (2 * tid == keep) sh_total = ex0;
    if (2 * tid + 1 == keep) sh_total = ex1;
    if (tid == 0 && keep >= n) sh_total = total;
    __syncthreads();
    // ---- inverse CDF: the first kept j whose inclusive mass exceeds u * kept mass ----
    const float target = u[blockIdx.x] * sh_total;
    if (tid == 0) sh_count = keep - 1;                      // fallback (u * mass rounding up to the whole mass)
    __syncthreads();
    {
        const int j = 2 * tid;
        if (j < keep && ex0 + p0 > target) atomicMin(&sh_count, j);
        else if (j + 1 < keep && ex1 + p1 > target) atomicMin(&sh_count, j + 1);
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = (int64_t)c_idx[sh_count];
}

}  // namespace

extern "C" int lvq_layernorm(const float *x, const float *add, int add_rows, int add_group, const float *gamma,
                             const float *beta, float eps, int64_t rows, int d, const float *post_add, int64_t post_rows,
                             float *y_f32, lvq_bf16 *y_bf16, lvq_bf16 *y_lo, lvq_stream_t stream) {
    if (rows < 0 || d <= 0 || !gamma || (!y_f32 && !y_bf16) || (y_lo && !y_bf16)) return LVQ_EINVAL;
    if (add && (add_rows <= 0 || add_group <= 0)) return LVQ_EINVAL;
    if (post_add && post_rows <= 0) return LVQ_EINVAL;
    if (rows == 0) return LVQ_OK;
    if (!x) return LVQ_EINVAL;
    if (launch_norm_vec<false>(x, add, add_rows, add_group < 1 ? 1 : add_group, gamma, beta, eps, rows, d, post_add,
                               post_rows < 1 ? 1 : post_rows, y_f32, y_bf16, y_lo, lvq_s(stream)))
        return lvq_launch_status();
    hipLaunchKernelGGL(k_norm<false>, dim3((unsigned)lvq_cdiv(rows, 4)), dim3(256), 0, lvq_s(stream), x, add, add_rows,
                       add_group < 1 ? 1 : add_group, gamma, beta, eps, rows, d, post_add, post_rows < 1 ? 1 : post_rows, y_f32,
                       y_bf16, y_lo);
    return lvq_launch_status();
}

extern "C" int lvq_rmsnorm(const float *x, const float *gamma, float eps, int64_t rows, int d, float *y_f32, lvq_bf16 *y_bf16,
                           lvq_bf16 *y_lo, lvq_stream_t stream) {
    if (rows < 0 || d <= 0 || !gamma || (!y_f32 && !y_bf16) || (y_lo && !y_bf16)) return LVQ_EINVAL;
    if (rows == 0) return LVQ_OK;
    if (!x) return LVQ_EINVAL;
    if (launch_norm_vec<true>(x, nullptr, 1, 1, gamma, nullptr, eps, rows, d, nullptr, 1, y_f32, y_bf16, y_lo, lvq_s(stream)))
        return lvq_launch_status();
    hipLaunchKernelGGL(k_norm<true>, dim3((unsigned)lvq_cdiv(rows, 4)), dim3(256), 0, lvq_s(stream), x, (const float *)nullptr,
                       1, 1, gamma, (const float *)nullptr, eps, rows, d, (const float *)nullptr, (int64_t)1, y_f32, y_bf16, y_lo);
    return lvq_launch_status();
}

extern "C" int lvq_dwconv3x3_gelu(const float *bev, const float *w9, const float *bias, int batch, int ch, int h, int w,
                                  lvq_bf16 *tokens_hi, lvq_bf16 *tokens_lo, lvq_stream_t stream) {
    if (batch <= 0 || ch <= 0 || h <= 0 || w <= 0 || !bev || !w9 || !tokens_hi) return LVQ_EINVAL;
    if (ch % 8) return LVQ_EUNSUPPORTED;                      // 16-byte token stores
    if (((uintptr_t)tokens_hi | (uintptr_t)tokens_lo) & 15) return LVQ_EUNSUPPORTED;
    const int cblocks = (ch + DW_C - 1) / DW_C;
    if ((int64_t)batch * cblocks > 65535 || lvq_cdiv(h, DW_RY) > 65535) return LVQ_EUNSUPPORTED;
    dim3 grid((unsigned)lvq_cdiv(w, DW_PX), (unsigned)lvq_cdiv(h, DW_RY), (unsigned)(batch * cblocks));
    if (tokens_lo) hipLaunchKernelGGL(k_dwconv3x3_gelu<true>, grid, dim3(256), 0, lvq_s(stream), bev, w9, bias, ch, h, w, tokens_hi, tokens_lo);
    else           hipLaunchKernelGGL(k_dwconv3x3_gelu<false>, grid, dim3(256), 0, lvq_s(stream), bev, w9, bias, ch, h, w, tokens_hi, tokens_lo);
    return lvq_launch_status();
}

// PointPillarScatter as an INDEX map (pointpillar_scatter.py:14-37 without the canvas): idx[b, y, x] = pillar row or -1.
extern "C" int lvq_pillar_index_map(const int32_t *coords_bzyx, int64_t m_cap, const int32_t *n_voxels_dev, int batch, int ny, int nx,
                                    int32_t *idx_map, lvq_stream_t stream) {
    if (m_cap < 0 || batch <= 0 || ny <= 0 || nx <= 0 || !idx_map || (m_cap > 0 && !coords_bzyx)) return LVQ_EINVAL;
    if ((int64_t)batch * ny * nx >= (1ll << 31)) return LVQ_EOVERFLOW;
    hipStream_t st = lvq_s(stream);
    hipMemsetAsync(idx_map, 0xff, sizeof(int32_t) * (size_t)batch * ny * nx, st);
    if (m_cap > 0)
        hipLaunchKernelGGL(k_pillar_index, dim3((unsigned)lvq_cdiv(m_cap, 256)), dim3(256), 0, st, coords_bzyx, m_cap, n_voxels_dev, batch, ny, nx, idx_map);
    return lvq_launch_status();
}

extern "C" size_t lvq_pillar_dwconv_workspace_bytes(int batch, int ny, int nx) {
    return lvq_align((size_t)batch * ny * nx * sizeof(int32_t)) + 256;
}

extern "C" int lvq_pillar_dwconv3x3_gelu(const float *feat, const int32_t *coords_bzyx, int64_t m_cap, const int32_t *n_voxels_dev,
                                         int ch, int batch, int ny, int nx, const float *w9, const float *bias, lvq_bf16 *tokens_hi,
                                         lvq_bf16 *tokens_lo, void *ws, size_t ws_bytes, lvq_stream_t stream) {
    if (batch <= 0 || ch <= 0 || ny <= 0 || nx <= 0 || m_cap < 0 || !w9 || !tokens_hi) return LVQ_EINVAL;
    if (m_cap > 0 && (!feat || !coords_bzyx)) return LVQ_EINVAL;
    if (ch % 8) return LVQ_EUNSUPPORTED;
    if ((((uintptr_t)tokens_hi | (uintptr_t)tokens_lo | (uintptr_t)coords_bzyx) & 15) || m_cap > 0x7fffffff) return LVQ_EUNSUPPORTED;
    LvqArena arena(ws, ws_bytes);
    int32_t *idx = arena.take<int32_t>((size_t)batch * ny * nx);
    if (!arena.ok) return LVQ_EWORKSPACE;
    const int cblocks = (ch + DW_C - 1) / DW_C;
    if ((int64_t)batch * cblocks > 65535 || lvq_cdiv(ny, DW_RY) > 65535) return LVQ_EUNSUPPORTED;
    hipStream_t st = lvq_s(stream);
    hipMemsetAsync(idx, 0xFF, (size_t)batch * ny * nx * sizeof(int32_t), st);           // -1 = empty cell
    if (m_cap > 0)
        hipLaunchKernelGGL(k_pillar_index, dim3((unsigned)lvq_cdiv(m_cap, 256)), dim3(256), 0, st, coords_bzyx, m_cap, n_voxels_dev, batch, ny, nx, idx);
    dim3 grid((unsigned)lvq_cdiv(nx, DW_PX), (unsigned)lvq_cdiv(ny, DW_RY), (unsigned)(batch * cblocks));
    if (tokens_lo) hipLaunchKernelGGL(k_pillar_dwconv_gelu<true>, grid, dim3(256), 0, st, feat, idx, w9, bias, ch, ny, nx, tokens_hi, tokens_lo);
    else           hipLaunchKernelGGL(k_pillar_dwconv_gelu<false>, grid, dim3(256), 0, st, feat, idx, w9, bias, ch, ny, nx, tokens_hi, tokens_lo);
    return lvq_launch_status();
}

extern "C" int lvq_scale_add_rows(const float *x, const float *add, int64_t add_rows, float alpha, int64_t rows, int d,
                                  float *out, lvq_stream_t stream) {
    if (rows < 0 || d <= 0 || (add && add_rows <= 0)) return LVQ_EINVAL;
    if (rows == 0) return LVQ_OK;
    if (!x || !out) return LVQ_EINVAL;
    hipLaunchKernelGGL(k_scale_add_rows, dim3((unsigned)lvq_cdiv(rows * d, 256)), dim3(256), 0, lvq_s(stream), x, add,
                       add_rows < 1 ? 1 : add_rows, alpha, rows, d, out);
    return lvq_launch_status();
}

extern "C" int lvq_rope_inplace(lvq_bf16 *x, lvq_bf16 *x_lo, int64_t rows, int seq_len, int n_heads, int dh, int64_t ld,
                                float theta, lvq_stream_t stream) {
    if (rows < 0 || seq_len <= 0 || n_heads <= 0 || dh <= 0 || (dh & 1) || ld < (int64_t)n_heads * dh) return LVQ_EINVAL;
    if (rows == 0) return LVQ_OK;
    if (!x) return LVQ_EINVAL;
    const int64_t n = rows * n_heads * (dh / 2);
    hipLaunchKernelGGL(k_rope, dim3((unsigned)lvq_cdiv(n, 256)), dim3(256), 0, lvq_s(stream), x, x_lo, rows, seq_len, n_heads,
                       dh, ld, theta, 0);
    return lvq_launch_status();
}

// decode-time form: the rows are new positions pos0 .. pos0 + seq_len - 1 of every sequence (seq_len = 1: one new token each)
extern "C" int lvq_rope_inplace_at(lvq_bf16 *x, lvq_bf16 *x_lo, int64_t rows, int seq_len, int pos0, int n_heads, int dh, int64_t ld,
                                   float theta, lvq_stream_t stream) {
    if (rows < 0 || seq_len <= 0 || pos0 < 0 || n_heads <= 0 || dh <= 0 || (dh & 1) || ld < (int64_t)n_heads * dh) return LVQ_EINVAL;
    if (rows == 0) return LVQ_OK;
    if (!x) return LVQ_EINVAL;
    const int64_t n = rows * n_heads * (dh / 2);
    hipLaunchKernelGGL(k_rope, dim3((unsigned)lvq_cdiv(n, 256)), dim3(256), 0, lvq_s(stream), x, x_lo, rows, seq_len, n_heads,
                       dh, ld, theta, pos0);
    return lvq_launch_status();
}

extern "C" size_t lvq_colsum_workspace_bytes(int64_t rows, int d) { return rows >= 4096 ? (size_t)64 * d * sizeof(float) + 256 : 256; }

extern "C" int lvq_colsum(const float *x, int64_t rows, int d, float *out, void *ws, size_t ws_bytes, lvq_stream_t stream) {
    if (rows < 0 || d <= 0 || !out || (rows > 0 && !x)) return LVQ_EINVAL;
    const unsigned cx = (unsigned)lvq_cdiv(d, 64);
    if (rows < 4096) {
        hipLaunchKernelGGL(k_colsum, dim3(cx, 1), dim3(1024), 0, lvq_s(stream), x, rows, d, out);
        return lvq_launch_status();
    }
    if (!ws || ws_bytes < lvq_colsum_workspace_bytes(rows, d)) return LVQ_EWORKSPACE;
    float *part = (float *)ws;
    hipLaunchKernelGGL(k_colsum, dim3(cx, 64), dim3(1024), 0, lvq_s(stream), x, rows, d, part);
    hipLaunchKernelGGL(k_colsum, dim3(cx, 1), dim3(1024), 0, lvq_s(stream), (const float *)part, (int64_t)64, d, out);
    return lvq_launch_status();
}

extern "C" int lvq_sample_rows(const float *logits, int64_t rows, int vocab, float temperature, int top_k, float top_p, const float *u,
                               int64_t *out_idx, lvq_stream_t stream) {
    if (rows < 0 || vocab <= 0 || !(temperature > 0.f) || !(top_p > 0.f) || top_p > 1.f) return LVQ_EINVAL;
    if (rows == 0) return LVQ_OK;
    if (!logits || !u || !out_idx) return LVQ_EINVAL;
    if (top_k <= 0 || top_k > vocab) top_k = vocab;         // transformers: top_k = 0 / None disables the filter
    if (top_k > 1024 || rows > 0x7fffffff) return LVQ_EUNSUPPORTED;
    hipLaunchKernelGGL(k_sample_rows, dim3((unsigned)rows), dim3(SAMPLE_NT), 0, lvq_s(stream), logits, vocab, 1.0f / temperature, top_k, top_p, u,
                       out_idx);
    return lvq_launch_status();
}

// greedy decoding: index of the first maximum of every row (torch.argmax semantics on finite logits)
extern "C" int lvq_argmax_rows(const float *x, int64_t rows, int n, int64_t *out_idx, lvq_stream_t stream) {
    if (rows < 0 || n <= 0) return LVQ_EINVAL;
    if (rows == 0) return LVQ_OK;
    if (!x || !out_idx) return LVQ_EINVAL;
    if (rows > 65535) return LVQ_EUNSUPPORTED;
    hipStream_t st = lvq_s(stream);
    unsigned long long *packed = reinterpret_cast<unsigned long long *>(out_idx);
    hipMemsetAsync(packed, 0, sizeof(unsigned long long) * rows, st);
    hipLaunchKernelGGL(k_argmax_partial, dim3((unsigned)lvq_cdiv(n, ARGMAX_CHUNK), (unsigned)rows), dim3(256), 0, st, x, n, packed);
    hipLaunchKernelGGL(k_argmax_final, dim3((unsigned)lvq_cdiv(rows, 256)), dim3(256), 0, st, packed, rows);
    return lvq_launch_status();
}

extern "C" int lvq_swiglu(const float *gate_up, int64_t rows, int inter, lvq_bf16 *out_hi, lvq_bf16 *out_lo,
                          lvq_stream_t stream) {
    if (rows < 0 || inter <= 0) return LVQ_EINVAL;
    if (rows == 0) return LVQ_OK;
    if (!gate_up || !out_hi) return LVQ_EINVAL;
    hipLaunchKernelGGL(k_swiglu, dim3((unsigned)lvq_cdiv(rows * inter, 256)), dim3(256), 0, lvq_s(stream), gate_up, rows,
                       inter, out_hi, out_lo);
    return lvq_launch_status();
}

extern "C" int lvq_cross_entropy(const float *logits, const int64_t *labels, int64_t rows, int vocab, float *loss_sum_cnt,
                                 lvq_stream_t stream) {
    if (rows < 0 || vocab <= 0 || !loss_sum_cnt) return LVQ_EINVAL;
    if (rows == 0) return LVQ_OK;
    if (!logits || !labels) return LVQ_EINVAL;
    hipLaunchKernelGGL(k_cross_entropy, dim3((unsigned)lvq_cdiv(rows, 4)), dim3(256), 0, lvq_s(stream), logits, labels, rows,
                       vocab, loss_sum_cnt);
    return lvq_launch_status();
}
