// csrc/gemm.hip -- nn.Linear / 1x1-conv / MHA projections on bf16 MFMA tiles (gfx950).
//
//   C[m,n] = epi( sum_k A[m,k] * W[n,k] )      A [M,K] and W [N,K] both K-contiguous ("TN"), which is
// exactly the v_mfma_f32_16x16x32_bf16 operand shape: lane l of a wave supplies 8 consecutive k of
// row (l & 15) for A and of column (l & 15) for B, k-block (l >> 4) -- one 16-byte LDS read each, no
// transposes anywhere.  Replaces torch's F.linear under VATBlock / VATLiDAR / VATVision
// (vat_blocks.py:20-34, vat_lidar.py:88-97,117-120, vat_vision.py:118-137, build_linear.py:18-19).
//
// Structure: BMxBN tile per 256-thread workgroup (4 waves as 2x2, each (BM/2)x(BN/2) = TMxTN MFMA
// tiles), BK = 64, two LDS stages.  Global->register loads of tile t+1 are issued before the MFMAs of
// tile t and written to the other LDS stage after them (issue-early / write-late), one barrier per
// K-tile.  LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row>>1)&7 so that a
// ds_read_b128 lane group (rows 0-3,12-15 at chunk c and rows 4-11 at chunk c+1) hits 16 distinct
// 16-byte slots of the 256-byte bank row.
//
// Precision modes (SURVEY 7 "1e-3 in bf16"): accumulation is always fp32.  With plain bf16 operands
// the operand rounding alone (2^-9 relative) puts VATBlock outputs ~2e-3*max|out| from the fp32 CPU
// reference, outside the 1e-3 parity bar.  "bf16x3" therefore feeds each operand as hi + lo bf16 parts
// and runs three MFMA passes (hi*hi + hi*lo + lo*hi) over the same accumulators -- still bf16 MFMA
// tiles, ~2^-17 relative operand error.  It is implemented as K-segments of one loop, so both modes
// share every line of the kernel.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct GemmArgs {
    const uint16_t *a[3];
    const uint16_t *w[3];
    int nseg;
    const float *bias, *residual, *rowtab;
    int64_t rowtab_rows;
    float alpha;
    int flags;
    int64_t M;
    int N, K;
    int64_t lda, ldw, ldc;
    int64_t a_bs, w_bs, c_bs;
    float *c32;
    uint16_t *c16, *c16lo;
};

template <int BM, int BN>
__global__ void __launch_bounds__(256) k_gemm_bf16(GemmArgs g) {
    constexpr int BK = 64;
    constexpr int TM = BM / 32, TN = BN / 32;      // MFMA tiles per wave
    constexpr int LA = BM / 32, LB = BN / 32;      // 16-byte chunks per thread per stage
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int64_t z = blockIdx.z;
    const int nk = (g.K + BK - 1) / BK;
    const int n_it = nk * g.nseg;

    uint4 ra[LA], rb[LB];
    auto gload = [&](int it) {
        const int seg = it / nk, k0 = (it - seg * nk) * BK;
        const uint16_t *A = g.a[seg] + z * g.a_bs;
        const uint16_t *W = g.w[seg] + z * g.w_bs;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int q = tid + 256 * i, row = q >> 3, kk = k0 + (q & 7) * 8;
            const int64_t gm = m0 + row;
            ra[i] = (gm < g.M && kk < g.K) ? *reinterpret_cast<const uint4 *>(A + gm * g.lda + kk) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int q = tid + 256 * i, row = q >> 3, kk = k0 + (q & 7) * 8;
            const int gn = n0 + row;
            rb[i] = (gn < g.N && kk < g.K) ? *reinterpret_cast<const uint4 *>(W + (int64_t)gn * g.ldw + kk) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
        uint8_t *sa = smem + buf * STAGE, *sb = sa + BM * 128;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int q = tid + 256 * i, row = q >> 3, ch = (q & 7) ^ ((row >> 1) & 7);
            *reinterpret_cast<uint4 *>(sa + row * 128 + ch * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int q = tid + 256 * i, row = q >> 3, ch = (q & 7) ^ ((row >> 1) & 7);
            *reinterpret_cast<uint4 *>(sb + row * 128 + ch * 16) = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(0);
    lstore(0);
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        const int buf = it & 1;
        if (it + 1 < n_it) gload(it + 1);
        const uint8_t *sa = smem + buf * STAGE, *sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / 2) + i * 16 + (lane & 15);
                const int ch = (ks * 4 + (lane >> 4)) ^ ((row >> 1) & 7);
                af[i] = *reinterpret_cast<const bf16x8 *>(sa + row * 128 + ch * 16);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / 2) + j * 16 + (lane & 15);
                const int ch = (ks * 4 + (lane >> 4)) ^ ((row >> 1) & 7);
                bfr[j] = *reinterpret_cast<const bf16x8 *>(sb + row * 128 + ch * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (it + 1 < n_it) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of 16x16x32: lane l holds rows (l>>4)*4 + r, column l & 15
    const bool gelu = g.flags & LVQ_GEMM_GELU;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
            if (col >= g.N) continue;
            const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bias;
                if (gelu) v = gelu_erf(v);
                v *= g.alpha;
                const int64_t o = z * g.c_bs + row * g.ldc + col;
                if (g.residual) v += g.residual[o];
                if (g.rowtab) v += g.rowtab[(row % g.rowtab_rows) * g.N + col];
                if (g.c32) g.c32[o] = v;
                if (g.c16) {
                    const uint16_t h = f32_to_bf16(v);
                    g.c16[o] = h;
                    if (g.c16lo) g.c16lo[o] = f32_to_bf16(v - bf16_to_f32(h));
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_cast_bf16(const float *__restrict__ x, int64_t n, uint16_t *__restrict__ hi,
                                                   uint16_t *__restrict__ lo) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 v = *reinterpret_cast<const float4 *>(x + i);
        ushort4 h = make_ushort4(f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w));
        *reinterpret_cast<ushort4 *>(hi + i) = h;
        if (lo) {
            ushort4 l = make_ushort4(f32_to_bf16(v.x - bf16_to_f32(h.x)), f32_to_bf16(v.y - bf16_to_f32(h.y)),
                                     f32_to_bf16(v.z - bf16_to_f32(h.z)), f32_to_bf16(v.w - bf16_to_f32(h.w)));
            *reinterpret_cast<ushort4 *>(lo + i) = l;
        }
    } else {
        for (; i < n; ++i) {
            uint16_t h = f32_to_bf16(x[i]);
            hi[i] = h;
            if (lo) lo[i] = f32_to_bf16(x[i] - bf16_to_f32(h));
        }
    }
}

__global__ void __launch_bounds__(256) k_bf16_to_f32(const uint16_t *__restrict__ hi, const uint16_t *__restrict__ lo, int64_t n,
                                                     float alpha, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = (bf16_to_f32(hi[i]) + (lo ? bf16_to_f32(lo[i]) : 0.f)) * alpha;
}

}  // namespace

extern "C" int lvq_gemm_bf16(const lvq_bf16 *a, const lvq_bf16 *a_lo, const lvq_bf16 *w, const lvq_bf16 *w_lo,
                             const float *bias, const float *residual, const float *rowtab, int64_t rowtab_rows,
                             float alpha, int flags, int64_t m, int n, int k, int64_t lda, int64_t ldw, int64_t ldc,
                             int batch, int64_t a_bs, int64_t w_bs, int64_t c_bs, float *c_f32, lvq_bf16 *c_bf16,
                             lvq_bf16 *c_lo, lvq_stream_t stream) {
    if (m < 0 || n <= 0 || k <= 0 || batch <= 0 || !a || !w || (!c_f32 && !c_bf16)) return LVQ_EINVAL;
    if ((a_lo == nullptr) != (w_lo == nullptr)) return LVQ_EINVAL;
    if (c_lo && !c_bf16) return LVQ_EINVAL;
    if (rowtab && rowtab_rows <= 0) return LVQ_EINVAL;
    if (m == 0) return LVQ_OK;
    // 16-byte operand loads: K, leading dims and batch strides in multiples of 8 elements, 16-B aligned bases
    if ((k & 7) || (lda & 7) || (ldw & 7) || (a_bs & 7) || (w_bs & 7) || lda < k || ldw < k || ldc < n)
        return LVQ_EUNSUPPORTED;
    if (((uintptr_t)a | (uintptr_t)w | (uintptr_t)a_lo | (uintptr_t)w_lo) & 15) return LVQ_EUNSUPPORTED;
    GemmArgs g;
    g.nseg = a_lo ? 3 : 1;
    g.a[0] = a; g.w[0] = w;
    g.a[1] = a; g.w[1] = w_lo;
    g.a[2] = a_lo; g.w[2] = w;
    g.bias = bias; g.residual = residual; g.rowtab = rowtab; g.rowtab_rows = rowtab_rows;
    g.alpha = alpha; g.flags = flags; g.M = m; g.N = n; g.K = k;
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.a_bs = a_bs; g.w_bs = w_bs; g.c_bs = c_bs;
    g.c32 = c_f32; g.c16 = c_bf16; g.c16lo = c_lo;
    hipStream_t st = lvq_s(stream);
    const int64_t big_tiles = lvq_cdiv(m, 128) * lvq_cdiv(n, 128) * batch;
    if (big_tiles >= 192) {
        dim3 grid((unsigned)lvq_cdiv(n, 128), (unsigned)lvq_cdiv(m, 128), (unsigned)batch);
        if (grid.y > 65535) return LVQ_EUNSUPPORTED;
        hipLaunchKernelGGL((k_gemm_bf16<128, 128>), grid, dim3(256), 2 * (128 + 128) * 128, st, g);
    } else {
        dim3 grid((unsigned)lvq_cdiv(n, 64), (unsigned)lvq_cdiv(m, 64), (unsigned)batch);
        if (grid.y > 65535) return LVQ_EUNSUPPORTED;
        hipLaunchKernelGGL((k_gemm_bf16<64, 64>), grid, dim3(256), 2 * (64 + 64) * 128, st, g);
    }
    return lvq_launch_status();
}

extern "C" int lvq_cast_bf16(const float *x, int64_t n, lvq_bf16 *hi, lvq_bf16 *lo, lvq_stream_t stream) {
    if (n < 0 || (n > 0 && (!x || !hi))) return LVQ_EINVAL;
    if (n == 0) return LVQ_OK;
    if (((uintptr_t)x & 15) || ((uintptr_t)hi & 7) || ((uintptr_t)lo & 7)) return LVQ_EUNSUPPORTED;
    hipLaunchKernelGGL(k_cast_bf16, dim3((unsigned)lvq_cdiv(n, 1024)), dim3(256), 0, lvq_s(stream), x, n, hi, lo);
    return lvq_launch_status();
}

extern "C" int lvq_bf16_to_f32(const lvq_bf16 *hi, const lvq_bf16 *lo, int64_t n, float alpha, float *out, lvq_stream_t stream) {
    if (n < 0 || (n > 0 && (!hi || !out))) return LVQ_EINVAL;
    if (n == 0) return LVQ_OK;
    hipLaunchKernelGGL(k_bf16_to_f32, dim3((unsigned)lvq_cdiv(n, 256)), dim3(256), 0, lvq_s(stream), hi, lo, n, alpha, out);
    return lvq_launch_status();
}
