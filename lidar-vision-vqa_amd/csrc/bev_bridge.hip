// csrc/bev_bridge.hip -- the data formats either side of the sparse backbone, gfx950.
//
//   lvq_f16_to_f32          f2  training/data/dataset.py:139-146: `torch.from_numpy(np.load(path)).float()` -- the stored
//                               fp16 BEV [C,H,W] (writer: get-data/precompute_bev_features.py:391-395) up-cast on load.
//                               Exact (every fp16 value is an fp32 value); HBM-bound, 6 bytes per element.
//   lvq_sparse_bev_merge    f3  spconv_backbone_voxelnext.py:149-164 `bev_out`: indices[:, [0,2,3]] -> torch.unique(dim=0,
//                               return_inverse) (rows sorted lexicographically by (b, y, x)) -> index_add_ of the features.
//                               The unique step IS the dynamic voxeliser's problem (integer cells instead of points), so it
//                               runs on the same slab-binned kernels: key = (b*ny + y)*nx + x ascending == the row order of
//                               torch.unique(dim=0).  index_add_ = per-output-row contributor lists (CSR) summed in ascending input-row
//                               order: deterministic and bit-identical to the CPU index_add_ (the first version used one fp32 atomic
//                               per (row, channel), as the reference's CUDA index_add_ does; same speed, unspecified order).
//   lvq_sparse_to_dense     f3  map_to_bev/height_compression.py:10-26: SparseConvTensor.dense() [N,C,D,H,W] viewed as
//                               [N, C*D, H, W]; with D == 1 and (b, y, x) indices it is the `.dense()` that
//                               precompute_bev_features.py stores.  Zero-fill + scatter.
#include "common.h"
#include <hip/hip_fp16.h>

namespace {

__device__ __forceinline__ float h2f(unsigned short u) {
    __half_raw r;
    r.x = u;
    return __half2float(__half(r));
}

__global__ void __launch_bounds__(256) k_f16_to_f32(const uint16_t *__restrict__ src, float *__restrict__ dst, int64_t n) {
    // 8 halves (16 bytes) in, 32 bytes out per thread
    const int64_t i8 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i8 + 8 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + i8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[2 * k] = h2f((unsigned short)(w[k] & 0xffffu));
            o[2 * k + 1] = h2f((unsigned short)(w[k] >> 16));
        }
        float4 *d4 = reinterpret_cast<float4 *>(dst + i8);
        d4[0] = make_float4(o[0], o[1], o[2], o[3]);
        d4[1] = make_float4(o[4], o[5], o[6], o[7]);
    } else {
        for (int64_t i = i8; i < n; ++i) dst[i] = h2f(src[i]);
    }
}

// (b, z, y, x) int32 rows -> the "points" the dynamic voxeliser takes: (batch, x' = y, y' = x, 0) with unit cells
__global__ void __launch_bounds__(256) k_idx_to_pts(const int4 *__restrict__ idx, int64_t m, float4 *__restrict__ pts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int4 v = idx[i];
    pts[i] = make_float4((float)v.x, (float)v.z, (float)v.w, 0.0f);
}

// dynamic voxeliser cells (b, 0, cy' = x, cx' = y) -> unique index rows (b, y, x)
__global__ void __launch_bounds__(256) k_cells_to_byx(const int4 *__restrict__ cells, const int32_t *__restrict__ counts, int64_t cap,
                                                      int32_t *__restrict__ out_byx) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t m = counts[0];
    if (m > cap) m = cap;
    if (v >= m) return;
    const int4 c = cells[v];
    out_byx[v * 3] = c.x;
    out_byx[v * 3 + 1] = c.w;
    out_byx[v * 3 + 2] = c.z;
}

// ---- features_unique.index_add_(0, inv, features) without atomics ----
// The first version issued one fp32 atomic per (row, channel): 12.6 M atomics for 98 729 x 128, order unspecified.  The unique step
// already knows every output row's contributor count, so: (1) block-local exclusive prefix of the counts + per-block totals,
// (2) every input row drops its index into its output row's list (one int atomic per ROW; every block rescans the few hundred
// block totals), (3) one wave per output row orders its (few) contributors by input index and sums them in that order -- the
// order of the CPU index_add_ the reference's semantics come from, so the sums are deterministic and bit-identical to it.
constexpr int CSR_BLK = 1024;                     // counts per block of the prefix kernel
constexpr int CSR_ROWS_PER_WAVE = 2;              // measured 2 vs 16: 126 / 153 us (4 scenes); the atomic version took 119 us
__global__ void __launch_bounds__(256) k_csr_local(const int32_t *__restrict__ cnt, const int32_t *__restrict__ counts, int32_t *__restrict__ off_local,
                                                   int32_t *__restrict__ btot) {
    __shared__ int wt[4];
    const int m2 = counts[0];
    const int base = blockIdx.x * CSR_BLK + threadIdx.x * 4;
    int c[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = base + j < m2 ? cnt[base + j] : 0; s += c[j]; }
    int incl = s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wt[wid] = incl;
    __syncthreads();
    int wb = 0;
    for (int w = 0; w < wid; ++w) wb += wt[w];
    int ex = wb + incl - s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (base + j < m2) off_local[base + j] = ex;
        ex += c[j];
    }
    if (threadIdx.x == 255) btot[blockIdx.x] = wb + incl;
}

// exclusive scan of the block totals into LDS (nblk <= 4096), by every block that needs global offsets
__device__ __forceinline__ void csr_block_prefix(const int32_t *__restrict__ btot, int nblk, int *bpre, int *wt) {
    const int per = (nblk + 255) / 256;
    int s = 0;
    for (int j = 0; j < per; ++j) {
        const int b = threadIdx.x * per + j;
        if (b < nblk) s += btot[b];
    }
    int incl = s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wt[wid] = incl;
    __syncthreads();
    int ex = incl - s;
    for (int w = 0; w < wid; ++w) ex += wt[w];
    for (int j = 0; j < per; ++j) {
        const int b = threadIdx.x * per + j;
        if (b < nblk) { bpre[b] = ex; ex += btot[b]; }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_csr_fill(const int32_t *__restrict__ inv, int64_t m, const int32_t *__restrict__ off_local,
                                                  const int32_t *__restrict__ btot, int nblk, int32_t *__restrict__ fillc, int32_t *__restrict__ list) {
    extern __shared__ int bpre[];
    __shared__ int wt[4];
    csr_block_prefix(btot, nblk, bpre, wt);
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const int v = inv[r];
    if (v < 0) return;                            // row outside the grid (the reference never produces one)
    list[bpre[v / CSR_BLK] + off_local[v] + atomicAdd(&fillc[v], 1)] = (int32_t)r;
}

__global__ void __launch_bounds__(256) k_csr_sum(const float *__restrict__ feats, int c, const int32_t *__restrict__ cnt, const int32_t *__restrict__ counts,
                                                 const int32_t *__restrict__ off_local, const int32_t *__restrict__ btot, int nblk,
                                                 const int32_t *__restrict__ list, float *__restrict__ out) {
    extern __shared__ int bpre[];
    __shared__ int wt[4];
    csr_block_prefix(btot, nblk, bpre, wt);
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < CSR_ROWS_PER_WAVE; ++i) {               // 64 output rows per block amortise the prefix above
    const int v = (blockIdx.x * 4 + (threadIdx.x >> 6)) * CSR_ROWS_PER_WAVE + i;
    if (v >= counts[0]) return;
    const int n = cnt[v];
    const int32_t *l = list + bpre[v / CSR_BLK] + off_local[v];
    // contributors in ascending input-row order: repeatedly take the smallest index above the last one (n is a handful)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};          // channels lane, lane + 64, ... (c <= 256 per pass)
    for (int c0 = 0; c0 < c; c0 += 256) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = 0.f;
        int last = -1;
        for (int k = 0; k < n; ++k) {
            int nxt = 0x7fffffff;
            for (int q = 0; q < n; ++q) {
                const int cand = l[q];
                if (cand > last && cand < nxt) nxt = cand;
            }
            last = nxt;
            const float *row = feats + (int64_t)nxt * c + c0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = lane + 64 * j;
                if (c0 + ch < c) acc[j] += row[ch];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = lane + 64 * j;
            if (c0 + ch < c) out[(int64_t)v * c + c0 + ch] = acc[j];
        }
    }
    }
}

// ---- dense(): index map + gather.  (The first version zero-filled the output and scattered 4-byte stores at plane stride:
// 177 us for [16,128,180,180]; one coalesced pass over the output is the byte roofline of this op.) ----
// map[b, z, y, x] = row or -1
__global__ void __launch_bounds__(256) k_map_rows(const int32_t *__restrict__ idx, int icols, int64_t m_cap, const int32_t *__restrict__ n_live,
                                                  int batch, int d, int h, int w, int32_t *__restrict__ map) {
    int64_t m = n_live ? (int64_t)*n_live : m_cap;
    if (m > m_cap) m = m_cap;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const int32_t *row = idx + r * icols;
    const int b = row[0];
    const int z = icols == 4 ? row[1] : 0, y = row[icols - 2], x = row[icols - 1];
    if (b < 0 || b >= batch || z < 0 || z >= d || y < 0 || y >= h || x < 0 || x >= w) return;
    map[(((int64_t)b * d + z) * h + y) * w + x] = (int)r;
}

// one workgroup per (b, z, y) line: gather the line's feature rows 32 channels at a time into an LDS tile [32][w] (coalesced
// 128-byte reads per row), write every channel's w contiguous floats of out[b, ch*d + z, y, :]
constexpr int DENSE_CH = 32;
__global__ void __launch_bounds__(256) k_dense_lines(const float *__restrict__ feats, const int32_t *__restrict__ map, int c, int d, int h,
                                                     int w, float *__restrict__ out) {
    extern __shared__ float tile[];                 // [DENSE_CH][w + 1] | int lmap[w]
    const int line = blockIdx.x;                    // (b * d + z) * h + y
    const int y = line % h, bz = line / h, z = bz % d, b = bz / d;
    const int32_t *mrow = map + (int64_t)line * w;
    const int64_t plane = (int64_t)h * w;
    const int wp = w + 1;
    int *lmap = reinterpret_cast<int *>(tile + DENSE_CH * wp);
    for (int x = threadIdx.x; x < w; x += 256) lmap[x] = mrow[x];
    __syncthreads();
    for (int ch0 = 0; ch0 < c; ch0 += DENSE_CH) {
        const int nch = c - ch0 < DENSE_CH ? c - ch0 : DENSE_CH;
        // gather: 32 consecutive lanes read 128 contiguous bytes of one row, 8 rows per sweep, 8 sweeps' loads in flight
        // (one load per iteration left every iteration waiting on its own L2 / HBM round trip: 219 us instead of ~90)
        {
            const int k = threadIdx.x & (DENSE_CH - 1);
            for (int x0 = threadIdx.x / DENSE_CH; x0 < w; x0 += 64) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int x = x0 + 8 * u;
                    const int r = x < w ? lmap[x] : -1;
                    v[u] = (r >= 0 && k < nch) ? feats[(int64_t)r * c + ch0 + k] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int x = x0 + 8 * u;
                    if (x < w) tile[k * wp + x] = v[u];
                }
            }
        }
        __syncthreads();
        for (int x = threadIdx.x; x < w; x += 256) {
            float *dst = out + (((int64_t)b * c + ch0) * d + z) * plane + (int64_t)y * w + x;
            for (int k = 0; k < nch; ++k) dst[(int64_t)k * d * plane] = tile[k * wp + x];
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int lvq_f16_to_f32(const uint16_t *src, float *dst, int64_t n, lvq_stream_t stream) {
    if (n < 0) return LVQ_EINVAL;
    if (n == 0) return LVQ_OK;
    if (!src || !dst) return LVQ_EINVAL;
    if ((((uintptr_t)src) & 15) || (((uintptr_t)dst) & 15)) return LVQ_EUNSUPPORTED;
    hipLaunchKernelGGL(k_f16_to_f32, dim3((unsigned)lvq_cdiv(lvq_cdiv(n, 8), 256)), dim3(256), 0, lvq_s(stream), src, dst, n);
    return lvq_launch_status();
}

namespace {
struct MergeWs {
    float4 *pts;
    int32_t *unq_key, *unq_cnt, *cells;
    int32_t *off_local, *btot, *fillc, *list;
    void *dyn;
    size_t dyn_bytes;
};
template <typename A> void merge_layout(A &a, MergeWs &w, int64_t m, size_t dyn_bytes) {
    w.pts = a.template take<float4>(m + 1);
    w.unq_key = a.template take<int32_t>(m + 1);
    w.unq_cnt = a.template take<int32_t>(m + 1);
    w.cells = a.template take<int32_t>(4 * (m + 1));
    w.off_local = a.template take<int32_t>(m + 1);
    w.btot = a.template take<int32_t>(m / CSR_BLK + 2);
    w.fillc = a.template take<int32_t>(m + 1);
    w.list = a.template take<int32_t>(m + 1);
    w.dyn = a.template take<char>(dyn_bytes);
    w.dyn_bytes = dyn_bytes;
}
struct SizerA {
    LvqSizer s;
    template <typename T> T *take(size_t n) { s.template take<T>(n); return nullptr; }
};
}  // namespace

extern "C" size_t lvq_sparse_bev_merge_workspace_bytes(int64_t m, int batch, int ny, int nx) {
    if (m < 0 || batch <= 0 || ny <= 0 || nx <= 0) return 0;
    const int32_t grid[3] = {ny, nx, 1};
    const size_t dyn = lvq_voxelize_dynamic_workspace_bytes(m, batch, grid, 2);
    if (dyn == 0) return 0;
    SizerA a;
    MergeWs w;
    merge_layout(a, w, m, dyn);
    return a.s.total();
}

extern "C" int lvq_sparse_bev_merge(const int32_t *indices_bzyx, const float *feats, int64_t m, int c, int batch, int ny, int nx,
                                    int32_t *out_indices_byx, float *out_feats, int32_t *unq_inv, int32_t *counts, void *ws,
                                    size_t ws_bytes, lvq_stream_t stream) {
    if (m < 0 || c <= 0 || batch <= 0 || ny <= 0 || nx <= 0 || !counts) return LVQ_EINVAL;
    if (ny >= (1 << 24) || nx >= (1 << 24) || batch >= (1 << 24)) return LVQ_EUNSUPPORTED;     // indices pass through fp32 exactly
    hipStream_t st = lvq_s(stream);
    if (m == 0) {
        hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), st);
        return lvq_launch_status();
    }
    if (!indices_bzyx || !feats || !out_indices_byx || !out_feats || !unq_inv || (((uintptr_t)indices_bzyx) & 15)) return LVQ_EINVAL;
    const int32_t grid[3] = {ny, nx, 1};
    const size_t dyn = lvq_voxelize_dynamic_workspace_bytes(m, batch, grid, 2);
    if (dyn == 0) return LVQ_EOVERFLOW;
    LvqArena arena(ws, ws_bytes);
    MergeWs w;
    merge_layout(arena, w, m, dyn);
    if (!arena.ok) return LVQ_EWORKSPACE;
    const unsigned nb = (unsigned)lvq_cdiv(m, 256);
    hipLaunchKernelGGL(k_idx_to_pts, dim3(nb), dim3(256), 0, st, reinterpret_cast<const int4 *>(indices_bzyx), m, w.pts);
    const float range[6] = {0.f, 0.f, 0.f, (float)ny, (float)nx, 1.f};
    const float vsize[3] = {1.f, 1.f, 1.f};
    const int rc = lvq_voxelize_dynamic(reinterpret_cast<const float *>(w.pts), m, 4, batch, range, vsize, grid, 2, unq_inv, nullptr,
                                        w.unq_key, w.unq_cnt, w.cells, counts, w.dyn, w.dyn_bytes, stream);
    if (rc != LVQ_OK) return rc;
    hipLaunchKernelGGL(k_cells_to_byx, dim3(nb), dim3(256), 0, st, reinterpret_cast<const int4 *>(w.cells), counts, m, out_indices_byx);
    const int nblk = (int)lvq_cdiv(m, CSR_BLK);
    if (nblk > 8192) return LVQ_EUNSUPPORTED;                   // block totals are rescanned in LDS (m <= 8 M rows)
    hipMemsetAsync(w.fillc, 0, sizeof(int32_t) * (size_t)m, st);
    hipLaunchKernelGGL(k_csr_local, dim3((unsigned)nblk), dim3(256), 0, st, w.unq_cnt, counts, w.off_local, w.btot);
    hipLaunchKernelGGL(k_csr_fill, dim3(nb), dim3(256), sizeof(int) * nblk, st, unq_inv, m, w.off_local, w.btot, nblk, w.fillc, w.list);
    hipLaunchKernelGGL(k_csr_sum, dim3((unsigned)lvq_cdiv(m, 4 * CSR_ROWS_PER_WAVE)), dim3(256), sizeof(int) * nblk, st, feats, c, w.unq_cnt, counts, w.off_local, w.btot,
                       nblk, w.list, out_feats);
    return lvq_launch_status();
}

extern "C" size_t lvq_sparse_to_dense_workspace_bytes(int batch, int d, int h, int w) {
    if (batch <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
    return lvq_align((size_t)batch * d * h * w * sizeof(int32_t)) + 256;
}

extern "C" int lvq_sparse_to_dense(const float *feats, const int32_t *indices, int index_cols, int64_t m_cap, const int32_t *n_live_dev,
                                   int c, int batch, int d, int h, int w, float *out, void *ws, size_t ws_bytes, lvq_stream_t stream) {
    if (m_cap < 0 || c <= 0 || batch <= 0 || d <= 0 || h <= 0 || w <= 0 || !out || (index_cols != 3 && index_cols != 4)) return LVQ_EINVAL;
    if (index_cols == 3 && d != 1) return LVQ_EINVAL;
    if ((m_cap > 0 && (!feats || !indices))) return LVQ_EINVAL;
    const int64_t lines = (int64_t)batch * d * h;
    if (lines >= (1ll << 31) || w > 8192) return LVQ_EUNSUPPORTED;
    const size_t map_bytes = (size_t)lines * w * sizeof(int32_t);
    if (!ws || ws_bytes < map_bytes) return LVQ_EWORKSPACE;
    hipStream_t st = lvq_s(stream);
    int32_t *map = reinterpret_cast<int32_t *>(ws);
    hipMemsetAsync(map, 0xff, map_bytes, st);                     // -1 everywhere
    if (m_cap > 0)
        hipLaunchKernelGGL(k_map_rows, dim3((unsigned)lvq_cdiv(m_cap, 256)), dim3(256), 0, st, indices, index_cols, m_cap, n_live_dev, batch, d,
                           h, w, map);
    hipLaunchKernelGGL(k_dense_lines, dim3((unsigned)lines), dim3(256), sizeof(float) * (DENSE_CH * (w + 1) + w), st, feats, map, c, d, h, w, out);
    return lvq_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// Dense BEV canvas -> its occupied cells (the inverse of PointPillarScatter): what lets VATLiDAR.forward(bev), the reference's own entry
// point (vat_lidar.py:187, fed by the fp16 .npy canvases of precompute_bev_features.py:391-395), take the sparse key stream of
// bev_tiles.hip.  A cell whose C channels are all zero is exactly an absent pillar: the refine conv sees zeros there either way.
// One thread per cell: the C channel planes are read coalesced along x, a wave allots its occupied cells contiguous rows with ONE atomic,
// and the rows leave through an LDS transpose ([channel][cell], pitch 65: conflict-free), two cells per step, as 128-byte row segments.  Row order
// follows the atomics: downstream addresses pillars through the coordinate -> row map only, so results do not depend on it.
// ---------------------------------------------------------------------------------------------------------
namespace {
// LDS hand-off between lanes of ONE wave (the LDS queue is in order per wave: only the compiler must be kept from reordering)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__global__ void __launch_bounds__(256) k_bev_cells(const float *__restrict__ bev, int batch, int c, int h, int w, int64_t cap,
                                                   float *__restrict__ feats, int32_t *__restrict__ coords, int32_t *__restrict__ n_cells) {
    __shared__ float stage[4][32][65];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t hw = (int64_t)h * w, total = (int64_t)batch * hw;
    const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = cell < total;
    const int64_t cc = in ? cell : total - 1;
    const int b = (int)(cc / hw);
    const int64_t yx = cc - (int64_t)b * hw;
    const float *src = bev + (int64_t)b * c * hw + yx;
    bool nz = false;
    for (int k0 = 0; k0 < c; k0 += 8) {                          // eight independent loads in flight
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (k0 + u < c) ? src[(int64_t)(k0 + u) * hw] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) nz = nz || (v[u] != 0.f);
    }
    nz = nz && in;
    const unsigned long long mask = __ballot(nz);
    if (mask == 0ull) return;                                    // (wave-uniform)
    const int cnt = __popcll(mask);
    int base = 0;
    if (lane == 0) base = atomicAdd(n_cells, cnt);
    base = __shfl(base, 0);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    const int64_t row = (int64_t)base + rank;
    if (nz && row < cap) reinterpret_cast<int4 *>(coords)[row] = make_int4(b, 0, (int)(yx / w), (int)(yx % w));
    // the wave's occupied cells in rank order (the lanes that hold them), two per step: lanes 0..31 and 32..63 write 32 channels each
    const int half = lane >> 5, l31 = lane & 31;
    for (int k0 = 0; k0 < c; k0 += 32) {                         // 32 channels at a time through the transpose
        const int kc = c - k0 < 32 ? c - k0 : 32;
        for (int k = 0; k < kc; ++k) stage[wid][k][lane] = nz ? src[(int64_t)(k0 + k) * hw] : 0.f;       // (L2-hot: just read)
        wave_sync();
        unsigned long long m = mask;
        for (int r = 0; r < cnt; r += 2) {
            const int c0 = __builtin_ctzll(m);
            m &= m - 1;
            int c1 = c0;
            if (r + 1 < cnt) { c1 = __builtin_ctzll(m); m &= m - 1; }
            const int cl = half ? c1 : c0;
            const int64_t ro = (int64_t)base + r + half;
            if (l31 < kc && r + half < cnt && ro < cap) feats[ro * c + k0 + l31] = stage[wid][l31][cl];
        }
        wave_sync();
    }
}
}  // namespace

extern "C" int lvq_bev_occupied_cells(const float *bev, int batch, int c, int h, int w, int64_t cap, float *feats, int32_t *coords_bzyx,
                                      int32_t *n_cells, lvq_stream_t stream) {
    if (!bev || batch <= 0 || c <= 0 || h <= 0 || w <= 0 || cap < 0 || !feats || !coords_bzyx || !n_cells) return LVQ_EINVAL;
    const int64_t total = (int64_t)batch * h * w;
    if (total >= (1ll << 31) || (((uintptr_t)coords_bzyx) & 15)) return LVQ_EUNSUPPORTED;
    hipStream_t st = lvq_s(stream);
    if (hipMemsetAsync(n_cells, 0, sizeof(int32_t), st) != hipSuccess) return LVQ_ELAUNCH;
    hipLaunchKernelGGL(k_bev_cells, dim3((unsigned)lvq_cdiv(total, 256)), dim3(256), 0, st, bev, batch, c, h, w, cap, feats, coords_bzyx, n_cells);
    return lvq_launch_status();
}
