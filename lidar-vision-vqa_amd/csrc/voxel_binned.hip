// csrc/voxel_binned.hip -- slab-binned voxelisation: coalesced passes + LDS-local hashing (gfx950).
//
// The first implementation (voxel.hip) is exact but bound by OPERATIONS, not bytes: 3 scattered memory-side
// atomics per point (~20 G/s chip-wide) and 2-3 dependent beyond-L2 gathers per point (~10 G/s) -> 283 us for
// 524k points where the byte roofline is ~15 us.  This file restructures the work the MI355X way:
//
//   pass A  (k_bin_hist / k_bin_scatter)  one counting sort of the points by SLAB = a contiguous range of 2^LOGSLAB
//           linear keys (key = ((b*nx + cx)*ny + cy)*nz + cz, the reference's ascending torch.unique order).
//           Histograms are built in LDS per 4096-point block (LDS atomics), only non-empty bins touch global memory,
//           and the scatter writes (orig index, in-slab offset[, 16-byte point]) runs grouped per (block, slab).
//   pass B  one workgroup per slab, everything in LDS: a DENSE occupancy bitmap of the slab (<= 16 KB), a popcount
//           scan that ranks the occupied cells (= sorted-unique order inside the slab, and slabs are key-ordered, so
//           the dynamic path's unq_key / coords / counts leave as SEQUENTIAL stores), per-voxel counters in LDS.
//   global  only tiny scans over slabs (<= 8192 entries).
//
// Caps: a slab with more voxels than the LDS counters hold falls back to global atomics for that slab only.
// Shapes the binned path does not take (key space > 2^30, C != 4 payloads, ...) use the voxel.hip kernels.
#include "common.h"

namespace vb {

struct Geom {
    float lo[3];
    float vs[3];
    int grid[3];
};

struct BinCfg {
    int logslab;       // keys per slab = 1 << logslab
    int nslabs;        // <= 8192
    int ndim;          // 2 or 3 (dynamic); hard is always 3
    int n_scenes;
    int mode;          // 0: dynamic layout (col 0 = batch idx, xyz in 1..3); 1: hard layout (scene from scene_off, xyz in 0..2)
};

constexpr int MAX_SLABS = 8192;
constexpr int BIN_BLOCK = 1024;         // threads
constexpr int BIN_PPT = 4;              // points per thread

__device__ __forceinline__ int find_scene(const int32_t *off, int n_scenes, int i) {
    int lo = 0, hi = n_scenes;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// linear key of point i or -1; cc = cell coordinates (cx, cy, cz)
__device__ __forceinline__ int64_t key_of(const float *__restrict__ pts, int i, int c, const Geom &g, const BinCfg &cfg,
                                         const int32_t *__restrict__ scene_off, int cc[3]) {
    const float *p = pts + (int64_t)i * c;
    const float *xyz = cfg.mode == 0 ? p + 1 : p;
    bool ok = true;
    cc[0] = cc[1] = cc[2] = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < cfg.ndim) {
            float d = xyz[j] - g.lo[j];
            float q = d / g.vs[j];
            float f = floorf(q);
            bool in = (f >= 0.0f) && (f < (float)g.grid[j]);
            ok = ok && in;
            cc[j] = in ? (int)f : -1;
        }
    }
    int b;
    if (cfg.mode == 0) {
        b = (int)p[0];
        ok = ok && b >= 0 && b < cfg.n_scenes;
    } else {
        b = find_scene(scene_off, cfg.n_scenes, i);
    }
    if (!ok) return -1;
    int64_t key = ((int64_t)b * g.grid[0] + cc[0]) * g.grid[1] + cc[1];
    if (cfg.ndim == 3) key = key * g.grid[2] + cc[2];
    return key;
}

// ---- pass A1: per-slab histogram (LDS), pt_coords / invalid markers ----
__global__ void __launch_bounds__(BIN_BLOCK) k_bin_hist(const float *__restrict__ pts, int n, int c, Geom g, BinCfg cfg,
                                                        const int32_t *__restrict__ scene_off, int32_t *__restrict__ ghist,
                                                        int32_t *__restrict__ pt_coords, int32_t *__restrict__ inv_or_null) {
    extern __shared__ int32_t lh[];
    for (int b = threadIdx.x; b < cfg.nslabs; b += BIN_BLOCK) lh[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * (BIN_BLOCK * BIN_PPT);
#pragma unroll
    for (int u = 0; u < BIN_PPT; ++u) {
        const int i = base + u * BIN_BLOCK + threadIdx.x;
        if (i < n) {
            int cc[3];
            const int64_t key = key_of(pts, i, c, g, cfg, scene_off, cc);
            if (pt_coords) { pt_coords[i * 3] = cc[0]; pt_coords[i * 3 + 1] = cc[1]; pt_coords[i * 3 + 2] = cc[2]; }
            if (key >= 0) atomicAdd(&lh[(int)(key >> cfg.logslab)], 1);
            else if (inv_or_null) inv_or_null[i] = -1;
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < cfg.nslabs; b += BIN_BLOCK)
        if (lh[b]) atomicAdd(&ghist[b], lh[b]);
}

// ---- single-workgroup exclusive scan of <= MAX_SLABS ints: out[0..n] ; also zeroes `zero_me` ----
__global__ void __launch_bounds__(1024) k_scan_small(const int32_t *__restrict__ in, int n, int32_t *__restrict__ out,
                                                     int32_t *__restrict__ zero_me, int32_t *__restrict__ total_out) {
    __shared__ int wave_tot[16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int running = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? in[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            const int t = wave_tot[w];
            if (w < wid) wbase += t;
            tot += t;
        }
        __syncthreads();
        if (i < n) {
            out[i] = running + wbase + incl - v;
            if (zero_me) zero_me[i] = 0;
        }
        running += tot;
    }
    if (threadIdx.x == 0) {
        out[n] = running;
        if (total_out) *total_out = running;
    }
}

// ---- pass A3: scatter (orig idx, in-slab offset[, point]) grouped by slab ----
// scan_here != 0 (dynamic path): the block scans the global histogram itself (<= 8192 entries, L2-resident) instead of a
// separate single-workgroup scan kernel; block 0 publishes gstart[0..nslabs] and the number of valid points.
__global__ void __launch_bounds__(BIN_BLOCK) k_bin_scatter(const float *__restrict__ pts, int n, int c, Geom g, BinCfg cfg,
                                                           const int32_t *__restrict__ scene_off,
                                                           int32_t *__restrict__ gstart, int32_t *__restrict__ cursor,
                                                           int32_t *__restrict__ sidx, int32_t *__restrict__ soff,
                                                           float4 *__restrict__ spts, const int32_t *__restrict__ ghist_or_null,
                                                           int32_t *__restrict__ total_out) {
    extern __shared__ int32_t lds[];
    __shared__ int wave_tot_s[BIN_BLOCK / 64];
    int32_t *lh = lds, *lb = lds + cfg.nslabs, *gs = lds + 2 * cfg.nslabs;
    if (ghist_or_null) {
        const int per = (cfg.nslabs + BIN_BLOCK - 1) / BIN_BLOCK;
        int csum = 0;
        for (int j = 0; j < per; ++j) {
            const int b = threadIdx.x * per + j;
            if (b < cfg.nslabs) csum += ghist_or_null[b];
        }
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        int incl = csum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot_s[wid] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < BIN_BLOCK / 64; ++w) {
            const int t = wave_tot_s[w];
            if (w < wid) wbase += t;
            tot += t;
        }
        int ex = wbase + incl - csum;
        for (int j = 0; j < per; ++j) {
            const int b = threadIdx.x * per + j;
            if (b < cfg.nslabs) {
                gs[b] = ex;
                if (blockIdx.x == 0) gstart[b] = ex;
                ex += ghist_or_null[b];
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            gstart[cfg.nslabs] = tot;
            if (total_out) *total_out = tot;
        }
    }
    for (int b = threadIdx.x; b < cfg.nslabs; b += BIN_BLOCK) lh[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * (BIN_BLOCK * BIN_PPT);
    int64_t keys[BIN_PPT];
#pragma unroll
    for (int u = 0; u < BIN_PPT; ++u) {
        const int i = base + u * BIN_BLOCK + threadIdx.x;
        keys[u] = -1;
        if (i < n) {
            int cc[3];
            keys[u] = key_of(pts, i, c, g, cfg, scene_off, cc);
            if (keys[u] >= 0) atomicAdd(&lh[(int)(keys[u] >> cfg.logslab)], 1);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < cfg.nslabs; b += BIN_BLOCK) {
        const int h = lh[b];
        lb[b] = h ? (ghist_or_null ? gs[b] : gstart[b]) + atomicAdd(&cursor[b], h) : 0;
        lh[b] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BIN_PPT; ++u) {
        const int i = base + u * BIN_BLOCK + threadIdx.x;
        if (keys[u] >= 0) {
            const int s = (int)(keys[u] >> cfg.logslab);
            const int pos = lb[s] + atomicAdd(&lh[s], 1);
            sidx[pos] = i;
            soff[pos] = (int)(keys[u] & ((1ll << cfg.logslab) - 1));
            if (spts) spts[pos] = reinterpret_cast<const float4 *>(pts)[i];
        }
    }
}

constexpr int SLAB_NT = 1024;     // threads per slab workgroup (dense slabs set the tail; 256 threads were 2x slower)

// ---- block-wide exclusive scan helper (blockDim.x = SLAB_NT) ----
template <int NT = SLAB_NT>
__device__ __forceinline__ int block_excl_scan256(int v, int *wave_tot, int &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const int t = wave_tot[w];
        if (w < wid) wbase += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return wbase + incl - v;
}

// floor(a / d) for 0 <= a < 2^31, d >= 1.  (The first version estimated the quotient in fp32 with one correction step each way;
// above ~2^29 the estimate is off by more than one divisor and keys of scenes >= 12 on the 0.1 m grid decoded to wrong
// (z, y, x) -- found by tests/test_gpu_lidar.py::test_hard_voxelizer_paths_agree_at_scale.)  The fp32 estimate is within a few
// units of the quotient (relative error 2^-23 of a quotient < 2^26); the correction LOOPS make it exact for every a.
__device__ __forceinline__ int fdiv(int a, int d, float rd) {
    int q = (int)((float)a * rd);
    int r = a - q * d;
    while (r < 0) { --q; r += d; }
    while (r >= d) { ++q; r -= d; }
    return q;
}

__device__ __forceinline__ int popc_below(uint64_t m, int bit) {
    return __popcll(m & ((bit == 0) ? 0ull : (~0ull >> (64 - bit))));
}

constexpr int DYN_NT = 512;       // threads per slab workgroup of the dynamic path (four workgroups per CU instead of two at 1024)
// ---- dynamic pass B: occupied cells per slab ----
__global__ void __launch_bounds__(DYN_NT) k_dyn_slab_count(BinCfg cfg, const int32_t *__restrict__ gstart,
                                                        const int32_t *__restrict__ soff, int32_t *__restrict__ slab_cnt) {
    extern __shared__ unsigned long long bm[];
    __shared__ int wave_tot[DYN_NT / 64];
    const int s = blockIdx.x;
    const int p0 = gstart[s], p1 = gstart[s + 1];
    if (p0 == p1) {
        if (threadIdx.x == 0) slab_cnt[s] = 0;
        return;
    }
    const int nw = 1 << (cfg.logslab - 6);
    for (int w = threadIdx.x; w < nw; w += DYN_NT) bm[w] = 0ull;
    __syncthreads();
    for (int p = p0 + threadIdx.x; p < p1; p += DYN_NT) {
        const int off = soff[p];
        atomicOr(&bm[off >> 6], 1ull << (off & 63));
    }
    __syncthreads();
    int c = 0;
    for (int w = threadIdx.x; w < nw; w += DYN_NT) c += __popcll(bm[w]);
    int tot;
    block_excl_scan256<DYN_NT>(c, wave_tot, tot);
    if (threadIdx.x == 0) slab_cnt[s] = tot;
}

constexpr int DYN_VCAP = 4096;   // per-slab voxel counters held in LDS

// ---- dynamic pass B': ranks, inverse map, sequential unique outputs ----
__global__ void __launch_bounds__(DYN_NT) k_dyn_slab_write(BinCfg cfg, Geom g, const int32_t *__restrict__ gstart,
                                                        const int32_t *__restrict__ sidx, const int32_t *__restrict__ soff,
                                                        const int32_t *__restrict__ slab_cnt, int32_t *__restrict__ unq_inv,
                                                        int32_t *__restrict__ unq_key, int32_t *__restrict__ unq_cnt,
                                                        int32_t *__restrict__ coords_bzyx, int32_t *__restrict__ m_out) {
    extern __shared__ unsigned long long smem64[];
    __shared__ int wave_tot[DYN_NT / 64];
    const int s = blockIdx.x;
    const int p0 = gstart[s], p1 = gstart[s + 1];
    if (p0 == p1 && s != 0) return;
    // first voxel of this slab = occupied cells of all slabs before it: every block sums the <= 8192 per-slab counts itself
    // (no single-workgroup scan kernel in between); block 0 also publishes the total M
    int before = 0, all = 0;
    for (int b = threadIdx.x; b < cfg.nslabs; b += DYN_NT) {
        const int v = slab_cnt[b];
        all += v;
        if (b < s) before += v;
    }
    int vbase, total;
    block_excl_scan256<DYN_NT>(before, wave_tot, vbase);
    if (s == 0) {
        block_excl_scan256<DYN_NT>(all, wave_tot, total);
        if (threadIdx.x == 0) *m_out = total;
        if (p0 == p1) return;
    }
    const int nvox = slab_cnt[s];
    const int nw = 1 << (cfg.logslab - 6);
    unsigned long long *bm = smem64;                               // [nw]
    int32_t *wpre = reinterpret_cast<int32_t *>(smem64 + nw);      // [nw]
    int32_t *cnt_l = wpre + nw;                                    // [DYN_VCAP]
    const bool lds_cnt = nvox <= DYN_VCAP;
    if (!lds_cnt) {                                   // oversized slab: counters live in this slab's range of unq_cnt
        for (int v = threadIdx.x; v < nvox; v += DYN_NT) unq_cnt[vbase + v] = 0;
        __threadfence();
    }
    for (int w = threadIdx.x; w < nw; w += DYN_NT) bm[w] = 0ull;
    for (int v = threadIdx.x; v < DYN_VCAP; v += DYN_NT) cnt_l[v] = 0;
    __syncthreads();
    for (int p = p0 + threadIdx.x; p < p1; p += DYN_NT) {
        const int off = soff[p];
        atomicOr(&bm[off >> 6], 1ull << (off & 63));
    }
    __syncthreads();
    // exclusive popcount scan over the bitmap words (thread t owns words t*per .. t*per+per-1)
    const int per = (nw + DYN_NT - 1) / DYN_NT;
    int c = 0;
    for (int j = 0; j < per; ++j) {
        const int w = threadIdx.x * per + j;
        if (w < nw) c += __popcll(bm[w]);
    }
    int tot;
    int ex = block_excl_scan256<DYN_NT>(c, wave_tot, tot);
    for (int j = 0; j < per; ++j) {
        const int w = threadIdx.x * per + j;
        if (w < nw) { wpre[w] = ex; ex += __popcll(bm[w]); }
    }
    __syncthreads();
    // per point: rank of its voxel, inverse map, count
    for (int p = p0 + threadIdx.x; p < p1; p += DYN_NT) {
        const int off = soff[p];
        const int lr = wpre[off >> 6] + popc_below(bm[off >> 6], off & 63);
        unq_inv[sidx[p]] = vbase + lr;
        if (lds_cnt) atomicAdd(&cnt_l[lr], 1);
        else atomicAdd(&unq_cnt[vbase + lr], 1);          // oversized slab: counters in global memory (pre-zeroed)
    }
    __syncthreads();
    // unique outputs in ascending key order.  Voxel v of the slab belongs to thread v % DYN_NT: consecutive lanes write consecutive rows
    // (full-line stores) and every thread gets the same number of voxels.  (Round 2 walked the bitmap instead, four words per thread:
    // the occupied cells of a LiDAR slab are clustered, so a few threads held up to 256 voxels -- each with three reciprocal divisions --
    // while most held none, and the slab's time was that serial tail: 45 of the path's 82 us.)
    const int64_t key0 = (int64_t)s << cfg.logslab;
    const float rx = 1.0f / (float)g.grid[0], ry = 1.0f / (float)g.grid[1], rz = 1.0f / (float)g.grid[2];
    for (int v = threadIdx.x; v < nvox; v += DYN_NT) {
        // the word that holds the v-th occupied cell: the LAST w with wpre[w] <= v (wpre = exclusive popcount prefix; empty words repeat it)
        int lo = 0, hi = nw - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (wpre[mid] <= v) lo = mid; else hi = mid - 1;
        }
        const unsigned long long bits = bm[lo];
        int k = v - wpre[lo];                                      // its k-th set bit (0-based), by halving
        unsigned int x = (unsigned int)bits;
        int pos = 0, c = __popc(x);
        if (k >= c) { k -= c; x = (unsigned int)(bits >> 32); pos = 32; }
        c = __popc(x & 0xffffu); if (k >= c) { k -= c; x >>= 16; pos += 16; } x &= 0xffffu;
        c = __popc(x & 0xffu);   if (k >= c) { k -= c; x >>= 8;  pos += 8; }  x &= 0xffu;
        c = __popc(x & 0xfu);    if (k >= c) { k -= c; x >>= 4;  pos += 4; }  x &= 0xfu;
        c = __popc(x & 0x3u);    if (k >= c) { k -= c; x >>= 2;  pos += 2; }  x &= 0x3u;
        if (k >= (int)(x & 1u)) pos += 1;
        const int key = (int)(key0 + lo * 64 + pos);
        const int vo = vbase + v;
        unq_key[vo] = key;
        if (lds_cnt) unq_cnt[vo] = cnt_l[v];
        // key -> (b, cx, cy, cz) with float-reciprocal division (exact for key < 2^30 after the fix-up step)
        int bq, cx, cy, cz = 0, t = key;
        if (cfg.ndim == 3) { const int q = fdiv(t, g.grid[2], rz); cz = t - q * g.grid[2]; t = q; }
        { const int q = fdiv(t, g.grid[1], ry); cy = t - q * g.grid[1]; t = q; }
        { const int q = fdiv(t, g.grid[0], rx); cx = t - q * g.grid[0]; bq = q; }
        reinterpret_cast<int4 *>(coords_bzyx)[vo] = make_int4(bq, cz, cy, cx);
    }
}

// slab size: average ~256-512 points per slab, dense bitmap <= 16 KB (logslab <= 17), at most MAX_SLABS slabs.  max_ls = 18 (dynamic path:
// 32-KB bitmaps, key spaces up to 2^31 -- e.g. 32 scenes of the 0.1 m nuScenes grid -- instead of handing those to the bitmap kernels of voxel.hip)
static bool choose_cfg(int64_t keyspace, int64_t n, BinCfg &cfg, int max_ls = 17) {
    if (keyspace <= 0 || keyspace > ((int64_t)MAX_SLABS << max_ls) || keyspace >= (1ll << 31)) return false;
    int64_t want = n / 384 + 1;
    if (want > MAX_SLABS) want = MAX_SLABS;
    int ls = 10;
    while (ls < 17 && ((keyspace + (1ll << ls) - 1) >> ls) > want) ++ls;      // (the preferred size stays <= 17)
    while (((keyspace + (1ll << ls) - 1) >> ls) > MAX_SLABS) {
        if (ls >= max_ls) return false;
        ++ls;
    }
    cfg.logslab = ls;
    cfg.nslabs = (int)((keyspace + (1ll << ls) - 1) >> ls);
    return true;
}

struct DynBinWs {
    int32_t *ghist, *gstart, *cursor, *slab_cnt, *slab_vbase, *sidx, *soff;
};

template <typename A> void dyn_layout(A &a, DynBinWs &w, int64_t n) {
    w.ghist = a.template take<int32_t>(2 * (MAX_SLABS + 64));
    w.cursor = w.ghist ? w.ghist + MAX_SLABS + 64 : nullptr;
    w.gstart = a.template take<int32_t>(MAX_SLABS + 2);
    w.slab_cnt = a.template take<int32_t>(MAX_SLABS + 1);
    w.slab_vbase = a.template take<int32_t>(MAX_SLABS + 2);
    w.sidx = a.template take<int32_t>(n + 1);
    w.soff = a.template take<int32_t>(n + 1);
}

struct SizerAdapter {
    LvqSizer s;
    template <typename T> T *take(size_t n) { s.template take<T>(n); return nullptr; }
};

// =================================================================================================================
// hard voxeliser on the same binning pass
// =================================================================================================================
constexpr int HARD_VCAP = 4096;    // voxels of a slab with LDS-resident per-voxel state
constexpr int HARD_PCAP = 8192;    // points of a slab with an LDS-resident bucket array

struct HardBinWs {
    int32_t *ghist, *gstart, *cursor;           // [MAX_SLABS+..]
    int32_t *sidx, *soff;                       // [n] sorted by slab
    float4 *spts;                               // [n]
    int32_t *pfirst, *pslot, *pcnt;             // [n] per sorted position: first point index of its voxel, slot, (first point) count
    uint8_t *fbytes;                            // [n] 1 = "first point of its cell" (by ORIGINAL index)
    uint64_t *fmask;                            // [nwords+1]
    int32_t *wcnt, *wpre;                       // [nwords+2]
    int32_t *sfr;                               // [n_scenes+1]
    int32_t *g_first, *g_cnt, *g_vstart, *g_fill, *g_bucket;   // [n] global stand-ins for oversized slabs
};

template <typename A> void hard_layout(A &a, HardBinWs &w, int64_t n, int n_scenes) {
    const int64_t nwords = (n + 63) / 64;
    w.ghist = a.template take<int32_t>(MAX_SLABS + 1);
    w.gstart = a.template take<int32_t>(MAX_SLABS + 2);
    w.cursor = a.template take<int32_t>(MAX_SLABS + 1);
    w.sidx = a.template take<int32_t>(n + 1);
    w.soff = a.template take<int32_t>(n + 1);
    w.spts = a.template take<float4>(n + 1);
    w.pfirst = a.template take<int32_t>(n + 1);
    w.pslot = a.template take<int32_t>(n + 1);
    w.pcnt = a.template take<int32_t>(n + 1);
    w.fbytes = a.template take<uint8_t>(n + 64);
    w.fmask = a.template take<uint64_t>(nwords + 2);
    w.wcnt = a.template take<int32_t>(nwords + 2);
    w.wpre = a.template take<int32_t>(nwords + 3);
    w.sfr = a.template take<int32_t>(n_scenes + 2);
    w.g_first = a.template take<int32_t>(n + 1);
    w.g_cnt = a.template take<int32_t>(n + 1);
    w.g_vstart = a.template take<int32_t>(n + 1);
    w.g_fill = a.template take<int32_t>(n + 1);
    w.g_bucket = a.template take<int32_t>(n + 1);
}

constexpr int SMALL_PCAP = 256;   // slabs up to this many points take the brute-force LDS kernel

// ---- hard pass B, ordinary slabs: every point scans the slab's (cell, index) pairs in LDS once ----
// slot = #points of the same cell with a smaller index, first = smallest index of the cell, count = size of the cell.
// No atomics, no scans, two barriers; O(n_s^2) LDS broadcast reads with n_s ~ a few hundred.
__global__ void __launch_bounds__(256) k_hard_slab_small(BinCfg cfg, HardBinWs w) {
    __shared__ __attribute__((aligned(16))) int2 pr[SMALL_PCAP + 8];
    const int s = blockIdx.x;
    const int p0 = w.gstart[s], np = w.gstart[s + 1] - p0;
    if (np == 0 || np > SMALL_PCAP) return;
    const int np8 = (np + 7) & ~7;
    for (int j = threadIdx.x; j < np8; j += 256)
        pr[j] = j < np ? make_int2(w.soff[p0 + j], w.sidx[p0 + j]) : make_int2(-1, 0x7fffffff);   // padding never matches
    __syncthreads();
    for (int j = threadIdx.x; j < np; j += 256) {
        const int2 me = pr[j];
        int r = 0, cnt = 0, f = 0x7fffffff;
        for (int k = 0; k < np8; k += 8) {               // 4 x 16-byte broadcast reads in flight per step
            int4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const int4 *>(&pr[k + 2 * u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool s0 = q[u].x == me.x, s1 = q[u].z == me.x;
                r += (s0 && q[u].y < me.y) + (s1 && q[u].w < me.y);
                cnt += s0 + s1;
                f = s0 ? min(f, q[u].y) : f;
                f = s1 ? min(f, q[u].w) : f;
            }
        }
        w.pfirst[p0 + j] = f;
        w.pslot[p0 + j] = r;
        if (me.y == f) {
            w.pcnt[p0 + j] = cnt;
            w.fbytes[f] = 1;
        }
    }
}

// ---- hard pass B, dense slabs (> SMALL_PCAP points): one workgroup per slab ----
// LDS: bitmap | word prefix | first[VCAP] cnt[VCAP] vstart[VCAP] fill[VCAP] | bucket[PCAP].  A slab that exceeds a cap
// uses the same code on global stand-in arrays (generic pointers), bounded but slow -- only pathological densities.
__global__ void __launch_bounds__(SLAB_NT) k_hard_slab(BinCfg cfg, int T, HardBinWs w) {
    extern __shared__ unsigned long long smem64[];
    __shared__ int wave_tot[SLAB_NT / 64];
    const int s = blockIdx.x;
    const int p0 = w.gstart[s], p1 = w.gstart[s + 1];
    const int np = p1 - p0;
    if (np <= SMALL_PCAP) return;                       // ordinary slabs were done by k_hard_slab_small
    const int nw = 1 << (cfg.logslab - 6);
    unsigned long long *bm = smem64;
    int32_t *wpre = reinterpret_cast<int32_t *>(smem64 + nw);
    int32_t *l_first = wpre + nw, *l_cnt = l_first + HARD_VCAP, *l_vstart = l_cnt + HARD_VCAP, *l_fill = l_vstart + HARD_VCAP;
    int32_t *l_bucket = l_fill + HARD_VCAP;
    const int tid = threadIdx.x;

    for (int x = tid; x < nw; x += SLAB_NT) bm[x] = 0ull;
    __syncthreads();
    for (int p = p0 + tid; p < p1; p += SLAB_NT) {
        const int off = w.soff[p];
        atomicOr(&bm[off >> 6], 1ull << (off & 63));
    }
    __syncthreads();
    const int per = (nw + SLAB_NT - 1) / SLAB_NT;
    int c = 0;
    for (int j = 0; j < per; ++j) {
        const int x = tid * per + j;
        if (x < nw) c += __popcll(bm[x]);
    }
    int nvox;
    int ex = block_excl_scan256(c, wave_tot, nvox);
    for (int j = 0; j < per; ++j) {
        const int x = tid * per + j;
        if (x < nw) { wpre[x] = ex; ex += __popcll(bm[x]); }
    }
    // per-voxel state: LDS when it fits, else this slab's range of the global stand-ins (generic pointers)
    const bool vfit = nvox <= HARD_VCAP, pfit = np <= HARD_PCAP;
    int32_t *first = vfit ? l_first : w.g_first + p0;
    int32_t *cnt = vfit ? l_cnt : w.g_cnt + p0;
    int32_t *vstart = vfit ? l_vstart : w.g_vstart + p0;
    int32_t *fill = vfit ? l_fill : w.g_fill + p0;
    int32_t *bucket = pfit ? l_bucket : w.g_bucket + p0;
    for (int v = tid; v < nvox; v += SLAB_NT) { first[v] = 0x7fffffff; cnt[v] = 0; fill[v] = 0; }
    const bool glob = !(vfit && pfit);       // global stand-ins need device-scope visibility between the phases
    if (glob) __threadfence();
    __syncthreads();
    // first index and count per voxel
    for (int p = p0 + tid; p < p1; p += SLAB_NT) {
        const int off = w.soff[p];
        const int lr = wpre[off >> 6] + popc_below(bm[off >> 6], off & 63);
        atomicMin(&first[lr], w.sidx[p]);
        atomicAdd(&cnt[lr], 1);
    }
    if (glob) __threadfence();
    __syncthreads();
    // bucket offsets: exclusive scan of the counts (chunks of SLAB_NT voxels)
    int running = 0;
    for (int base = 0; base < nvox; base += SLAB_NT) {
        const int v = base + tid;
        const int x = v < nvox ? cnt[v] : 0;
        int tot;
        const int e2 = block_excl_scan256(x, wave_tot, tot);
        if (v < nvox) vstart[v] = running + e2;
        running += tot;
    }
    if (glob) __threadfence();
    __syncthreads();
    for (int p = p0 + tid; p < p1; p += SLAB_NT) {
        const int off = w.soff[p];
        const int lr = wpre[off >> 6] + popc_below(bm[off >> 6], off & 63);
        bucket[vstart[lr] + atomicAdd(&fill[lr], 1)] = w.sidx[p];
    }
    if (glob) __threadfence();
    __syncthreads();
    // slot of every point = number of smaller indices in its voxel (stop at T unless it is the first point)
    for (int p = p0 + tid; p < p1; p += SLAB_NT) {
        const int off = w.soff[p], idx = w.sidx[p];
        const int lr = wpre[off >> 6] + popc_below(bm[off >> 6], off & 63);
        const int f = first[lr], n_in = cnt[lr];
        int r = 0;
        if (n_in > 1 && idx != f) {
            const int32_t *b = bucket + vstart[lr];
            for (int j = 0; j < n_in; ++j) {
                r += (b[j] < idx);
                if (r >= T) break;
            }
        }
        w.pfirst[p] = f;
        w.pslot[p] = r;
        if (idx == f) {
            w.pcnt[p] = n_in;
            w.fbytes[f] = 1;         // first-appearance flag by ORIGINAL index (plain byte store, no atomic)
        }
    }
}

// first-flag bytes -> 64-bit mask words + per-word counts
__global__ void __launch_bounds__(256) k_flags_to_words(const uint8_t *__restrict__ fbytes, int n, uint64_t *__restrict__ fmask,
                                                        int32_t *__restrict__ wcnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool f = i < n && fbytes[i];
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0 && i < n) { fmask[i >> 6] = m; wcnt[i >> 6] = __popcll(m); }
}

__device__ __forceinline__ int first_rank(const HardBinWs &w, int i) {
    return w.wpre[i >> 6] + popc_below(w.fmask[i >> 6], i & 63);
}

__global__ void k_hard_scene_offsets2(const int32_t *__restrict__ scene_off, int n, int n_scenes, int max_voxels, HardBinWs w,
                                      int32_t *__restrict__ scene_voxel_off) {
    const int nwords = (n + 63) / 64;
    for (int s = threadIdx.x; s <= n_scenes; s += blockDim.x) {
        const int i = scene_off[s];
        w.sfr[s] = (i >> 6) < nwords ? first_rank(w, i) : w.wpre[nwords];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int s = 0; s < n_scenes; ++s) {
            scene_voxel_off[s] = acc;
            const int tot = w.sfr[s + 1] - w.sfr[s];
            acc += tot < max_voxels ? tot : max_voxels;
        }
        scene_voxel_off[n_scenes] = acc;
    }
}

// ---- hard pass C: placement, one thread per sorted position (coalesced reads, 16-byte row stores) ----
__global__ void __launch_bounds__(256) k_hard_place(int n_sorted, BinCfg cfg, Geom g, int T, int max_voxels,
                                                    const int32_t *__restrict__ scene_off, HardBinWs w,
                                                    const int32_t *__restrict__ scene_voxel_off, float *__restrict__ voxels,
                                                    int32_t *__restrict__ coords_bzyx, int32_t *__restrict__ num_pts) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_sorted || p >= w.gstart[cfg.nslabs]) return;     // only the binned (in-range) points have sorted positions
    const int f = w.pfirst[p], r = w.pslot[p], idx = w.sidx[p];
    const int s = find_scene(scene_off, cfg.n_scenes, f);
    const int rank = first_rank(w, f) - w.sfr[s];
    if (rank >= max_voxels) return;                       // voxel never created (`continue` semantics)
    const int v = scene_voxel_off[s] + rank;
    float4 *row = reinterpret_cast<float4 *>(voxels) + (int64_t)v * T;
    if (r < T) row[r] = w.spts[p];
    if (idx == f) {
        const int cntv = w.pcnt[p];
        const int npv = cntv < T ? cntv : T;
        num_pts[v] = npv;
        for (int k = npv; k < T; ++k) row[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        // cell coordinates from the key: key = ((b*nx + cx)*ny + cy)*nz + cz
        // (slab id recovered from the sorted position via binary search over gstart)
        int lo = 0, hi = cfg.nslabs;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (w.gstart[mid] <= p) lo = mid; else hi = mid;
        }
        const int64_t key = ((int64_t)lo << cfg.logslab) + w.soff[p];
        const float rx = 1.0f / (float)g.grid[0], ry = 1.0f / (float)g.grid[1], rz = 1.0f / (float)g.grid[2];
        int t = (int)key;
        int q = fdiv(t, g.grid[2], rz); const int cz = t - q * g.grid[2]; t = q;
        q = fdiv(t, g.grid[1], ry); const int cy = t - q * g.grid[1]; t = q;
        q = fdiv(t, g.grid[0], rx); const int cx = t - q * g.grid[0];
        reinterpret_cast<int4 *>(coords_bzyx)[v] = make_int4(s, cz, cy, cx);
    }
}

}  // namespace vb

// -------------------------------------------------------------------------------------------------
// entry points used by voxel.hip's lvq_voxelize_dynamic (returns LVQ_EUNSUPPORTED when the binned path does not apply)
// -------------------------------------------------------------------------------------------------
size_t lvq_binned_dynamic_workspace_bytes(int64_t n) {
    vb::SizerAdapter a;
    vb::DynBinWs w;
    vb::dyn_layout(a, w, n);
    return a.s.total();
}

int lvq_binned_voxelize_dynamic(const float *pts, int64_t n, int c, int batch_size, const float *range_host,
                                const float *vsize_host, const int32_t *grid_host, int ndim, int32_t *unq_inv, int32_t *pt_coords,
                                int32_t *unq_key, int32_t *unq_cnt, int32_t *coords_bzyx, int32_t *counts, void *ws, size_t ws_bytes,
                                hipStream_t st) {
    using namespace vb;
    int64_t keyspace = (int64_t)batch_size * grid_host[0] * grid_host[1];
    if (ndim == 3) keyspace *= grid_host[2];
    BinCfg cfg;
    if (!choose_cfg(keyspace, n, cfg, 18)) return LVQ_EUNSUPPORTED;
    cfg.ndim = ndim; cfg.n_scenes = batch_size; cfg.mode = 0;
    LvqArena arena(ws, ws_bytes);
    DynBinWs w;
    dyn_layout(arena, w, n);
    if (!arena.ok) return LVQ_EWORKSPACE;
    Geom g;
    for (int j = 0; j < 3; ++j) { g.lo[j] = range_host[j]; g.vs[j] = vsize_host[j]; g.grid[j] = grid_host[j]; }
    const unsigned nb = (unsigned)lvq_cdiv(n, BIN_BLOCK * BIN_PPT);
    const int64_t cap = n < keyspace ? n : keyspace;
    hipMemsetAsync(w.ghist, 0, sizeof(int32_t) * 2 * (MAX_SLABS + 64), st);      // histogram + cursors (contiguous)
    hipLaunchKernelGGL(k_bin_hist, dim3(nb), dim3(BIN_BLOCK), sizeof(int32_t) * cfg.nslabs, st, pts, (int)n, c, g, cfg,
                       (const int32_t *)nullptr, w.ghist, pt_coords, unq_inv);
    hipLaunchKernelGGL(k_bin_scatter, dim3(nb), dim3(BIN_BLOCK), 3 * sizeof(int32_t) * cfg.nslabs, st, pts, (int)n, c, g, cfg,
                       (const int32_t *)nullptr, w.gstart, w.cursor, w.sidx, w.soff, (float4 *)nullptr, (const int32_t *)w.ghist,
                       &counts[1]);
    const int nw = 1 << (cfg.logslab - 6);
    hipLaunchKernelGGL(k_dyn_slab_count, dim3(cfg.nslabs), dim3(DYN_NT), sizeof(unsigned long long) * nw, st, cfg, w.gstart, w.soff,
                       w.slab_cnt);
    const size_t lds = sizeof(unsigned long long) * nw + sizeof(int32_t) * nw + sizeof(int32_t) * DYN_VCAP;
    static LvqLdsOnce once_dyn;
    if (lds > 48 * 1024 && !lvq_ensure_lds(once_dyn, {(const void *)k_dyn_slab_write}, 80 * 1024)) return LVQ_ELAUNCH;
    hipLaunchKernelGGL(k_dyn_slab_write, dim3(cfg.nslabs), dim3(DYN_NT), lds, st, cfg, g, w.gstart, w.sidx, w.soff, w.slab_cnt, unq_inv,
                       unq_key, unq_cnt, coords_bzyx, &counts[0]);
    return lvq_launch_status();
}

size_t lvq_binned_hard_workspace_bytes(int64_t n, int n_scenes) {
    vb::SizerAdapter a;
    vb::HardBinWs w;
    vb::hard_layout(a, w, n, n_scenes);
    return a.s.total();
}

// `continue` cap semantics, C == 4 payloads, key space <= 2^30; anything else -> LVQ_EUNSUPPORTED (legacy kernels)
int lvq_binned_voxelize_hard(const float *pts, const int32_t *scene_off, int64_t n, int n_scenes, int c, const float *range_host,
                             const float *vsize_host, const int32_t *grid_host, int max_pts, int max_voxels, float *voxels,
                             int32_t *coords_bzyx, int32_t *num_pts, int32_t *scene_voxel_off, void *ws, size_t ws_bytes,
                             hipStream_t st) {
    using namespace vb;
    if (c != 4 || (((uintptr_t)pts | (uintptr_t)voxels) & 15)) return LVQ_EUNSUPPORTED;
    const int64_t keyspace = (int64_t)n_scenes * grid_host[0] * grid_host[1] * grid_host[2];
    BinCfg cfg;
    if (!choose_cfg(keyspace, n, cfg)) return LVQ_EUNSUPPORTED;
    cfg.ndim = 3; cfg.n_scenes = n_scenes; cfg.mode = 1;
    LvqArena arena(ws, ws_bytes);
    HardBinWs w;
    hard_layout(arena, w, n, n_scenes);
    if (!arena.ok) return LVQ_EWORKSPACE;
    Geom g;
    for (int j = 0; j < 3; ++j) { g.lo[j] = range_host[j]; g.vs[j] = vsize_host[j]; g.grid[j] = grid_host[j]; }
    const unsigned nb = (unsigned)lvq_cdiv(n, BIN_BLOCK * BIN_PPT);
    const int64_t nwords = (n + 63) / 64;
    hipMemsetAsync(w.ghist, 0, sizeof(int32_t) * (cfg.nslabs + 1), st);
    hipMemsetAsync(w.fbytes, 0, (size_t)n + 64, st);
    hipLaunchKernelGGL(k_bin_hist, dim3(nb), dim3(BIN_BLOCK), sizeof(int32_t) * cfg.nslabs, st, pts, (int)n, c, g, cfg, scene_off, w.ghist,
                       (int32_t *)nullptr, (int32_t *)nullptr);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, st, w.ghist, cfg.nslabs, w.gstart, w.cursor, (int32_t *)nullptr);
    hipLaunchKernelGGL(k_bin_scatter, dim3(nb), dim3(BIN_BLOCK), 3 * sizeof(int32_t) * cfg.nslabs, st, pts, (int)n, c, g, cfg, scene_off,
                       w.gstart, w.cursor, w.sidx, w.soff, w.spts, (const int32_t *)nullptr, (int32_t *)nullptr);
    const int nw = 1 << (cfg.logslab - 6);
    const size_t lds = sizeof(unsigned long long) * nw + sizeof(int32_t) * nw + sizeof(int32_t) * (4 * HARD_VCAP + HARD_PCAP);
    static LvqLdsOnce once;
    if (!lvq_ensure_lds(once, {(const void *)k_hard_slab}, 160 * 1024 - 256)) return LVQ_ELAUNCH;
    hipLaunchKernelGGL(k_hard_slab_small, dim3(cfg.nslabs), dim3(256), 0, st, cfg, w);
    if (n > SMALL_PCAP) hipLaunchKernelGGL(k_hard_slab, dim3(cfg.nslabs), dim3(SLAB_NT), lds, st, cfg, max_pts, w);
    hipLaunchKernelGGL(k_flags_to_words, dim3((unsigned)lvq_cdiv(nwords * 64, 256)), dim3(256), 0, st, w.fbytes, (int)(nwords * 64), w.fmask,
                       w.wcnt);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, st, w.wcnt, (int)nwords, w.wpre, (int32_t *)nullptr, (int32_t *)nullptr);
    hipLaunchKernelGGL(k_hard_scene_offsets2, dim3(1), dim3(256), 0, st, scene_off, (int)n, n_scenes, max_voxels, w, scene_voxel_off);
    // the number of binned (valid) points is only known on the device (gstart[nslabs]); launch over n and let extra threads exit
    hipLaunchKernelGGL(k_hard_place, dim3((unsigned)lvq_cdiv(n, 256)), dim3(256), 0, st, (int)n, cfg, g, max_pts, max_voxels, scene_off, w,
                       scene_voxel_off, voxels, coords_bzyx, num_pts);
    return lvq_launch_status();
}
