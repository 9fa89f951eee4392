// csrc/voxel_dyn.hip -- dynamic voxelisation (dynamic_mean_vfe.py:53-72, dynamic_pillar_vfe.py:93-104, dynamic_voxel_vfe.py:60-72:
// key = b*S + cx*Sx + cy*Sy + cz, `torch.unique(key, return_inverse, return_counts)` = ascending keys), second design (gfx950).
//
// The first slab-binned version (voxel_binned.hip) needed memset + 4 kernels (histogram, scatter, slab count, slab write:
// 88 us for 8 x 65 536 points = 3.4 % of the HBM roofline): every dependent stream operation costs 2-4 us on top of its
// work, and each slab built its occupancy bitmap twice (once to learn how many voxels it holds so that the slab bases could be
// scanned, once to write).  This file keeps what the sorted output needs -- KEY-CONTIGUOUS slabs, ranked by a dense LDS bitmap
// -- and removes the passes:
//   k_dbin   ONE binning pass: every slab owns a fixed region of DCAP entries; a block counts its points per slab in LDS (the LDS
//            atomic's return value is the rank inside the block's run), reserves the runs with one global atomic per (block,
//            slab) and stores (original index, offset inside the slab).  Entries that do not fit go to a shared overflow list
//            (only clouds with thousands of points in one 2^logslab-key range get there; their slabs re-scan that list).
//   k_dslab  one workgroup per slab, in ticket order: occupancy bitmap in LDS -> popcount ranks -> the slab's voxel count is
//            PUBLISHED and the base (= number of voxels in all earlier slabs = the global rank offset, which is what makes
//            the output torch.unique-ordered) comes from a decoupled LOOK-BACK over the earlier slabs' published counts, in the
//            same kernel -> keys, counts, coords and the inverse map leave in one pass.
// memset (cursors, tickets, look-back words: one contiguous block) + 2 kernels.
#include "common.h"

namespace vd {

struct Geom {
    float lo[3];
    float vs[3];
    int grid[3];
};

constexpr int MAX_SLABS = 8192;
constexpr int DCAP = 4096;              // entries of a slab's fixed region (mean fill <= ~800; Dist-C's densest 1.3 m strips hold ~1600)
constexpr int BIN_NT = 1024, BIN_PPT = 4;
constexpr int SLAB_NT = 512;
constexpr int VCAP = 2048;              // voxels of a slab counted in LDS (more: global atomics on unq_cnt)
constexpr uint32_t ST_AGG = 1u << 30, ST_PRE = 2u << 30, ST_MASK = 3u << 30, VAL_MASK = ~ST_MASK;

struct Ws {
    int32_t *cursor;                    // [MAX_SLABS] points per slab          | one memset covers cursor .. lookback
    int32_t *misc;                      // [64] 0: overflow count, 1: ticket, 2: valid points
    uint32_t *lookback;                 // [MAX_SLABS] status | value
    int2 *region;                       // [nslabs * DCAP] (original index, offset in slab)
    int2 *ovf;                          // [n] (original index, key)
};

template <typename A> void layout(A &a, Ws &w, int64_t n, int nslabs) {
    w.cursor = a.template take<int32_t>(2 * MAX_SLABS + 64);
    w.misc = w.cursor ? w.cursor + MAX_SLABS : nullptr;
    w.lookback = w.cursor ? reinterpret_cast<uint32_t *>(w.cursor + MAX_SLABS + 64) : nullptr;
    w.region = a.template take<int2>((int64_t)nslabs * DCAP + 1);
    w.ovf = a.template take<int2>(n + 1);
}
struct SizerAdapter {
    LvqSizer s;
    template <typename T> T *take(size_t n) { s.template take<T>(n); return nullptr; }
};

// slab geometry: 2^logslab keys per slab, bitmap <= 64 KiB of LDS (logslab <= 19), ~800 points per slab on average: the binning
// pass pays one global atomic per (block, slab), so fewer, larger slabs (first version: 1280 slabs of ~410 points, 30 us of binning)
static bool choose(int64_t keyspace, int64_t n, int &logslab, int &nslabs) {
    if (keyspace <= 0 || keyspace >= (1ll << 31)) return false;
    int64_t want = n / 800 + 1;
    if (want > MAX_SLABS) want = MAX_SLABS;
    int ls = 10;
    while (ls < 19 && ((keyspace + (1ll << ls) - 1) >> ls) > want) ++ls;
    if (((keyspace + (1ll << ls) - 1) >> ls) > MAX_SLABS) return false;
    logslab = ls;
    nslabs = (int)((keyspace + (1ll << ls) - 1) >> ls);
    return true;
}

__device__ __forceinline__ int popc_below(unsigned long long m, int bit) {
    return __popcll(m & (bit == 0 ? 0ull : (~0ull >> (64 - bit))));
}

// ---- K1: keys + single-pass binning ----
__global__ void __launch_bounds__(BIN_NT) k_dbin(const float *__restrict__ pts, int n, int c, Geom g, int ndim, int batch, int logslab, int nslabs,
                                                 Ws w, int32_t *__restrict__ pt_coords, int32_t *__restrict__ unq_inv) {
    extern __shared__ int32_t lds[];
    int32_t *lh = lds, *lb = lds + nslabs;
    __shared__ int nvalid_blk;
    for (int b = threadIdx.x; b < nslabs; b += BIN_NT) lh[b] = 0;
    if (threadIdx.x == 0) nvalid_blk = 0;
    __syncthreads();
    const int base = blockIdx.x * (BIN_NT * BIN_PPT);
    int64_t key[BIN_PPT];
    int rk[BIN_PPT];
    int nv = 0;
#pragma unroll
    for (int u = 0; u < BIN_PPT; ++u) {
        const int i = base + u * BIN_NT + threadIdx.x;
        key[u] = -1;
        rk[u] = 0;
        if (i < n) {
            const float *p = pts + (int64_t)i * c;
            int cc[3] = {0, 0, 0};
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < ndim) {
                    const float d = p[1 + j] - g.lo[j];
                    const float q = d / g.vs[j];
                    const float f = floorf(q);
                    const bool in = (f >= 0.0f) && (f < (float)g.grid[j]);
                    ok = ok && in;
                    cc[j] = in ? (int)f : -1;
                }
            }
            const int b = (int)p[0];
            ok = ok && b >= 0 && b < batch;
            if (pt_coords) { pt_coords[i * 3] = cc[0]; pt_coords[i * 3 + 1] = cc[1]; pt_coords[i * 3 + 2] = cc[2]; }
            if (ok) {
                int64_t k = ((int64_t)b * g.grid[0] + cc[0]) * g.grid[1] + cc[1];
                if (ndim == 3) k = k * g.grid[2] + cc[2];
                key[u] = k;
                rk[u] = atomicAdd(&lh[(int)(k >> logslab)], 1);
                ++nv;
            } else {
                unq_inv[i] = -1;
            }
        }
    }
    if (nv) atomicAdd(&nvalid_blk, nv);
    __syncthreads();
    for (int b = threadIdx.x; b < nslabs; b += BIN_NT) {
        const int h = lh[b];
        if (h) lb[b] = atomicAdd(&w.cursor[b], h);
    }
    if (threadIdx.x == 0 && nvalid_blk) atomicAdd(&w.misc[2], nvalid_blk);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BIN_PPT; ++u) {
        const int i = base + u * BIN_NT + threadIdx.x;
        if (key[u] >= 0) {
            const int s = (int)(key[u] >> logslab);
            const int pos = lb[s] + rk[u];
            if (pos < DCAP) w.region[(int64_t)s * DCAP + pos] = make_int2(i, (int)(key[u] & ((1ll << logslab) - 1)));
            else w.ovf[atomicAdd(&w.misc[0], 1)] = make_int2(i, (int)key[u]);
        }
    }
}

// floor(a / d) for 0 <= a < 2^31, d >= 1 (fp32 estimate + integer correction loops)
__device__ __forceinline__ int fdiv(int a, int d, float rd) {
    int q = (int)((float)a * rd);
    int r = a - q * d;
    while (r < 0) { --q; r += d; }
    while (r >= d) { ++q; r -= d; }
    return q;
}

// ---- K2: one workgroup per slab (ticket order): bitmap -> ranks -> look-back base -> outputs ----
__global__ void __launch_bounds__(SLAB_NT) k_dslab(Geom g, int ndim, int logslab, int nslabs, Ws w, int32_t *__restrict__ unq_inv,
                                                   int32_t *__restrict__ unq_key, int32_t *__restrict__ unq_cnt, int32_t *__restrict__ coords_bzyx,
                                                   int32_t *__restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int wave_tot[SLAB_NT / 64];
    __shared__ int l_slab, l_base;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) l_slab = atomicAdd(&w.misc[1], 1);          // ticket: slabs START in order, so every predecessor is running or done
    __syncthreads();
    const int s = l_slab;
    const int nw = 1 << (logslab - 6);
    unsigned long long *bm = reinterpret_cast<unsigned long long *>(smem);          // [nw] occupancy words
    int32_t *wpre = reinterpret_cast<int32_t *>(bm + nw);                           // [nw] voxel rank at the start of each word
    int32_t *lcnt = wpre + nw;                                                      // [VCAP] points per voxel
    const int np_all = w.cursor[s];
    const int np = np_all < DCAP ? np_all : DCAP;
    const int novf = np_all > DCAP ? w.misc[0] : 0;                                 // this slab has entries in the overflow list
    const int2 *reg = w.region + (int64_t)s * DCAP;
    for (int x = tid; x < nw; x += SLAB_NT) bm[x] = 0ull;
    __syncthreads();
    for (int p = tid; p < np; p += SLAB_NT) {
        const int off = reg[p].y;
        atomicOr(&bm[off >> 6], 1ull << (off & 63));
    }
    for (int o = tid; o < novf; o += SLAB_NT) {
        const int2 e = w.ovf[o];
        if ((e.y >> logslab) == s) { const int off = e.y & ((1 << logslab) - 1); atomicOr(&bm[off >> 6], 1ull << (off & 63)); }
    }
    __syncthreads();
    // voxel ranks inside the slab: exclusive popcount scan over the words
    const int per = (nw + SLAB_NT - 1) / SLAB_NT;
    int c = 0;
    for (int j = 0; j < per; ++j) {
        const int x = tid * per + j;
        if (x < nw) c += __popcll(bm[x]);
    }
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int wb = 0, nvox = 0;
    for (int q = 0; q < SLAB_NT / 64; ++q) {
        const int t = wave_tot[q];
        if (q < wid) wb += t;
        nvox += t;
    }
    int ex = wb + incl - c;
    for (int j = 0; j < per; ++j) {
        const int x = tid * per + j;
        if (x < nw) { wpre[x] = ex; ex += __popcll(bm[x]); }
    }
    // ---- publish this slab's voxel count, look back for the base (decoupled look-back: the words carry their own payload, so
    // relaxed agent-scope atomics are all the protocol needs) ----
    // The first version walked back one slab at a time from one lane: with every workgroup in flight nobody has a prefix yet, so
    // slab s read s words one dependent L2 round trip after the other (104 us for 1280 slabs).  Here wave 0 reads 64 predecessors
    // per step and stops at the nearest published prefix.
    if (wid == 0) {
        if (lane == 0) __hip_atomic_store(&w.lookback[s], ST_AGG | (uint32_t)nvox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int base = 0;
        for (int q0 = s - 1; q0 >= 0;) {
            const int q = q0 - lane;
            const uint32_t v = q >= 0 ? __hip_atomic_load(&w.lookback[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_PRE;   // before slab 0: prefix 0
            const unsigned long long ready = __ballot((v & ST_MASK) != 0u), pre = __ballot((v & ST_MASK) == ST_PRE);
            const int p = pre ? __builtin_ctzll(pre) : 64;                      // nearest predecessor that already holds a prefix
            const unsigned long long need = p >= 63 ? ~0ull : ((1ull << (p + 1)) - 1ull);
            if ((ready & need) != need) { __builtin_amdgcn_s_sleep(1); continue; }   // somebody in the window has not published yet
            int part = lane <= p ? (int)(v & VAL_MASK) : 0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
            base += part;
            if (p < 64) break;
            q0 -= 64;
        }
        if (lane == 0) {
            __hip_atomic_store(&w.lookback[s], ST_PRE | (uint32_t)(base + nvox), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            l_base = base;
            if (s == nslabs - 1) { counts[0] = base + nvox; counts[1] = w.misc[2]; }
        }
    }
    const bool vfit = nvox <= VCAP;
    if (vfit) for (int v = tid; v < nvox; v += SLAB_NT) lcnt[v] = 0;
    __syncthreads();
    const int base = l_base;
    if (!vfit) {                                                // rare: counts of this slab go through global atomics
        for (int v = tid; v < nvox; v += SLAB_NT) unq_cnt[base + v] = 0;
        __threadfence();
        __syncthreads();
    }
    // ---- per point: inverse index, per-voxel count ----
    for (int p = tid; p < np; p += SLAB_NT) {
        const int2 e = reg[p];
        const int lr = wpre[e.y >> 6] + popc_below(bm[e.y >> 6], e.y & 63);
        unq_inv[e.x] = base + lr;
        if (vfit) atomicAdd(&lcnt[lr], 1); else atomicAdd(&unq_cnt[base + lr], 1);
    }
    for (int o = tid; o < novf; o += SLAB_NT) {
        const int2 e = w.ovf[o];
        if ((e.y >> logslab) == s) {
            const int off = e.y & ((1 << logslab) - 1);
            const int lr = wpre[off >> 6] + popc_below(bm[off >> 6], off & 63);
            unq_inv[e.x] = base + lr;
            if (vfit) atomicAdd(&lcnt[lr], 1); else atomicAdd(&unq_cnt[base + lr], 1);
        }
    }
    __syncthreads();
    // ---- per voxel: key, count, (b, z, y, x) -- sequential stores, ascending key ----
    const float rx = 1.0f / (float)g.grid[0], ry = 1.0f / (float)g.grid[1], rz = 1.0f / (float)g.grid[2];
    for (int x = tid; x < nw; x += SLAB_NT) {
        unsigned long long m = bm[x];
        int r = wpre[x];
        while (m) {
            const int bit = __builtin_ctzll(m);
            m &= m - 1;
            const int key = (s << logslab) + (x << 6) + bit;
            unq_key[base + r] = key;
            if (vfit) unq_cnt[base + r] = lcnt[r];
            int t = key, cz = 0;
            if (ndim == 3) { const int q = fdiv(t, g.grid[2], rz); cz = t - q * g.grid[2]; t = q; }
            int q = fdiv(t, g.grid[1], ry);
            const int cy = t - q * g.grid[1];
            t = q;
            q = fdiv(t, g.grid[0], rx);
            const int cx = t - q * g.grid[0];
            reinterpret_cast<int4 *>(coords_bzyx)[base + r] = make_int4(q, cz, cy, cx);
            ++r;
        }
    }
}

}  // namespace vd

size_t lvq_dyn2_workspace_bytes(int64_t n, int64_t keyspace) {
    int ls, ns;
    if (!vd::choose(keyspace, n, ls, ns)) return 0;
    vd::SizerAdapter a;
    vd::Ws w;
    vd::layout(a, w, n, ns);
    return a.s.total();
}

int lvq_dyn2_voxelize(const float *pts, int64_t n, int c, int batch_size, const float *range_host, const float *vsize_host,
                      const int32_t *grid_host, int ndim, int32_t *unq_inv, int32_t *pt_coords, int32_t *unq_key, int32_t *unq_cnt,
                      int32_t *coords_bzyx, int32_t *counts, void *ws, size_t ws_bytes, hipStream_t st) {
    using namespace vd;
    int64_t keyspace = (int64_t)batch_size * grid_host[0] * grid_host[1];
    if (ndim == 3) keyspace *= grid_host[2];
    int logslab, nslabs;
    if (!choose(keyspace, n, logslab, nslabs) || n >= (1ll << 30) || (((uintptr_t)coords_bzyx) & 15)) return LVQ_EUNSUPPORTED;
    LvqArena arena(ws, ws_bytes);
    Ws w;
    layout(arena, w, n, nslabs);
    if (!arena.ok) return LVQ_EWORKSPACE;
    Geom g;
    for (int j = 0; j < 3; ++j) { g.lo[j] = range_host[j]; g.vs[j] = vsize_host[j]; g.grid[j] = grid_host[j]; }
    hipMemsetAsync(w.cursor, 0, sizeof(int32_t) * (2 * MAX_SLABS + 64), st);
    const unsigned nb = (unsigned)lvq_cdiv(n, BIN_NT * BIN_PPT);
    hipLaunchKernelGGL(k_dbin, dim3(nb), dim3(BIN_NT), 2 * sizeof(int32_t) * nslabs, st, pts, (int)n, c, g, ndim, batch_size, logslab, nslabs, w,
                       pt_coords, unq_inv);
    const int nw = 1 << (logslab - 6);
    const size_t lds = (sizeof(unsigned long long) + sizeof(int32_t)) * nw + sizeof(int32_t) * VCAP;
    static LvqLdsOnce once;
    if (lds > 64 * 1024 && !lvq_ensure_lds(once, {(const void *)k_dslab}, 112 * 1024)) return LVQ_ELAUNCH;
    hipLaunchKernelGGL(k_dslab, dim3(nslabs), dim3(SLAB_NT), lds, st, g, ndim, logslab, nslabs, w, unq_inv, unq_key, unq_cnt, coords_bzyx, counts);
    return lvq_launch_status();
}
