// csrc/voxel_hashed.hip -- hard voxeliser, third design: hash-balanced slabs + input-order placement (gfx950).
//
// Reference semantics (data_processor.py:16-61,133-180 -> spconv Point2VoxelCPU3d): voxel id = first
// appearance in input order, first T points of a cell in input order, `continue` cap.  Order-independent formulation:
//   voxel id(cell)  = rank of the cell's FIRST point among all first points   (prefix popcount over 1 flag per point)
//   slot(point)     = number of points of the same cell with a smaller index
//
// The slab-binned version (voxel_binned.hip, 11 stream operations, 128 us for 8 x 65 536 points) lost its time in
//   (1) ~2500 key-contiguous slabs -> one global atomic per (block, slab) pair in the histogram and in the scatter
//       (260 k memory-side atomics each), plus hot slabs around the sensor origin that need a second, dense kernel;
//   (2) placement in slab order: 16-byte stores scattered over first-appearance ranks + a 13-step binary search;
//   (3) launches: three memsets and three single-workgroup scans.
// This file keeps the idea (sort points into LDS-sized groups that hold whole cells, rank inside LDS) and changes:
//   * slab = top bits of a BIJECTIVE 32-bit mix of the key: every cell still lands in exactly one slab, but slabs are
//     statistically balanced (no hot slabs), a power-of-two count of ~1000-point slabs is enough (4x fewer global
//     atomics), and ONE slab kernel serves every slab: an LDS open-addressing table (CAS insert, atomicMin first
//     index, count), the cells' points bucketed contiguously by a scan over the table -> O(points) LDS work instead of
//     the O(n_s^2) compare loop (a per-cell linked list was tried first: dependent LDS reads, 40 us on the pillar grid).
//   * balanced slabs need no histogram pass: every slab owns a fixed 2048-entry region (mean fill <= 1024) that is
//     filled in ONE binning pass; the rare entries that do not fit go to a shared overflow list and their slab is ranked
//     on global arrays (it would not fit LDS anyway).
//   * flags -> words keeps only block-local prefixes (k_words); the scan over the per-block totals, the first-rank at
//     the scene starts and the capped scene offsets are recomputed in the prologue of every placement block
//     (a few hundred L2-resident values) -- no single-workgroup scan kernels are left.
//   * placement runs in INPUT order: consecutive first points own consecutive output rows (that is what first-appearance
//     order means), so a wave stages its first points in LDS and streams rows*T*16 contiguous bytes with full 1 KB
//     wave stores (zero padding included); only non-first points (multi-point cells) issue scattered 16-byte stores.
//     coords / num_points leave as dense stores, cell coordinates are recomputed from the point (no key decode).
//   4 kernels + 1 memset (k_bin, k_slab, k_words, k_place): 54 us for 8 x 65 536 points (k_place 22 us = the 93 MB output
//   stream at ~4.5 TB/s; every stream operation costs ~4.5 us of dependent-launch latency on top of its work, which is
//   why the count of operations was the first thing to cut).
//   Unsupported shapes (C != 4, T > 127, key space >= 2^31, N > 4 M, > 1024 scenes, `break` cap) -> LVQ_EUNSUPPORTED
//   and the caller falls back to voxel_binned.hip / the hash kernels in voxel.hip.
#include "common.h"

namespace vh {

struct Geom {
    float lo[3];
    float vs[3];
    int grid[3];
};

constexpr int MAX_SLABS = 4096;
constexpr int BIN_NT = 1024;          // threads of a binning block
constexpr int BIN_PPT = 4;            // points per thread (1, 2 and 4 time the same: the pass is latency-, not issue-bound)
constexpr int SLAB_NT = 512;          // threads of a slab workgroup
constexpr int SLAB_CAP = 2048;        // points of a slab ranked in LDS (larger slabs: same code on global arrays)
constexpr int SLAB_TS = 2 * SLAB_CAP; // table slots in LDS

struct Ws {
    int32_t *cursor;                  // [MAX_SLABS] points per slab, then ovf_count, galloc: one memset
    int32_t *ovf_count, *galloc;
    int2 *sent;                       // [nslabs * SLAB_CAP] (original index, mixed key): slab s owns [s * SLAB_CAP, (s+1) * SLAB_CAP)
    int2 *ovf;                        // [n] entries that did not fit their slab's region
    uint8_t *fb;                      // [n+64] by ORIGINAL index: min(count, T) for the first point of a cell, else 0
    int2 *rec;                        // [n]    by ORIGINAL index, non-first points: (first index of the cell, slot)
    float4 *tmean;                    // [n]    voxelise->mean form: by ORIGINAL first index, mean of a multi-point cell
    uint64_t *fmask;                  // [nwords+1] first-point flags, 64 points per word
    int32_t *wloc;                    // [nwords+2] exclusive popcount prefix of a word inside its 4096-point block
    int32_t *btot;                    // [MAX_KB+2] first points per 4096-point block
    int32_t *g_idx;                   // [n]      oversize slabs: contiguous point lists + global stand-ins for the LDS arrays
    uint32_t *g_mix;
    int32_t *g_bucket, *g_sorted;     // [n]
    uint32_t *g_key;                  // [2n+64]
    int32_t *g_cnt, *g_first, *g_start;
};

// power-of-two slab count with a mean fill in (512, 1024] points (regions hold SLAB_CAP = 2048)
static int slab_count(int64_t n) {
    int lg = 0;
    while (((int64_t)1024 << lg) < n) ++lg;
    return 1 << lg;
}

template <typename A> void layout(A &a, Ws &w, int64_t n, int n_scenes) {
    const int64_t nwords = (n + 63) / 64;
    const int64_t region = (int64_t)slab_count(n) * SLAB_CAP;
    w.cursor = a.template take<int32_t>(MAX_SLABS + 64);
    w.ovf_count = w.cursor ? w.cursor + MAX_SLABS : nullptr;
    w.galloc = w.cursor ? w.cursor + MAX_SLABS + 1 : nullptr;
    w.sent = a.template take<int2>(region + 1);
    w.ovf = a.template take<int2>(n + 1);
    w.fb = a.template take<uint8_t>(n + 64);
    w.rec = a.template take<int2>(n + 1);
    w.tmean = a.template take<float4>(n + 1);
    w.fmask = a.template take<uint64_t>(nwords + 1);
    w.wloc = a.template take<int32_t>(nwords + 2);
    w.btot = a.template take<int32_t>(4096);
    w.g_idx = a.template take<int32_t>(n + 1);
    w.g_mix = a.template take<uint32_t>(n + 1);
    w.g_bucket = a.template take<int32_t>(n + 1);
    w.g_sorted = a.template take<int32_t>(n + 1);
    w.g_key = a.template take<uint32_t>(2 * n + 64);
    w.g_cnt = a.template take<int32_t>(2 * n + 64);
    w.g_first = a.template take<int32_t>(2 * n + 64);
    w.g_start = a.template take<int32_t>(2 * n + 64);
}

struct SizerAdapter {
    LvqSizer s;
    template <typename T> T *take(size_t n) { s.template take<T>(n); return nullptr; }
};

__device__ __forceinline__ int find_scene(const int32_t *off, int n_scenes, int i) {
    int lo = 0, hi = n_scenes;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

// cell coordinates (cx, cy, cz) of a point: floor((p - lo) / vs) in IEEE fp32, in-range test as the reference does it
__device__ __forceinline__ bool cell_of(float x, float y, float z, const Geom &g, int cc[3]) {
    const float p[3] = {x, y, z};
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float d = p[j] - g.lo[j];
        const float q = d / g.vs[j];
        const float f = floorf(q);
        const bool in = (f >= 0.0f) && (f < (float)g.grid[j]);
        ok = ok && in;
        cc[j] = in ? (int)f : -1;
    }
    return ok;
}

// bijective 32-bit mix (odd multiplies and xor-shifts); mix(0) == 0, and keys enter as key + 1, so a mixed key is never 0
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// mixed key of point i, 0 for a point outside the grid
__device__ __forceinline__ uint32_t mixed_key(const float4 p, int scene, const Geom &g) {
    int cc[3];
    if (!cell_of(p.x, p.y, p.z, g, cc)) return 0u;
    const uint32_t key = (uint32_t)(((scene * g.grid[0] + cc[0]) * g.grid[1] + cc[1]) * g.grid[2] + cc[2]);   // < 2^31 (host check)
    return mix32(key + 1u);
}

__device__ __forceinline__ int block_excl_scan(int v, int *wave_tot, int nwaves, int &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int wbase = 0, tot = 0;
    for (int w = 0; w < nwaves; ++w) {
        const int t = wave_tot[w];
        if (w < wid) wbase += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return wbase + incl - v;
}

// ---- K1: single-pass binning.  Hash-balanced slabs need no histogram pass: every slab owns a fixed region of SLAB_CAP
// entries (mean fill <= 1024); a block counts its points per slab in LDS (the LDS atomic's return value is the point's
// rank inside the block's run), reserves the runs with one global atomic per (block, slab) and writes (index, mixed key).
// Entries that do not fit their region go to one shared overflow list -- their slab cannot be ranked in LDS anyway.
// Points outside the grid get their flag byte cleared here.
__global__ void __launch_bounds__(BIN_NT) k_bin(const float4 *__restrict__ pts, int n, Geom g, int n_scenes, int shift, int nslabs,
                                                const int32_t *__restrict__ scene_off, Ws w) {
    extern __shared__ int32_t lds[];
    int32_t *lh = lds, *lb = lds + nslabs;
    for (int b = threadIdx.x; b < nslabs; b += BIN_NT) lh[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * (BIN_NT * BIN_PPT);
    uint32_t mk[BIN_PPT];
    int rk[BIN_PPT];
#pragma unroll
    for (int u = 0; u < BIN_PPT; ++u) {
        const int i = base + u * BIN_NT + threadIdx.x;
        mk[u] = 0u;
        rk[u] = 0;
        if (i < n) {
            mk[u] = mixed_key(pts[i], find_scene(scene_off, n_scenes, i), g);
            if (mk[u]) rk[u] = atomicAdd(&lh[shift >= 32 ? 0 : (int)(mk[u] >> shift)], 1);
            else w.fb[i] = 0;
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nslabs; b += BIN_NT) {
        const int h = lh[b];
        if (h) lb[b] = atomicAdd(&w.cursor[b], h);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BIN_PPT; ++u) {
        const int i = base + u * BIN_NT + threadIdx.x;
        if (mk[u]) {
            const int s = shift >= 32 ? 0 : (int)(mk[u] >> shift);
            const int pos = lb[s] + rk[u];
            if (pos < SLAB_CAP) w.sent[(int64_t)s * SLAB_CAP + pos] = make_int2(i, (int)mk[u]);     // one 8-byte store
            else w.ovf[atomicAdd(w.ovf_count, 1)] = make_int2(i, (int)mk[u]);
        }
    }
}

// ---- K2: one workgroup per slab: open-addressing table of the slab's cells -> first index / count per cell, then the
// cell's points are bucketed contiguously (exclusive scan of the counts over the table slots) so that the slot of a
// point = number of smaller indices in its bucket is a run of independent reads (no pointer chasing).
// slot of point idx = number of smaller indices in its cell's bucket b[0..cnt) (stops at T: the point is dropped then)
__device__ __forceinline__ int bucket_rank(const int32_t *b, int cnt, int idx, int T) {
    int r = 0;
    for (int k = 0; k < cnt && r < T; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k + u < cnt) r += (b[k + u] < idx);
    }
    return r;
}

// mean of the first min(cnt, T) points of a cell in slot order: MeanVFE.forward (mean_vfe.py:25-29) on the row the padded
// tensor would hold -- the same fp32 additions in the same order as lvq_mean_vfe (zeros of the padding add nothing)
__device__ __forceinline__ float4 cell_mean(const float4 *__restrict__ pts, const int32_t *sorted, int npv) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < npv; ++k) {
        const float4 p = pts[sorted[k]];
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    const float d = (float)npv;
    return make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
}

// LDS form (np <= SLAB_CAP): a thread keeps its <= 4 points (index, key, table slot) in registers across the phases
template <bool MEAN>
__device__ __forceinline__ void slab_rank_lds(const float4 *__restrict__ pts, const int2 *__restrict__ sent, int np, int T,
                                              const Ws &w, int32_t *bucket, uint32_t *t_key, int32_t *t_cnt, int32_t *t_first,
                                              int32_t *t_start, int ts, int *wave_tot) {
    constexpr int PPT = SLAB_CAP / SLAB_NT;
    const int tid = threadIdx.x;
    const uint32_t tmask = (uint32_t)ts - 1u;
    int idx[PPT];
    uint32_t m[PPT], h[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const int j = tid + u * SLAB_NT;
        const int2 e = j < np ? sent[j] : make_int2(0, 0);
        idx[u] = e.x;
        m[u] = (uint32_t)e.y;
    }
    for (int x = tid; x < ts; x += SLAB_NT) { t_key[x] = 0u; t_cnt[x] = 0; t_first[x] = 0x7fffffff; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        if (m[u]) {
            uint32_t hh = m[u] & tmask;
            while (true) {
                const uint32_t prev = atomicCAS(&t_key[hh], 0u, m[u]);
                if (prev == 0u || prev == m[u]) break;
                hh = (hh + 1u) & tmask;
            }
            h[u] = hh;
            atomicMin(&t_first[hh], idx[u]);
            atomicAdd(&t_cnt[hh], 1);
        }
    }
    __syncthreads();
    const int per = (ts + SLAB_NT - 1) / SLAB_NT;
    int c = 0;
    for (int k = 0; k < per; ++k) {
        const int x = tid * per + k;
        if (x < ts) c += t_cnt[x];
    }
    int tot;
    int ex = block_excl_scan(c, wave_tot, SLAB_NT / 64, tot);
    for (int k = 0; k < per; ++k) {
        const int x = tid * per + k;
        if (x < ts) { t_start[x] = ex; ex += t_cnt[x]; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PPT; ++u)
        if (m[u] && t_cnt[h[u]] > 1) bucket[atomicAdd(&t_start[h[u]], 1)] = idx[u];    // t_start doubles as the cursor
    __syncthreads();
    int r[PPT], cn[PPT], bs[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        r[u] = 0; cn[u] = 0; bs[u] = 0;
        if (m[u]) {
            cn[u] = t_cnt[h[u]];
            bs[u] = t_start[h[u]] - cn[u];
            const int f = t_first[h[u]];
            if (idx[u] == f) {
                w.fb[idx[u]] = (uint8_t)(cn[u] < T ? cn[u] : T);
            } else {
                r[u] = bucket_rank(bucket + bs[u], cn[u], idx[u], T);
                w.fb[idx[u]] = 0;
                if (!MEAN) w.rec[idx[u]] = make_int2(f, r[u]);      // r >= T: the point is not stored (only "r < T" is used)
            }
        }
    }
    if (MEAN) {
        // the bucket becomes the cell's first min(cnt, T) indices in slot order (every slot below that is claimed by exactly
        // one point), then the first point of every multi-point cell sums them
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PPT; ++u)
            if (m[u] && cn[u] > 1 && r[u] < T) bucket[bs[u] + r[u]] = idx[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PPT; ++u)
            if (m[u] && cn[u] > 1 && idx[u] == t_first[h[u]]) w.tmean[idx[u]] = cell_mean(pts, bucket + bs[u], cn[u] < T ? cn[u] : T);
    }
}

// global form: the same phases on global arrays (a slab with more than SLAB_CAP points: only inputs with ~thousands of
// points in single cells get there; bounded, slow, exact)
template <bool MEAN>
__device__ __forceinline__ void slab_rank_glob(const float4 *__restrict__ pts, int32_t *sorted, const int32_t *sidx, const uint32_t *smix, int np, int T, const Ws &w, int32_t *bucket,
                                               uint32_t *t_key, int32_t *t_cnt, int32_t *t_first, int32_t *t_start, int ts,
                                               int *wave_tot) {
    const int tid = threadIdx.x;
    const uint32_t tmask = (uint32_t)ts - 1u;
    for (int x = tid; x < ts; x += SLAB_NT) { t_key[x] = 0u; t_cnt[x] = 0; t_first[x] = 0x7fffffff; }
    __threadfence();
    __syncthreads();
    for (int j = tid; j < np; j += SLAB_NT) {
        const int idx = sidx[j];
        const uint32_t m = smix[j];
        uint32_t h = m & tmask;
        while (true) {
            const uint32_t prev = atomicCAS(&t_key[h], 0u, m);
            if (prev == 0u || prev == m) break;
            h = (h + 1u) & tmask;
        }
        atomicMin(&t_first[h], idx);
        atomicAdd(&t_cnt[h], 1);
    }
    __threadfence();
    __syncthreads();
    const int per = (ts + SLAB_NT - 1) / SLAB_NT;
    int c = 0;
    for (int k = 0; k < per; ++k) {
        const int x = tid * per + k;
        if (x < ts) c += t_cnt[x];
    }
    int tot;
    int ex = block_excl_scan(c, wave_tot, SLAB_NT / 64, tot);
    for (int k = 0; k < per; ++k) {
        const int x = tid * per + k;
        if (x < ts) { t_start[x] = ex; ex += t_cnt[x]; }
    }
    __threadfence();
    __syncthreads();
    for (int j = tid; j < np; j += SLAB_NT) {
        const uint32_t m = smix[j];
        uint32_t h = m & tmask;
        while (t_key[h] != m) h = (h + 1u) & tmask;
        if (t_cnt[h] > 1) bucket[atomicAdd(&t_start[h], 1)] = sidx[j];
    }
    __threadfence();
    __syncthreads();
    for (int j = tid; j < np; j += SLAB_NT) {
        const uint32_t m = smix[j];
        uint32_t h = m & tmask;
        while (t_key[h] != m) h = (h + 1u) & tmask;
        const int cnt = t_cnt[h], idx = sidx[j], f = t_first[h], bs = t_start[h] - cnt;
        if (idx == f) {
            w.fb[idx] = (uint8_t)(cnt < T ? cnt : T);
            if (MEAN && cnt > 1) sorted[bs] = idx;
        } else {
            const int r = bucket_rank(bucket + bs, cnt, idx, T);
            w.fb[idx] = 0;
            if (!MEAN) w.rec[idx] = make_int2(f, r);
            else if (r < T) sorted[bs + r] = idx;
        }
    }
    if (MEAN) {
        __threadfence();
        __syncthreads();
        for (int j = tid; j < np; j += SLAB_NT) {
            const uint32_t m = smix[j];
            uint32_t h = m & tmask;
            while (t_key[h] != m) h = (h + 1u) & tmask;
            const int cnt = t_cnt[h], idx = sidx[j];
            if (cnt > 1 && idx == t_first[h]) w.tmean[idx] = cell_mean(pts, sorted + (t_start[h] - cnt), cnt < T ? cnt : T);
        }
    }
}

template <bool MEAN>
__global__ void __launch_bounds__(SLAB_NT) k_slab(const float4 *__restrict__ pts, int T, int shift, Ws w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int wave_tot[SLAB_NT / 64];
    __shared__ int l_goff, l_fill;
    const int s = blockIdx.x, tid = threadIdx.x;
    const int np = w.cursor[s];                 // every point of the slab: region entries + its share of the overflow list
    if (np == 0) return;
    const int2 *sent = w.sent + (int64_t)s * SLAB_CAP;
    if (np <= SLAB_CAP) {
        int32_t *bucket = reinterpret_cast<int32_t *>(smem);
        uint32_t *t_key = reinterpret_cast<uint32_t *>(bucket + SLAB_CAP);
        int32_t *t_cnt = reinterpret_cast<int32_t *>(t_key + SLAB_TS);
        int32_t *t_first = t_cnt + SLAB_TS, *t_start = t_first + SLAB_TS;
        int ts = 64;
        while (ts < 2 * np) ts <<= 1;
        slab_rank_lds<MEAN>(pts, sent, np, T, w, bucket, t_key, t_cnt, t_first, t_start, ts, wave_tot);
    } else {
        // oversize: gather the slab's points (region + matching overflow entries) into a contiguous global list
        if (tid == 0) { l_goff = atomicAdd(w.galloc, np); l_fill = SLAB_CAP; }
        __syncthreads();
        const int64_t goff = l_goff;
        int32_t *gi = w.g_idx + goff;
        uint32_t *gm = w.g_mix + goff;
        for (int j = tid; j < SLAB_CAP; j += SLAB_NT) { const int2 e = sent[j]; gi[j] = e.x; gm[j] = (uint32_t)e.y; }
        const int novf = *w.ovf_count;
        for (int o = tid; o < novf; o += SLAB_NT) {
            const int2 e = w.ovf[o];
            const uint32_t m = (uint32_t)e.y;
            if ((shift >= 32 ? 0 : (int)(m >> shift)) == s) {
                const int q = atomicAdd(&l_fill, 1);
                gi[q] = e.x;
                gm[q] = m;
            }
        }
        __threadfence();
        __syncthreads();
        int ts = 64;
        while (ts < np) ts <<= 1;                // ts < 2 np: tables at [2 goff, 2 goff + 2 np)
        slab_rank_glob<MEAN>(pts, w.g_sorted + goff, gi, gm, np, T, w, w.g_bucket + goff, w.g_key + 2 * goff, w.g_cnt + 2 * goff, w.g_first + 2 * goff,
                       w.g_start + 2 * goff, ts, wave_tot);
    }
}

// ---- K4: flag bytes -> 64-point words; per word the popcount prefix INSIDE its 4096-point block, per block the total.
// (A single-workgroup scan over all flag bytes was tried first: one CU streams ~16 GB/s, 34 us for 524 KB.)
constexpr int WORDS_NT = 256;
constexpr int WORDS_PTS = WORDS_NT * 16;          // 4096 points = 64 words per block
constexpr int MAX_KB = 2048;                      // blocks of 4096 points (n <= 8 M)

__global__ void __launch_bounds__(WORDS_NT) k_words(int n, Ws w) {
    __shared__ int lcnt[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t base = (int64_t)blockIdx.x * WORDS_PTS + tid * 16;
    const uint4 v = base < n ? *reinterpret_cast<const uint4 *>(w.fb + base) : make_uint4(0u, 0u, 0u, 0u);
    // flag bytes are <= 127 (T <= 127): +0x7f sets a byte's top bit iff the byte is nonzero, no carries between
    // bytes; the multiply gathers the four top bits into one nibble
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t nz = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t y = ((x[k] + 0x7f7f7f7fu) & 0x80808080u) >> 7;
        nz |= ((y * 0x01020408u) >> 24) << (4 * k);
    }
    const int64_t left = (int64_t)n - base;       // bytes past n are not flags
    if (left < 16) nz = left <= 0 ? 0u : (nz & ((1u << left) - 1u));
    unsigned long long word = (unsigned long long)nz << (16 * (lane & 3));
    word |= __shfl_xor(word, 1);
    word |= __shfl_xor(word, 2);
    const int nwords = (n + 63) >> 6;
    const int wl = tid >> 2, wg = blockIdx.x * 64 + wl;
    if ((lane & 3) == 0) {
        lcnt[wl] = __popcll(word);
        if (wg < nwords) w.fmask[wg] = word;
    }
    __syncthreads();
    if (tid < 64) {
        const int c = lcnt[tid];
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (tid >= o) incl += t;
        }
        const int wi = blockIdx.x * 64 + tid;
        if (wi < nwords) w.wloc[wi] = incl - c;
        if (tid == 63) w.btot[blockIdx.x] = incl;
    }
}

__device__ __forceinline__ int popc_below(unsigned long long m, int bit) {
    return __popcll(m & (bit == 0 ? 0ull : (~0ull >> (64 - bit))));
}

// floor(a / d) for 0 <= a < 2^23, d >= 1
__device__ __forceinline__ int fdiv(int a, int d, float rd) {
    int q = (int)((float)a * rd);
    int r = a - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

// ---- K5: placement in input order.  A wave = 64 consecutive points = one flag word.
// Prologue (every block, a few hundred L2-resident values): exclusive scan of the per-block totals, first-rank at the
// scene starts, per-scene output offsets with the max_voxels cap; block 0 publishes scene_voxel_off.
constexpr int PLACE_NT = 1024;
constexpr int MAX_SCENES = 1024;

__global__ void __launch_bounds__(PLACE_NT) k_place(const float4 *__restrict__ pts, int n, Geom g, int n_scenes, int T, int max_voxels,
                                                    const int32_t *__restrict__ scene_off, Ws w, float4 *__restrict__ voxels,
                                                    int4 *__restrict__ coords_bzyx, int32_t *__restrict__ num_pts,
                                                    int32_t *__restrict__ scene_voxel_off) {
    __shared__ float4 l_pt[PLACE_NT / 64][64];
    __shared__ int l_np[PLACE_NT / 64][64];
    __shared__ int bpre[MAX_KB + 2];
    __shared__ int l_sfr[MAX_SCENES + 2], l_svo[MAX_SCENES + 2];
    __shared__ int wave_tot[PLACE_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nwords = (n + 63) >> 6;
    const int nkb = (n + WORDS_PTS - 1) / WORDS_PTS;
    // my point first: its loads fly while the prologue runs
    const int i = blockIdx.x * PLACE_NT + tid;
    const bool live = i < n;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    int fbv = 0;
    if (live) { p = pts[i]; fbv = w.fb[i]; }
    {   // exclusive scan of btot[0..nkb): two entries per thread
        const int a0 = 2 * tid < nkb ? w.btot[2 * tid] : 0, a1 = 2 * tid + 1 < nkb ? w.btot[2 * tid + 1] : 0;
        int tot;
        const int ex = block_excl_scan(a0 + a1, wave_tot, PLACE_NT / 64, tot);
        if (2 * tid <= nkb) bpre[2 * tid] = ex;
        if (2 * tid + 1 <= nkb) bpre[2 * tid + 1] = ex + a0;
        __syncthreads();
        for (int s = tid; s <= n_scenes; s += PLACE_NT) {
            const int si = scene_off[s], wi = si >> 6;
            l_sfr[s] = wi < nwords ? bpre[wi >> 6] + w.wloc[wi] + popc_below(w.fmask[wi], si & 63) : tot;
        }
        __syncthreads();
        int running = 0;
        for (int s0 = 0; s0 < n_scenes; s0 += PLACE_NT) {
            const int s = s0 + tid;
            int t = 0;
            if (s < n_scenes) { t = l_sfr[s + 1] - l_sfr[s]; t = t < max_voxels ? t : max_voxels; }
            int ct;
            const int e2 = block_excl_scan(t, wave_tot, PLACE_NT / 64, ct);
            if (s < n_scenes) l_svo[s] = running + e2;
            running += ct;
        }
        if (tid == 0) l_svo[n_scenes] = running;
        __syncthreads();
        if (blockIdx.x == 0)
            for (int s = tid; s <= n_scenes; s += PLACE_NT) scene_voxel_off[s] = l_svo[s];
    }
    int s = 0, cc[3] = {0, 0, 0};
    bool inr = false;
    if (live) {
        s = find_scene(scene_off, n_scenes, i);
        inr = cell_of(p.x, p.y, p.z, g, cc);
    }
    const bool first = fbv > 0;
    const unsigned long long fm = __ballot(first);
    int wi = (blockIdx.x * PLACE_NT + wv * 64) >> 6;                 // this wave's flag word (wave-uniform)
    const int wbase = wi < nwords ? bpre[wi >> 6] + w.wloc[wi] : 0;   // waves past the end hold no live lane
    const int sbase = l_sfr[s], vbase = l_svo[s];
    const int rs = wbase + popc_below(fm, lane) - sbase;              // rank of my cell among its scene's cells
    const bool kept = first && rs < max_voxels;
    const unsigned long long km = __ballot(kept);
    const int nk = __popcll(km);
    const int cpos = popc_below(km, lane);
    const int v = vbase + rs;
    if (kept) {
        l_pt[wv][cpos] = p;
        l_np[wv][cpos] = fbv;
        num_pts[v] = fbv;
        coords_bzyx[v] = make_int4(s, cc[2], cc[1], cc[0]);
    }
    __syncthreads();
    if (nk) {
        const int v0 = __shfl(v, __builtin_ctzll(km));                // kept first points of a wave own consecutive rows
        float4 *dst = voxels + (int64_t)v0 * T;
        const float rT = 1.0f / (float)T;
        const int total = nk * T;
        for (int e = lane; e < total; e += 64) {
            const int row = fdiv(e, T, rT), slot = e - row * T;
            // (non-temporal stores were tried here: 57 instead of 53 us for the call -- the plain stores combine better in L2)
            if (slot == 0) dst[e] = l_pt[wv][row];
            else if (slot >= l_np[wv][row]) dst[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // non-first points of multi-point cells: scattered 16-byte stores into their cell's row
    if (live && inr && !first) {
        const int2 fr = w.rec[i];
        if (fr.y < T) {
            const int f = fr.x, fw = f >> 6;
            const int rf = bpre[fw >> 6] + w.wloc[fw] + popc_below(w.fmask[fw], f & 63) - sbase;
            if (rf < max_voxels) voxels[(int64_t)(vbase + rf) * T + fr.y] = p;
        }
    }
}

// ---- K4': placement of the fused voxelise -> MeanVFE form: one 16-byte feature row per voxel instead of the padded
// [T, C] block (16 N + M (4 C + 16) bytes).  Same prologue as k_place; every store is dense.
__global__ void __launch_bounds__(PLACE_NT) k_place_mean(const float4 *__restrict__ pts, int n, Geom g, int n_scenes, int max_voxels,
                                                         const int32_t *__restrict__ scene_off, Ws w, float4 *__restrict__ feats,
                                                         int4 *__restrict__ coords_bzyx, int32_t *__restrict__ num_pts,
                                                         int32_t *__restrict__ scene_voxel_off) {
    __shared__ int bpre[MAX_KB + 2];
    __shared__ int l_sfr[MAX_SCENES + 2], l_svo[MAX_SCENES + 2];
    __shared__ int wave_tot[PLACE_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nwords = (n + 63) >> 6;
    const int nkb = (n + WORDS_PTS - 1) / WORDS_PTS;
    const int i = blockIdx.x * PLACE_NT + tid;
    const bool live = i < n;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    int fbv = 0;
    if (live) { p = pts[i]; fbv = w.fb[i]; }
    {
        const int a0 = 2 * tid < nkb ? w.btot[2 * tid] : 0, a1 = 2 * tid + 1 < nkb ? w.btot[2 * tid + 1] : 0;
        int tot;
        const int ex = block_excl_scan(a0 + a1, wave_tot, PLACE_NT / 64, tot);
        if (2 * tid <= nkb) bpre[2 * tid] = ex;
        if (2 * tid + 1 <= nkb) bpre[2 * tid + 1] = ex + a0;
        __syncthreads();
        for (int s = tid; s <= n_scenes; s += PLACE_NT) {
            const int si = scene_off[s], wi = si >> 6;
            l_sfr[s] = wi < nwords ? bpre[wi >> 6] + w.wloc[wi] + popc_below(w.fmask[wi], si & 63) : tot;
        }
        __syncthreads();
        int running = 0;
        for (int s0 = 0; s0 < n_scenes; s0 += PLACE_NT) {
            const int s = s0 + tid;
            int t = 0;
            if (s < n_scenes) { t = l_sfr[s + 1] - l_sfr[s]; t = t < max_voxels ? t : max_voxels; }
            int ct;
            const int e2 = block_excl_scan(t, wave_tot, PLACE_NT / 64, ct);
            if (s < n_scenes) l_svo[s] = running + e2;
            running += ct;
        }
        if (tid == 0) l_svo[n_scenes] = running;
        __syncthreads();
        if (blockIdx.x == 0)
            for (int s = tid; s <= n_scenes; s += PLACE_NT) scene_voxel_off[s] = l_svo[s];
    }
    if (fbv == 0) return;                                            // only first points write
    const int s = find_scene(scene_off, n_scenes, i);
    int cc[3];
    cell_of(p.x, p.y, p.z, g, cc);
    const unsigned long long fm = __ballot(true);                     // the first points of this wave (all others have left)
    const int wi = (blockIdx.x * PLACE_NT + wv * 64) >> 6;
    const int rs = bpre[wi >> 6] + w.wloc[wi] + popc_below(fm, lane) - l_sfr[s];
    if (rs >= max_voxels) return;
    const int v = l_svo[s] + rs;
    // a single-point cell's mean is (0 + p) / 1: the same two operations lvq_mean_vfe performs on its padded row
    const float4 mean = fbv == 1 ? make_float4((0.f + p.x) / 1.f, (0.f + p.y) / 1.f, (0.f + p.z) / 1.f, (0.f + p.w) / 1.f) : w.tmean[i];
    feats[v] = mean;
    num_pts[v] = fbv;
    coords_bzyx[v] = make_int4(s, cc[2], cc[1], cc[0]);
}

}  // namespace vh

size_t lvq_hashed_hard_workspace_bytes(int64_t n, int n_scenes) {
    vh::SizerAdapter a;
    vh::Ws w;
    vh::layout(a, w, n, n_scenes);
    return a.s.total();
}

// `continue` cap semantics, C == 4 payloads, T <= 127, key space < 2^31; anything else -> LVQ_EUNSUPPORTED.
// mean != 0: `out` is voxel_features [cap, 4] (voxelise -> MeanVFE fused), else the padded voxels [cap, T, 4].
static int hashed_voxelize(bool mean, const float *pts, const int32_t *scene_off, int64_t n, int n_scenes, int c, const float *range_host,
                           const float *vsize_host, const int32_t *grid_host, int max_pts, int max_voxels, float *out,
                           int32_t *coords_bzyx, int32_t *num_pts, int32_t *scene_voxel_off, void *ws, size_t ws_bytes, hipStream_t st) {
    using namespace vh;
    if (c != 4 || max_pts > 127 || (((uintptr_t)pts | (uintptr_t)out | (uintptr_t)coords_bzyx) & 15)) return LVQ_EUNSUPPORTED;
    const int64_t keyspace = (int64_t)n_scenes * grid_host[0] * grid_host[1] * grid_host[2];
    if (keyspace <= 0 || keyspace >= (1ll << 31) - 1 || n > (int64_t)MAX_KB * WORDS_PTS || n_scenes > MAX_SCENES)
        return LVQ_EUNSUPPORTED;
    const int nslabs = slab_count(n);
    if (nslabs > MAX_SLABS) return LVQ_EUNSUPPORTED;          // N <= 4 M (k_bin holds two counters per slab in 32 KB of LDS)
    int lg = 0;
    while ((1 << lg) < nslabs) ++lg;
    const int shift = 32 - lg;
    LvqArena arena(ws, ws_bytes);
    Ws w;
    layout(arena, w, n, n_scenes);
    if (!arena.ok) return LVQ_EWORKSPACE;
    Geom g;
    for (int j = 0; j < 3; ++j) { g.lo[j] = range_host[j]; g.vs[j] = vsize_host[j]; g.grid[j] = grid_host[j]; }
    static LvqLdsOnce once;
    if (!lvq_ensure_lds(once, {(const void *)k_slab<false>, (const void *)k_slab<true>}, 96 * 1024)) return LVQ_ELAUNCH;
    const unsigned nb = (unsigned)lvq_cdiv(n, BIN_NT * BIN_PPT);
    const float4 *p4 = reinterpret_cast<const float4 *>(pts);
    hipMemsetAsync(w.cursor, 0, sizeof(int32_t) * (MAX_SLABS + 64), st);
    hipLaunchKernelGGL(k_bin, dim3(nb), dim3(BIN_NT), 2 * sizeof(int32_t) * nslabs, st, p4, (int)n, g, n_scenes, shift, nslabs, scene_off, w);
    const size_t slab_lds = sizeof(int32_t) * SLAB_CAP + 4 * sizeof(int32_t) * SLAB_TS;
    if (mean) hipLaunchKernelGGL(k_slab<true>, dim3(nslabs), dim3(SLAB_NT), slab_lds, st, p4, max_pts, shift, w);
    else hipLaunchKernelGGL(k_slab<false>, dim3(nslabs), dim3(SLAB_NT), slab_lds, st, p4, max_pts, shift, w);
    hipLaunchKernelGGL(k_words, dim3((unsigned)lvq_cdiv(n, WORDS_PTS)), dim3(WORDS_NT), 0, st, (int)n, w);
    if (mean)
        hipLaunchKernelGGL(k_place_mean, dim3((unsigned)lvq_cdiv(n, PLACE_NT)), dim3(PLACE_NT), 0, st, p4, (int)n, g, n_scenes, max_voxels,
                           scene_off, w, reinterpret_cast<float4 *>(out), reinterpret_cast<int4 *>(coords_bzyx), num_pts, scene_voxel_off);
    else
        hipLaunchKernelGGL(k_place, dim3((unsigned)lvq_cdiv(n, PLACE_NT)), dim3(PLACE_NT), 0, st, p4, (int)n, g, n_scenes, max_pts, max_voxels,
                           scene_off, w, reinterpret_cast<float4 *>(out), reinterpret_cast<int4 *>(coords_bzyx), num_pts,
                           scene_voxel_off);
    return lvq_launch_status();
}

int lvq_hashed_voxelize_hard(const float *pts, const int32_t *scene_off, int64_t n, int n_scenes, int c, const float *range_host,
                             const float *vsize_host, const int32_t *grid_host, int max_pts, int max_voxels, float *voxels,
                             int32_t *coords_bzyx, int32_t *num_pts, int32_t *scene_voxel_off, void *ws, size_t ws_bytes,
                             hipStream_t st) {
    return hashed_voxelize(false, pts, scene_off, n, n_scenes, c, range_host, vsize_host, grid_host, max_pts, max_voxels, voxels, coords_bzyx,
                           num_pts, scene_voxel_off, ws, ws_bytes, st);
}

int lvq_hashed_voxelize_mean(const float *pts, const int32_t *scene_off, int64_t n, int n_scenes, int c, const float *range_host,
                             const float *vsize_host, const int32_t *grid_host, int max_pts, int max_voxels, float *voxel_features,
                             int32_t *coords_bzyx, int32_t *num_pts, int32_t *scene_voxel_off, void *ws, size_t ws_bytes,
                             hipStream_t st) {
    return hashed_voxelize(true, pts, scene_off, n, n_scenes, c, range_host, vsize_host, grid_host, max_pts, max_voxels, voxel_features,
                           coords_bzyx, num_pts, scene_voxel_off, ws, ws_bytes, st);
}
