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
#include <stdlib.h>
#include <type_traits>

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
    int nsplit;       // KV splits (flash-decoding style) for few-queries x many-keys shapes
    int nqt;          // query tiles (of 64*QT queries)
    float *part;      // [B*H][nsplit][Nq][dh + 2] fp32 partial (unnormalised O | m | l) when nsplit > 1
    // tiled key stream (k_attn32 only; bev_tiles.hip): key slot r (0..63) of tile t of batch b is row row_src[(b * ntiles + t) * 64 + r]
    // of k / v -- ONE buffer that holds the per-model table rows and the computed rows of every batch (k_bs / v_bs are not applied)
    const int32_t *row_src;
    // signed pair stream (k_attn32<., ., 2>, lvq_attention_bf16_tiled_signed): batch b with pair_info[2 b + 1] != 0 streams the
    // pair_info[2 b] tiles of pair_src + b * pair_cap * 64 (bev_tiles.hip: k_scene_pairs) and SUBTRACTS keys 32..63 of every tile; the
    // per-model totals (unnormalised O | m | l per head and query, shared by all batches) are added by the combine kernel
    const int32_t *pair_src, *pair_info;
    int pair_cap;
    const float *totals;
    int32_t *flags;           // out, per (b, h): the signed result of some query is unusable (non-finite, or cancellation too deep)
    const int32_t *pred;      // in, per (b, h): workgroups / rows of (b, h) with pred == 0 do nothing (the predicated full re-run)
    int force_part;           // write partials even when nsplit == 1
    int32_t *stat_slots;      // [STAT_SLOTS][64] scratch of the guard statistics (per slot: count | max ratio bits), zeroed per call; with pred set: reduced into stats
    int32_t *stats;           // optional [4] (accumulated): rows whose own (dirty) keys dominate the softmax mass | (batch, head) pairs re-run | max l / l_table (float bits) | -
};

constexpr int KVB = 64;  // keys per tile

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
// two fp32 -> packed fp16 pair, round to nearest even (v_cvt_f16_f32 x 2; NOT v_cvt_pkrtz, which truncates)
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
    f16x2_t p = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, p);
}

// two fp32 -> packed bf16 pair (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    bf16x2_t p = {(__bf16)a, (__bf16)b};
    return *reinterpret_cast<uint32_t *>(&p);
}

typedef __attribute__((ext_vector_type(4))) int i32x4;

// raw v_exp_f32 (no denormal fix-up sequence: arguments here are <= 0 and tiny results may flush to zero)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max over the four lane groups g = lane>>4 that hold different keys of the same query (VALU permlane swaps, no LDS)
__device__ __forceinline__ float max_over_groups(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// QT = 16-query column tiles per wave: a wave owns 16*QT queries and reuses every K / V fragment QT times.
// The main loop is written for a lean instruction stream (the first version issued ~1100 instructions per 18
// MFMAs and was VALU-issue-bound): K/V tiles arrive through buffer loads whose per-lane offset is loop
// invariant and whose tile offset is a scalar (hardware bounds check = zero fill past Nkv), exponentials are raw
// v_exp_f32, the row maximum uses v_max3 chains + two permlane swaps, the row sum is an extra MFMA row, and
// bias / causal / ragged-tile handling lives in a separate slow path taken only by the tiles that need it.
// NW = waves per workgroup (4, 8 or 12): all of them share one K/V stream, so a 12-wave workgroup (192 queries)
// reads K/V from HBM/MALL three times less often than three 4-wave workgroups (PMC: 26 GB per launch = 8x the
// algorithmic bytes with NW = 4 on the 576 x 262144 shape).
template <int DHP, int NSPLIT, int QT, int NW>
__global__ void __launch_bounds__(NW * 64) k_attn(AttnArgs a) {
    constexpr int NS = (NSPLIT == 3) ? 2 : 1;
    // K / V rows in LDS.  DHP == 64: unpadded 128-byte rows with the 16-byte chunk index XORed by (row & 6) -- conflict-free
    // for BOTH the ds_read_b128 K reads (16-lane groups {0-3,12-15,20-27}, ...) and the ds_read_b64_tr_b16 V reads (the +16 B
    // pad left both 2-way: SQ_LDS_BANK_CONFLICT was 40 % of the LDS cycles).  Other head dims keep the padded rows.
    constexpr bool SWZ = DHP == 64 && NW >= 8;
    constexpr int KROW = SWZ ? DHP : DHP + 8;       // bf16 elements per K / V row
    constexpr int NC = DHP / 32;        // 32-wide k chunks of the head dim
    constexpr int ND = DHP / 16;        // 16-wide output tiles of the head dim
    constexpr int QW = 16 * QT;         // queries per wave
    constexpr int CH = DHP / 8;                      // 16-byte chunks per row
    constexpr int NT = NW * 64;
    constexpr int NLD = (KVB * CH + NT - 1) / NT;    // chunks per thread per operand
    constexpr int TILE_E = 2 * NS * KVB * KROW;      // bf16 elements of one K+V stage
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    // per stage: Ks [NS][KVB][KROW] row-major keys | Vs [NS][KVB][KROW] row-major values (read transposed by
    // ds_read_b64_tr_b16); two stages

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, g = lane >> 4, l15 = lane & 15;
    // 1-D grid.  The nqt workgroups that stream the same K/V range (same batch, head, split) should share an L2: hardware deals
    // block ids round-robin over the 8 XCDs, so the long-stream forms (NW >= 8) give them ids 8 apart (same XCD, back to back;
    // the group count is padded to a multiple of 8 by the launcher).  Short forms keep plain order (query tile fastest).
    int grp, qtile;
    if (NW >= 8) {
        const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
        qtile = pos % a.nqt;
        grp = (pos / a.nqt) * 8 + xcd;
        if (grp >= a.B * a.H * a.nsplit) return;                // padding workgroup (whole workgroup leaves)
    } else {
        grp = blockIdx.x / a.nqt;
        qtile = blockIdx.x - grp * a.nqt;
    }
    const int sp = grp % a.nsplit, h = (grp / a.nsplit) % a.H, b = grp / (a.nsplit * a.H);
    const int hk = h / (a.H / a.Hkv);
    const int q0 = qtile * (NW * QW) + wid * QW;
    // plain-bf16 mode: Q is pre-multiplied by scale*log2(e) once (re-rounded to bf16), so the MFMA output is already the
    // exponent argument and the per-score v_fma disappears from the softmax (the loop is VALU-issue-bound: ~17 exp + ~75
    // other vector issues against 18 MFMAs per 64-key tile).  bf16x3 keeps exact operands and scales in fp32.
    constexpr bool FAST = NSPLIT == 1;
    const float cexp0 = a.scale * 1.4426950408889634f;  // scores are exponentiated in the log2 domain
    const float cexp = FAST ? 1.0f : cexp0;
    const float bias_mul = FAST ? 1.4426950408889634f : 1.0f / a.scale;   // bias enters before the cexp multiply

    // Q fragments (B operand of S^T = K Q^T): lane supplies Q[q0 + qt*16 + l15][c*32 + 8g .. +7]
    bf16x8 qf[NS][QT][NC];
    {
        const uint16_t *src[2] = {a.q, a.ql};
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int kk = c * 32 + g * 8, qi = q0 + qt * 16 + l15;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (qi < a.Nq && kk < a.dh)
                        v = *reinterpret_cast<const uint4 *>(src[s] + (int64_t)b * a.q_bs + (int64_t)qi * a.ldq + (int64_t)h * a.q_hs + kk);
                    if (FAST) {
                        uint32_t *w = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            w[e] = pack_bf16(__uint_as_float(w[e] << 16) * cexp0, __uint_as_float(w[e] & 0xffff0000u) * cexp0);
                    }
                    qf[s][qt][c] = *reinterpret_cast<bf16x8 *>(&v);
                }
    }

    // o[ND] is the row-sum tile: V^T is extended by a row of ones, so l = sum_j p_j falls out of the same MFMAs
    f32x4 o[ND + 1][QT];
#pragma unroll
    for (int n = 0; n <= ND; ++n)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[n][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) m_run[qt] = -INFINITY;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;

    const int n_tiles = (a.Nkv + KVB - 1) / KVB;
    const int tps = (n_tiles + a.nsplit - 1) / a.nsplit;
    const int t0 = sp * tps;
    int t1 = t0 + tps < n_tiles ? t0 + tps : n_tiles;
    if (a.causal) {   // keys beyond the last visible one of this workgroup's last query are never needed
        int qlast = qtile * (NW * QW) + NW * QW - 1;
        if (qlast > a.Nq - 1) qlast = a.Nq - 1;
        const int tl = (qlast + a.Nkv - a.Nq) / KVB + 1;
        if (t1 > tl) t1 = tl;
    }

    // ---- K/V staging: buffer descriptors cover exactly the valid rows of this (batch, kv-head), so rows past Nkv
    // read as zero in hardware; the per-lane byte offset is loop invariant, the tile offset is a scalar ----
    const int wave_b = __builtin_amdgcn_readfirstlane(b), wave_hk = __builtin_amdgcn_readfirstlane(hk);
    __amdgpu_buffer_rsrc_t rk[NS], rv[NS];
    {
        const uint16_t *kb[2] = {a.k, a.kl}, *vb[2] = {a.v, a.vl};
        const uint32_t kbytes = (uint32_t)(((int64_t)(a.Nkv - 1) * a.ldk + a.dh) * 2);
        const uint32_t vbytes = (uint32_t)(((int64_t)(a.Nkv - 1) * a.ldv + a.dh) * 2);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            rk[s] = __builtin_amdgcn_make_buffer_rsrc((void *)(kb[s] + (int64_t)wave_b * a.k_bs + (int64_t)wave_hk * a.k_hs), 0, kbytes, 0x00020000);
            rv[s] = __builtin_amdgcn_make_buffer_rsrc((void *)(vb[s] + (int64_t)wave_b * a.v_bs + (int64_t)wave_hk * a.v_hs), 0, vbytes, 0x00020000);
        }
    }
    uint32_t koff[NLD], voff[NLD], lso[NLD];
    bool live[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + NT * i, r = e / CH, kk = (e - r * CH) * 8;
        live[i] = (e < KVB * CH) && (kk < a.dh);        // head-dim padding chunks stay zero
        koff[i] = (uint32_t)((r * a.ldk + kk) * 2);
        voff[i] = (uint32_t)((r * a.ldv + kk) * 2);
        lso[i] = (uint32_t)(r * KROW + (SWZ ? ((kk >> 3) ^ (r & 6)) << 3 : kk));
    }
    const uint32_t ktile = (uint32_t)(KVB * a.ldk * 2), vtile = (uint32_t)(KVB * a.ldv * 2);
    // two register sets: the loads of tile t+2 are in flight while tile t is computed and tile t+1 (the other set) is written to
    // LDS -- with one set the loop ran at one HBM/Infinity-Cache round trip per 64-key tile (~1.3 us)
    i32x4 skA[NS][NLD], svA[NS][NLD], skB[NS][NLD], svB[NS][NLD];
    auto gload = [&](int t, i32x4 (&sk)[NS][NLD], i32x4 (&sv)[NS][NLD]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                sk[s][i] = i32x4{0, 0, 0, 0};
                sv[s][i] = i32x4{0, 0, 0, 0};
                if (live[i]) {
                    sk[s][i] = __builtin_amdgcn_raw_buffer_load_b128(rk[s], koff[i], t * ktile, 0);
                    sv[s][i] = __builtin_amdgcn_raw_buffer_load_b128(rv[s], voff[i], t * vtile, 0);
                }
            }
    };
    auto lstore = [&](int buf, i32x4 (&sk)[NS][NLD], i32x4 (&sv)[NS][NLD]) __attribute__((always_inline)) {
        uint16_t *kd = smem + buf * TILE_E, *vd = kd + NS * KVB * KROW;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (tid + NT * i < KVB * CH) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    *reinterpret_cast<i32x4 *>(kd + s * KVB * KROW + lso[i]) = sk[s][i];
                    *reinterpret_cast<i32x4 *>(vd + s * KVB * KROW + lso[i]) = sv[s][i];
                }
            }
    };

    // ---- one 64-key tile.  MASKED = false: every key is valid and visible, no bias (the common case) ----
    auto tile = [&](const uint16_t *Ks, const uint16_t *Vs, int t, auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        // element offset inside a V row of the 4 columns that lane 4q+p (q = l15>>2, p = l15&3) addresses for d-block n; the
        // row is (multiple of 16) + 4g + q, so its swizzle key (row & 6) is (4g + q) & 6 for every sub-tile
        auto vcol = [&](int n) { return SWZ ? ((((2 * n + ((l15 & 3) >> 1)) ^ ((4 * g + (l15 >> 2)) & 6)) << 3) + 4 * (l15 & 1)) : n * 16 + 4 * (l15 & 3); };
        const int kleft = a.Nkv - t * KVB;
        const int nkt = (!MASKED || kleft >= KVB) ? 4 : (kleft + 15) >> 4;   // 16-key sub-tiles holding a valid key
        // DEFER (unmasked plain-bf16 tiles): the score accumulators start at -m_run, so the MFMA result is s - m_run and goes
        // straight into v_exp_f32; the running max only moves (and O / l are only rescaled) when a tile exceeds it by more than
        // DEFER_THR in the log2 domain, i.e. P <= 2^DEFER_THR -- fp32 accumulation and bf16's 8-bit exponent have the headroom.
        // The rare rescale tile runs its own copy of the P V product, so the common path accumulates O in place.
        constexpr bool DEFER = FAST && !MASKED && QT == 1 && NW >= 8;   // (4-wave kernels: short K/V, registers go to occupancy)
        constexpr float DEFER_THR = 6.0f;
        const bool fresh = DEFER && m_run[0] == -INFINITY;        // first tile of this split: no reference yet
        const float sinit = (DEFER && !fresh) ? -m_run[0] : 0.f;
        f32x4 sc[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) sc[kt][qt] = f32x4{sinit, sinit, sinit, sinit};
            if (!MASKED || kt < nkt) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int kcol = SWZ ? (((c * 4 + g) ^ (l15 & 6)) << 3) : c * 32 + g * 8;
                    const bf16x8 kh = *reinterpret_cast<const bf16x8 *>(Ks + (kt * 16 + l15) * KROW + kcol);
                    bf16x8 kl;
                    if (NSPLIT == 3) kl = *reinterpret_cast<const bf16x8 *>(Ks + (KVB + kt * 16 + l15) * KROW + kcol);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        sc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qf[0][qt][c], sc[kt][qt], 0, 0, 0);
                        if (NSPLIT == 3) {
                            sc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qf[NS - 1][qt][c], sc[kt][qt], 0, 0, 0);
                            sc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qf[0][qt][c], sc[kt][qt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // V^T fragments of the whole tile are fetched BEFORE the softmax VALU section (32 VGPRs at dh = 64): issued after
        // it, every PV MFMA sat behind its own ds_read + lgkmcnt(0)
        constexpr bool PRE = !MASKED && NSPLIT == 1 && QT == 1 && ND <= 4 && NW >= 8;
        bf16x8 vpre[PRE ? 2 : 1][PRE ? ND : 1];
        if (PRE) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const uint16_t *vbase = Vs + ((2 * s2) * 16 + 4 * g + (l15 >> 2)) * KROW;
#pragma unroll
                for (int n = 0; n < ND; ++n) {
                    const uint16_t *va = vbase + vcol(n);
                    bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4 *)va);
                    bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4 *)(va + 16 * KROW));
                    vpre[s2][n] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t pk[QT][4][2], pkl[QT][4][2];   // packed bf16 P^T [qt][kt][pair], hi and (bf16x3) lo parts

        // O^T += V^T P^T : A = V^T[d][keys] via the transposing LDS read, B = P^T straight from the registers.
        // k index of step s2, element j of lane group g  <->  key (2*s2 + (j>>2))*16 + 4g + (j&3)   (both operands)
        auto pv = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (!MASKED || 2 * s2 < nkt) {
                    bf16x8 pf[QT], pfl[QT];
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        uint4 u = make_uint4(pk[qt][2 * s2][0], pk[qt][2 * s2][1], pk[qt][2 * s2 + 1][0], pk[qt][2 * s2 + 1][1]);
                        pf[qt] = *reinterpret_cast<bf16x8 *>(&u);
                        if (NSPLIT == 3) {
                            uint4 ul = make_uint4(pkl[qt][2 * s2][0], pkl[qt][2 * s2][1], pkl[qt][2 * s2 + 1][0], pkl[qt][2 * s2 + 1][1]);
                            pfl[qt] = *reinterpret_cast<bf16x8 *>(&ul);
                        }
                    }
                    // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of its group's 4-key x 16-d block
                    const uint16_t *vbase = Vs + ((2 * s2) * 16 + 4 * g + (l15 >> 2)) * KROW;
#pragma unroll
                    for (int n = 0; n < ND; ++n) {
                        const uint16_t *va = vbase + vcol(n);
                        bf16x8 vh;
                        if (PRE) {
                            vh = vpre[PRE ? s2 : 0][PRE ? n : 0];
                        } else {
                            bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4 *)va);
                            bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4 *)(va + 16 * KROW));
                            vh = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                        bf16x8 vlo;
                        if (NSPLIT == 3) {
                            bf16x4 w0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4 *)(va + KVB * KROW));
                            bf16x4 w1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4 *)(va + (KVB + 16) * KROW));
                            vlo = __builtin_shufflevector(w0, w1, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
                            o[n][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pf[qt], o[n][qt], 0, 0, 0);
                            if (NSPLIT == 3) {
                                o[n][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pfl[qt], o[n][qt], 0, 0, 0);
                                o[n][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vlo, pf[qt], o[n][qt], 0, 0, 0);
                            }
                        }
                    }
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {     // row sums: the ones-row of V^T
                        o[ND][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qt], o[ND][qt], 0, 0, 0);
                        if (NSPLIT == 3) o[ND][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pfl[qt], o[ND][qt], 0, 0, 0);
                    }
                }
            }
        };
        auto row_max = [&](int qt) __attribute__((always_inline)) {
            float tmax = fmaxf(fmaxf(sc[0][qt][0], sc[0][qt][1]), fmaxf(sc[0][qt][2], sc[0][qt][3]));
#pragma unroll
            for (int kt = 1; kt < 4; ++kt)
                tmax = fmaxf(fmaxf(fmaxf(tmax, sc[kt][qt][0]), fmaxf(sc[kt][qt][1], sc[kt][qt][2])), sc[kt][qt][3]);
            return max_over_groups(tmax);
        };
        auto exp_pack = [&](int qt) __attribute__((always_inline)) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                pk[qt][kt][0] = pack_bf16(fast_exp2(sc[kt][qt][0]), fast_exp2(sc[kt][qt][1]));
                pk[qt][kt][1] = pack_bf16(fast_exp2(sc[kt][qt][2]), fast_exp2(sc[kt][qt][3]));
            }
        };

        if (DEFER) {
            const float tmax = row_max(0);                        // relative to m_run (absolute on the first tile)
            if (__any(fresh || tmax > DEFER_THR)) {               // rare: move the reference of the rows that need it
                const float delta = fresh ? tmax : fmaxf(tmax, 0.f);
                const float al = fresh ? 1.0f : fast_exp2(-delta);
                m_run[0] = (fresh ? 0.f : m_run[0]) + delta;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[kt][0][r] -= delta;
#pragma unroll
                for (int n = 0; n <= ND; ++n) o[n][0] *= al;
                exp_pack(0);
                pv();
            } else {
                exp_pack(0);
                pv();
            }
            return;
        }

        bool any_grow = false;
        float alpha[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (MASKED) {
                const int qi = q0 + qt * 16 + l15;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = t * KVB + kt * 16 + g * 4 + r;
                        if (a.bias && key < a.Nkv && qi < a.Nq)   // folded so that one fma(x, cexp, -m) serves both terms
                            sc[kt][qt][r] += a.bias[(((int64_t)b * a.H + h) * a.Nq + qi) * a.Nkv + key] * bias_mul;
                        if (key >= a.Nkv || (a.causal && key > qi + a.Nkv - a.Nq)) sc[kt][qt][r] = -INFINITY;
                    }
            }
            const float tmax = row_max(qt);
            const float m_new = fmaxf(m_run[qt], tmax * cexp);
            const float m_safe = (MASKED && m_new == -INFINITY) ? 0.f : m_new;
            alpha[qt] = fast_exp2(m_run[qt] - m_safe);
            any_grow = any_grow || (m_new != m_run[qt]);
            m_run[qt] = m_new;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                float p[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = fast_exp2(fmaf(sc[kt][qt][r], cexp, -m_safe));
                pk[qt][kt][0] = pack_bf16(p[0], p[1]);
                pk[qt][kt][1] = pack_bf16(p[2], p[3]);
                if (NSPLIT == 3) {
                    pkl[qt][kt][0] = pack_bf16(p[0] - __uint_as_float(pk[qt][kt][0] << 16), p[1] - __uint_as_float(pk[qt][kt][0] & 0xffff0000u));
                    pkl[qt][kt][1] = pack_bf16(p[2] - __uint_as_float(pk[qt][kt][1] << 16), p[3] - __uint_as_float(pk[qt][kt][1] & 0xffff0000u));
                }
            }
        }
        // rescale the accumulators (masked tiles: only when some running max moved -- alpha == 1 exactly otherwise; the
        // unmasked form is unconditional: the branch made the compiler copy all of O around it every tile)
        if (MASKED ? __any(any_grow) : true) {
#pragma unroll
            for (int n = 0; n <= ND; ++n)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[n][qt] *= alpha[qt];
        }
        pv();
    };

    // K/V tiles are double-buffered in LDS and double-buffered again in registers: tile t+2 is requested before the MFMAs of
    // tile t, tile t+1 (requested one tile earlier) is written to the other LDS buffer after them; one barrier per tile.
    const bool special = a.bias != nullptr || a.causal;
    auto step = [&](int t, int buf) __attribute__((always_inline)) {
        const uint16_t *Ks = smem + buf * TILE_E, *Vs = Ks + NS * KVB * KROW;
        if (special || (t + 1) * KVB > a.Nkv) tile(Ks, Vs, t, std::true_type{});
        else tile(Ks, Vs, t, std::false_type{});
    };
    constexpr bool DEEP = NW >= 8;       // the many-wave forms serve the long K/V streams; short ones keep registers for occupancy
    if (t0 < t1) { gload(t0, skA, svA); lstore(0, skA, svA); }
    if (DEEP) {
        if (t0 + 1 < t1) gload(t0 + 1, skB, svB);
        __syncthreads();
        for (int t = t0; t < t1; t += 2) {
            if (t + 2 < t1) gload(t + 2, skA, svA);
            step(t, 0);
            if (t + 1 < t1) lstore(1, skB, svB);
            __syncthreads();
            if (t + 1 >= t1) break;
            if (t + 3 < t1) gload(t + 3, skB, svB);
            step(t + 1, 1);
            if (t + 2 < t1) lstore(0, skA, svA);
            __syncthreads();
        }
    } else {
        __syncthreads();
        for (int t = t0; t < t1; ++t) {
            const int buf = (t - t0) & 1;
            if (t + 1 < t1) gload(t + 1, skA, svA);
            step(t, buf);
            if (t + 1 < t1) lstore(buf ^ 1, skA, svA);
            __syncthreads();
        }
    }

    // ---- write back: lane holds O[q0 + qt*16 + l15][n*16 + g*4 + r]; o[ND][qt][*] = l ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qi = q0 + qt * 16 + l15;
        if (qi >= a.Nq) continue;
        const float l_run = o[ND][qt][0];
        if (a.nsplit > 1) {
            float *pr = a.part + ((((int64_t)b * a.H + h) * a.nsplit + sp) * a.Nq + qi) * (a.dh + 2);
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                const int d0 = n * 16 + g * 4;
                if (d0 < a.dh) *reinterpret_cast<f32x4 *>(pr + d0) = o[n][qt];
            }
            if (g == 0) { pr[a.dh] = m_run[qt]; pr[a.dh + 1] = l_run; }
            continue;
        }
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        uint16_t *dst = a.o + (int64_t)b * a.o_bs + (int64_t)qi * a.ldo + (int64_t)h * a.o_hs;
        uint16_t *dl = a.ol ? a.ol + (int64_t)b * a.o_bs + (int64_t)qi * a.ldo + (int64_t)h * a.o_hs : nullptr;
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            const int d0 = n * 16 + g * 4;
            if (d0 >= a.dh) continue;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = o[n][qt][r] * inv;
            uint2 hv = make_uint2(pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]));
            *reinterpret_cast<uint2 *>(dst + d0) = hv;
            if (dl) {
                uint2 lv = make_uint2(pack_bf16(y[0] - __uint_as_float(hv.x << 16), y[1] - __uint_as_float(hv.x & 0xffff0000u)),
                                      pack_bf16(y[2] - __uint_as_float(hv.y << 16), y[3] - __uint_as_float(hv.y & 0xffff0000u)));
                *reinterpret_cast<uint2 *>(dl + d0) = lv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_attn32: head_dim 64, plain bf16, long K/V stream, no bias / mask, Nkv % 64 == 0 -- on 32x32x16 MFMAs.
// Why a second formulation: the 16x16x32 loop above is bound by the SIMD's vector ISSUE port, not by the matrix pipe.  An
// MFMA holds the port for 8 cycles whatever its shape, so a 16-cycle 16x16x32 leaves 8 free cycles (two VALU slots) and a
// 32-cycle 32x32x16 leaves 24 (six slots) for the same FLOPs per cycle; and with 32 queries per wave each K / V fragment
// read from LDS serves twice the queries, the row maximum needs one cross-lane step per 32 keys instead of two per 16.
// Layout (S^T = K Q^T as before): A = K rows (lane: key l&31, dh 16 ks + 8 hi + j, hi = lane >> 5), B = Q^T (lane: query
// l&31, same dh) -> C: lane (query, hi) holds keys 8 (i>>2) + 4 hi + (i&3), i = 0..15, of each 32-key block.  Registers
// 8 s .. 8 s + 7 of a block are therefore exactly the B operand of O^T += V^T P^T for the 16-key step s under the key
// permutation  k-index 8 hi + j  <->  key 16 s + 8 (j>>2) + 4 hi + (j&3),  and the A operand V^T follows the same permutation
// with two ds_read_b64_tr_b16 (rows 16 s + 4 hi + 0..3 and + 8) -- nothing is shuffled between the two products.
// LDS rows are unpadded 128-byte rows, chunk index XORed by f(row) = bit1(row) << 2 | (row >> 2) & 3: conflict-free for the
// ds_read_b128 K reads (16 distinct rows per 16-lane group) and for the transposed V reads (4 rows x 64 B per half wave).
// The row sum is a VALU add per score (a ones-row would cost a whole 32-row MFMA); deferred rescale as in k_attn.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ds_read_b64_tr_b16 as opaque asm: the builtin carries no alias information, so behind an LDS-DMA (global_load_lds) the
// compiler's waitcnt pass puts s_waitcnt vmcnt(0) in front of it -- i.e. it waits for the K/V request that was JUST made for two
// tiles ahead, and the stream runs at one memory round trip per tile.  The asm form is invisible to that pass; the consumer
// waits on lgkmcnt itself (lds_tr_wait) -- hardware counters are in order, so the compiler's own counted waits stay sufficient.
__device__ __forceinline__ bf16x4 lds_tr_read(uint32_t lds_byte_addr) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_byte_addr) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ bf16x4 lds_tr_read_o(uint32_t lds_byte_addr) {      // same, with a 16-bit immediate byte offset
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_tr_wait(bf16x4 &a, bf16x4 &b, bf16x4 &c, bf16x4 &d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
// same, leaving the four newest LDS reads in flight (the next step's fragments)
__device__ __forceinline__ void lds_tr_wait4(bf16x4 &a, bf16x4 &b, bf16x4 &c, bf16x4 &d) {
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

// fragment reads of the pipelined stream (k_attn32<..., PIPE>): four K fragments (ds_read_b128) / the eight transposed V fragments of
// one 32-key block, as asm for the same reason; waited for by lds_wait_k / lds_wait_v
template <int OFF>
__device__ __forceinline__ void lds_k_read4(bf16x8 (&kf)[4], uint32_t sb, const uint32_t (&kl)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[ks]) : "v"(sb + kl[ks]), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_v_read8(bf16x4 (&va)[8], uint32_t e0, uint32_t e1, uint32_t o0, uint32_t o1) {
    va[0] = lds_tr_read_o<OFF>(e0);        va[1] = lds_tr_read_o<OFF + 1024>(o0);  va[2] = lds_tr_read_o<OFF>(e1);        va[3] = lds_tr_read_o<OFF + 1024>(o1);
    va[4] = lds_tr_read_o<OFF + 2048>(e0); va[5] = lds_tr_read_o<OFF + 3072>(o0);  va[6] = lds_tr_read_o<OFF + 2048>(e1); va[7] = lds_tr_read_o<OFF + 3072>(o1);
}
__device__ __forceinline__ void lds_wait_k(bf16x8 (&kf)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3])::"memory");
}
__device__ __forceinline__ void lds_wait_v(bf16x4 (&va)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]), "+v"(va[4]), "+v"(va[5]), "+v"(va[6]), "+v"(va[7])::"memory");
}

// QS = 1 ("mixed" precision): Q arrives as hi + lo and every score is K . (Q_hi + Q_lo) -- two MFMAs per K fragment on
// the same accumulator -- while K, V and P stay plain bf16.  Rounding Q is the same perturbation for every key of a row, so it
// does not average out over the stream the way the per-key roundings of K, V and P do (tools/precision_study.py: 2.8e-3 of the
// 1e-3 budget at 262 144 keys); splitting it costs 4 of 12 MFMAs per 32-key block and 16 VGPRs, nothing in LDS or HBM.
// QS = 2 ("mixed16"): Q as ONE fp16 operand (hi + lo summed, scaled, rounded to 11 bits) against K stored as fp16 -- one MFMA per K
// fragment at the bf16 rate; P and V stay bf16.  Q's rounding is common to all keys of a row, so the result carries ~3.5x the error of
// QS = 1 (3.7e-4 instead of 1.1e-4 on the bench scene), for 2/3 of its MFMA passes.
// PIPE: the fast stream as ONE software-pipelined instruction stream per wave (stream_pipe below): the score MFMAs of the next 32-key
// block are issued between the exponentials of the current one, so a wave overlaps its own matrix and vector work instead of relying
// on the other waves of its SIMD to do so; 4 LDS slots, 2 waves per SIMD (256 registers).
template <int NW, int QS, int TL = 0, bool PIPE = false>
__global__ void __launch_bounds__(NW * 64, PIPE ? 2 : 3) k_attn32(AttnArgs a) {
    constexpr int DH = 64, KROW = 64, CH = 8, NT = NW * 64;
    constexpr int NLD = (KVB * CH + NT - 1) / NT;
    constexpr int TILE_E = 2 * KVB * KROW;                     // K + V of one stage (bf16 elements)
    constexpr int NSLOT = PIPE ? 4 : 3;                        // LDS slots of the K|V ring
    static_assert(!PIPE || NW == 4, "the pipelined stream is written for 4 waves (4 DMA pieces per wave and tile)");
    constexpr float THR = 6.0f;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    // wid through readfirstlane: the compiler does not know tid >> 6 is wave-uniform, and everything derived from it (which DMA
    // piece, K or V, the piece's source row) then turns into exec-masked vector code -- in the tiled stream even a global load of
    // ldk / ldv with a vmcnt(0) behind it, inside the loop that lives on counted vmcnt
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l31 = lane & 31;
    // Workgroup order: hardware deals block ids round-robin over the 8 XCDs, and each XCD has its own L2.  The nqt workgroups
    // that stream the SAME K/V range (same batch, head, split; different query tiles) get block ids 8 apart, i.e. the same XCD
    // back to back, so the range leaves HBM once and the repeats are L2 hits (in plain id order the three copies went to three XCDs).
    const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
    const int qtile = pos % a.nqt, grp = (pos / a.nqt) * 8 + xcd;
    if (grp >= a.B * a.H * a.nsplit) return;                    // padding of the group count to a multiple of 8 (whole workgroup)
    const int sp = grp % a.nsplit, h = (grp / a.nsplit) % a.H, b = grp / (a.nsplit * a.H);
    const int hk = h / (a.H / a.Hkv);
    if (a.pred && a.pred[b * a.H + h] == 0) return;             // predicated re-run: nothing to redo for this (batch, head)
    const int q0 = qtile * (NW * 32) + wid * 32;
    const int qi = q0 + l31;
    const float cexp0 = a.scale * 1.4426950408889634f;

    // Q^T fragments, pre-multiplied by scale * log2(e): lane supplies Q[qi][16 ks + 8 hi .. +7]
    bf16x8 qf[4], qfl[(QS == 1 || QS == 3) ? 4 : 1];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0), vl = make_uint4(0, 0, 0, 0);
        const int64_t qo = (int64_t)b * a.q_bs + (int64_t)qi * a.ldq + (int64_t)h * a.q_hs + 16 * ks + 8 * hi;
        if (qi < a.Nq) {
            v = *reinterpret_cast<const uint4 *>(a.q + qo);
            if (QS && a.ql) vl = *reinterpret_cast<const uint4 *>(a.ql + qo);
        }
        uint32_t *w = reinterpret_cast<uint32_t *>(&v), *wl = reinterpret_cast<uint32_t *>(&vl);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (QS == 3) {      // fp16 hi + fp16 lo (22 bits) against fp16 K: the per-model totals of the mixed16 mode
                const float x0 = (__uint_as_float(w[e] << 16) + __uint_as_float(wl[e] << 16)) * cexp0;
                const float x1 = (__uint_as_float(w[e] & 0xffff0000u) + __uint_as_float(wl[e] & 0xffff0000u)) * cexp0;
                const f16x2_t hh = {(_Float16)x0, (_Float16)x1};
                w[e] = __builtin_bit_cast(uint32_t, hh);
                wl[e] = pack_f16(x0 - (float)hh[0], x1 - (float)hh[1]);
            } else if (QS == 2) {      // (hi + lo) * c in fp32 -> fp16 (round to nearest even)
                const float x0 = (__uint_as_float(w[e] << 16) + __uint_as_float(wl[e] << 16)) * cexp0;
                const float x1 = (__uint_as_float(w[e] & 0xffff0000u) + __uint_as_float(wl[e] & 0xffff0000u)) * cexp0;
                w[e] = pack_f16(x0, x1);
            } else if (QS) {    // (hi + lo) * c in fp32, split again: hi' = bf16(x), lo' = bf16(x - hi')
                const float x0 = (__uint_as_float(w[e] << 16) + __uint_as_float(wl[e] << 16)) * cexp0;
                const float x1 = (__uint_as_float(w[e] & 0xffff0000u) + __uint_as_float(wl[e] & 0xffff0000u)) * cexp0;
                const uint32_t hb = pack_bf16(x0, x1);
                w[e] = hb;
                wl[e] = pack_bf16(x0 - __uint_as_float(hb << 16), x1 - __uint_as_float(hb & 0xffff0000u));
            } else {
                w[e] = pack_bf16(__uint_as_float(w[e] << 16) * cexp0, __uint_as_float(w[e] & 0xffff0000u) * cexp0);
            }
        }
        qf[ks] = *reinterpret_cast<bf16x8 *>(&v);
        if (QS == 1 || QS == 3) qfl[ks] = *reinterpret_cast<bf16x8 *>(&vl);
    }
    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
    float mref = 0.f, lsum = 0.f;
    bool fresh = true;

    // TL == 2: this batch's stream is its pair list (signed) or, when that would not be shorter, its full tile list (unsigned)
    const int pair_mode = TL == 2 ? __builtin_amdgcn_readfirstlane(a.pair_info[2 * b + 1]) : 0;
    const uint32_t smask = pair_mode ? 0x80008000u : 0u;        // sign bits of a packed bf16 pair
    const uint32_t lsign = pair_mode ? 0x80000000u : 0u;
    const int n_tiles = pair_mode ? __builtin_amdgcn_readfirstlane(a.pair_info[2 * b]) : a.Nkv / KVB;
    const int tps = (n_tiles + a.nsplit - 1) / a.nsplit;
    const int t0 = sp * tps;
    const int t1 = t0 + tps < n_tiles ? t0 + tps : n_tiles;

    // ---- K/V staging by LDS-DMA (global_load_lds, 16 B per lane, no VGPRs -- registers are what limits occupancy here).
    // A tile's K and V are 8 + 8 pieces of 8 rows x 128 B; wave w moves pieces w, w + NW, ...  The DMA writes lane-linear
    // (lane 8 r + c -> row r, chunk position c of the piece), so position c of row `row` is filled with SOURCE chunk c ^ f(row).
    // Three LDS slots: tile t+2 is requested at the top of tile t into the slot tile t-1 just left; tile t+1 has landed when a
    // counted vmcnt leaves exactly this wave's newest request in flight.
    const int wave_b = __builtin_amdgcn_readfirstlane(b), wave_hk = __builtin_amdgcn_readfirstlane(hk);
    constexpr bool tiled = TL != 0;                          // the tiled key stream of bev_tiles.hip (a.row_src etc.)
    const uint16_t *kbase = a.k + (tiled ? (int64_t)0 : (int64_t)wave_b * a.k_bs) + (int64_t)wave_hk * a.k_hs;
    const uint16_t *vbase = a.v + (tiled ? (int64_t)0 : (int64_t)wave_b * a.v_bs) + (int64_t)wave_hk * a.v_hs;
    const int r8 = lane >> 3, pch = lane & 7;
    constexpr int NPC = (16 + NW - 1) / NW;                    // DMA pieces per wave per tile (K pieces 0..7, V pieces 8..15)
    // per piece: a loop-invariant 32-bit lane offset; the tile base is a scalar 64-bit add (a per-tile 64-bit multiply per
    // lane and piece was ~60 vector issues per tile on a loop that is issue-bound)
    uint32_t doff[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int pc = wid + NW * j, isv = (pc >> 3) & 1, piece = pc & 7, row = piece * 8 + r8;
        const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
        doff[j] = (uint32_t)(row * (isv ? a.ldv : a.ldk) + ((pch ^ f) << 3));
    }
    const int64_t kstep = (int64_t)KVB * a.ldk, vstep = (int64_t)KVB * a.ldv;
    // tiled stream: every key slot of a tile has its own source ROW (bev_tiles.hip: a computed row or a row of the per-model table,
    // both in ONE buffer).  The row words of 8 tiles (512 words) sit in an LDS window, two windows deep, filled by wave 0 with two
    // LDS-DMAs per window (no registers and no vmcnt wait on a VGPR in a loop that lives on counted vmcnt).  A lane reads the
    // words of ITS rows (row r8 of each of its pieces; K and V pieces of the same key rows share a word) right after the previous
    // tile's requests went out; the address is one v_mad_u64_u32 per piece on top of the piece's scalar base.
    constexpr int WIN_T = 8, WIN_W = WIN_T * KVB;              // tiles / words per window
    const int32_t *tsrc_b = !tiled ? nullptr : pair_mode ? a.pair_src + (int64_t)wave_b * a.pair_cap * KVB : a.row_src + (int64_t)wave_b * (a.Nkv / KVB) * KVB;
    int32_t *win = reinterpret_cast<int32_t *>(smem + NSLOT * TILE_E) + 4;      // [2][512] behind the ring and the redo flag
    auto win_dma = [&](int w) __attribute__((always_inline)) {                   // wave 0: window w = tiles t0 + 8 w .. + 7
        const int n_e = n_tiles * KVB;
#pragma unroll
        for (int i = 0; i < WIN_W / 256; ++i) {
            int e = (t0 + WIN_T * w) * KVB + i * 256 + lane * 4;
            e = e > n_e - 4 ? n_e - 4 : e;                                      // words past the stream's end are never used
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tsrc_b + e),
                                             (__attribute__((address_space(3))) void *)(win + (w & 1) * WIN_W + i * 256), 16, 0, 0);
        }
    };
    constexpr int NRS = NW == 4 ? 2 : NPC;                     // distinct row words per lane and tile (4 waves: pieces w, w + 4 for K and for V)
    int rs[NRS];                                               // source rows of the tile of the next dma() call
#pragma unroll
    for (int j = 0; j < NRS; ++j) rs[j] = 0;
    auto win_read = [&](int t) __attribute__((always_inline)) {
        const int r = t - t0;
        const int32_t *wp = win + ((r >> 3) & 1) * WIN_W + (r & (WIN_T - 1)) * KVB + r8;
#pragma unroll
        for (int j = 0; j < NRS; ++j) rs[j] = wp[((wid + NW * j) & 7) * 8];
    };
    const uint32_t ld_t = (uint32_t)a.ldk;                                      // tiled: ldk == ldv < 2^31 (checked by the host)
    // tiled form of doff: the swizzled chunk only (the row comes from the window).  f(row) of row = 8 piece + r8 depends on the
    // piece's parity alone, and all pieces of a wave have the parity of wid (NW is even): ONE lane offset for every piece
    static_assert(NW % 2 == 0, "pieces of a wave share their parity");
    const int f_t = (((r8 >> 1) & 1) << 2) | ((2 * (wid & 1) + (r8 >> 2)) & 3);
    const uint32_t dofft = (uint32_t)((pch ^ f_t) << 3);
    uint64_t lbj[NPC];                                         // per DMA piece of this wave: the K or V base of this head
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int pc = wid + NW * j, isv = (pc >> 3) & 1;
        lbj[j] = reinterpret_cast<uint64_t>(isv ? vbase : kbase);
    }
    // one DMA piece (j-th of this wave) of tile t into `slot`; dma_head / dma_tail are the window bookkeeping around a tile's pieces
    auto dma_head = [&](int t) __attribute__((always_inline)) {
        // tiled: refill the other window once every wave is 4 tiles into the current one (nobody reads the old buffer any more)
        if (tiled && wid == 0 && ((t - t0) & (WIN_T - 1)) == 4) win_dma(((t - t0) >> 3) + 1);
    };
    auto dma_one = [&](int t, int slot, int j) __attribute__((always_inline)) {
        const int pc = wid + NW * j;                           // wave-uniform
        if (pc < 16) {
            const int isv = pc >> 3, piece = pc & 7;
            const uint16_t *src;
            if (tiled) {
                const uint32_t row = (uint32_t)rs[NW == 4 ? (j & 1) : j];
                src = reinterpret_cast<const uint16_t *>(lbj[j]) + ((uint64_t)row * ld_t + dofft);
            } else {
                src = (isv ? vbase + t * vstep : kbase + t * kstep) + doff[j];
            }
            uint16_t *dst = smem + slot * TILE_E + isv * (KVB * KROW) + piece * 512;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    auto dma_tail = [&](int t) __attribute__((always_inline)) {
        if (tiled) win_read(t + 1);                            // in flight until the next tile's pieces (LDS returns in order)
    };
    auto dma = [&](int t, int slot) __attribute__((always_inline)) {
        dma_head(t);
#pragma unroll
        for (int j = 0; j < NPC; ++j) dma_one(t, slot, j);
        dma_tail(t);
    };
    const int my_pieces = (16 - wid + NW - 1) / NW;            // this wave's DMA instructions per tile

    // per-lane LDS offsets (elements).  K A-operand: row l31 (+32 kb), chunk 2 ks + hi.
    const int fk = (((l31 >> 1) & 1) << 2) | ((l31 >> 2) & 3);           // f(row) of rows l31 and l31 + 32
    // V^T A-operand through the transposing read: this lane ADDRESSES row 16 s + 4 hi + q' (+8) and columns
    // 32 dblk + 16 (g16 & 1) + 4 p .. + 3 (q' = (lane & 15) >> 2, p = lane & 3, g16 = lane >> 4) and RECEIVES column
    // 32 dblk + (lane & 31) of the block's four rows
    const int qp = (lane & 15) >> 2, pp = lane & 3, gh = (lane >> 4) & 1;
    auto v_off = [&](int row, int dblk) __attribute__((always_inline)) {   // row inside the 64-row tile
        const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
        const int chunk = dblk * 4 + gh * 2 + (pp >> 1);
        return row * KROW + ((chunk ^ f) << 3) + 4 * (pp & 1);
    };

    auto scores = [&](const uint16_t *Ks, float sinit, f32x16 (&sc)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[kb][i] = sinit;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(Ks + (kb * 32 + l31) * KROW + (((2 * ks + hi) ^ fk) << 3));
                if (QS >= 2) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf), __builtin_bit_cast(f16x8, qf[ks]), sc[kb], 0, 0, 0);
                else         sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sc[kb], 0, 0, 0);
                if (QS == 1) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfl[ks], sc[kb], 0, 0, 0);
                if (QS == 3) sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf), __builtin_bit_cast(f16x8, qfl[ks]), sc[kb], 0, 0, 0);
            }
        }
    };
    // row maximum over the tile's 64 keys: 32 in this lane, 32 in lane ^ 32
    auto row_max = [&](const f32x16 (&sc)[2]) __attribute__((always_inline)) {
        float tmax = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
#pragma unroll
        for (int i = 4; i < 16; i += 2) tmax = fmaxf(fmaxf(tmax, sc[0][i]), sc[0][i + 1]);
#pragma unroll
        for (int i = 0; i < 16; i += 2) tmax = fmaxf(fmaxf(tmax, sc[1][i]), sc[1][i + 1]);
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    // P = exp2(scores), row sums, O^T += V^T P^T.  Per 32-key block: the eight transposed V reads of its two 16-key steps are issued
    // first (they land under the 16 exps), then each step waits only for its own four.
    auto finish = [&](const uint16_t *Vs, const f32x16 (&sc)[2]) __attribute__((always_inline)) {
        const uint32_t vb = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t *)Vs);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int r0 = kb * 32 + 4 * hi + qp;
            bf16x4 a00 = lds_tr_read(vb + 2 * v_off(r0, 0)), a01 = lds_tr_read(vb + 2 * v_off(r0 + 8, 0));
            bf16x4 a10 = lds_tr_read(vb + 2 * v_off(r0, 1)), a11 = lds_tr_read(vb + 2 * v_off(r0 + 8, 1));
            bf16x4 b00 = lds_tr_read(vb + 2 * v_off(r0 + 16, 0)), b01 = lds_tr_read(vb + 2 * v_off(r0 + 24, 0));
            bf16x4 b10 = lds_tr_read(vb + 2 * v_off(r0 + 16, 1)), b11 = lds_tr_read(vb + 2 * v_off(r0 + 24, 1));
            float p[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) p[i] = fast_exp2(sc[kb][i]);
            const float s0 = (p[0] + p[1]) + (p[2] + p[3]), s1 = (p[4] + p[5]) + (p[6] + p[7]);
            const float s2 = (p[8] + p[9]) + (p[10] + p[11]), s3 = (p[12] + p[13]) + (p[14] + p[15]);
            lsum += (s0 + s1) + (s2 + s3);
            uint4 u0 = make_uint4(pack_bf16(p[0], p[1]), pack_bf16(p[2], p[3]), pack_bf16(p[4], p[5]), pack_bf16(p[6], p[7]));
            uint4 u1 = make_uint4(pack_bf16(p[8], p[9]), pack_bf16(p[10], p[11]), pack_bf16(p[12], p[13]), pack_bf16(p[14], p[15]));
            const bf16x8 pf0 = *reinterpret_cast<bf16x8 *>(&u0), pf1 = *reinterpret_cast<bf16x8 *>(&u1);
            lds_tr_wait4(a00, a01, a10, a11);
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(a00, a01, 0, 1, 2, 3, 4, 5, 6, 7), pf0, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(a10, a11, 0, 1, 2, 3, 4, 5, 6, 7), pf0, o[1], 0, 0, 0);
            lds_tr_wait(b00, b01, b10, b11);
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(b00, b01, 0, 1, 2, 3, 4, 5, 6, 7), pf1, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(b10, b11, 0, 1, 2, 3, 4, 5, 6, 7), pf1, o[1], 0, 0, 0);
        }
    };

    // steady-state tile: one 32-key block at a time (16 score registers live instead of 32).  The 16 transposed-read addresses of
    // a tile are 4 per-lane bases + immediates: rows are rb + 8 j (rb = 4 hi + q'), the swizzle key f(row) only depends on j's
    // parity (f_odd = f_even ^ 2) and the d-block flips chunk bit 2, so {f_even, f_odd} x {dblk 0, 1} are the only distinct lane terms.
    const int rb = 4 * hi + qp;
    const int f_even = (((rb >> 1) & 1) << 2) | ((rb >> 2) & 3);
    const int ch0 = gh * 2 + (pp >> 1);
    uint32_t vlane[2][2];                                       // [row parity][dblk] byte offsets inside a V tile
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int d = 0; d < 2; ++d)
            vlane[par][d] = (uint32_t)(2 * (rb * KROW + (((d * 4 + ch0) ^ (f_even ^ (par << 1))) << 3) + 4 * (pp & 1)));
    auto tile_fast = [&](const uint16_t *Ks, const uint16_t *Vs) __attribute__((always_inline)) {
        const uint32_t vb = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t *)Vs);
        const uint32_t e0 = vb + vlane[0][0], e1 = vb + vlane[0][1], o0 = vb + vlane[1][0], o1 = vb + vlane[1][1];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            // scores straight from a ZERO accumulator (an inline constant of the MFMA: no 16 v_mov per block): the fast stream uses
            // the fixed softmax reference 0, i.e. P = 2^score
            f32x16 sc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(Ks + (kb * 32 + l31) * KROW + (((2 * ks + hi) ^ fk) << 3));
                if (QS >= 2) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf), __builtin_bit_cast(f16x8, qf[ks]), sc, 0, 0, 0);
                else         sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sc, 0, 0, 0);
                if (QS == 1) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfl[ks], sc, 0, 0, 0);
                if (QS == 3) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf), __builtin_bit_cast(f16x8, qfl[ks]), sc, 0, 0, 0);
            }
            // rows rb + {0, 8} (step 0) and rb + {16, 24} (step 1) of this 32-key block: byte immediates (32 kb + 8 j) * 128
            bf16x4 a00, a01, a10, a11, b00, b01, b10, b11;
            if (kb == 0) {
                a00 = lds_tr_read_o<0>(e0);        a01 = lds_tr_read_o<1024>(o0);  a10 = lds_tr_read_o<0>(e1);        a11 = lds_tr_read_o<1024>(o1);
                b00 = lds_tr_read_o<2048>(e0);     b01 = lds_tr_read_o<3072>(o0);  b10 = lds_tr_read_o<2048>(e1);     b11 = lds_tr_read_o<3072>(o1);
            } else {
                a00 = lds_tr_read_o<4096>(e0);     a01 = lds_tr_read_o<5120>(o0);  a10 = lds_tr_read_o<4096>(e1);     a11 = lds_tr_read_o<5120>(o1);
                b00 = lds_tr_read_o<6144>(e0);     b01 = lds_tr_read_o<7168>(o0);  b10 = lds_tr_read_o<6144>(e1);     b11 = lds_tr_read_o<7168>(o1);
            }
            uint32_t pk[8];
            float ls = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x0 = fast_exp2(sc[2 * e]), x1 = fast_exp2(sc[2 * e + 1]);
                ls += x0 + x1;
                pk[e] = pack_bf16(x0, x1);
                if (TL == 2 && kb == 1) pk[e] ^= smask;        // -P for the subtracted half of a pair tile
            }
            lsum += (TL == 2 && kb == 1) ? __uint_as_float(__float_as_uint(ls) ^ lsign) : ls;   // the subtracted half's row sum
            uint4 u0 = make_uint4(pk[0], pk[1], pk[2], pk[3]), u1 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            const bf16x8 pf0 = *reinterpret_cast<bf16x8 *>(&u0), pf1 = *reinterpret_cast<bf16x8 *>(&u1);
            lds_tr_wait4(a00, a01, a10, a11);
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(a00, a01, 0, 1, 2, 3, 4, 5, 6, 7), pf0, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(a10, a11, 0, 1, 2, 3, 4, 5, 6, 7), pf0, o[1], 0, 0, 0);
            lds_tr_wait(b00, b01, b10, b11);
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(b00, b01, 0, 1, 2, 3, 4, 5, 6, 7), pf1, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(b10, b11, 0, 1, 2, 3, 4, 5, 6, 7), pf1, o[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- PIPE: the pipelined fast stream -------------------------------------------------------------------------------------
    // One tile = block A (keys 0..31) and block B (keys 32..63).  Per iteration t, with S_A(t) already in sc0:
    //   P1  S_B(t) -> sc1 (8 MFMAs; 4 without the Q split), between them exp / pack of sc0          [V_A(t) fragments requested]
    //   B1  tile t+1 has landed (counted vmcnt, barrier); tile t+3 requested into the slot of t-1;   [K_A(t+1) fragments requested]
    //   P2  O += V_A P_A (4 MFMAs), between them the row-sum chain of block A
    //   P3  S_A(t+1) -> sc0, between them exp / pack of sc1                                           [V_B(t) fragments requested]
    //   P4  O += V_B P_B, row-sum chain of block B                                                    [K_B(t+1) fragments requested]
    // Every MFMA is followed by at most ~24 cycles of vector issue (MI355X: an MFMA holds the issue port for 8 of its 32 cycles) and a
    // sched_barrier pins that placement.  All LDS fragment reads are asm (invisible to the compiler's waitcnt pass, see lds_tr_read)
    // and are waited for with lgkmcnt(0) one phase (>= 128 cycles) after their issue.  Arithmetic and summation order are those of
    // tile_fast: the two forms are bit-identical.
    [[maybe_unused]] auto stream_pipe = [&]() __attribute__((always_inline)) {
        if (!(t0 < t1)) return;
        constexpr int NM = (QS == 1 || QS == 3) ? 8 : 4;           // score MFMAs per 32-key block
        // softmax groups (2 scores each) of the finished block per score-MFMA gap: group g runs in gap m with slot_lo(m) <= g < slot_lo(m + 1)
        // (NM == 8: one per gap; NM == 4: 2, 1, 2, 1 -- the last two groups run in the PV gaps, see pv)
        auto slot_lo = [](int m) { return NM == 8 ? m : (m * 3 + 1) / 2; };
        const uint32_t sbase = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint16_t *)smem);
        uint32_t kl[4];                                            // K fragment byte offsets inside a slot (block A; block B: + 4096)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kl[ks] = (uint32_t)(2 * (l31 * KROW + (((2 * ks + hi) ^ fk) << 3)));
        bf16x8 kf[4];
        bf16x4 va[8];
        auto k_reads = [&](uint32_t sb, auto kb_tag) __attribute__((always_inline)) { lds_k_read4<decltype(kb_tag)::value * 4096>(kf, sb, kl); };
        auto v_reads = [&](uint32_t sb, auto kb_tag) __attribute__((always_inline)) {
            lds_v_read8<2 * KVB * KROW + decltype(kb_tag)::value * 4096>(va, sb + vlane[0][0], sb + vlane[0][1], sb + vlane[1][0], sb + vlane[1][1]);
        };
        auto wait_k = [&]() __attribute__((always_inline)) { lds_wait_k(kf); };
        auto wait_v = [&]() __attribute__((always_inline)) { lds_wait_v(va); };
        auto s_mfma = [&](f32x16 &sc, int m) __attribute__((always_inline)) {     // m-th score MFMA of a block (m compile-time after unrolling)
            const int ks = NM == 8 ? (m >> 1) : m;
            const bool lo = NM == 8 && (m & 1);
            if (QS >= 2) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf[ks]), __builtin_bit_cast(f16x8, lo ? qfl[QS == 3 ? ks : 0] : qf[ks]), sc, 0, 0, 0);
            else         sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], lo ? qfl[QS == 1 ? ks : 0] : qf[ks], sc, 0, 0, 0);
        };
        uint32_t pk[8];
        float tg[8], xa[2], xb[2];
        // exp of the two scores of group g of a finished block (soft_e), their pair sum and packed bf16 pair one gap later (soft_f: a
        // transcendental's result costs a wait state when the very next vector instruction reads it).  neg: the pair goes into a
        // SUBTRACTED block -- source modifiers of the conversion, no extra instruction
        auto soft_e = [&](const f32x16 &sc, int g) __attribute__((always_inline)) {
            xa[g & 1] = fast_exp2(sc[2 * g]);
            xb[g & 1] = fast_exp2(sc[2 * g + 1]);
            asm volatile("" : "+v"(xa[g & 1]), "+v"(xb[g & 1]));   // computed HERE (see soft_f)
        };
        auto soft_f = [&](int g, auto neg_tag) __attribute__((always_inline)) {
            const float x0 = xa[g & 1], x1 = xb[g & 1];
            tg[g] = x0 + x1;
            pk[g] = decltype(neg_tag)::value ? pack_bf16(-x0, -x1) : pack_bf16(x0, x1);
            asm volatile("" : "+v"(tg[g]), "+v"(pk[g]));           // computed HERE: the IR sinks unpinned values past B1 to their first use
        };
        // O += V P of one block with the row-sum chain in the gaps (the last group's soft_f first)
        // (single-pass score forms, NM == 4: a block has 4 score MFMAs for the same 16 exponentials, so the last two groups of a block's
        // softmax run in the first two PV gaps -- the second half of P, pf1, is only needed by the third PV MFMA)
        auto pv = [&](auto neg_tag, auto &&after_second, const f32x16 &sc) __attribute__((always_inline)) {
            constexpr bool NEG = decltype(neg_tag)::value;
            if (NM == 8) soft_f(7, neg_tag);
            uint4 u0 = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            const bf16x8 pf0 = *reinterpret_cast<bf16x8 *>(&u0);
            float ls = 0.f;
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(va[0], va[1], 0, 1, 2, 3, 4, 5, 6, 7), pf0, o[0], 0, 0, 0);
            ls += tg[0]; ls += tg[1];
            if (NM == 4) { soft_e(sc, 6); soft_f(5, neg_tag); }
            __builtin_amdgcn_sched_barrier(0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(va[2], va[3], 0, 1, 2, 3, 4, 5, 6, 7), pf0, o[1], 0, 0, 0);
            ls += tg[2]; ls += tg[3];
            if (NM == 4) { soft_e(sc, 7); soft_f(6, neg_tag); }
            after_second();
            __builtin_amdgcn_sched_barrier(0);
            if (NM == 4) soft_f(7, neg_tag);
            uint4 u1 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            const bf16x8 pf1 = *reinterpret_cast<bf16x8 *>(&u1);
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(va[4], va[5], 0, 1, 2, 3, 4, 5, 6, 7), pf1, o[0], 0, 0, 0);
            ls += tg[4]; ls += tg[5];
            __builtin_amdgcn_sched_barrier(0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(va[6], va[7], 0, 1, 2, 3, 4, 5, 6, 7), pf1, o[1], 0, 0, 0);
            ls += tg[6]; ls += tg[7];
            lsum = NEG ? lsum - ls : lsum + ls;
            __builtin_amdgcn_sched_barrier(0);
        };
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 sc0, sc1;

        // ---- prologue: window, tiles t0 .. t0+2 requested, S_A(t0)
        if (tiled) {
            if (wid == 0) win_dma(0);                              // window 1 follows at tile t0 + 4
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            win_read(t0);
        }
        dma(t0, 0);
        if (t0 + 1 < t1) dma(t0 + 1, 1);
        if (t0 + 2 < t1) dma(t0 + 2, 2);
        if (t0 + 2 < t1)      __builtin_amdgcn_s_waitcnt(8 | 0x70);         // tile t0 landed, two requests (4 pieces each) may fly
        else if (t0 + 1 < t1) __builtin_amdgcn_s_waitcnt(4 | 0x70);
        else                  __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        if (q0 >= a.Nq) {
            // a wave without queries (the padding of the last query tile: 2 of 20 waves at 576 queries) only keeps its share of the
            // K|V stream moving: the same waits, barriers and requests as the loop below, no matrix or vector work
            for (int t = t0; t + 1 < t1; ++t) {
                if (t + 2 < t1) __builtin_amdgcn_s_waitcnt(4 | 0x70);
                else            __builtin_amdgcn_s_waitcnt(0x0070);
                __builtin_amdgcn_s_barrier();
                if (t + 3 < t1) dma(t + 3, (t - t0 + 3) & 3);
            }
            return;
        }
        k_reads(sbase, std::integral_constant<int, 0>{});
        wait_k();
        sc0 = zero16;
#pragma unroll
        for (int m = 0; m < NM; ++m) s_mfma(sc0, m);
        k_reads(sbase, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);

        auto body = [&](int t, auto more_tag, auto neg_tag) __attribute__((always_inline)) {
            constexpr bool MORE = decltype(more_tag)::value;       // a tile t+1 exists
            const int sl = (t - t0) & 3;
            const uint32_t sb = sbase + (uint32_t)sl * (uint32_t)(TILE_E * 2);
            const uint32_t sb1 = sbase + (uint32_t)((sl + 1) & 3) * (uint32_t)(TILE_E * 2);
            // P1
            wait_k();                                              // K_B(t)
            v_reads(sb, std::integral_constant<int, 0>{});         // V_A(t)
            sc1 = zero16;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                s_mfma(sc1, m);
#pragma unroll
                for (int g = slot_lo(m); g < slot_lo(m + 1); ++g) {
                    soft_e(sc0, g);
                    if (g > 0) soft_f(g - 1, std::false_type{});
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            wait_v();
            if (MORE) {
                // B1
                if (t + 2 < t1) __builtin_amdgcn_s_waitcnt(4 | 0x70);        // tile t+1 landed, the request for t+2 may fly
                else            __builtin_amdgcn_s_waitcnt(0x0070);
                __builtin_amdgcn_s_barrier();
                // tile t+3 into the slot of t-1, one piece at a time over P2 / P3: four back-to-back requests per wave right after a
                // barrier (16 per workgroup at once) queue up in front of the texture path and hold the waves that issue them
                if (t + 3 < t1) { dma_head(t + 3); dma_one(t + 3, (sl + 3) & 3, 0); }
                k_reads(sb1, std::integral_constant<int, 0>{});    // K_A(t+1)
                __builtin_amdgcn_sched_barrier(0);
            }
            // P2
            pv(std::false_type{}, [&]() __attribute__((always_inline)) { if (MORE && t + 3 < t1) dma_one(t + 3, (sl + 3) & 3, 1); }, sc0);
            // P3
            if (MORE) wait_k();
            v_reads(sb, std::integral_constant<int, 1>{});         // V_B(t)
            if (MORE) sc0 = zero16;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (MORE) s_mfma(sc0, m);
#pragma unroll
                for (int g = slot_lo(m); g < slot_lo(m + 1); ++g) {
                    soft_e(sc1, g);
                    if (g > 0) soft_f(g - 1, neg_tag);
                }
                if (MORE && m == NM / 4 && t + 3 < t1) dma_one(t + 3, (sl + 3) & 3, 2);
                if (MORE && m == NM / 4 + NM / 2 && t + 3 < t1) { dma_one(t + 3, (sl + 3) & 3, 3); dma_tail(t + 3); }
                __builtin_amdgcn_sched_barrier(0);
            }
            wait_v();
            if (MORE) k_reads(sb1, std::integral_constant<int, 1>{});   // K_B(t+1)
            // P4
            pv(neg_tag, []() __attribute__((always_inline)) {}, sc1);
        };
        int t = t0;
        if (TL == 2 && pair_mode) {                                // keys 32..63 of every tile are subtracted
            for (; t + 1 < t1; ++t) body(t, std::true_type{}, std::true_type{});
            body(t, std::false_type{}, std::true_type{});
        } else {
            for (; t + 1 < t1; ++t) body(t, std::true_type{}, std::false_type{});
            body(t, std::false_type{}, std::false_type{});
        }
    };

    // DMA bookkeeping shared by the three loop forms below
    int slot = 0;
    auto pre = [&](int t) __attribute__((always_inline)) {
        const int s2 = slot == 0 ? 2 : slot - 1;               // (slot + 2) % 3: the slot tile t-1 left at the last barrier
        if (t + 2 < t1) dma(t + 2, s2);
    };
    auto post = [&](int t) __attribute__((always_inline)) {
        // tile t+1 landed; this wave's request for tile t+2 (if any) stays in flight across the barrier
        if (t + 2 < t1) {
            if (my_pieces == NPC) __builtin_amdgcn_s_waitcnt((NPC & 15) | 0x70 | ((NPC >> 4) << 14));
            else                  __builtin_amdgcn_s_waitcnt(((NPC - 1) & 15) | 0x70);
        } else {
            __builtin_amdgcn_s_waitcnt(0x0070);
        }
        __builtin_amdgcn_s_barrier();      // raw: __syncthreads() would drain vmcnt(0), i.e. wait for the request just made
        slot = slot == 2 ? 0 : slot + 1;
    };

    // The fast stream uses the FIXED softmax reference 0 (P = 2^score in the log2 domain): fp32 O / l and bf16 P keep their
    // relative precision whatever the exponent, so the only hazards are overflow (a score above ~100) and total underflow (every
    // score of a row below ~-100) -- both detected at the end (row sum outside [2^-100, 2^100] or not finite) and answered by
    // running the stream again in the classic form (first tile sets the reference, maximum tracked, O rescaled).  The steady-state
    // loop therefore has no row maximum, no branch, no rescale and no accumulator initialisation: a rescale inside it, even behind
    // a never-taken branch, made the compiler copy the 32 accumulator registers every tile, and the 17-step v_max3 chain sat
    // between the score MFMAs and the first exp.
    uint32_t *redo_flag = reinterpret_cast<uint32_t *>(smem + NSLOT * TILE_E);
    auto stream = [&](auto slow_tag) __attribute__((always_inline)) {
        constexpr bool SLOW = decltype(slow_tag)::value;
        if constexpr (PIPE && !SLOW) { stream_pipe(); return; }
        slot = 0;
        if (tiled && t0 < t1) {                                // (an empty range reads nothing: a pair list may have no tiles at all)
            if (wid == 0) win_dma(0);                          // window 1 follows at tile t0 + 4
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            win_read(t0);
        }
        if (t0 < t1) dma(t0, 0);
        if (t0 + 1 < t1) dma(t0 + 1, 1);
        if (t0 + 1 < t1) {                                     // tile t0 landed, tile t0+1 may still fly
            if (my_pieces == NPC) __builtin_amdgcn_s_waitcnt((NPC & 15) | 0x70 | ((NPC >> 4) << 14));
            else                  __builtin_amdgcn_s_waitcnt(((NPC - 1) & 15) | 0x70);
        } else {
            __builtin_amdgcn_s_waitcnt(0x0070);
        }
        __builtin_amdgcn_s_barrier();
        int t = t0;
        if (SLOW && t < t1) {                                  // classic form, first tile: absolute scores set the reference
            f32x16 sc[2];
            pre(t);
            scores(smem, 0.f, sc);
            mref = row_max(sc);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) sc[kb][i] -= mref;
            finish(smem + KVB * KROW, sc);
            post(t);
            fresh = false;
            ++t;
        }
        for (; t < t1; ++t) {
            pre(t);
            if (!SLOW) {
                tile_fast(smem + slot * TILE_E, smem + slot * TILE_E + KVB * KROW);
                post(t);
                continue;
            }
            f32x16 sc[2];
            scores(smem + slot * TILE_E, -mref, sc);
            if (SLOW) {
                const float delta = fmaxf(row_max(sc), 0.f);
                if (__any(delta > 0.f)) {
                    const float al = fast_exp2(-delta);
                    mref += delta;
                    lsum *= al;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int i = 0; i < 16; ++i) sc[kb][i] -= delta;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int i = 0; i < 16; ++i) o[d][i] *= al;
                }
            }
            finish(smem + slot * TILE_E + KVB * KROW, sc);
            post(t);
        }
    };
    if (tid == 0) *redo_flag = 0;
    stream(std::false_type{});
    fresh = !(t0 < t1);
    if (t0 < t1 && !pair_mode && !(PIPE && q0 >= a.Nq)) {       // a signed partial may be anything; the combine kernel judges the total
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
        const float ltot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (!(ltot < 1.2676506e30f) || !(ltot > 7.8886091e-31f)) *redo_flag = 1;      // outside [2^-100, 2^100]; also catches inf / NaN
    }
    __syncthreads();
    if (*redo_flag) {                                          // workgroup-uniform
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
        lsum = 0.f; mref = 0.f; fresh = true;
        stream(std::true_type{});
    }

    // ---- write back: lane (query, hi) holds O[qi][32 dblk + 8 (i>>2) + 4 hi + (i&3)]; the row sum is split over hi ----
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
        lsum = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (qi >= a.Nq) return;
    if (a.nsplit > 1 || a.force_part) {
        float *pr = a.part + ((((int64_t)b * a.H + h) * a.nsplit + sp) * a.Nq + qi) * (a.dh + 2);
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                *reinterpret_cast<f32x4 *>(pr + 32 * d + 8 * i4 + 4 * hi) = f32x4{o[d][4 * i4], o[d][4 * i4 + 1], o[d][4 * i4 + 2], o[d][4 * i4 + 3]};
        if (hi == 0) { pr[a.dh] = fresh ? -INFINITY : mref; pr[a.dh + 1] = lsum; }
        return;
    }
    const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
    uint16_t *dst = a.o + (int64_t)b * a.o_bs + (int64_t)qi * a.ldo + (int64_t)h * a.o_hs;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const float y0 = o[d][4 * i4] * inv, y1 = o[d][4 * i4 + 1] * inv, y2 = o[d][4 * i4 + 2] * inv, y3 = o[d][4 * i4 + 3] * inv;
            const uint2 hv = make_uint2(pack_bf16(y0, y1), pack_bf16(y2, y3));
            *reinterpret_cast<uint2 *>(dst + 32 * d + 8 * i4 + 4 * hi) = hv;
            if (a.ol) {
                const uint2 lv = make_uint2(pack_bf16(y0 - __uint_as_float(hv.x << 16), y1 - __uint_as_float(hv.x & 0xffff0000u)),
                                            pack_bf16(y2 - __uint_as_float(hv.y << 16), y3 - __uint_as_float(hv.y & 0xffff0000u)));
                *reinterpret_cast<uint2 *>(a.ol + (dst - a.o) + 32 * d + 8 * i4 + 4 * hi) = lv;
            }
        }
}

// Measured dead end: a bf16x3 form of k_attn32 (hi + lo operands, three MFMAs per product, two 32-KiB LDS slots) is correct
// but needs 246 VGPRs (two sets of Q fragments, hi/lo P and V fragments): two waves per SIMD, i.e. one 6-wave workgroup per
// CU, and 8.4 ms against 6.6 ms for the 12-wave 16x16 bf16x3 kernel; capped at 168 VGPRs it spills 224 bytes.
// Measured dead end: a short-K/V companion of k_attn32 (all of K/V of one head staged once in LDS, every wave walking its own
// 32-query tiles with no barrier -- the shape of the cross-attention onto 196 image patches) was correct but slower than the
// 4-wave 16x16 form on the headline shape (65 vs 58 us): with ~2.5 waves per SIMD in a single dispatch round the time is one
// wave's serial latency (~4000 cycles per 32x64 tile step), not issue throughput; the 16-query form has 4x more, shorter waves.
// merge the KV splits: out = sum_s 2^(m_s - M) O_s / sum_s 2^(m_s - M) l_s
__global__ void __launch_bounds__(256) k_attn_combine(AttnArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int dq = a.dh / 4;
    const int64_t total = (int64_t)a.B * a.H * a.Nq * dq;
    if (idx >= total) return;
    const int d0 = (int)(idx % dq) * 4;
    const int64_t row = idx / dq;                    // (b*H + h)*Nq + qi
    const int qi = (int)(row % a.Nq);
    const int64_t bh = row / a.Nq;
    const int h = (int)(bh % a.H), b = (int)(bh / a.H);
    const int stride = a.dh + 2;
    float M = -INFINITY;
    for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, a.part[((bh * a.nsplit + s) * a.Nq + qi) * stride + a.dh]);
    const float Ms = (M == -INFINITY) ? 0.f : M;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, l = 0.f;
    for (int s = 0; s < a.nsplit; ++s) {
        const float *pr = a.part + ((bh * a.nsplit + s) * a.Nq + qi) * stride;
        const float w = exp2f(pr[a.dh] - Ms);
        l += w * pr[a.dh + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += w * pr[d0 + r];
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = acc[r] * inv;
    const int64_t off = (int64_t)b * a.o_bs + (int64_t)qi * a.ldo + (int64_t)h * a.o_hs + d0;
    uint2 hv = make_uint2(pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]));
    *reinterpret_cast<uint2 *>(a.o + off) = hv;
    if (a.ol) {
        uint2 lv = make_uint2(pack_bf16(y[0] - __uint_as_float(hv.x << 16), y[1] - __uint_as_float(hv.x & 0xffff0000u)),
                              pack_bf16(y[2] - __uint_as_float(hv.y << 16), y[3] - __uint_as_float(hv.y & 0xffff0000u)));
        *reinterpret_cast<uint2 *>(a.ol + off) = lv;
    }
}

// Signed pair stream (k_attn32<., ., 2>): merge the KV splits of (b, h), add the per-model totals when the batch ran its pair list,
// normalise -- and JUDGE the result: a signed sum is only as good as what is left after the subtraction.  With T the total row sum
// over all table keys and l the final one, fp32 accumulation error is ~2^-20 T, so l < T / 16 (or a non-finite / non-positive l)
// flags (b, h) for the predicated full re-run.  With pred set this is that re-run's merge (rows of unflagged (b, h) are left alone).
constexpr int STAT_SLOTS = 256;        // same-address atomics serialise (and device-scope loads of one address do too: every XCD's L2 is bypassed)
__global__ void __launch_bounds__(256) k_attn_combine_signed(AttnArgs a) {
    __shared__ int s_over[4];
    __shared__ unsigned int s_bits[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (a.pred && a.stats && a.stat_slots && blockIdx.x == 0) {
        // the re-run's merge doubles as the last step of the guard statistics: the first launch left one (count, max) pair per slot
        int over = a.stat_slots[threadIdx.x * 64];
        unsigned int bits = (unsigned int)a.stat_slots[threadIdx.x * 64 + 1];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            over += __shfl_xor(over, o);
            const unsigned int t = (unsigned int)__shfl_xor((int)bits, o);
            bits = bits > t ? bits : t;
        }
        if (lane == 0) { s_over[wid] = over; s_bits[wid] = bits; }
        __syncthreads();
        if (threadIdx.x == 0) {
            over = s_over[0] + s_over[1] + s_over[2] + s_over[3];
            bits = max(max(s_bits[0], s_bits[1]), max(s_bits[2], s_bits[3]));
            if (over) atomicAdd(&a.stats[0], over);
            if (bits) atomicMax(reinterpret_cast<unsigned int *>(&a.stats[2]), bits);
        }
    }
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int dq = a.dh / 4;
    const int64_t total = (int64_t)a.B * a.H * a.Nq * dq;
    const bool slots = !a.pred && a.stat_slots != nullptr;       // (uniform) every lane stays to the block reduction below; lanes past the end compute a copy, store nothing
    const int64_t idc = idx < total ? idx : total - 1;
    const int d0 = (int)(idc % dq) * 4;
    const int64_t row = idc / dq;                    // (b*H + h)*Nq + qi
    const int qi = (int)(row % a.Nq);
    const int64_t bh = row / a.Nq;
    const int h = (int)(bh % a.H), b = (int)(bh / a.H);
    const bool live = idx < total && !(a.pred && a.pred[bh] == 0);
    if (!live && !slots) return;                                // (the predicated re-run's merge: rows of unflagged (b, h) leave at once)
    const int stride = a.dh + 2;
    const bool signed_b = !a.pred && a.pair_info && a.pair_info[2 * b + 1] != 0;
    const float *tot = signed_b ? a.totals + ((int64_t)h * a.Nq + qi) * stride : nullptr;
    float M = tot ? tot[a.dh] : -INFINITY;
    for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, a.part[((bh * a.nsplit + s) * a.Nq + qi) * stride + a.dh]);
    const float Ms = (M == -INFINITY) ? 0.f : M;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, l = 0.f, lt = 0.f;
    if (tot) {
        const float w = exp2f(tot[a.dh] - Ms);
        lt = w * tot[a.dh + 1];
        l = lt;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = w * tot[d0 + r];
    }
    for (int s = 0; s < a.nsplit; ++s) {
        const float *pr = a.part + ((bh * a.nsplit + s) * a.Nq + qi) * stride;
        const float w = exp2f(pr[a.dh] - Ms);
        l += w * pr[a.dh + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += w * pr[d0 + r];
    }
    const bool judge = live && signed_b && d0 == 0;
    if (judge && a.flags && !(l > 0.0625f * lt && l < 3.0e38f)) {
        a.flags[bh] = 1;
        if (a.stats && qi == 0) atomicAdd(&a.stats[1], 1);
    }
    // per-launch half of the plain-stream guard (lvq_stream_guard is the per-model half): rows whose softmax mass sits mostly on the
    // scene's own (dirty) keys, which the per-model statistic has not seen -> a count and the largest l / l_table.  One atomic per row on
    // one address serialised 221 000 of them (0.8 ms); reading the address first was worse (1.2 ms).  So: reduce over the workgroup, one
    // atomic pair per workgroup into one of STAT_SLOTS lines, and the re-run's merge (above) folds the slots into stats.
    if (slots) {
        float ratio = (judge && lt > 0.f) ? fminf(l / lt, 3.0e38f) : 0.f;
        int over = (judge && lt > 0.f && l > 1.5f * lt) ? 1 : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            ratio = fmaxf(ratio, __shfl_xor(ratio, o));
            over += __shfl_xor(over, o);
        }
        if (lane == 0) { s_over[wid] = over; s_bits[wid] = __float_as_uint(ratio); }      // non-negative floats order like their bits
        __syncthreads();
        if (threadIdx.x == 0) {
            over = s_over[0] + s_over[1] + s_over[2] + s_over[3];
            const unsigned int bits = max(max(s_bits[0], s_bits[1]), max(s_bits[2], s_bits[3]));
            int32_t *slot = a.stat_slots + (blockIdx.x & (STAT_SLOTS - 1)) * 64;
            if (over) atomicAdd(&slot[0], over);
            if (bits) atomicMax(reinterpret_cast<unsigned int *>(&slot[1]), bits);
        }
    }
    if (!live) return;
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = acc[r] * inv;
    const int64_t off = (int64_t)b * a.o_bs + (int64_t)qi * a.ldo + (int64_t)h * a.o_hs + d0;
    uint2 hv = make_uint2(pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]));
    *reinterpret_cast<uint2 *>(a.o + off) = hv;
    if (a.ol) {
        uint2 lv = make_uint2(pack_bf16(y[0] - __uint_as_float(hv.x << 16), y[1] - __uint_as_float(hv.x & 0xffff0000u)),
                              pack_bf16(y[2] - __uint_as_float(hv.y << 16), y[3] - __uint_as_float(hv.y & 0xffff0000u)));
        *reinterpret_cast<uint2 *>(a.ol + off) = lv;
    }
}

// merge the KV splits of a B = 1 run WITHOUT normalising: totals[h][qi] = (sum_s w_s O_s | M | sum_s w_s l_s)
__global__ void __launch_bounds__(256) k_attn_combine_raw(AttnArgs a, float *__restrict__ totals) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int dq = a.dh / 4;
    const int64_t total = (int64_t)a.B * a.H * a.Nq * dq;
    if (idx >= total) return;
    const int d0 = (int)(idx % dq) * 4;
    const int64_t row = idx / dq;
    const int qi = (int)(row % a.Nq);
    const int64_t bh = row / a.Nq;
    const int stride = a.dh + 2;
    float M = -INFINITY;
    for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, a.part[((bh * a.nsplit + s) * a.Nq + qi) * stride + a.dh]);
    const float Ms = (M == -INFINITY) ? 0.f : M;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, l = 0.f;
    for (int s = 0; s < a.nsplit; ++s) {
        const float *pr = a.part + ((bh * a.nsplit + s) * a.Nq + qi) * stride;
        const float w = exp2f(pr[a.dh] - Ms);
        l += w * pr[a.dh + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += w * pr[d0 + r];
    }
    float *dst = totals + (bh * a.Nq + qi) * stride;
    *reinterpret_cast<f32x4 *>(dst + d0) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    if (d0 == 0) { dst[a.dh] = Ms; dst[a.dh + 1] = l; }
}

// launch geometry shared by lvq_attention_workspace_bytes and lvq_attention_bf16
struct AttnPlan { int qt, nw, nqt, nsplit, k32; };
// k_attn32 geometry (0 = not applicable): 32 queries per wave, 4 or 6 waves, at most 1/8 of the query slots padding
int plan_k32_waves(int nq, int nkv, int dh, bool split) {      // split = K / V carry lo parts (full bf16x3): not this kernel
    if (split || dh != 64 || nkv < 4096 || (nkv % KVB) != 0 || lvq_tune().attn_no32) return 0;
    const int force = lvq_tune().attn32_nw;
    for (int nw : {4, 6}) {       // measured on 576 x 262144: 4 waves 2.29 ms, 6 waves 2.63, 3 waves 2.71 (three workgroups per CU at 4)
        if (force && nw != force) continue;
        const int64_t tile = 32 * nw, padded = (nq + tile - 1) / tile * tile;
        if ((padded - nq) * 8 <= nq) return nw;
    }
    return 0;
}
// the pipelined form of k_attn32 (2 workgroups per CU) is the default for 4 waves and a split (hi + lo) or single-fp16 query: bit-identical
// to the other form, same-box +4-10 % on the signed stream and +2.5 % on the bench step (mixed16: +1.4 % once the last two softmax groups
// of a block moved into the PV gaps).  With a plain bf16 query (4 score MFMAs per block for the same 16 exponentials, no second pass to
// hide behind) the two forms measure within 2 % of each other either way and the 3-waves-per-SIMD form stays.  LVQ_ATTN_NO_PIPE=1 / LVQ_ATTN_PIPE=1 force one form for every
// query kind (A/B: tools/ab_attn_pipe.py; DESIGN 3.2 has the measurements)
bool k32_pipe(int nw, int qs) {
    if (nw != 4 || lvq_tune().attn_pipe < 0) return false;
    return qs != 0 || lvq_tune().attn_pipe > 0;
}
AttnPlan plan_attn(int batch, int n_heads, int nq, int nkv, int dh, bool split, bool allow32 = false, int qs = 0) {
    AttnPlan p;
    p.k32 = allow32 ? plan_k32_waves(nq, nkv, dh, split) : 0;
    if (p.k32) {
        p.qt = 2; p.nw = p.k32;
        p.nqt = (nq + 32 * p.nw - 1) / (32 * p.nw);
        const int64_t base = (int64_t)p.nqt * n_heads * batch;
        const int n_tiles = nkv / KVB;
        int ns = (int)((4096 + base - 1) / base);
        if (ns > n_tiles / 8) ns = n_tiles / 8;
        if (ns < 1) ns = 1;
        if (ns > 64) ns = 64;
        {
            // dispatch-round quantisation: 3 (4-wave) or 2 (6-wave) workgroups fit per CU; among the split counts near the
            // target pick the one whose last round is fullest (576 queries, 4 scenes: 18 splits = 5.6 rounds, 16 = 5.0)
            const int64_t slots = (int64_t)256 * (p.k32 == 4 && !k32_pipe(p.k32, qs) ? 3 : 2);
            double best = 1e30;
            int best_ns = ns;
            for (int c = ns > 4 ? ns - 4 : 1; c <= ns + 2 && c <= 64 && c <= (n_tiles / 8 > 0 ? n_tiles / 8 : 1); ++c) {
                const int64_t wgs = base * c;
                const double waste = (double)((wgs + slots - 1) / slots * slots) / (double)wgs;
                // (2 workgroups per CU: among equal fills the finer split ran faster -- 8 scenes: 8 splits 5.43 ms, 5 splits 5.96)
                if (wgs >= 3 * slots && (waste < best - 1e-9 || (slots == 512 && waste < best + 1e-9))) { best = waste; best_ns = c; }
            }
            ns = best_ns;
        }
        if (lvq_tune().attn_nsplit) {       // test hook: force the KV split count (1 = direct output path)
            const int f = lvq_tune().attn_nsplit;
            if (f >= 1 && f <= 64 && f <= n_tiles) ns = f;
        }
        p.nsplit = ns;
        return p;
    }
    const int dhp = (dh + 31) / 32 * 32;
    int qmax = (dhp <= 64 && !split) ? 4 : 2;
    if (dhp > 64 && split) qmax = 1;
    // Measured on MI355X (tools/bench_kernels.py attn, LVQ_ATTN_QT sweep): one 16-query tile per wave (4 waves/SIMD
    // resident) beats 2 or 4 tiles per wave on every shape tried -- occupancy hides the softmax VALU and the
    // staging latency better than K/V fragment reuse saves LDS reads.
    p.qt = (dhp >= 96 && !split) ? 2 : 1;             // head_dim 96/128: two tiles per wave measured faster (377 vs 267 TFLOP/s)
    if (lvq_tune().attn_qt) {      // tuning knob (tools/bench_kernels.py); not used in production
        const int f = lvq_tune().attn_qt;
        if ((f == 1 || f == 2 || f == 4) && f <= qmax) p.qt = f;
    }
    // waves per workgroup: long K/V streams are bandwidth-bound on re-reads -> as many queries per stream as fit
    p.nw = 4;
    if (p.qt == 1 && nkv >= 4096 && dhp <= 64) {
        for (int nw : {12, 8}) {
            const int64_t tile = 16 * nw, padded = (nq + tile - 1) / tile * tile;
            if ((padded - nq) * 8 <= nq) { p.nw = nw; break; }
        }
    }
    if (lvq_tune().attn_nw) {
        const int f = lvq_tune().attn_nw;
        if ((f == 4 || f == 8 || f == 12) && p.qt == 1 && dhp <= 64) p.nw = f;
    }
    p.nqt = (nq + 16 * p.nw * p.qt - 1) / (16 * p.nw * p.qt);
    const int64_t base = (int64_t)p.nqt * n_heads * batch;
    const int n_tiles = (nkv + KVB - 1) / KVB;
    // enough workgroups for >= ~5 dispatch rounds (3 resident workgroups per CU): with ~1.1 rounds the straggler round
    // doubled the kernel time (4 ms vs 8 ms run to run)
    int ns = (int)((4096 + base - 1) / base);
    // at least 8 KV tiles per split -- 3 for a decode step (a handful of queries: B*H workgroups walking ~14 tiles each in series
    // took 17 us per layer at 1 x 870 keys; 4 splits + combine: 1.39 -> 1.22 ms per token of the 24-layer decoder)
    const int min_tiles = (p.nqt == 1 && nq <= 16) ? 3 : 8;
    if (ns > n_tiles / min_tiles) ns = n_tiles / min_tiles;
    // head_dim 96 / 128: a partial is (dh + 2) fp32 per query and split -- with the 5-round rule the partials of the reference's
    // own VATLiDAR geometry (576 x 32 400, head_dim 112, 8 heads: 63 splits) were 132 MB against 116 MB of K|V.  One full
    // round of workgroups is enough there (measured: 63 splits 0.481 ms, 32: 0.416, 19: 0.395, 12: 0.392, 8: 0.466 for the
    // whole cross-attention sub-path).
    if (dhp >= 96) {
        const int64_t one_round = (256 * 3) / base;
        if (one_round >= 1 && ns > one_round) ns = (int)one_round;
    }
    if (ns < 1) ns = 1;
    if (ns > 64) ns = 64;
    if (lvq_tune().attn_nsplit) {           // test / tuning hook, as for k_attn32
        const int f = lvq_tune().attn_nsplit;
        if (f >= 1 && f <= 64 && f <= n_tiles) ns = f;
    }
    p.nsplit = ns;
    return p;
}

template <int DHP, int NSPLIT, int QT, int NW> int launch_attn_qt(AttnArgs &a, hipStream_t st) {
    constexpr int NS = (NSPLIT == 3) ? 2 : 1;
    const size_t lds = (size_t)(2 * 2 * NS * KVB * ((DHP == 64 && NW >= 8) ? DHP : DHP + 8)) * sizeof(uint16_t);   // two K+V stages
    if (lds > 64 * 1024)
        hipFuncSetAttribute((const void *)k_attn<DHP, NSPLIT, QT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int64_t ngrp = (int64_t)a.B * a.H * a.nsplit;
    const int64_t nwg = (NW >= 8 ? (ngrp + 7) / 8 * 8 : ngrp) * a.nqt;      // NW >= 8: XCD-aware id mapping in the kernel
    if (nwg > 0x7fffffff) return LVQ_EUNSUPPORTED;
    dim3 grid((unsigned)nwg);
    hipLaunchKernelGGL((k_attn<DHP, NSPLIT, QT, NW>), grid, dim3(NW * 64), lds, st, a);
    if (a.nsplit > 1) {
        const int64_t total = (int64_t)a.B * a.H * a.Nq * (a.dh / 4);
        hipLaunchKernelGGL(k_attn_combine, dim3((unsigned)lvq_cdiv(total, 256)), dim3(256), 0, st, a);
    }
    return lvq_launch_status();
}

template <int DHP, int NSPLIT> int launch_attn(AttnArgs &a, const AttnPlan &pl, hipStream_t st) {
    if (pl.qt == 1 && DHP <= 64) {        // the many-wave forms exist for the head dims of the VAT blocks (<= 64)
        if (pl.nw == 12) return launch_attn_qt<DHP, NSPLIT, 1, (DHP <= 64) ? 12 : 4>(a, st);
        if (pl.nw == 8) return launch_attn_qt<DHP, NSPLIT, 1, (DHP <= 64) ? 8 : 4>(a, st);
    }
    if (!(DHP > 64 && NSPLIT == 3) && pl.qt >= 2) return launch_attn_qt<DHP, NSPLIT, (DHP > 64 && NSPLIT == 3) ? 1 : 2, 4>(a, st);
    return launch_attn_qt<DHP, NSPLIT, 1, 4>(a, st);
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

// Per-model half of the plain-stream guard: one workgroup per score row s[0 .. n) (raw dot products; softmax weights exp(scale s)):
//   g = (1 + max |scale s|) / sqrt(N_eff),   N_eff = (sum p)^2 / sum p^2
// -- the quantity the error of plain bf16 K / V / P on a key stream scales with (tools/mixed_guard_study.py: a per-key 2^-9 rounding
// moves the output by ~ |score| 2^-9 per unit of softmax mass and averages out over N_eff keys).
__global__ void __launch_bounds__(256) k_stream_guard_rows(const float *__restrict__ s, int64_t n, float scale, float *__restrict__ g) {
    __shared__ float red[3][4];
    const float *sr = s + (int64_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float mx = -INFINITY, amax = 0.f;
    for (int64_t k = tid * 4; k < n; k += 1024) {
        const float4 v = *reinterpret_cast<const float4 *>(sr + k);
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o)), amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) red[0][w] = mx, red[1][w] = amax;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])) * scale;
    amax = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])) * fabsf(scale);
    __syncthreads();
    float l1 = 0.f, l2 = 0.f;
    for (int64_t k = tid * 4; k < n; k += 1024) {
        const float4 v = *reinterpret_cast<const float4 *>(sr + k);
        const float p0 = expf(v.x * scale - mx), p1 = expf(v.y * scale - mx), p2 = expf(v.z * scale - mx), p3 = expf(v.w * scale - mx);
        l1 += (p0 + p1) + (p2 + p3);
        l2 += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l1 += __shfl_xor(l1, o), l2 += __shfl_xor(l2, o);
    if (lane == 0) red[0][w] = l1, red[1][w] = l2;
    __syncthreads();
    if (tid == 0) {
        l1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        l2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        g[blockIdx.x] = (1.0f + amax) * sqrtf(l2) / l1;
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

// (A dedicated one-query kernel for decode steps -- one workgroup per (batch, head), fp32 FMAs, scores in LDS -- was built and
//  measured at 23-42 us per call against 16 us for the split-KV tile path below at 1 x ~900 keys: 14 workgroups of strided row
//  reads cannot hide their latency.  Removed; tests/test_gpu_head.py::test_decode_attention_one_query covers the shape.)

}  // namespace

// the long-stream plan with the larger KV split of the two query kinds (workspace queries do not know which one will be launched)
static AttnPlan plan_k32_any(int batch, int n_heads, int nq, int nkv, int dh) {
    const AttnPlan a = plan_attn(batch, n_heads, nq, nkv, dh, false, true, 0), b = plan_attn(batch, n_heads, nq, nkv, dh, false, true, 1);
    return b.nsplit > a.nsplit ? b : a;
}

extern "C" size_t lvq_attention_workspace_bytes(int batch, int n_heads, int nq, int nkv, int dh, int precision) {
    if (dh <= 128 && (dh & 15) == 0) {              // fused kernel: workspace only for the KV-split partials
        const AttnPlan p = plan_attn(batch, n_heads, nq, nkv, dh, precision == 3);
        const AttnPlan p32 = plan_attn(batch, n_heads, nq, nkv, dh, precision == 3, true);   // chosen when there is no bias / mask
        const AttnPlan p32q = plan_attn(batch, n_heads, nq, nkv, dh, precision == 3, true, 1);   // ... and the query is split
        int ns = p.nsplit > p32.nsplit ? p.nsplit : p32.nsplit;
        if (p32q.nsplit > ns) ns = p32q.nsplit;
        if (ns == 1) return 256;
        return lvq_align((size_t)batch * n_heads * ns * nq * (dh + 2) * sizeof(float)) + 256;
    }
    const int64_t nkp = (nkv + 7) / 8 * 8;
    const int ns = precision == 3 ? 2 : 1;
    size_t s = (size_t)n_heads * nq * nkv * sizeof(float);            // scores of one batch element
    size_t p = (size_t)ns * n_heads * nq * nkp * sizeof(uint16_t);    // probabilities (hi, lo)
    size_t vt = (size_t)ns * n_heads * dh * nkp * sizeof(uint16_t);   // V^T (hi, lo)
    (void)batch;
    return lvq_align(s) + lvq_align(p) + lvq_align(vt) + 1024;
}

extern "C" int lvq_attention_stream_ok(int nq, int nkv, int dh) { return plan_k32_waves(nq, nkv, dh, false) != 0; }

namespace {
constexpr size_t K32_LDS = (size_t)3 * 2 * KVB * 64 * sizeof(uint16_t) + 16;               // three K+V slots + the redo flag
constexpr size_t K32_LDS_TILED = K32_LDS + 2 * 8 * KVB * sizeof(int32_t);                  // + two windows of 8 tiles x 64 row words
constexpr size_t K32_LDS_TILED_PIPE = K32_LDS_TILED + (size_t)2 * KVB * 64 * sizeof(uint16_t);   // the pipelined form's fourth slot
template <int QS, int TL> void launch_k32_pipe(const AttnArgs &a, int64_t nwg, size_t lds, hipStream_t st) {
    const size_t l = lds + (size_t)2 * KVB * 64 * sizeof(uint16_t);                         // a fourth K+V slot
    static bool attr_set[64] = {};                                                          // per device (the attribute is per device)
    int devi = 0;
    hipGetDevice(&devi);
    if (devi < 0 || devi >= 64 || !attr_set[devi]) {
        hipFuncSetAttribute((const void *)k_attn32<4, QS, TL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(K32_LDS_TILED_PIPE));
        if (devi >= 0 && devi < 64) attr_set[devi] = true;
    }
    hipLaunchKernelGGL((k_attn32<4, QS, TL, true>), dim3((unsigned)nwg), dim3(256), l, st, a);
}
template <int TL> void launch_k32(const AttnArgs &a, int nw, int qs, int64_t nwg, size_t lds, hipStream_t st) {
    if (k32_pipe(nw, qs)) {
        if (qs == 2)  launch_k32_pipe<2, TL>(a, nwg, lds, st);
        else if (qs)  launch_k32_pipe<1, TL>(a, nwg, lds, st);
        else          launch_k32_pipe<0, TL>(a, nwg, lds, st);
        return;
    }
    if (qs == 2) {
        if (nw == 6) hipLaunchKernelGGL((k_attn32<6, 2, TL>), dim3((unsigned)nwg), dim3(384), lds, st, a);
        else         hipLaunchKernelGGL((k_attn32<4, 2, TL>), dim3((unsigned)nwg), dim3(256), lds, st, a);
    } else if (qs) {
        if (nw == 6) hipLaunchKernelGGL((k_attn32<6, 1, TL>), dim3((unsigned)nwg), dim3(384), lds, st, a);
        else         hipLaunchKernelGGL((k_attn32<4, 1, TL>), dim3((unsigned)nwg), dim3(256), lds, st, a);
    } else {
        if (nw == 6) hipLaunchKernelGGL((k_attn32<6, 0, TL>), dim3((unsigned)nwg), dim3(384), lds, st, a);
        else         hipLaunchKernelGGL((k_attn32<4, 0, TL>), dim3((unsigned)nwg), dim3(256), lds, st, a);
    }
}
}  // namespace

// VATLiDAR's cross-attention over the TILED key stream of bev_tiles.hip: batch b attends to n_tiles x 64 keys; key slot r of tile t is row
// row_src[(b * n_tiles + t) * 64 + r] of ONE K|V buffer that holds the per-model table rows and the computed rows of every batch.
// Long-stream kernel only (head_dim 64, n_tiles * 64 >= 4096, lvq_attention_stream_ok(nq, 64 n_tiles, 64)); q plain or hi + lo, K / V plain.
extern "C" int lvq_attention_bf16_tiled(const lvq_bf16 *q, const lvq_bf16 *q_lo, const lvq_bf16 *k_rows, const lvq_bf16 *v_rows,
                                        const int32_t *row_src, int batch, int n_heads, int nq, int n_tiles, int dh, int64_t q_bstride, int64_t ldq,
                                        int64_t q_hstride, int64_t ldkv, int64_t kv_hstride, int64_t o_bstride, int64_t ldo, int64_t o_hstride,
                                        float scale, int k_fp16, lvq_bf16 *o, lvq_bf16 *o_lo, void *ws, size_t ws_bytes, lvq_stream_t stream) {
    if (batch <= 0 || n_heads <= 0 || nq < 0 || n_tiles <= 0 || !q || !k_rows || !v_rows || !row_src || !o) return LVQ_EINVAL;
    if (nq == 0) return LVQ_OK;
    const int64_t nkv64 = (int64_t)n_tiles * KVB;
    if (nkv64 > 0x7fffffff) return LVQ_EUNSUPPORTED;
    const int nkv = (int)nkv64;
    if (dh != 64 || (ldq & 7) || (ldkv & 7) || ldkv <= 0 || ldkv > 0x7fffffff || (q_hstride & 7) || (kv_hstride & 7) || (q_bstride & 7) || (ldo & 3) || (o_hstride & 3) || (o_bstride & 3))
        return LVQ_EUNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)q_lo | (uintptr_t)k_rows | (uintptr_t)v_rows | (uintptr_t)row_src) & 15) return LVQ_EUNSUPPORTED;
    if (((uintptr_t)o | (uintptr_t)o_lo) & 7) return LVQ_EUNSUPPORTED;
    if (n_heads > 65535 || batch > 65535) return LVQ_EUNSUPPORTED;
    AttnArgs a{};
    a.q = q; a.ql = q_lo; a.k = k_rows; a.kl = nullptr; a.v = v_rows; a.vl = nullptr; a.bias = nullptr;
    a.B = batch; a.H = n_heads; a.Hkv = n_heads; a.Nq = nq; a.Nkv = nkv; a.dh = dh;
    a.q_bs = q_bstride; a.ldq = ldq; a.q_hs = q_hstride; a.k_bs = 0; a.ldk = ldkv; a.k_hs = kv_hstride;
    a.v_bs = 0; a.ldv = ldkv; a.v_hs = kv_hstride; a.o_bs = o_bstride; a.ldo = ldo; a.o_hs = o_hstride;
    a.scale = scale; a.causal = 0; a.o = o; a.ol = o_lo;
    a.row_src = row_src;
    const int qs = k_fp16 ? 2 : (q_lo != nullptr ? 1 : 0);
    const AttnPlan pl = plan_attn(batch, n_heads, nq, nkv, dh, false, true, qs);
    if (!pl.k32) return LVQ_EUNSUPPORTED;
    a.nsplit = pl.nsplit; a.nqt = pl.nqt; a.part = nullptr;
    if (pl.nsplit > 1) {
        LvqArena arena(ws, ws_bytes);
        a.part = arena.take<float>((size_t)batch * n_heads * pl.nsplit * nq * (dh + 2));
        if (!arena.ok) return LVQ_EWORKSPACE;
    }
    hipStream_t st = lvq_s(stream);
    const int64_t ngrp = (int64_t)a.B * a.H * a.nsplit;
    const int64_t nwg = (ngrp + 7) / 8 * 8 * a.nqt;
    if (nwg > 0x7fffffff) return LVQ_EUNSUPPORTED;
    launch_k32<1>(a, pl.k32, qs, nwg, K32_LDS_TILED, st);
    if (a.nsplit > 1) {
        const int64_t total = (int64_t)a.B * a.H * a.Nq * (a.dh / 4);
        hipLaunchKernelGGL(k_attn_combine, dim3((unsigned)lvq_cdiv(total, 256)), dim3(256), 0, st, a);
    }
    return lvq_launch_status();
}


// ---- the signed pair stream: attention over the LIVE pieces only -------------------------------------------------------------
// When the queries do not depend on the batch (VATLiDAR's first block: learned queries -> self-attention -> ca_ln -> W_q), the
// contribution of a TABLE key to the fixed-reference softmax sums (l, O) is the same for every batch.  With TOTALS = the sums over
// all table keys (once per model: lvq_attention_bf16_stream_totals), a batch's result is
//     TOTALS - sum over its live positions of the table contribution + sum over its live rows,
// i.e. a stream over 2 x (live fraction) of the keys instead of all of them.  The subtracted terms are recomputed with the same
// operand bits as inside TOTALS, so the cancellation is exact up to fp32 accumulation order (not in the mixed16 mode when the totals
// were taken with the full fp16 hi + lo query, k_fp16 = 2: there the per-scene streams subtract with the once-rounded query and the
// cancellation holds to 2^-11 of the subtracted scores only -- deliberate, see include/lvq.h).  Batches whose pair list would not be
// shorter run their full tile list; (batch, head) pairs whose signed result is unusable are redone by a predicated full launch.

// totals [n_heads, nq, dh + 2] fp32 (unnormalised O | m | l) of ONE batch of queries over a dense key stream (the table).
extern "C" size_t lvq_attention_stream_totals_workspace_bytes(int n_heads, int nq, int nkv, int dh) {
    const AttnPlan pl = plan_k32_any(1, n_heads, nq, nkv, dh);
    if (!pl.k32) return 0;
    return (size_t)n_heads * pl.nsplit * nq * (dh + 2) * sizeof(float) + 256;
}
extern "C" int lvq_attention_bf16_stream_totals(const lvq_bf16 *q, const lvq_bf16 *q_lo, const lvq_bf16 *k, const lvq_bf16 *v, int n_heads, int nq,
                                                int nkv, int dh, int64_t ldq, int64_t q_hstride, int64_t ldkv, int64_t kv_hstride, float scale,
                                                int k_fp16, float *totals, void *ws, size_t ws_bytes, lvq_stream_t stream) {
    if (n_heads <= 0 || nq <= 0 || nkv <= 0 || !q || !k || !v || !totals) return LVQ_EINVAL;
    if (dh != 64 || (ldq & 7) || (ldkv & 7) || (q_hstride & 7) || (kv_hstride & 7) || n_heads > 65535) return LVQ_EUNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)q_lo | (uintptr_t)k | (uintptr_t)v | (uintptr_t)totals) & 15) return LVQ_EUNSUPPORTED;
    if (((int64_t)nkv * ldkv + dh) * 2 >= (1ll << 32)) return LVQ_EUNSUPPORTED;
    const int qs = k_fp16 == 2 ? 0 : k_fp16 ? 2 : (q_lo != nullptr ? 1 : 0);            // (the fp16 hi + lo form has no pipelined variant)
    const AttnPlan pl = plan_attn(1, n_heads, nq, nkv, dh, false, true, qs);
    if (!pl.k32) return LVQ_EUNSUPPORTED;
    AttnArgs a{};
    a.q = q; a.ql = q_lo; a.k = k; a.v = v;
    a.B = 1; a.H = n_heads; a.Hkv = n_heads; a.Nq = nq; a.Nkv = nkv; a.dh = dh;
    a.ldq = ldq; a.q_hs = q_hstride; a.ldk = ldkv; a.k_hs = kv_hstride; a.ldv = ldkv; a.v_hs = kv_hstride;
    a.scale = scale; a.nsplit = pl.nsplit; a.nqt = pl.nqt; a.force_part = 1;
    LvqArena arena(ws, ws_bytes);
    a.part = arena.take<float>((size_t)n_heads * pl.nsplit * nq * (dh + 2));
    if (!arena.ok) return LVQ_EWORKSPACE;
    hipStream_t st = lvq_s(stream);
    const size_t lds = K32_LDS;
    const int64_t ngrp = (int64_t)a.H * a.nsplit;
    if (k_fp16 == 2) {       // fp16 K against the query as fp16 hi + lo: the totals keep the full query, only the per-batch streams round it
        const int64_t nwg = (ngrp + 7) / 8 * 8 * a.nqt;
        if (pl.k32 == 6) hipLaunchKernelGGL((k_attn32<6, 3, 0>), dim3((unsigned)nwg), dim3(384), lds, st, a);
        else             hipLaunchKernelGGL((k_attn32<4, 3, 0>), dim3((unsigned)nwg), dim3(256), lds, st, a);
    } else {
        launch_k32<0>(a, pl.k32, qs, (ngrp + 7) / 8 * 8 * a.nqt, lds, st);
    }
    const int64_t total = (int64_t)a.H * a.Nq * (a.dh / 4);
    hipLaunchKernelGGL(k_attn_combine_raw, dim3((unsigned)lvq_cdiv(total, 256)), dim3(256), 0, st, a, totals);
    return lvq_launch_status();
}

extern "C" size_t lvq_attention_tiled_signed_workspace_bytes(int batch, int n_heads, int nq, int n_tiles, int dh) {
    const int64_t nkv = (int64_t)n_tiles * KVB;
    if (nkv > 0x7fffffff) return 0;
    const AttnPlan pl = plan_k32_any(batch, n_heads, nq, (int)nkv, dh);
    if (!pl.k32) return 0;
    return (size_t)batch * n_heads * pl.nsplit * nq * (dh + 2) * sizeof(float) + (size_t)batch * n_heads * sizeof(int32_t) +
           (size_t)STAT_SLOTS * 64 * sizeof(int32_t) + 768;
}

// As lvq_attention_bf16_tiled, with the per-batch pair lists of lvq_bev_scene_pairs and the per-model totals.  The queries must be the
// ones the totals were computed with (pass q_bstride = 0 to share one copy between the batches).
extern "C" int lvq_attention_bf16_tiled_signed(const lvq_bf16 *q, const lvq_bf16 *q_lo, const lvq_bf16 *k_rows, const lvq_bf16 *v_rows,
                                               const int32_t *row_src, const int32_t *pair_src,
                                               const int32_t *pair_info, int pair_cap_tiles, const float *totals, int batch, int n_heads, int nq,
                                               int n_tiles, int dh, int64_t q_bstride, int64_t ldq, int64_t q_hstride, int64_t ldkv,
                                               int64_t kv_hstride, int64_t o_bstride, int64_t ldo, int64_t o_hstride, float scale, int k_fp16,
                                               lvq_bf16 *o, lvq_bf16 *o_lo, int32_t *stats, void *ws, size_t ws_bytes, lvq_stream_t stream) {
    if (batch <= 0 || n_heads <= 0 || nq < 0 || n_tiles <= 0 || pair_cap_tiles <= 0 || !q || !k_rows || !v_rows || !row_src || !pair_src ||
        !pair_info || !totals || !o)
        return LVQ_EINVAL;
    if (nq == 0) return LVQ_OK;
    const int64_t nkv64 = (int64_t)n_tiles * KVB;
    if (nkv64 > 0x7fffffff) return LVQ_EUNSUPPORTED;
    const int nkv = (int)nkv64;
    if (dh != 64 || (ldq & 7) || (ldkv & 7) || ldkv <= 0 || ldkv > 0x7fffffff || (q_hstride & 7) || (kv_hstride & 7) || (q_bstride & 7) || (ldo & 3) ||
        (o_hstride & 3) || (o_bstride & 3))
        return LVQ_EUNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)q_lo | (uintptr_t)k_rows | (uintptr_t)v_rows | (uintptr_t)row_src | (uintptr_t)pair_src | (uintptr_t)totals) & 15)
        return LVQ_EUNSUPPORTED;
    if (((uintptr_t)o | (uintptr_t)o_lo) & 7) return LVQ_EUNSUPPORTED;
    if (n_heads > 65535 || batch > 65535) return LVQ_EUNSUPPORTED;
    const int qs = k_fp16 ? 2 : (q_lo != nullptr ? 1 : 0);
    const AttnPlan pl = plan_attn(batch, n_heads, nq, nkv, dh, false, true, qs);
    if (!pl.k32) return LVQ_EUNSUPPORTED;
    AttnArgs a{};
    a.q = q; a.ql = q_lo; a.k = k_rows; a.v = v_rows;
    a.B = batch; a.H = n_heads; a.Hkv = n_heads; a.Nq = nq; a.Nkv = nkv; a.dh = dh;
    a.q_bs = q_bstride; a.ldq = ldq; a.q_hs = q_hstride; a.ldk = ldkv; a.k_hs = kv_hstride; a.ldv = ldkv; a.v_hs = kv_hstride;
    a.o_bs = o_bstride; a.ldo = ldo; a.o_hs = o_hstride; a.scale = scale; a.o = o; a.ol = o_lo;
    a.row_src = row_src;
    a.pair_src = pair_src; a.pair_info = pair_info; a.pair_cap = pair_cap_tiles; a.totals = totals;
    a.nsplit = pl.nsplit; a.nqt = pl.nqt; a.force_part = 1;
    LvqArena arena(ws, ws_bytes);
    a.part = arena.take<float>((size_t)batch * n_heads * pl.nsplit * nq * (dh + 2));
    int32_t *flags = arena.take<int32_t>((size_t)batch * n_heads);
    int32_t *stat_slots = stats ? arena.take<int32_t>((size_t)STAT_SLOTS * 64) : nullptr;
    if (!arena.ok) return LVQ_EWORKSPACE;
    hipStream_t st = lvq_s(stream);
    const size_t lds = K32_LDS_TILED;
    const int64_t ngrp = (int64_t)a.B * a.H * a.nsplit;
    const int64_t nwg = (ngrp + 7) / 8 * 8 * a.nqt;
    if (nwg > 0x7fffffff) return LVQ_EUNSUPPORTED;
    const int64_t total = (int64_t)a.B * a.H * a.Nq * (a.dh / 4);
    if (hipMemsetAsync(flags, 0, (size_t)batch * n_heads * sizeof(int32_t), st) != hipSuccess) return LVQ_ELAUNCH;
    if (stat_slots && hipMemsetAsync(stat_slots, 0, (size_t)STAT_SLOTS * 64 * sizeof(int32_t), st) != hipSuccess) return LVQ_ELAUNCH;
    a.flags = flags;
    a.stats = stats;
    a.stat_slots = stat_slots;
    launch_k32<2>(a, pl.k32, qs, nwg, lds, st);
    hipLaunchKernelGGL(k_attn_combine_signed, dim3((unsigned)lvq_cdiv(total, 256)), dim3(256), 0, st, a);
    // predicated full re-run of the flagged (batch, head) pairs: every workgroup of an unflagged pair returns at once
    a.flags = nullptr; a.pred = flags; a.pair_src = nullptr; a.pair_info = nullptr; a.totals = nullptr;      // (stats / stat_slots stay: folded by this merge)
    launch_k32<1>(a, pl.k32, qs, nwg, lds, st);
    hipLaunchKernelGGL(k_attn_combine_signed, dim3((unsigned)lvq_cdiv(total, 256)), dim3(256), 0, st, a);
    return lvq_launch_status();
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
    // precision modes: plain (no lo parts), bf16x3 (q, k, v all hi + lo), "mixed" stream form (q hi + lo, k / v plain: the
    // long-stream kernel k_attn32<., 1> only -- lvq_attention_stream_ok says whether a shape takes it)
    const bool split = k_lo != nullptr;
    const bool qsplit = q_lo != nullptr && !split;
    if (split != (v_lo != nullptr) || (split && !q_lo)) return LVQ_EINVAL;
    if (nq == 0) return LVQ_OK;
    if ((dh & 7) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (q_hstride & 7) || (k_hstride & 7) || (v_hstride & 7) ||
        (q_bstride & 7) || (k_bstride & 7) || (v_bstride & 7) || (ldo & 3) || (o_hstride & 3) || (o_bstride & 3))
        return LVQ_EUNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)q_lo | (uintptr_t)k_lo | (uintptr_t)v_lo) & 15)
        return LVQ_EUNSUPPORTED;
    if (((uintptr_t)o | (uintptr_t)o_lo) & 7) return LVQ_EUNSUPPORTED;
    if (n_heads > 65535 || batch > 65535) return LVQ_EUNSUPPORTED;
    // the fused kernel addresses one (batch, kv-head) K / V slab with 32-bit buffer offsets
    if (((int64_t)nkv * ldk + dh) * 2 >= (1ll << 32) || ((int64_t)nkv * ldv + dh) * 2 >= (1ll << 32)) return LVQ_EUNSUPPORTED;
    hipStream_t st = lvq_s(stream);
    if (dh <= 128 && (dh & 15) == 0) {
        AttnArgs a{};
        a.q = q; a.ql = q_lo; a.k = k; a.kl = k_lo; a.v = v; a.vl = v_lo; a.bias = bias;
        a.B = batch; a.H = n_heads; a.Hkv = n_kv_heads; a.Nq = nq; a.Nkv = nkv; a.dh = dh;
        a.q_bs = q_bstride; a.ldq = ldq; a.q_hs = q_hstride; a.k_bs = k_bstride; a.ldk = ldk; a.k_hs = k_hstride;
        a.v_bs = v_bstride; a.ldv = ldv; a.v_hs = v_hstride; a.o_bs = o_bstride; a.ldo = ldo; a.o_hs = o_hstride;
        a.scale = scale; a.causal = causal; a.o = o; a.ol = o_lo;
        const int dhp = (dh + 31) / 32 * 32;
        const AttnPlan pl = plan_attn(batch, n_heads, nq, nkv, dh, split, bias == nullptr && !causal && (o_lo == nullptr || qsplit), qsplit ? 1 : 0);
        if (qsplit && !pl.k32) return LVQ_EUNSUPPORTED;
        a.nsplit = pl.nsplit; a.nqt = pl.nqt; a.part = nullptr;
        if (pl.nsplit > 1) {
            LvqArena arena(ws, ws_bytes);
            a.part = arena.take<float>((size_t)batch * n_heads * pl.nsplit * nq * (dh + 2));
            if (!arena.ok) return LVQ_EWORKSPACE;
        }
        if (pl.k32) {
            const int64_t ngrp = (int64_t)a.B * a.H * a.nsplit;
            const int64_t nwg = (ngrp + 7) / 8 * 8 * a.nqt;       // groups padded to the 8 XCDs (see the kernel's id mapping)
            if (nwg > 0x7fffffff) return LVQ_EUNSUPPORTED;
            launch_k32<0>(a, pl.k32, qsplit ? 1 : 0, nwg, K32_LDS, st);
            if (a.nsplit > 1) {
                const int64_t total = (int64_t)a.B * a.H * a.Nq * (a.dh / 4);
                hipLaunchKernelGGL(k_attn_combine, dim3((unsigned)lvq_cdiv(total, 256)), dim3(256), 0, st, a);
            }
            return lvq_launch_status();
        }
        if (!split) {
            switch (dhp) {
                case 32: return launch_attn<32, 1>(a, pl, st);
                case 64: return launch_attn<64, 1>(a, pl, st);
                case 96: return launch_attn<96, 1>(a, pl, st);
                default: return launch_attn<128, 1>(a, pl, st);
            }
        } else {
            switch (dhp) {
                case 32: return launch_attn<32, 3>(a, pl, st);
                case 64: return launch_attn<64, 3>(a, pl, st);
                case 96: return launch_attn<96, 3>(a, pl, st);
                default: return launch_attn<128, 3>(a, pl, st);
            }
        }
    }
    // ---- split path for large head dims: S = Q K^T (GEMM) -> row softmax -> O = P V (GEMM on V^T) ----
    if (n_heads != n_kv_heads || qsplit) return LVQ_EUNSUPPORTED;
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

extern "C" size_t lvq_stream_guard_workspace_bytes(int nq, int64_t nkv) { return (size_t)nq * (size_t)nkv * sizeof(float) + 256; }

// g[h * nq + i] = (1 + max_k |scale q_i . k_k|) / sqrt(N_eff) of head h, query i over the nkv keys (head_dim 64, nkv % 4 == 0, plain bf16
// operands: a guard statistic, not a result).  The caller reduces g (its maximum) and decides the route.
extern "C" int lvq_stream_guard(const lvq_bf16 *q, const lvq_bf16 *k_rows, int n_heads, int nq, int64_t nkv, int64_t ldq, int64_t ldk, float scale,
                                float *g, void *ws, size_t ws_bytes, lvq_stream_t stream) {
    if (!q || !k_rows || !g || !ws || n_heads <= 0 || nq <= 0 || nkv <= 0) return LVQ_EINVAL;
    if ((nkv & 3) || nkv > 0x7fffffff) return LVQ_EUNSUPPORTED;
    if (ws_bytes < lvq_stream_guard_workspace_bytes(nq, nkv)) return LVQ_EWORKSPACE;
    float *S = (float *)ws;
    hipStream_t st = lvq_s(stream);
    for (int h = 0; h < n_heads; ++h) {
        const int rc = lvq_gemm_bf16(q + (int64_t)h * 64, nullptr, k_rows + (int64_t)h * 64, nullptr, nullptr, nullptr, nullptr, 0, 1.0f, 0, nq, (int)nkv, 64,
                                     ldq, ldk, nkv, 1, 0, 0, 0, S, nullptr, nullptr, stream);
        if (rc != LVQ_OK) return rc;
        hipLaunchKernelGGL(k_stream_guard_rows, dim3((unsigned)nq), dim3(256), 0, st, S, nkv, scale, g + (int64_t)h * nq);
    }
    return lvq_launch_status();
}
