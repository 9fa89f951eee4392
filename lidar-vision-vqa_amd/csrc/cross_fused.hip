// csrc/cross_fused.hip -- the short-K/V cross-attention sub-path of VATBlock as ONE kernel (gfx950):
//
//     out = q + ca(ca_ln(q), kv, kv)            encoder-decoder/training/models/vat_blocks.py:41-42
//
// at the shapes BASELINE.json's metric names: (B, Nq, Nkv, d, h) = (1, 32768, 196, 768, 12) "32k pts x 196 patches", and the
// resampled-token forms (B, 576, 196), i.e. many queries against a few hundred keys.  The unfused sequence (LayerNorm, Q projection,
// K|V projection, attention, out projection + residual: five launches) moves 24-36 B per activation element through HBM for a problem
// whose algorithmic traffic is 8 B (fp32 q in, fp32 out); here the activation never leaves the CU between the fp32 load and the fp32 store.
//
// Structure.  A workgroup = 4 waves, one per SIMD, each with the whole 512-register file; a wave owns 32 query rows and keeps, as MFMA
// operand fragments IN REGISTERS: LayerNorm(x) of its rows (192 registers), then Q of all heads (192), overwritten head by head with the
// attention output O.  Every product is computed TRANSPOSED (C^T = W . X^T): the query sits on the lane, so
//   * LayerNorm is two lanes per row and one permlane swap,
//   * the Q^T accumulator tile IS the B operand of S^T = K Q^T, the exponentiated S^T IS the B operand of O^T = V^T P^T and O^T IS
//     the B operand of out^T = W_o O^T (MI355X guide: "an accumulator tile as the next MFMA's operand") -- nothing is ever shuffled
//     between lanes or staged through LDS, and softmax statistics are per-lane scalars + one permlane swap.
// All A operands (W_q, K, V^T, W_o) are PRE-PACKED in MFMA fragment order (1 KiB = one wave-instruction per 32 x 16 fragment, with the
// k permutation pi(ks, half, j) = 16 ks + 8 (j >> 2) + 4 half + (j & 3) that the accumulator-as-operand identity imposes) and arrive as
// ONE linear fragment stream per workgroup through an LDS ring filled by LDS-DMA (global_load_lds, 16 B per lane): the four waves share
// every fragment, so the stream leaves L2 once per 128 rows (3 MB per workgroup; 0.8 GB per call at the headline shape).  The fp32
// residual rows come back the same way (LDS-DMA with a per-lane gather address into a wave-private LDS area), so the steady state issues
// no register-destination global load at all and every wait in the stream is a counted vmcnt / lgkmcnt placed by hand.
// The K|V projection of the few hundred kv tokens is a small kernel of its own that writes K and V^T directly in fragment order.
//
// Operand type: bf16 or IEEE fp16 (same MFMA rate).  tools/precision_study_ca.py: with every operand rounded once to fp16 the sub-path
// is 4.3e-4 from the fp64 result at the headline shape (bf16: 3.6e-3; tolerance 1e-3), so the fp16 form is the parity-true one.
// fp16's range is guarded where a value is not bounded by construction (kv tokens, K, V, scaled Q: clamped to +-65504).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

typedef short h16x8 __attribute__((ext_vector_type(8)));       // eight 16-bit MFMA operands (bf16 or fp16 bit patterns)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

constexpr int D = 768, NH = 12, DH = 64;
constexpr int NKS = D / 16;                 // 48 k-steps of a 768-deep product
constexpr int FRAG = 1024;                  // bytes of one fragment = one wave-instruction of 16 B per lane
constexpr int GROUP = 16;                   // fragments per ring group (one barrier per group)
constexpr int NSLOT = 4;                    // ring slots
constexpr int NW = 4;                       // waves per workgroup
constexpr int DPG = GROUP / NW;             // LDS-DMA instructions per wave and group
constexpr int RING_BYTES = NSLOT * GROUP * FRAG;         // 64 KiB
constexpr int QL = 5;                       // heads whose Q / O fragments wait in LDS during phases A and B
constexpr int RESID_BYTES = (4 * QL > GROUP ? 4 * QL : GROUP) * FRAG;   // per wave: parked Q / O fragments, then (phase C) 32 rows x 128 columns fp32
constexpr int TAB_OFF = RING_BYTES + NW * RESID_BYTES;   // three [768] fp32 tables: bias' of Q, row sums of W_q', b_o
constexpr int LDS_BYTES = TAB_OFF + 3 * D * 4;
constexpr int RPITCH = FRAG + 16;            // LDS pitch of a residual piece (two 512-byte row segments + pad)
constexpr int NBC = 4;                      // 32-column blocks per out-projection chunk (128 columns)
constexpr int NCHUNK = D / (32 * NBC);      // 6

template <int B, int E, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

template <bool F16> __device__ __forceinline__ f32x16 mfma(const h16x8 &a, const h16x8 &b, const f32x16 &c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// two fp32 -> one packed pair, round to nearest even (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32)
template <bool F16> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    if constexpr (F16) {
        f16x2_t p = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(uint32_t, p);
    } else {
        bf16x2_t p = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(uint32_t, p);
    }
}
__device__ __forceinline__ float clamp16(float x) { return __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f); }
template <bool F16, bool CLAMP = false> __device__ __forceinline__ h16x8 pack8(const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = v[2 * i], b = v[2 * i + 1];
        if (F16 && CLAMP) a = clamp16(a), b = clamp16(b);
        w[i] = pack2<F16>(a, b);
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 u = {w[0], w[1], w[2], w[3]};
    return __builtin_bit_cast(h16x8, u);
}

// LDS reads as opaque asm (the compiler's waitcnt pass would put vmcnt(0) in front of any LDS read it can see while an LDS-DMA is in
// flight, i.e. drain the ring); the consumer waits on lgkmcnt itself (hardware returns LDS data in order)
template <int OFF> __device__ __forceinline__ void lds_read(h16x8 &d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_read(f32x4 &d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wait0(h16x8 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])::"memory");
}
__device__ __forceinline__ void lds_wait4(h16x8 (&f)[4]) {      // leaves the four newest reads (the next batch) in flight
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])::"memory");
}
__device__ __forceinline__ void lds_wait0(f32x4 (&f)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])::"memory");
}
// pins a value where it is computed: LLVM otherwise sinks pure arithmetic to its first use (for Q / O fragments: a whole phase later,
// with the accumulators and table values it depends on kept alive -- i.e. spilled -- until then)
__device__ __forceinline__ void pin(h16x8 &x) { asm volatile("" : "+v"(x)); }
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ float swap_sum(float x) {            // x(lane) + x(lane ^ 32)
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

struct CaArgs {
    const float *x;          // [rows, D] fp32: the queries (= the residual stream)
    float *out;              // [rows, D] fp32
    const char *wq;          // packed W_q' fragments [head][ks][dh block]      (gamma folded in)
    const char *wo;          // packed W_o fragments [chunk][ks][column block]
    const char *kv;          // packed K / V^T fragments [batch][head][CF]
    const float *tabs;       // [3][D] fp32: (b_q + W_q beta) * qscale | row sums of the rounded W_q' | b_o
    int nq, nkv;             // rows per batch element, keys
    int tiles_per_batch;     // ceil(nq / 128)
    int64_t kv_batch_bytes;
    float eps, qscale;       // qscale = log2(e) / sqrt(dh)
    unsigned long long *stamps;   // diagnostics (LVQ_CA_STAMPS): [workgroup][wave][8] s_memrealtime ticks (100 MHz) at the phase boundaries
};

template <int OFF> __device__ __forceinline__ void lds_write(uint32_t addr, const f32x4 &d) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(d), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_write(uint32_t addr, const h16x8 &d) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(d), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wait0_4(h16x8 &a, h16x8 &b, h16x8 &c, h16x8 &d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
__device__ __forceinline__ void lds_wait0(f32x4 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])::"memory");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The fused kernel.  KC = 32-key blocks of the K|V stream (all scores of a row live in registers, one softmax pass; nkv <= 32 KC).
// Fragment stream of one workgroup (groups of 16 fragments; every consumer step is a "batch" of 4 fragments = 4 MFMAs per wave):
//   A  12 heads x [48 k-steps x 2 dh-blocks] of W_q'            72 groups    loop over head pairs, 48 batches per iteration
//   B  12 heads x [K: KC x 4 | V^T: KC x 4 | pad]               12 x CG groups   loop over heads, 4 CG batches per iteration
//   C   6 chunks x [48 k-steps x 4 column blocks] of W_o        72 groups    loop over chunks, 48 batches per iteration
// Every loop body covers a whole number of ring revolutions (4 groups = 16 batches), so ring slots, fragment offsets and the
// double-buffer parity are compile-time constants inside a body while the code stays a few thousand instructions per phase.
// Registers (512 per lane, one wave per SIMD): the row's 16-bit operand fragments xd (192, phase A), Q / O of heads QL..11 (112;
// heads 0..QL-1 live in this wave's residual area in LDS until phase C has room for them), scores (112) and P (56) of one head.
// ---------------------------------------------------------------------------------------------------------------------------------

// compile-time description of a consumer step: batches per loop body, residual pieces ride along (phase C), vector code in the body
template <int BODY, bool RESID, bool LATE> struct StepCfg {
    static constexpr int body_batches = BODY;
    static constexpr bool resid = RESID, late = LATE;
    // vector-memory operations OTHER than ring pieces that batch lb of the body issues (phase C: one residual piece in batches 16 .. 31
    // of a 48-batch chunk, four row stores in batches 9, 11, 13, 15 of every chunk but the first).  vmcnt retires in order, so the counted
    // wait for a ring piece must allow for every younger operation -- otherwise it waits for the stores' round trip to HBM as well.
    static constexpr int other_ops(int lb, bool stores) {
        if (!RESID || lb < 0) return 0;
        const int l = lb % (D / 16), c = lb / (D / 16);
        return ((l >= 16 && l < 32) ? 1 : 0) + ((stores && c > 0 && l >= 9 && l < 16 && (l & 1)) ? 4 : 0);
    }
    // ... issued after the last ring piece of the group that the wait at the start of batch lb (lb % 4 == 3) is for: that piece went out
    // in batch lb - 9 (before that batch's fourth vector slot); counted from batch lb - 8 on (a lower bound is safe, an upper bound is not)
    // (`stores`: a wave without rows of its own issues no stores)
    static constexpr int younger_other_ops(int lb, bool stores) {
        int n = 0;
        for (int b = lb - 8; b < lb; ++b) n += other_ops(b, stores);
        return n;
    }
};

// DBG: timing experiments only (CA_DEBUG_VARIANTS builds; results are wrong): 1 no MFMA, 2 no LDS-DMA after the first ring fill,
// 4 no fragment reads, 8 no barriers, 16 no softmax / epilogue vector code
template <bool F16, int KC, int DBG = 0>
__global__ void __launch_bounds__(NW * 64, 1) k_ca_fused(CaArgs a) {
    constexpr int CF = (8 * KC + GROUP - 1) / GROUP * GROUP;    // fragments per (batch, head) of the K|V stream
    constexpr int CG = CF / GROUP;
    static_assert(CG == NSLOT, "a head of phase B is one ring revolution");
    constexpr int GA = NH * 6, GB = NH * CG, GC = NCHUNK * 12, GTOT = GA + GB + GC;
    constexpr int PA = GA * DPG, PB = GB * DPG, PTOT = GTOT * DPG;      // LDS-DMA pieces of this wave per segment
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q31 = lane & 31, h2 = lane >> 5;
    const int batch = blockIdx.x / a.tiles_per_batch, tile = blockIdx.x % a.tiles_per_batch;
    const int r0 = tile * (NW * 32) + wid * 32;                 // first row of this wave inside its batch element
    const bool active = r0 < a.nq;                              // nq % 32 == 0: a wave is whole or absent (absent waves still load and sync)
    const int64_t row = (int64_t)batch * a.nq + (active ? r0 + q31 : q31);
    const uint32_t sbase = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char *)smem);
    const uint32_t fr_addr = sbase + lane * 16;                                      // ring: + slot * 16 KiB + fragment * 1 KiB (immediates)
    const uint32_t rs_addr = sbase + RING_BYTES + wid * RESID_BYTES + lane * 16;     // this wave's residual rows / parked Q, O fragments
    const uint32_t tb_addr = sbase + TAB_OFF + h2 * 16;                              // tables: + table * 3 KiB + column * 4
    const uint32_t voff = (uint32_t)(wid * FRAG + lane * 16);                        // this lane's bytes of this wave's fragments
    const char *segA = a.wq, *segC = a.wo, *segB = a.kv + (int64_t)batch * a.kv_batch_bytes;
    auto mfma_d = [&](const h16x8 &x, const h16x8 &y, const f32x16 &c) __attribute__((always_inline)) {
        if constexpr (DBG & 1) return c;
        else return mfma<F16>(x, y, c);
    };
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if (a.stamps && lane == 0) a.stamps[((int64_t)blockIdx.x * NW + wid) * 8 + i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);

    // ---- LayerNorm, streaming: two lanes per row, one pass over the row, straight into B-operand fragments ----
    // lane (q, h2) holds x[q][16 ks + 8 (j >> 2) + 4 h2 + (j & 3)], j = 0..7, of every k-step (half a row).  The fragments carry
    // d = x - c rounded to 16 bits, c = the mean of the row's first 128 elements (so |d| is a few standard deviations whatever the
    // row's offset); the exact statistics of d (fp32) turn the product into LayerNorm in the Q epilogue:
    //   W' ((d - mu_d) rstd) = rstd (W' d) - (mu_d rstd) rowsum(W')          (gamma is folded into W', beta into the bias)
    // No LDS-DMA is in flight yet: next to one, the compiler waits vmcnt(0) for every ordinary load, one HBM round trip per chunk.
    h16x8 xd[NKS];
    float q_a, q_b;                            // rstd * qscale, mu_d * rstd * qscale
    {
        float tv[3 * D / (NW * 64)];           // the per-column tables ride along (stored to LDS after the row)
#pragma unroll
        for (int i = 0; i < 3 * D / (NW * 64); ++i) tv[i] = a.tabs[tid + i * NW * 64];
        const float *xr = a.x + row * D + 4 * h2;
        constexpr int CH = 8;                  // k-steps per load chunk (16 float4 per lane)
        f32x4 raw[2][2 * CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            raw[0][2 * i] = *reinterpret_cast<const f32x4 *>(xr + 16 * i);
            raw[0][2 * i + 1] = *reinterpret_cast<const f32x4 *>(xr + 16 * i + 8);
        }
        float cshift = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NKS / CH; ++c) {
            if (c + 1 < NKS / CH) {            // next chunk in flight while this one is converted
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    raw[(c + 1) & 1][2 * i] = *reinterpret_cast<const f32x4 *>(xr + 16 * (CH * (c + 1) + i));
                    raw[(c + 1) & 1][2 * i + 1] = *reinterpret_cast<const f32x4 *>(xr + 16 * (CH * (c + 1) + i) + 8);
                }
            }
            if (c == 0) {
                float c0 = 0.f;
#pragma unroll
                for (int i = 0; i < 2 * CH; ++i) c0 += (raw[0][i][0] + raw[0][i][1]) + (raw[0][i][2] + raw[0][i][3]);
                cshift = swap_sum(c0) * (1.0f / (16 * CH));
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = raw[c & 1][2 * i][j] - cshift, e[4 + j] = raw[c & 1][2 * i + 1][j] - cshift;
#pragma unroll
                for (int j = 0; j < 8; ++j) s1 += e[j], s2 += e[j] * e[j];
                xd[CH * c + i] = pack8<F16, true>(e);
                pin(xd[CH * c + i]);
            }
        }
        const float mu = swap_sum(s1) * (1.0f / D);
        const float var = __builtin_fmaxf(swap_sum(s2) * (1.0f / D) - mu * mu, 0.f);
        const float rstd = 1.0f / sqrtf(var + a.eps);
        q_a = rstd * a.qscale;
        q_b = mu * q_a;
        float *tab = reinterpret_cast<float *>(smem + TAB_OFF);
#pragma unroll
        for (int i = 0; i < 3 * D / (NW * 64); ++i) tab[tid + i * NW * 64] = tv[i];
    }
    __syncthreads();
    stamp(1);

    // ---- the LDS-DMA side of the stream: this wave moves piece j (fragment NW j + wid) of every group; within a segment consecutive
    // pieces are 4 KiB apart, so the source is a running (wave-uniform) pointer ----
    const char *sp = segA;                     // source of the next piece
    int pc = 0;                                // pieces issued
    // (no branch in here: the scalar bookkeeping sits between two MFMAs of a batch, and a taken branch costs more than their shadow)
    auto issue_piece = [&](auto si, auto ji) __attribute__((always_inline)) {        // next piece of the stream -> ring slot si, position ji
        constexpr int slot = decltype(si)::value, j = decltype(ji)::value;
        char *dst = smem + slot * (GROUP * FRAG) + (NW * j) * FRAG + wid * FRAG;
        if (!((DBG & 2) && pc >= NSLOT * DPG))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sp + voff), (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        ++pc;
        sp += NW * FRAG;
        sp = pc == PA ? segB : sp;
        sp = pc == PA + PB ? segC : sp;
    };
    static_for<0, NSLOT * DPG>([&](auto t) {                                         // the ring starts full: groups 0 .. NSLOT - 1
        issue_piece(std::integral_constant<int, decltype(t)::value / DPG>{}, std::integral_constant<int, decltype(t)::value % DPG>{});
    });

    // ---- the consumer side ----
    h16x8 FA[4], FB[4];                       // two batches of fragments (the next batch is read while the current one feeds MFMAs)
    h16x8 qf[NKS];                            // Q^T as B fragments, then O^T (same slots); slots of heads < QL are filled at the start of phase C
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // the residual rows of an out-projection chunk as 16 pieces of two FULL 512-byte row segments each: piece u = rows u and u + 16 of this
    // wave (lane l: row u + 16 (l >> 5), floats 4 (l & 31) .. + 3 of the chunk's 128 columns).  LDS image: piece u at u * RPITCH, i.e.
    // row r at (r & 15) * RPITCH + (r >> 4) * 512 -- the 16-byte pad per piece makes the epilogue's column-wise accesses (32 rows at
    // one column offset) conflict-free
    auto issue_resid = [&](int c, auto ui) __attribute__((always_inline)) {
        constexpr int u = decltype(ui)::value;
        const float *src = a.x + ((int64_t)batch * a.nq + (active ? r0 : 0) + u + 16 * h2) * D + 128 * c + 4 * q31;
        char *dst = smem + RING_BYTES + wid * RESID_BYTES + u * RPITCH;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    // One consumer step = batch LB (compile time) of the current loop body; every body starts on a ring revolution (16 batches).
    // The body issues its four MFMAs and calls mid(0), mid(1), mid(2) between them: the requests that keep the stream going ride in
    // the MFMAs' shadows (an MFMA holds the issue port 8 of its 32 cycles) --
    //   mid(0), mid(1)  the next batch's four fragment reads
    //   mid(2)          one LDS-DMA piece of the group three ahead (its slot was freed by the barrier at the end of the previous group)
    //                   and, in phase C, one piece of the chunk's residual rows
    // Before the body: this batch's reads have landed; at a group end (B4 == 3) also the counted wait for this wave's pieces of the
    // next group and the barrier that makes everyone's pieces visible and frees the slot just finished.
    // LATE: the body carries compiler-scheduled vector code (an epilogue, the softmax) -- the next batch is then requested AFTER the
    // body: a register that an in-flight asm ds_read is about to fill looks defined to the register allocator, which under pressure
    // may copy or spill it before the data is there (tools/check_asm_hazards.py).
    // last_iter: the last iteration of phase C, where the groups in flight run out (run-time, wave-uniform).
    auto step_impl = [&](auto lbi, auto cfg, bool first3, bool last_iter, int chunk, h16x8(&cur)[4], h16x8(&nxt)[4], auto &&body) __attribute__((always_inline)) {
        using Cfg = decltype(cfg);
        constexpr int LB = decltype(lbi)::value, B4 = LB % 4, BODY = Cfg::body_batches;
        constexpr bool LATE = Cfg::late, LASTB = LB == BODY - 1;
        lds_wait0(cur);
        if constexpr (B4 == 3) {
            constexpr int gl = LB / 4, groups = BODY / 4;                         // group inside the body
            if (last_iter && gl >= groups - 3) {                                  // stream tail: 1, then 0 groups in flight behind the next one
                if constexpr (gl == groups - 3) vm_wait<DPG>();
                else vm_wait<0>();
            } else {
                if constexpr (Cfg::younger_other_ops(LB, true) != Cfg::younger_other_ops(LB, false)) {
                    if (active) vm_wait<DPG *(NSLOT - 2) + Cfg::younger_other_ops(LB, true)>();
                    else vm_wait<DPG *(NSLOT - 2) + Cfg::younger_other_ops(LB, false)>();
                } else {
                    vm_wait<DPG *(NSLOT - 2) + Cfg::younger_other_ops(LB, false)>();
                }
            }
            if constexpr (!(DBG & 8)) if (!(last_iter && LASTB)) __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NB = (LB + 1) % BODY;                                       // the next batch (of this body, or batch 0 of the next iteration)
        constexpr int nbase = ((NB / 4) % NSLOT) * (GROUP * FRAG) + (NB % 4) * 4 * FRAG;
        auto mid = [&](auto mi) __attribute__((always_inline)) {
            constexpr int m = decltype(mi)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m == 0 && !LATE && !(DBG & 4)) {                        // all four right behind the first MFMA: three MFMAs of cover
                // (ahead of the first MFMA -- four MFMAs of cover -- measured 6 % slower in phase A: the wait at the batch start is not what costs)
                lds_read<nbase>(nxt[0], fr_addr);
                lds_read<nbase + FRAG>(nxt[1], fr_addr);
                lds_read<nbase + 2 * FRAG>(nxt[2], fr_addr);
                lds_read<nbase + 3 * FRAG>(nxt[3], fr_addr);
            }
            if constexpr (m == 2) {
                constexpr int u = (LB + 13) % 16;                                 // piece u % 4 of the group that takes slot u / 4
                // pieces go out at batches 3 .. PTOT + 2 of the stream: not in its first three batches, not in its last 13
                if constexpr (LB < 3) {
                    if (!first3) issue_piece(std::integral_constant<int, u / 4>{}, std::integral_constant<int, u % 4>{});
                } else if constexpr (LB >= BODY - 13) {
                    if (!last_iter) issue_piece(std::integral_constant<int, u / 4>{}, std::integral_constant<int, u % 4>{});
                } else {
                    issue_piece(std::integral_constant<int, u / 4>{}, std::integral_constant<int, u % 4>{});
                }
                if constexpr (Cfg::resid && LB % NKS >= 16 && LB % NKS < 32) issue_resid(LB / NKS, std::integral_constant<int, (LB % NKS >= 16 && LB % NKS < 32) ? LB % NKS - 16 : 0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        body(cur, mid);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LATE) {
            if (!(last_iter && LASTB)) {
                lds_read<nbase>(nxt[0], fr_addr);
                lds_read<nbase + FRAG>(nxt[1], fr_addr);
                lds_read<nbase + 2 * FRAG>(nxt[2], fr_addr);
                lds_read<nbase + 3 * FRAG>(nxt[3], fr_addr);
            }
        }
    };
    const std::integral_constant<int, 0> M0{};
    const std::integral_constant<int, 1> M1{};
    const std::integral_constant<int, 2> M2{};

    // group 0 has landed (this wave's pieces: counted; everyone's: barrier); first batch
    vm_wait<DPG *(NSLOT - 1)>();
    __builtin_amdgcn_s_barrier();
    lds_read<0>(FA[0], fr_addr);
    lds_read<FRAG>(FA[1], fr_addr);
    lds_read<2 * FRAG>(FA[2], fr_addr);
    lds_read<3 * FRAG>(FA[3], fr_addr);

    stamp(2);
    if (a.stamps && lane == 0) a.stamps[((int64_t)blockIdx.x * NW + wid) * 8 + 6] = __builtin_amdgcn_s_memtime();
    // ================================ A: Q^T = W_q' LN(x)^T, head by head ================================
    // The epilogue of head h (acc * q_a - rowsum(W') * q_b + bias' -> 16-bit B fragments of S^T = K Q^T) rides in the MFMA shadows of
    // head h + 1's first batches: accumulators are double-buffered by head parity.  Four units (dh-block blk, half sx); unit u reads
    // its four table vectors at the end of batch 2 u and computes in batch 2 u + 1 (the batch-start wait has retired the reads).
    f32x16 qacc[2][2];                                           // [head parity][dh-block]
    f32x4 tq4[2][4];                                             // table vectors of two units (set = unit parity): bias' (2), row sums (2)
    h16x8 qfr[4];
    auto q_unit_read = [&](auto hi, auto ui) __attribute__((always_inline)) {
        constexpr int h = decltype(hi)::value, blk = decltype(ui)::value >> 1, sx = decltype(ui)::value & 1;
        if constexpr (DBG & 16) return;
        constexpr int set = decltype(ui)::value & 1;
        lds_read<(64 * h + 32 * blk + 16 * sx) * 4>(tq4[set][0], tb_addr);
        lds_read<(64 * h + 32 * blk + 16 * sx + 8) * 4>(tq4[set][1], tb_addr);
        lds_read<(D + 64 * h + 32 * blk + 16 * sx) * 4>(tq4[set][2], tb_addr);
        lds_read<(D + 64 * h + 32 * blk + 16 * sx + 8) * 4>(tq4[set][3], tb_addr);
    };
    // quarter qq of unit u: two of its eight values per call... (row = dh of register r of block blk: 32 blk + (r & 3) + 8 (r >> 2) + 4 h2)
    uint32_t qw[4];
    auto q_unit_quarter = [&](auto hi, auto ui, auto qi) __attribute__((always_inline)) {
        constexpr int h = decltype(hi)::value, u = decltype(ui)::value, blk = u >> 1, sx = u & 1, qq = decltype(qi)::value;
        if constexpr (DBG & 16) return;
        constexpr int set = u & 1;
        if constexpr (qq == 0) asm volatile("" : "+v"(tq4[set][0]), "+v"(tq4[set][1]), "+v"(tq4[set][2]), "+v"(tq4[set][3]));      // behind the batch-start wait (asm order)
        const f32x16 &acc = qacc[h & 1][blk];
        float e0, e1;
        {
            constexpr int j = 2 * qq;
            e0 = __builtin_fmaf(acc[8 * sx + j], q_a, __builtin_fmaf(-tq4[set][2 + (j >> 2)][j & 3], q_b, tq4[set][j >> 2][j & 3]));
            e1 = __builtin_fmaf(acc[8 * sx + j + 1], q_a, __builtin_fmaf(-tq4[set][2 + ((j + 1) >> 2)][(j + 1) & 3], q_b, tq4[set][(j + 1) >> 2][(j + 1) & 3]));
        }
        if (F16) e0 = clamp16(e0), e1 = clamp16(e1);
        qw[qq] = pack2<F16>(e0, e1);
        if constexpr (qq == 3) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w4 = {qw[0], qw[1], qw[2], qw[3]};
            h16x8 fr = __builtin_bit_cast(h16x8, w4);
            pin(fr);
            if constexpr (h < QL) lds_write<(4 * h + u) * FRAG>(rs_addr, fr);
            else {
                qf[4 * h + u] = fr;
                pin(qf[4 * h + u]);
            }
        }
    };
    // one value (v = 0 .. 7) of unit u: the thin slice (<= 5 vector instructions) that hides in one MFMA gap
    float qe = 0.f;
    auto q_unit_value = [&](auto hi, auto ui, auto vi) __attribute__((always_inline)) {
        constexpr int h = decltype(hi)::value, u = decltype(ui)::value, blk = u >> 1, sx = u & 1, v = decltype(vi)::value, set = u & 1;
        if constexpr (DBG & 16) return;
        if constexpr (v == 0) asm volatile("" : "+v"(tq4[set][0]), "+v"(tq4[set][1]), "+v"(tq4[set][2]), "+v"(tq4[set][3]));      // behind a batch-start wait (asm order)
        const f32x16 &acc = qacc[h & 1][blk];
        float e = __builtin_fmaf(acc[8 * sx + v], q_a, __builtin_fmaf(-tq4[set][2 + (v >> 2)][v & 3], q_b, tq4[set][v >> 2][v & 3]));
        if (F16) e = clamp16(e);
        if constexpr (v % 2 == 0) qe = e;
        else qw[v >> 1] = pack2<F16>(qe, e);
        if constexpr (v == 7) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w4 = {qw[0], qw[1], qw[2], qw[3]};
            h16x8 fr = __builtin_bit_cast(h16x8, w4);
            pin(fr);
            if constexpr (h < QL) lds_write<(4 * h + u) * FRAG>(rs_addr, fr);
            else {
                qf[4 * h + u] = fr;
                pin(qf[4 * h + u]);
            }
        }
    };
    static_for<0, NH>([&](auto hi) {
        constexpr int h = decltype(hi)::value, par = h & 1;
        static_for<0, 24>([&](auto li) {
            constexpr int lb = decltype(li)::value, ks = 2 * lb, LB = 24 * h + lb;
            using Cfg = StepCfg<NH * 24, false, false>;
            auto body = [&](h16x8(&f)[4], auto &&mid) __attribute__((always_inline)) {
                // epilogue slices of the previous head: batches 0, 2, 4, 6 end with a unit's table reads, batches 1, 3, 5, 7 compute it
                auto vs = [&](auto si) __attribute__((always_inline)) {
                    constexpr int sl = decltype(si)::value;
                    // gap g = 4 lb + sl of this head: value g - 4 - 8 u of unit u of the previous head in gaps 4 + 8 u .. 11 + 8 u (batches 1 .. 8);
                    // unit u's table vectors are requested a whole batch or more ahead, into the register set its parity names: gaps 0, 4, 12, 20
                    if constexpr (h > 0 && lb < 9) {
                        constexpr int g = 4 * lb + sl;
                        using HP = std::integral_constant<int, (h > 0 ? h - 1 : 0)>;
                        if constexpr (g == 0) q_unit_read(HP{}, std::integral_constant<int, 0>{});
                        if constexpr (g == 4) q_unit_read(HP{}, std::integral_constant<int, 1>{});
                        if constexpr (g == 12) q_unit_read(HP{}, std::integral_constant<int, 2>{});
                        if constexpr (g == 20) q_unit_read(HP{}, std::integral_constant<int, 3>{});
                        if constexpr (g >= 4) q_unit_value(HP{}, std::integral_constant<int, (g >= 4 ? (g - 4) / 8 : 0)>{}, std::integral_constant<int, (g >= 4 ? (g - 4) % 8 : 0)>{});
                    }
                };
                if constexpr (lb == 0) qacc[par][0] = mfma_d(f[0], xd[ks], zero);
                else qacc[par][0] = mfma_d(f[0], xd[ks], qacc[par][0]);
                vs(M0);
                mid(M0);
                if constexpr (lb == 0) qacc[par][1] = mfma_d(f[1], xd[ks], zero);
                else qacc[par][1] = mfma_d(f[1], xd[ks], qacc[par][1]);
                vs(M1);
                mid(M1);
                qacc[par][0] = mfma_d(f[2], xd[ks + 1], qacc[par][0]);
                vs(M2);
                mid(M2);
                qacc[par][1] = mfma_d(f[3], xd[ks + 1], qacc[par][1]);
                vs(std::integral_constant<int, 3>{});
            };
            if constexpr (LB % 2 == 0) step_impl(std::integral_constant<int, LB>{}, Cfg{}, h == 0, false, 0, FA, FB, body);
            else step_impl(std::integral_constant<int, LB>{}, Cfg{}, h == 0, false, 0, FB, FA, body);
        });
    });
    // the last head's epilogue has no projection left to hide under (the next batch's fragment reads are in flight: requested by the last step)
    static_for<0, 4>([&](auto ui) {
        q_unit_read(std::integral_constant<int, NH - 1>{}, ui);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tq4[decltype(ui)::value & 1][0]), "+v"(tq4[decltype(ui)::value & 1][1]), "+v"(tq4[decltype(ui)::value & 1][2]), "+v"(tq4[decltype(ui)::value & 1][3])::"memory");
        static_for<0, 4>([&](auto qi) { q_unit_quarter(std::integral_constant<int, NH - 1>{}, ui, qi); });
        __builtin_amdgcn_sched_barrier(0);
    });

    stamp(3);
    if (a.stamps && lane == 0) a.stamps[((int64_t)blockIdx.x * NW + wid) * 8 + 7] = __builtin_amdgcn_s_memtime();
    // keys past nkv exist in the LAST key block only (32 (KC - 1) < nkv <= 32 KC, checked by the host): its score accumulators start at
    // -inf for those keys instead of 0, so the softmax needs no masking code at all
    f32x16 cpart;
#pragma unroll
    for (int r = 0; r < 16; ++r) cpart[r] = (32 * (KC - 1) + (r & 3) + 8 * (r >> 2) + 4 * h2) < a.nkv ? 0.f : -INFINITY;
    // ================================ B: attention, software-pipelined over the heads ================================
    // A head's softmax is ~440 vector instructions against 56 MFMAs, and a lone wave per SIMD overlaps the two only inside its own
    // instruction stream.  So the K|V stream is ordered K0 K1 V0 K2 V1 ... K11 V10 V11 (half-blocks of 8 batches: 7 key blocks + 1 pad)
    // and head h's softmax rides in the MFMA shadows of its neighbours' products:
    //   under  O(h-1) = V(h-1)^T P(h-1)   the row maximum of S(h)          (4 scores per MFMA)
    //   under  S(h+1) = K(h+1) Q(h+1)^T   exp2 / row sum / pack of P(h)    (4 scores per MFMA)
    // Scores and packed P are double-buffered by head parity.  Each batch issues: MFMA, vector slice, the stream's requests (mid), x 4.
    constexpr int B0 = 0;                                       // phase B is one 192-batch body for the step bookkeeping
    using CfgB = StepCfg<24 * 8, false, false>;
    f32x16 sacc[2][KC];
    uint32_t pw[2][KC][8];                                      // packed P of a head: [key block][word]: words 0..3 = k-step 0, 4..7 = k-step 1
    f32x16 o0, o1;
    float mrow[2] = {0.f, 0.f}, lsum[2] = {0.f, 0.f}, linv[2] = {0.f, 0.f};
    auto pfrag = [&](int par, int kb, int t) __attribute__((always_inline)) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 u = {pw[par][kb][4 * t], pw[par][kb][4 * t + 1], pw[par][kb][4 * t + 2], pw[par][kb][4 * t + 3]};
        return __builtin_bit_cast(h16x8, u);
    };
    // vector slices (i = 0 .. 4 KC - 1): four scores of key block i / 4, registers 4 (i % 4) .. + 3
    auto sm_max = [&](auto hi, auto ii) __attribute__((always_inline)) {
        constexpr int h = decltype(hi)::value, i = decltype(ii)::value, par = h & 1, kb = i / 4, q4 = (i % 4) * 4;
        if constexpr (DBG & 16) return;
        const f32x16 &sv = sacc[par][kb];
        float m = i == 0 ? -INFINITY : mrow[par];
        m = __builtin_fmaxf(__builtin_fmaxf(m, sv[q4]), sv[q4 + 1]);
        m = __builtin_fmaxf(__builtin_fmaxf(m, sv[q4 + 2]), sv[q4 + 3]);
        if constexpr (i == 4 * KC - 1) m = swap_max(m);
        mrow[par] = m;
    };
    auto sm_exp = [&](auto hi, auto ii) __attribute__((always_inline)) {
        constexpr int h = decltype(hi)::value, i = decltype(ii)::value, par = h & 1, kb = i / 4, q = i % 4;
        if constexpr (DBG & 16) return;
        const f32x16 &sv = sacc[par][kb];
        const float m = mrow[par];
        const float e0 = __builtin_amdgcn_exp2f(sv[4 * q] - m), e1 = __builtin_amdgcn_exp2f(sv[4 * q + 1] - m);
        const float e2 = __builtin_amdgcn_exp2f(sv[4 * q + 2] - m), e3 = __builtin_amdgcn_exp2f(sv[4 * q + 3] - m);
        const float part = (e0 + e1) + (e2 + e3);
        lsum[par] = i == 0 ? part : lsum[par] + part;
        pw[par][kb][2 * q] = pack2<F16>(e0, e1);
        pw[par][kb][2 * q + 1] = pack2<F16>(e2, e3);
        if constexpr (i == 4 * KC - 1) linv[par] = 1.0f / swap_sum(lsum[par]);
    };
    auto normalize = [&](auto hi) __attribute__((always_inline)) {      // O(h) / l -> B fragments of out^T = W_o O^T, into the slots of head h's Q
        constexpr int h = decltype(hi)::value;
        if constexpr (DBG & 16) return;
        h16x8 fr[4];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const f32x16 &o = blk ? o1 : o0;
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = o[8 * sx + j] * linv[h & 1];
                fr[2 * blk + sx] = pack8<F16>(e);
                pin(fr[2 * blk + sx]);
            }
        }
        if constexpr (h < QL) {
            lds_write<(4 * h + 0) * FRAG>(rs_addr, fr[0]);
            lds_write<(4 * h + 1) * FRAG>(rs_addr, fr[1]);
            lds_write<(4 * h + 2) * FRAG>(rs_addr, fr[2]);
            lds_write<(4 * h + 3) * FRAG>(rs_addr, fr[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qf[4 * h + i] = fr[i];
                pin(qf[4 * h + i]);
            }
        }
    };
    // one half-block of the stream = 8 batches; `hb` numbers the half-blocks of phase B (0 .. 23)
    auto run_batch = [&](auto lbi, auto &&body) __attribute__((always_inline)) {
        constexpr int LB = decltype(lbi)::value;
        if constexpr (LB % 2 == 0) step_impl(std::integral_constant<int, B0 + LB>{}, CfgB{}, false, false, 0, FA, FB, body);
        else step_impl(std::integral_constant<int, B0 + LB>{}, CfgB{}, false, false, 0, FB, FA, body);
    };
    h16x8 qh[4];                                                // Q^T fragments of the head whose scores are being computed
    auto load_q = [&](auto hi) __attribute__((always_inline)) {
        constexpr int h = decltype(hi)::value;
        if constexpr (h < QL) {                                 // parked Q: back from LDS (no vector code between the reads and their wait)
            lds_read<(4 * h + 0) * FRAG>(qh[0], rs_addr);
            lds_read<(4 * h + 1) * FRAG>(qh[1], rs_addr);
            lds_read<(4 * h + 2) * FRAG>(qh[2], rs_addr);
            lds_read<(4 * h + 3) * FRAG>(qh[3], rs_addr);
            lds_wait0(qh);                                      // (also retires the next batch's reads, requested just before: harmless)
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) qh[i] = qf[4 * h + i];
        }
    };
    // S(hs) = K(hs) Q(hs)^T over half-block hb, with softmax slices of head hv (exp pass) in the shadows when hv >= 0
    auto s_block = [&](auto hbi, auto hsi, auto hvi) __attribute__((always_inline)) {
        constexpr int hb = decltype(hbi)::value, hs = decltype(hsi)::value, hv = decltype(hvi)::value, par = hs & 1;
        load_q(hsi);
        static_for<0, 8>([&](auto li) {
            constexpr int lb = decltype(li)::value;
            run_batch(std::integral_constant<int, 8 * hb + lb>{}, [&](h16x8(&f)[4], auto &&mid) __attribute__((always_inline)) {
                auto vs = [&](auto si) __attribute__((always_inline)) {
                    constexpr int i = 4 * lb + decltype(si)::value;
                    if constexpr (hv >= 0 && i < 4 * KC) sm_exp(std::integral_constant<int, hv < 0 ? 0 : hv>{}, std::integral_constant<int, i < 4 * KC ? i : 0>{});
                };
                if constexpr (lb < KC) {                        // S^T block lb = K rows 32 lb .. + 31 against Q^T (4 k-steps over dh)
                    sacc[par][lb] = mfma_d(f[0], qh[0], lb == KC - 1 ? cpart : zero);
                    vs(M0);
                    mid(M0);
                    sacc[par][lb] = mfma_d(f[1], qh[1], sacc[par][lb]);
                    vs(M1);
                    mid(M1);
                    sacc[par][lb] = mfma_d(f[2], qh[2], sacc[par][lb]);
                    vs(M2);
                    mid(M2);
                    sacc[par][lb] = mfma_d(f[3], qh[3], sacc[par][lb]);
                    vs(std::integral_constant<int, 3>{});
                } else {                                        // padding of the half-block to whole groups
                    mid(M0);
                    mid(M1);
                    mid(M2);
                }
            });
        });
    };
    // O(hp) = V(hp)^T P(hp) over half-block hb, with the row maximum of head hv in the shadows when hv >= 0; normalises O(hp) at the end
    auto pv_block = [&](auto hbi, auto hpi, auto hvi) __attribute__((always_inline)) {
        constexpr int hb = decltype(hbi)::value, hp = decltype(hpi)::value, hv = decltype(hvi)::value, par = hp & 1;
        static_for<0, 8>([&](auto li) {
            constexpr int lb = decltype(li)::value;
            run_batch(std::integral_constant<int, 8 * hb + lb>{}, [&](h16x8(&f)[4], auto &&mid) __attribute__((always_inline)) {
                auto vs = [&](auto si) __attribute__((always_inline)) {
                    constexpr int i = 4 * lb + decltype(si)::value;
                    if constexpr (hv >= 0 && i < 4 * KC) sm_max(std::integral_constant<int, hv < 0 ? 0 : hv>{}, std::integral_constant<int, i < 4 * KC ? i : 0>{});
                };
                if constexpr (lb < KC) {                        // O^T += V^T P^T over key block lb (2 k-steps x 2 dh-blocks)
                    if constexpr (lb == 0) o0 = mfma_d(f[0], pfrag(par, lb, 0), zero);
                    else o0 = mfma_d(f[0], pfrag(par, lb, 0), o0);
                    vs(M0);
                    mid(M0);
                    if constexpr (lb == 0) o1 = mfma_d(f[1], pfrag(par, lb, 0), zero);
                    else o1 = mfma_d(f[1], pfrag(par, lb, 0), o1);
                    vs(M1);
                    mid(M1);
                    o0 = mfma_d(f[2], pfrag(par, lb, 1), o0);
                    vs(M2);
                    mid(M2);
                    o1 = mfma_d(f[3], pfrag(par, lb, 1), o1);
                    vs(std::integral_constant<int, 3>{});
                } else {
                    mid(M0);
                    mid(M1);
                    mid(M2);
                    normalize(hpi);
                }
            });
        });
    };
    constexpr std::integral_constant<int, -1> NONE{};
    s_block(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, NONE);              // K0
    static_for<0, 4 * KC>([&](auto i) { sm_max(std::integral_constant<int, 0>{}, i); });            // row maximum of head 0: no product to hide under
    s_block(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});     // K1 with exp(S0)
    static_for<1, NH>([&](auto hi) {
        constexpr int h = decltype(hi)::value;
        // V(h-1) with max(S(h)); then K(h+1) with exp(S(h)) -- or, for the last head, the exp pass alone
        pv_block(std::integral_constant<int, 2 * h>{}, std::integral_constant<int, h - 1>{}, hi);
        if constexpr (h + 1 < NH) s_block(std::integral_constant<int, 2 * h + 1>{}, std::integral_constant<int, h + 1>{}, hi);
        else static_for<0, 4 * KC>([&](auto i) { sm_exp(hi, i); });
    });
    pv_block(std::integral_constant<int, 23>{}, std::integral_constant<int, NH - 1>{}, NONE);       // V11

    stamp(4);
    // ================================ C: out^T = W_o O^T in chunks of 128 columns, + b_o + residual ================================
    // the parked O fragments of heads < QL come back first (their LDS area is about to receive the residual rows)
    static_for<0, 4 * QL>([&](auto fi) { lds_read<decltype(fi)::value * FRAG>(qf[decltype(fi)::value], rs_addr); });
#pragma unroll
    for (int h = 0; h < QL; ++h) lds_wait0_4(qf[4 * h], qf[4 * h + 1], qf[4 * h + 2], qf[4 * h + 3]);
    // The epilogue of chunk c rides in the MFMA shadows of chunk c + 1's first 16 batches (accumulators double-buffered by chunk parity):
    //   E1, column block nb (batches 2 nb, 2 nb + 1): out^T (lane = row, registers = columns) + bias is added INTO the residual tile in LDS,
    //       column-wise (the residual pieces were requested >= 16 batches before the chunk ended: every ring wait since then covered them);
    //   E2, piece group v (batches 8 + 2 v, 9 + 2 v): the tile leaves row-wise, every store instruction writes two full 512-byte row segments.
    // A unit's LDS reads are issued at the end of its first batch and consumed in its second (the batch-start wait has retired them).
    // The residual pieces of chunk c + 1 follow in batches 16 .. 31 (StepCfg::resid), once the tile of chunk c has left.
    f32x16 oacc[2][NBC];
    f32x4 te[8];
    const uint32_t tq = rs_addr - lane * 16 + (q31 & 15) * RPITCH + (q31 >> 4) * 512 + h2 * 16;     // this lane's row in the tile, + 16 B for h2
    float *const obase = a.out + ((int64_t)batch * a.nq + r0 + 16 * h2) * D + 4 * q31;
    auto e1_read = [&](auto ci, auto ni) __attribute__((always_inline)) {
        constexpr int c = decltype(ci)::value, nb = decltype(ni)::value;
        if constexpr (DBG & 16) return;
        static_for<0, 4>([&](auto gi) {
            constexpr int g = decltype(gi)::value;
            lds_read<(32 * nb + 8 * g) * 4>(te[g], tq);
            lds_read<(2 * D + 128 * c + 32 * nb + 8 * g) * 4>(te[4 + g], tb_addr);
        });
    };
    auto e1_quarter = [&](auto ci, auto ni, auto gi) __attribute__((always_inline)) {
        constexpr int c = decltype(ci)::value, nb = decltype(ni)::value, g = decltype(gi)::value;
        if constexpr (DBG & 16) return;
        if constexpr (g == 0) asm volatile("" : "+v"(te[0]), "+v"(te[1]), "+v"(te[2]), "+v"(te[3]), "+v"(te[4]), "+v"(te[5]), "+v"(te[6]), "+v"(te[7]));
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (oacc[c & 1][nb][4 * g + i] + te[4 + g][i]) + te[g][i];
        lds_write<(32 * nb + 8 * g) * 4>(tq, o);
    };
    auto e2_read = [&](auto vi) __attribute__((always_inline)) {
        constexpr int v4 = decltype(vi)::value;
        if constexpr (DBG & 16) return;
        static_for<0, 4>([&](auto ji) { lds_read<(4 * v4 + decltype(ji)::value) * RPITCH>(te[decltype(ji)::value], rs_addr); });
    };
    auto e2_store = [&](auto ci, auto vi, auto ji) __attribute__((always_inline)) {
        constexpr int c = decltype(ci)::value, v4 = decltype(vi)::value, j = decltype(ji)::value;
        if constexpr (DBG & 16) return;
        if constexpr (j == 0) asm volatile("" : "+v"(te[0]), "+v"(te[1]), "+v"(te[2]), "+v"(te[3]));
        if (active) __builtin_nontemporal_store(te[j], reinterpret_cast<f32x4 *>(obase + (int64_t)(4 * v4 + j) * D + 128 * c));
    };
    using CfgC = StepCfg<NCHUNK * NKS, true, false>;
    static_for<0, NCHUNK>([&](auto ci) {
        constexpr int c = decltype(ci)::value;
        static_for<0, NKS>([&](auto ki) {
            constexpr int ks = decltype(ki)::value, LB = NKS * c + ks;
            auto body = [&](h16x8(&f)[4], auto &&mid) __attribute__((always_inline)) {
                auto vs = [&](auto si) __attribute__((always_inline)) {       // epilogue slices of chunk c - 1
                    constexpr int sl = decltype(si)::value;
                    if constexpr (c > 0 && ks < 16) {
                        using CP = std::integral_constant<int, (c > 0 ? c - 1 : 0)>;
                        constexpr int u = (ks % 8) / 2;
                        if constexpr (ks < 8) {
                            if constexpr (ks % 2 == 1) e1_quarter(CP{}, std::integral_constant<int, u>{}, si);
                            else if constexpr (sl == 3) e1_read(CP{}, std::integral_constant<int, u>{});
                        } else {
                            if constexpr (ks % 2 == 1) e2_store(CP{}, std::integral_constant<int, u>{}, si);
                            else if constexpr (sl == 3) e2_read(std::integral_constant<int, u>{});
                        }
                    }
                };
                static_for<0, NBC>([&](auto ni) {
                    constexpr int nb = decltype(ni)::value;
                    if constexpr (ks == 0) oacc[c & 1][nb] = mfma_d(f[nb], qf[ks], zero);
                    else oacc[c & 1][nb] = mfma_d(f[nb], qf[ks], oacc[c & 1][nb]);
                    vs(ni);
                    if constexpr (nb < 3) mid(std::integral_constant<int, nb>{});
                });
            };
            if constexpr (LB % 2 == 0) step_impl(std::integral_constant<int, LB>{}, CfgC{}, false, true, c, FA, FB, body);
            else step_impl(std::integral_constant<int, LB>{}, CfgC{}, false, true, c, FB, FA, body);
        });
    });
    // the last chunk's epilogue has nothing left to hide under
    static_for<0, NBC>([&](auto ni) {
        e1_read(std::integral_constant<int, NCHUNK - 1>{}, ni);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(te[0]), "+v"(te[1]), "+v"(te[2]), "+v"(te[3]), "+v"(te[4]), "+v"(te[5]), "+v"(te[6]), "+v"(te[7])::"memory");
        static_for<0, 4>([&](auto gi) { e1_quarter(std::integral_constant<int, NCHUNK - 1>{}, ni, gi); });
        __builtin_amdgcn_sched_barrier(0);
    });
    static_for<0, 4>([&](auto vi) {
        e2_read(vi);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(te[0]), "+v"(te[1]), "+v"(te[2]), "+v"(te[3])::"memory");
        static_for<0, 4>([&](auto ji) { e2_store(std::integral_constant<int, NCHUNK - 1>{}, vi, ji); });
        __builtin_amdgcn_sched_barrier(0);
    });
    stamp(5);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K|V projection of the kv tokens straight into fragment order.  One workgroup per (key block, head, batch): K^T block = W_k kv^T
// (accumulator = the K fragment of the fused kernel's S^T = K Q^T: registers 8 s .. 8 s + 7 of dh-block b are fragment 2 b + s) and
// V block = kv W_v^T (registers 8 t .. 8 t + 7 of dh-block b are the V^T fragment (t, b)).  8 waves split the 48 k-steps, sums via LDS.
// ---------------------------------------------------------------------------------------------------------------------------------
struct KvArgs {
    const float *kv;        // [B, nkv, D] fp32
    const char *wk, *wv;    // packed [ks][24 row blocks] fragments
    const float *bk, *bv;   // [D]
    char *out;              // [B][NH][CF] fragments
    int nkv, kc, cf;
};

constexpr int KVW = 8;                       // waves per workgroup of the K|V projection: 6 k-steps each, every load of a wave in flight at once

template <bool F16> __global__ void __launch_bounds__(KVW * 64) k_ca_kvproj(KvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [wave][block][register][lane]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r31 = lane & 31, h2 = lane >> 5;
    const int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int key = 32 * kb + r31;
    const bool valid = key < a.nkv;
    const float *kr = a.kv + ((int64_t)b * a.nkv + (valid ? key : 0)) * D + 4 * h2;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 kt0 = zero, kt1 = zero, v0 = zero, v1 = zero;
    constexpr int NS = NKS / KVW;
    f32x4 x0[NS], x1[NS];
    h16x8 wk0[NS], wk1[NS], wv0[NS], wv1[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int ks = wid + KVW * i;
        x0[i] = *reinterpret_cast<const f32x4 *>(kr + 16 * ks);
        x1[i] = *reinterpret_cast<const f32x4 *>(kr + 16 * ks + 8);
        const int64_t fo = ((int64_t)(ks * 24 + 2 * h) * 64 + lane) * 16;
        wk0[i] = *reinterpret_cast<const h16x8 *>(a.wk + fo);
        wk1[i] = *reinterpret_cast<const h16x8 *>(a.wk + fo + FRAG);
        wv0[i] = *reinterpret_cast<const h16x8 *>(a.wv + fo);
        wv1[i] = *reinterpret_cast<const h16x8 *>(a.wv + fo + FRAG);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = valid ? x0[i][j] : 0.f, e[4 + j] = valid ? x1[i][j] : 0.f;
        const h16x8 fkv = pack8<F16, true>(e);
        kt0 = mfma<F16>(wk0[i], fkv, kt0);
        kt1 = mfma<F16>(wk1[i], fkv, kt1);
        v0 = mfma<F16>(fkv, wv0[i], v0);
        v1 = mfma<F16>(fkv, wv1[i], v1);
    }
    float *mine = red + (size_t)wid * 4 * 16 * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        mine[(0 * 16 + r) * 64 + lane] = kt0[r];
        mine[(1 * 16 + r) * 64 + lane] = kt1[r];
        mine[(2 * 16 + r) * 64 + lane] = v0[r];
        mine[(3 * 16 + r) * 64 + lane] = v1[r];
    }
    __syncthreads();
    if (wid >= 4) return;
    // wave w finishes block w: 0, 1 = K^T dh-blocks, 2, 3 = V dh-blocks
    float s[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < KVW; ++w) t += red[((size_t)(w * 4 + wid) * 16 + r) * 64 + lane];
        s[r] = t;
    }
    // stream order of the fused kernel's phase B: K0 K1 V0 K2 V1 ... K11 V10 V11, half-blocks of 32 fragments
    const int kpos = h < 2 ? h : 2 * h - 1, vpos = h < NH - 1 ? 2 + 2 * h : 2 * NH - 1;
    char *segk = a.out + ((int64_t)b * 2 * NH + kpos) * 32 * FRAG, *segv = a.out + ((int64_t)b * 2 * NH + vpos) * 32 * FRAG;
    if (wid < 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] += a.bk[64 * h + 32 * wid + (r & 3) + 8 * (r >> 2) + 4 * h2];       // row of the K^T block = dh
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = s[8 * sp + j];
            *reinterpret_cast<h16x8 *>(segk + (int64_t)(kb * 4 + 2 * wid + sp) * FRAG + lane * 16) = pack8<F16, true>(e);
        }
    } else {
        const int blk = wid - 2;
        const float bv = a.bv[64 * h + 32 * blk + r31];                                                       // column of the V block = dh
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = s[8 * t + j] + bv;
            *reinterpret_cast<h16x8 *>(segv + (int64_t)(kb * 4 + 2 * t + blk) * FRAG + lane * 16) = pack8<F16, true>(e);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight packing (once per weights version): W [N rows, D] fp32 (row scale optional) -> fragments of 32 rows x 16 k with the k
// permutation pi; fragment f of `order`: 0  W_q' [head][ks][blk] (row block 2 head + blk), 1  W_o [chunk][ks][nb] (row block 4 chunk + nb),
// 2  K|V weights [ks][row block]
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool F16> __global__ void __launch_bounds__(256) k_ca_pack(const float *__restrict__ w, const float *__restrict__ colscale, int order,
                                                                     char *__restrict__ out) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;        // 1152 fragments
    int ks, rb;
    if (order == 0) { const int hh = f / 96, rem = f % 96; ks = rem >> 1; rb = 2 * hh + (rem & 1); }
    else if (order == 1) { const int c = f / 192, rem = f % 192; ks = rem >> 2; rb = 4 * c + (rem & 3); }
    else { ks = f / 24; rb = f % 24; }
    const int n = 32 * rb + (lane & 31), hb = lane >> 5;
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * ks + 8 * (j >> 2) + 4 * hb + (j & 3);
        e[j] = w[(int64_t)n * D + k] * (colscale ? colscale[k] : 1.0f);
    }
    *reinterpret_cast<h16x8 *>(out + (int64_t)f * FRAG + lane * 16) = pack8<F16, true>(e);
}
// Per-row constants of the Q projection, one wave per row n: bias' = (b_n + W_n . beta) * qscale (LayerNorm's shift folded in, scaled like
// Q) and the row sum of the ROUNDED W'_n = round16(W_n * gamma) (the fused kernel subtracts mu * rowsum: it must be the sum of exactly the
// operand values the MFMA sees)
template <bool F16> __global__ void __launch_bounds__(256) k_ca_row_consts(const float *__restrict__ w, const float *__restrict__ b,
                                                                           const float *__restrict__ gamma, const float *__restrict__ beta, float qscale,
                                                                           float *__restrict__ bias_out, float *__restrict__ sum_out) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    float sb = 0.f, sw = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float wv = w[(int64_t)n * D + k];
        sb += wv * beta[k];
        float wg = wv * gamma[k];
        if (F16) wg = (float)(_Float16)clamp16(wg);
        else wg = (float)(__bf16)wg;
        sw += wg;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) sb += __shfl_xor(sb, o), sw += __shfl_xor(sw, o);
    if (lane == 0) {
        bias_out[n] = (b[n] + sb) * qscale;
        sum_out[n] = sw;
    }
}

constexpr size_t PK_W = (size_t)D * D * 2;                    // one packed matrix
constexpr size_t PK_WQ = 0, PK_WO = PK_W, PK_WK = 2 * PK_W, PK_WV = 3 * PK_W, PK_TABS = 4 * PK_W, PK_BK = PK_TABS + 3 * D * 4, PK_BV = PK_BK + D * 4,
                 PK_TOTAL = PK_BV + D * 4;
constexpr float QSCALE = 1.4426950408889634f * 0.125f;        // log2(e) / sqrt(64)

inline int kc_of(int nkv) { return (nkv + 31) / 32; }
inline int cf_of(int kc) { return (8 * kc + GROUP - 1) / GROUP * GROUP; }

LvqLdsOnce g_lds_once;

#ifdef CA_DEBUG_VARIANTS
template <int DBG> void launch_dbg(const CaArgs &a, int64_t nwg, hipStream_t st) {
    hipFuncSetAttribute((const void *)k_ca_fused<true, 7, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL((k_ca_fused<true, 7, DBG>), dim3((unsigned)nwg), dim3(NW * 64), LDS_BYTES, st, a);
}
#endif
template <bool F16> int launch_fused(const CaArgs &a, int kc, int64_t nwg, hipStream_t st) {
#ifdef CA_DEBUG_VARIANTS
    if (lvq_tune().ca_fused_variant) {
        switch (lvq_tune().ca_fused_variant) {
            case 2: launch_dbg<2>(a, nwg, st); return LVQ_OK;
            case 8: launch_dbg<8>(a, nwg, st); return LVQ_OK;
            case 10: launch_dbg<10>(a, nwg, st); return LVQ_OK;
            case 16: launch_dbg<16>(a, nwg, st); return LVQ_OK;
        }
    }
#endif
    switch (kc) {
#define CASE(K) case K: hipLaunchKernelGGL((k_ca_fused<F16, K>), dim3((unsigned)nwg), dim3(NW * 64), LDS_BYTES, st, a); return LVQ_OK;
        CASE(7)
#undef CASE
    }
    return LVQ_EUNSUPPORTED;
}

}  // namespace

extern "C" int lvq_ca_fused_ok(int batch, int nq, int nkv, int d, int n_heads) {
    return batch >= 1 && d == D && n_heads == NH && nq >= 32 && nq % 32 == 0 && nkv > 192 && nkv <= 224;   // KC = 7 instantiated
}
extern "C" size_t lvq_ca_fused_packed_bytes(int d, int n_heads) { return (d == D && n_heads == NH) ? PK_TOTAL : 0; }
extern "C" size_t lvq_ca_fused_workspace_bytes(int batch, int nq, int nkv, int d, int n_heads) {
    if (!lvq_ca_fused_ok(batch, nq, nkv, d, n_heads)) return 0;
    return (size_t)batch * NH * cf_of(7) * FRAG + 256;
}

extern "C" int lvq_ca_fused_pack(const float *ln_gamma, const float *ln_beta, const float *in_proj_w, const float *in_proj_b, const float *out_w,
                                 const float *out_b, int d, int n_heads, int f16, void *packed, size_t packed_bytes, lvq_stream_t stream) {
    if (d != D || n_heads != NH) return LVQ_EUNSUPPORTED;
    if (!ln_gamma || !ln_beta || !in_proj_w || !in_proj_b || !out_w || !out_b || !packed) return LVQ_EINVAL;
    if (packed_bytes < PK_TOTAL) return LVQ_EWORKSPACE;
    hipStream_t st = lvq_s(stream);
    char *p = (char *)packed;
    const int nblk = D * D / (32 * 16) / 4;                   // 1152 fragments, 4 per workgroup
    if (f16) {
        hipLaunchKernelGGL((k_ca_pack<true>), dim3(nblk), dim3(256), 0, st, in_proj_w, ln_gamma, 0, p + PK_WQ);
        hipLaunchKernelGGL((k_ca_pack<true>), dim3(nblk), dim3(256), 0, st, out_w, (const float *)nullptr, 1, p + PK_WO);
        hipLaunchKernelGGL((k_ca_pack<true>), dim3(nblk), dim3(256), 0, st, in_proj_w + (size_t)D * D, (const float *)nullptr, 2, p + PK_WK);
        hipLaunchKernelGGL((k_ca_pack<true>), dim3(nblk), dim3(256), 0, st, in_proj_w + (size_t)2 * D * D, (const float *)nullptr, 2, p + PK_WV);
    } else {
        hipLaunchKernelGGL((k_ca_pack<false>), dim3(nblk), dim3(256), 0, st, in_proj_w, ln_gamma, 0, p + PK_WQ);
        hipLaunchKernelGGL((k_ca_pack<false>), dim3(nblk), dim3(256), 0, st, out_w, (const float *)nullptr, 1, p + PK_WO);
        hipLaunchKernelGGL((k_ca_pack<false>), dim3(nblk), dim3(256), 0, st, in_proj_w + (size_t)D * D, (const float *)nullptr, 2, p + PK_WK);
        hipLaunchKernelGGL((k_ca_pack<false>), dim3(nblk), dim3(256), 0, st, in_proj_w + (size_t)2 * D * D, (const float *)nullptr, 2, p + PK_WV);
    }
    float *tabs = (float *)(p + PK_TABS);
    if (f16) hipLaunchKernelGGL((k_ca_row_consts<true>), dim3(D / 4), dim3(256), 0, st, in_proj_w, in_proj_b, ln_gamma, ln_beta, QSCALE, tabs, tabs + D);
    else hipLaunchKernelGGL((k_ca_row_consts<false>), dim3(D / 4), dim3(256), 0, st, in_proj_w, in_proj_b, ln_gamma, ln_beta, QSCALE, tabs, tabs + D);
    if (hipMemcpyAsync(tabs + 2 * D, out_b, D * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return LVQ_ELAUNCH;
    if (hipMemcpyAsync(p + PK_BK, in_proj_b + D, 2 * D * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return LVQ_ELAUNCH;
    return lvq_launch_status();
}

extern "C" int lvq_ca_fused(const float *q, const float *kv, const void *packed, float eps, int batch, int nq, int nkv, int d, int n_heads, int f16,
                            float *out, void *ws, size_t ws_bytes, lvq_stream_t stream) {
    if (!lvq_ca_fused_ok(batch, nq, nkv, d, n_heads)) return LVQ_EUNSUPPORTED;
    if (!q || !kv || !packed || !out || !ws) return LVQ_EINVAL;
    if (ws_bytes < lvq_ca_fused_workspace_bytes(batch, nq, nkv, d, n_heads)) return LVQ_EWORKSPACE;
    hipStream_t st = lvq_s(stream);
    const char *p = (const char *)packed;
    const int kc = kc_of(nkv), cf = cf_of(7);       // K / V^T fragments of kc key blocks in a 64-fragment segment per (batch, head)
    KvArgs ka{kv, p + PK_WK, p + PK_WV, (const float *)(p + PK_BK), (const float *)(p + PK_BV), (char *)ws, nkv, kc, cf};
    constexpr int KV_LDS = KVW * 4 * 16 * 64 * 4;
    static LvqLdsOnce kv_once;
    if (!lvq_ensure_lds(kv_once, {(const void *)k_ca_kvproj<true>, (const void *)k_ca_kvproj<false>}, KV_LDS)) return LVQ_ELAUNCH;
    if (f16) hipLaunchKernelGGL((k_ca_kvproj<true>), dim3(kc, NH, batch), dim3(KVW * 64), KV_LDS, st, ka);
    else hipLaunchKernelGGL((k_ca_kvproj<false>), dim3(kc, NH, batch), dim3(KVW * 64), KV_LDS, st, ka);
    if (!lvq_ensure_lds(g_lds_once, {(const void *)k_ca_fused<true, 7>, (const void *)k_ca_fused<false, 7>}, LDS_BYTES))
        return LVQ_ELAUNCH;
    CaArgs a;
    a.x = q;
    a.out = out;
    a.wq = p + PK_WQ;
    a.wo = p + PK_WO;
    a.kv = (const char *)ws;
    a.tabs = (const float *)(p + PK_TABS);
    a.nq = nq;
    a.nkv = nkv;
    a.tiles_per_batch = (nq + NW * 32 - 1) / (NW * 32);
    a.kv_batch_bytes = (int64_t)NH * cf * FRAG;
    a.eps = eps;
    a.qscale = QSCALE;
    a.stamps = (unsigned long long *)(uintptr_t)lvq_tune().ca_fused_stamps;     // diagnostics (include/lvq.h: lvq_tuning)
    const int64_t nwg = (int64_t)batch * a.tiles_per_batch;
    const int rc = f16 ? launch_fused<true>(a, kc, nwg, st) : launch_fused<false>(a, kc, nwg, st);
    if (rc != LVQ_OK) return rc;
    return lvq_launch_status();
}
