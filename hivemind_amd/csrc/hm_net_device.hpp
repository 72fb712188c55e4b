// hm_net_device.hpp — device side of the fused RISEv3 forward (hm_net.hip): fragment helpers and the one-position-per-workgroup body
// `narrow_position`, shared by the stand-alone forward kernels (hm_net.hip) and the evaluator role of the single-launch search
// (hm_search.hip: k_rollout).  See hm_net.hip for the design notes.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hivemind_amd.h"
#include "hm_queue.hpp"
#include "hm_policy.hpp"

#ifndef HM_NPF_PROJ
#define HM_NPF_PROJ 0      // the same for the 1x1 projection of the 384-channel variant (its accumulators are live across the chunks)
#endif
#ifndef HM_NPF_WIDE4
#define HM_NPF_WIDE4 24           // the same for the 4-wave form (narrow_position4: 512 registers per lane, two accumulators per fragment)
#endif
#ifndef HM_NPF_PROJ4
#define HM_NPF_PROJ4 0            // 4-wave form, 384-channel variant: 20 = the projection tiles of a full chunk (320 expanded channels, 20 k-steps) as one fragment stream (project_stream2) — measured: 207 spilled VGPRs, not used
#endif
#ifndef HM_NET_PREFETCH
#define HM_NET_PREFETCH 1         // 4-wave form, narrow trunks (C <= 128): the expand GEMM's first weight fragments are requested one phase ahead (FragQ): -1 k cycles per block
#endif
#ifndef HM_NPF_WIDE
#define HM_NPF_WIDE 10            // weight-fragment requests a wave keeps in flight in the 384-channel variant (build parameter for measurements)
#endif
namespace hmn {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16;

constexpr int MAXB = 16;
struct BlockDesc {
    int cop, k, eca, pad;
    unsigned w1, dw, w2, ecaw;     // fp16 buffer offsets (in halfs)
    unsigned b1, b2, b3, ecab;     // fp32 buffer offsets (in floats)
};
struct NetDesc {
    int C, nblocks, cv, cin_pad;   // trunk channels, #blocks, value-head channels, padded input channels (80)
    unsigned stem_w, ps_w, pp_w, v_w, vl_w;    // fp16 offsets
    unsigned stem_b, ps_b, v_b, vl_b;          // fp32 offsets
    BlockDesc blk[MAXB];
};

__device__ __forceinline__ floatx16 mfma(half8 a, half8 b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ floatx16 zero16() {
    floatx16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}
// packed fragment (kstep, tile) of a [K][N] matrix: lane l holds W[kstep*16 + 8*(l>>5) + j][tile*32 + (l&31)]
__device__ __forceinline__ half8 wfrag(const h16* w, int ntiles, int kstep, int tile, int lane) {
    return reinterpret_cast<const half8*>(w)[(size_t)(kstep * ntiles + tile) * 64 + lane];
}
// D element (reg) of a 32x32 tile: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
__device__ __forceinline__ int drow(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// GEMM with M = 64 squares (2 row tiles) and N = ncol output channels, A from an LDS activation tile.
//   conv3 = false: A[sq][k] = act[sq][k]                      (K = kdim)
//   conv3 = true : A[sq][tap*cin + ci] = act[sq + shift(tap)][ci] or 0   (3x3, zero padding; K = 9*cin)
// Wave w computes row tile (w & 1) and column tiles (w >> 1) + 2*i.  acc must hold ncol/64 tiles.
// im2col / plain row address of the A (or transposed B) fragment of k-step `ks` for square `sq`
__device__ __forceinline__ const h16* frag_src(const h16* act, int lda, int zeroRow, bool conv3, int cin, int ks, int kh, int sq) {
    const int k0 = ks * 16 + kh;
    int row = sq, col = k0;
    if (conv3) {
        const int tap = k0 / cin;
        col = k0 - tap * cin;
        const int dr = tap / 3 - 1, df = tap % 3 - 1;
        const int rr = (sq >> 3) + dr, ff = (sq & 7) + df;
        row = (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) ? rr * 8 + ff : zeroRow;
    }
    return act + (size_t)row * lda + col;
}

// k-steps are processed in groups of KG with a two-deep software pipeline: the weight-fragment
// loads (KG x tiles, 1 KiB per wave-load, L2-resident) and LDS activation-fragment reads of group
// g+1 are issued before the MFMAs of group g, so the L2 latency of one group hides behind the
// matrix work of the previous one instead of serialising load -> MFMA -> load.
template <int MAXT, int KG>
struct FragGroup { half8 a[KG]; half8 b[KG][MAXT]; };

template <int MAXT, int KG>
__device__ __forceinline__ void load_group_rows(FragGroup<MAXT, KG>& g, const h16* act, int lda, int zeroRow, bool conv3, int cin,
                                                const h16* w, int ntiles, int ks, int t0, int kh, int sq, int lane) {
#pragma unroll
    for (int u = 0; u < KG; ++u) {
#pragma unroll
        for (int i = 0; i < MAXT; ++i) g.b[u][i] = wfrag(w, ntiles, ks + u, t0 + 2 * i, lane);
    }
#pragma unroll
    for (int u = 0; u < KG; ++u) g.a[u] = *reinterpret_cast<const half8*>(frag_src(act, lda, zeroRow, conv3, cin, ks + u, kh, sq));
}
template <int MAXT, int KG>
__device__ __forceinline__ void mfma_group_rows(floatx16 (&acc)[MAXT], const FragGroup<MAXT, KG>& g) {
#pragma unroll
    for (int u = 0; u < KG; ++u) {
#pragma unroll
        for (int i = 0; i < MAXT; ++i) acc[i] = mfma(g.a[u], g.b[u][i], acc[i]);
    }
}

template <int MAXT>
__device__ __forceinline__ void gemm_rows(floatx16 (&acc)[MAXT], const h16* act, int lda, int zeroRow, bool conv3, int cin,
                                          int ksteps, const h16* w, int ncol, int wave, int lane) {
    constexpr int KG = MAXT <= 2 ? 4 : 2;
    const int ntiles = ncol >> 5;
    const int sq = (wave & 1) * 32 + (lane & 31);
    const int kh = 8 * (lane >> 5);
    const int t0 = wave >> 1;
    const int ngroups = ksteps / KG;
    FragGroup<MAXT, KG> g0, g1;
    if (ngroups > 0) load_group_rows<MAXT, KG>(g0, act, lda, zeroRow, conv3, cin, w, ntiles, 0, t0, kh, sq, lane);
    int gi = 0;
    for (; gi + 2 <= ngroups; gi += 2) {
        load_group_rows<MAXT, KG>(g1, act, lda, zeroRow, conv3, cin, w, ntiles, (gi + 1) * KG, t0, kh, sq, lane);
        mfma_group_rows<MAXT, KG>(acc, g0);
        if (gi + 2 < ngroups) load_group_rows<MAXT, KG>(g0, act, lda, zeroRow, conv3, cin, w, ntiles, (gi + 2) * KG, t0, kh, sq, lane);
        mfma_group_rows<MAXT, KG>(acc, g1);
    }
    if (gi < ngroups) mfma_group_rows<MAXT, KG>(acc, g0);
    for (int ks = ngroups * KG; ks < ksteps; ++ks) {
        const half8 a = *reinterpret_cast<const half8*>(frag_src(act, lda, zeroRow, conv3, cin, ks, kh, sq));
#pragma unroll
        for (int i = 0; i < MAXT; ++i) acc[i] = mfma(a, wfrag(w, ntiles, ks, t0 + 2 * i, lane), acc[i]);
    }
}

// Transposed GEMM: D[ch][sq] = sum_k W[k][ch] * act[sq][k]; one 32x32 tile per call
// (channel tile `wtile` of the packed matrix, square tile = wave & 1).  Same two-deep pipeline.
template <int KG>
struct FragGroupT { half8 a[KG]; half8 b[KG]; };
template <int KG>
__device__ __forceinline__ void load_group_cols(FragGroupT<KG>& g, const h16* act, int lda, int zeroRow, bool conv3, int cin,
                                                const h16* w, int ntilesTotal, int wtile, int ks, int kh, int sq, int lane) {
#pragma unroll
    for (int u = 0; u < KG; ++u) g.a[u] = wfrag(w, ntilesTotal, ks + u, wtile, lane);
#pragma unroll
    for (int u = 0; u < KG; ++u) g.b[u] = *reinterpret_cast<const half8*>(frag_src(act, lda, zeroRow, conv3, cin, ks + u, kh, sq));
}
__device__ __forceinline__ floatx16 gemm_cols(const h16* act, int lda, int zeroRow, bool conv3, int cin, int ksteps,
                                              const h16* w, int ntilesTotal, int wtile, int wave, int lane) {
    constexpr int KG = 4;
    floatx16 acc = zero16();
    const int sq = (wave & 1) * 32 + (lane & 31);
    const int kh = 8 * (lane >> 5);
    const int ngroups = ksteps / KG;
    FragGroupT<KG> g0, g1;
    if (ngroups > 0) load_group_cols<KG>(g0, act, lda, zeroRow, conv3, cin, w, ntilesTotal, wtile, 0, kh, sq, lane);
    int gi = 0;
    for (; gi + 2 <= ngroups; gi += 2) {
        load_group_cols<KG>(g1, act, lda, zeroRow, conv3, cin, w, ntilesTotal, wtile, (gi + 1) * KG, kh, sq, lane);
#pragma unroll
        for (int u = 0; u < KG; ++u) acc = mfma(g0.a[u], g0.b[u], acc);       // A = W^T fragment, B = act^T fragment
        if (gi + 2 < ngroups) load_group_cols<KG>(g0, act, lda, zeroRow, conv3, cin, w, ntilesTotal, wtile, (gi + 2) * KG, kh, sq, lane);
#pragma unroll
        for (int u = 0; u < KG; ++u) acc = mfma(g1.a[u], g1.b[u], acc);
    }
    if (gi < ngroups) {
#pragma unroll
        for (int u = 0; u < KG; ++u) acc = mfma(g0.a[u], g0.b[u], acc);
    }
    for (int ks = ngroups * KG; ks < ksteps; ++ks) {
        const half8 b = *reinterpret_cast<const half8*>(frag_src(act, lda, zeroRow, conv3, cin, ks, kh, sq));
        acc = mfma(wfrag(w, ntilesTotal, ks, wtile, lane), b, acc);
    }
    return acc;
}

// depthwise KxK (+bias, ReLU) for one channel and two board rows; weights in registers
template <int K>
__device__ __forceinline__ void depthwise_rows(const h16* y1 /*[66] row of this channel*/, h16* y2, int ch, int g, const h16* wd, float bias) {
    constexpr int H = K / 2;
    float wreg[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wreg[i] = (float)wd[i];
    // the K+1 input rows this thread needs, zero padded in both directions
    float in[K + 1][8 + 2 * H];
#pragma unroll
    for (int r = 0; r < K + 1; ++r) {
        const int y = 2 * g - H + r;
#pragma unroll
        for (int x = 0; x < 8 + 2 * H; ++x) {
            const int xx = x - H;
            in[r][x] = (y >= 0 && y < 8 && xx >= 0 && xx < 8) ? (float)y1[y * 8 + xx] : 0.0f;
        }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int ff = 0; ff < 8; ++ff) {
            float s = bias;
#pragma unroll
            for (int dy = 0; dy < K; ++dy) {
#pragma unroll
                for (int dx = 0; dx < K; ++dx) s += wreg[dy * K + dx] * in[o + dy][ff + dx];
            }
            y2[((2 * g + o) * 8 + ff) * 72 + ch] = (h16)fmaxf(s, 0.0f);
        }
    }
}

// =============================================================================================================
// Narrow trunks (C <= 128, the self-play "RISEv3-small"): a latency-first form of the same fused forward.
//
// The search hands the evaluator a few hundred positions at a time — at most one workgroup per CU — so what the lockstep
// loop waits for is the latency of ONE position through the network, not throughput.  The 4-wave kernel above walks a
// mobile block in 64-channel chunks (three barriers and three exposed weight-fetch latencies per chunk: 63 barriers for
// the small net, 153 us per position).  Here one position gets 8 waves (512 threads), every wave owns exactly one 32x32
// tile of an N = C GEMM, a block is three phases over ALL its expanded channels (expand -> depthwise -> project), and each
// GEMM keeps PF weight fragments (1 KiB wave-loads straight from L2) in flight, the first ones issued before the barrier
// that opens the phase.  im2col row offsets are computed once per convolution, not per k-step.
// LDS: Xs[65][C+8] | union{ Ss[65][C+8] , Y1[cop][66] + Y2[64][cop+8] } | per-block parameter stage | ECA scratch.
// =============================================================================================================
// weight fragments in flight per wave: 8 for the narrow trunks (no spills, one tile per wave), 16 for the 384-channel net
// (three tiles per wave, 24-step expansions: measured 0.90 -> 0.79 ms per position together with the dword depthwise reads)

// LDS element offset of the source row of 3x3 tap t for square sq (or of the zero row for off-board taps)
__device__ __forceinline__ int im2col_row(int sq, int t, int pitch, int zeroRow) {
    const int t3 = (t * 11) >> 5;                                       // t / 3 for t in 0..8
    const int rr = (sq >> 3) + t3 - 1, ff = (sq & 7) + (t - 3 * t3) - 1;
    return (((unsigned)rr < 8u && (unsigned)ff < 8u) ? rr * 8 + ff : zeroRow) * pitch;
}

// Fragments travel as four dwords (bit-cast to 8 halfs only at the MFMA): a half-typed vector carried around a loop is
// re-assembled element by element by the compiler (v_perm storms), and a conditional refill of the prefetch queue makes it
// drain the queue with vmcnt(0) at every step.  So: no branches around loads (the tail re-requests the last fragment
// instead — an L1 hit), straight-line bodies.
typedef int frag4 __attribute__((ext_vector_type(4)));
typedef h16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ half8 h8(frag4 x) { return __builtin_bit_cast(half8, x); }
__device__ __forceinline__ frag4 lds_frag(const h16* p) { return *reinterpret_cast<const frag4*>(p); }

// One 32x32 output tile, K = ksteps*16: operand X from LDS (`xrow` = this lane's row + 8*(lane>>5) halfs, k-step i at
// xrow + i*16), operand W = packed fragments from global memory (k-step i of `tile` at wp + i*ws) with NPF requests in
// flight.  WA = true: the weight fragment is the MFMA's A operand (transposed product D[ch][sq]), else its B operand.
template <bool WA, int Q>
__device__ __forceinline__ floatx16 gemm_tile(floatx16 acc, const h16* xrow, int ksteps, const h16* w, int ntiles, int tile, int lane) {
    const frag4* wp = reinterpret_cast<const frag4*>(w) + (size_t)tile * 64 + lane;
    const size_t ws = (size_t)ntiles * 64;
    const int last = ksteps - 1;
    frag4 q[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) q[j] = wp[(size_t)(j < last ? j : last) * ws];
    int base = 0;
    for (; base + Q < ksteps; base += Q) {                             // full groups: refill each slot right after its use
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const frag4 x = lds_frag(xrow + (base + j) * 16);
            const frag4 f = q[j];
            const int nxt = base + Q + j;
            q[j] = wp[(size_t)(nxt < last ? nxt : last) * ws];
            acc = WA ? mfma(h8(f), h8(x), acc) : mfma(h8(x), h8(f), acc);
        }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {                                       // last (possibly partial) group: nothing left to request
        if (base + j < ksteps) {
            const frag4 x = lds_frag(xrow + (base + j) * 16);
            acc = WA ? mfma(h8(q[j]), h8(x), acc) : mfma(h8(x), h8(q[j]), acc);
        }
    }
    return acc;
}
// A wave's tiles of the transposed expand GEMM as ONE fragment stream: tiles tile0, tile0 + tstride, ... (ntile of them), each
// K = ksteps*16 with ksteps % Q == 0.  The queue is refilled with the next tile's first fragments while the current tile's last
// group multiplies, so only the first tile pays the weight-fetch latency (the activation fragments are the same for every tile).
// epi(i, acc) stores tile i.  The last tile's last group re-requests its own first fragments instead of branching.
template <int Q, typename Epi>
__device__ __forceinline__ void expand_stream(const h16* xrow, int ksteps, const h16* w, int ntiles, int tile0, int tstride, int ntile, int lane, Epi&& epi) {
    const frag4* base = reinterpret_cast<const frag4*>(w) + lane;
    const size_t ws = (size_t)ntiles * 64;
    const frag4* wp = base + (size_t)tile0 * 64;
    frag4 q[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) q[j] = wp[(size_t)j * ws];
    for (int i = 0; i < ntile; ++i) {
        const frag4* wn = base + (size_t)(tile0 + (i + 1 < ntile ? i + 1 : i) * tstride) * 64;
        floatx16 acc = zero16();
        for (int b = 0; b < ksteps; b += Q) {
            const frag4* nx = b + Q < ksteps ? wp + (size_t)(b + Q) * ws : wn;
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const frag4 x = lds_frag(xrow + (b + j) * 16);
                const frag4 f = q[j];
                q[j] = nx[(size_t)j * ws];
                acc = mfma(h8(f), h8(x), acc);
            }
        }
        epi(i, acc);
        wp = wn;
    }
}
// 3x3 convolution as an implicit-im2col GEMM: K = 9*cin, k-step (tap, kk) reads X at act + row(tap) + kk*16 + kh.
// KPT = cin / 16 k-steps per tap = the depth of the weight-fragment queue (the fragments of tap t+1 are requested while
// tap t multiplies).
template <int KPT, bool WA>
__device__ __forceinline__ floatx16 gemm_conv3(floatx16 acc, const h16* act, int sq, int pitch, int zeroRow, int kh, const h16* w, int ntiles, int tile, int lane) {
    const frag4* wp = reinterpret_cast<const frag4*>(w) + (size_t)tile * 64 + lane;
    const size_t ws = (size_t)ntiles * 64;
    frag4 q[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) q[j] = wp[(size_t)j * ws];
    // taps unrolled (a rolled loop drains the queue at its back edge: the refilled slots are loop-carried copies); the
    // scheduling barriers keep one tap's loads from being hoisted over the previous taps, which would spill
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const h16* ap = act + im2col_row(sq, t, pitch, zeroRow) + kh;
#pragma unroll
        for (int kk = 0; kk < KPT; ++kk) {
            const frag4 x = lds_frag(ap + kk * 16);
            const frag4 f = q[kk];
            if (t < 8) q[kk] = wp[(size_t)((t + 1) * KPT + kk) * ws];
            acc = WA ? mfma(h8(f), h8(x), acc) : mfma(h8(x), h8(f), acc);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// ---- the same three GEMM forms with BOTH square halves of a channel tile on one wave (4-wave workgroups): every weight fragment
// is requested once per workgroup instead of once per square half (half the L2 traffic of the weight stream) and feeds two
// independent MFMA chains (acc0: squares 0..31, acc1: squares 32..63).  Per accumulator the order of the k-steps is the one of
// the single-tile forms, so the results are bit-identical to them.
template <bool WA, int Q>
__device__ __forceinline__ void gemm_tile2(floatx16& acc0, floatx16& acc1, const h16* xrow0, const h16* xrow1, int ksteps, const h16* w, int ntiles, int tile, int lane) {
    const frag4* wp = reinterpret_cast<const frag4*>(w) + (size_t)tile * 64 + lane;
    const size_t ws = (size_t)ntiles * 64;
    const int last = ksteps - 1;
    frag4 q[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) q[j] = wp[(size_t)(j < last ? j : last) * ws];
    int base = 0;
    for (; base + Q < ksteps; base += Q) {
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const frag4 x0 = lds_frag(xrow0 + (base + j) * 16);
            const frag4 x1 = lds_frag(xrow1 + (base + j) * 16);
            const frag4 f = q[j];
            const int nxt = base + Q + j;
            q[j] = wp[(size_t)(nxt < last ? nxt : last) * ws];
            acc0 = WA ? mfma(h8(f), h8(x0), acc0) : mfma(h8(x0), h8(f), acc0);
            acc1 = WA ? mfma(h8(f), h8(x1), acc1) : mfma(h8(x1), h8(f), acc1);
        }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        if (base + j < ksteps) {
            const frag4 x0 = lds_frag(xrow0 + (base + j) * 16);
            const frag4 x1 = lds_frag(xrow1 + (base + j) * 16);
            acc0 = WA ? mfma(h8(q[j]), h8(x0), acc0) : mfma(h8(x0), h8(q[j]), acc0);
            acc1 = WA ? mfma(h8(q[j]), h8(x1), acc1) : mfma(h8(x1), h8(q[j]), acc1);
        }
    }
}
template <int Q, typename Epi>
__device__ __forceinline__ void expand_stream2(const h16* xrow0, const h16* xrow1, int ksteps, const h16* w, int ntiles, int tile0, int tstride, int ntile, int lane, Epi&& epi) {
    const frag4* base = reinterpret_cast<const frag4*>(w) + lane;
    const size_t ws = (size_t)ntiles * 64;
    const frag4* wp = base + (size_t)tile0 * 64;
    frag4 q[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) q[j] = wp[(size_t)j * ws];
    for (int i = 0; i < ntile; ++i) {
        const frag4* wn = base + (size_t)(tile0 + (i + 1 < ntile ? i + 1 : i) * tstride) * 64;
        floatx16 acc0 = zero16(), acc1 = zero16();
        for (int b = 0; b < ksteps; b += Q) {
            const frag4* nx = b + Q < ksteps ? wp + (size_t)(b + Q) * ws : wn;
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const frag4 x0 = lds_frag(xrow0 + (b + j) * 16);
                const frag4 x1 = lds_frag(xrow1 + (b + j) * 16);
                const frag4 f = q[j];
                q[j] = nx[(size_t)j * ws];
                acc0 = mfma(h8(f), h8(x0), acc0);
                acc1 = mfma(h8(f), h8(x1), acc1);
            }
        }
        epi(i, acc0, acc1);
        wp = wn;
    }
}
template <int KPT, bool WA>
__device__ __forceinline__ void gemm_conv3_2(floatx16& acc0, floatx16& acc1, const h16* act, int sq0, int pitch, int zeroRow, int kh, const h16* w, int ntiles, int tile, int lane) {
    const frag4* wp = reinterpret_cast<const frag4*>(w) + (size_t)tile * 64 + lane;
    const size_t ws = (size_t)ntiles * 64;
    frag4 q[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) q[j] = wp[(size_t)j * ws];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const h16* ap0 = act + im2col_row(sq0, t, pitch, zeroRow) + kh;
        const h16* ap1 = act + im2col_row(sq0 + 32, t, pitch, zeroRow) + kh;
#pragma unroll
        for (int kk = 0; kk < KPT; ++kk) {
            const frag4 x0 = lds_frag(ap0 + kk * 16);
            const frag4 x1 = lds_frag(ap1 + kk * 16);
            const frag4 f = q[kk];
            if (t < 8) q[kk] = wp[(size_t)((t + 1) * KPT + kk) * ws];
            acc0 = WA ? mfma(h8(f), h8(x0), acc0) : mfma(h8(x0), h8(f), acc0);
            acc1 = WA ? mfma(h8(f), h8(x1), acc1) : mfma(h8(x1), h8(f), acc1);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// A weight-fragment queue requested AHEAD of the phase that multiplies it (4-wave form: a lane has 512 registers, so a queue can stay
// live across a barrier, a depthwise phase or an epilogue — the L2 round trip of a GEMM's first fragments then overlaps that work
// instead of opening the phase).  fragq_fill requests k-steps 0 .. Q-1 of `tile` (clamped to the last one, as gemm_tile2 does).
template <int Q>
struct FragQ { frag4 q[Q]; };
template <int Q>
__device__ __forceinline__ void fragq_fill(FragQ<Q>& f, const h16* w, int ntiles, int tile, int ksteps, int lane) {
    const frag4* wp = reinterpret_cast<const frag4*>(w) + (size_t)tile * 64 + lane;
    const size_t ws = (size_t)ntiles * 64;
    const int last = ksteps - 1;
#pragma unroll
    for (int j = 0; j < Q; ++j) f.q[j] = wp[(size_t)(j < last ? j : last) * ws];
}
// gemm_tile2 continuing from a queue fragq_fill has requested for the same (w, ntiles, tile, ksteps)
template <bool WA, int Q>
__device__ __forceinline__ void gemm_tile2_q(FragQ<Q>& f, floatx16& acc0, floatx16& acc1, const h16* xrow0, const h16* xrow1, int ksteps, const h16* w, int ntiles, int tile, int lane) {
    const frag4* wp = reinterpret_cast<const frag4*>(w) + (size_t)tile * 64 + lane;
    const size_t ws = (size_t)ntiles * 64;
    const int last = ksteps - 1;
    int base = 0;
    for (; base + Q < ksteps; base += Q) {
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const frag4 x0 = lds_frag(xrow0 + (base + j) * 16);
            const frag4 x1 = lds_frag(xrow1 + (base + j) * 16);
            const frag4 fr = f.q[j];
            const int nxt = base + Q + j;
            f.q[j] = wp[(size_t)(nxt < last ? nxt : last) * ws];
            acc0 = WA ? mfma(h8(fr), h8(x0), acc0) : mfma(h8(x0), h8(fr), acc0);
            acc1 = WA ? mfma(h8(fr), h8(x1), acc1) : mfma(h8(x1), h8(fr), acc1);
        }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        if (base + j < ksteps) {
            const frag4 x0 = lds_frag(xrow0 + (base + j) * 16);
            const frag4 x1 = lds_frag(xrow1 + (base + j) * 16);
            acc0 = WA ? mfma(h8(f.q[j]), h8(x0), acc0) : mfma(h8(x0), h8(f.q[j]), acc0);
            acc1 = WA ? mfma(h8(f.q[j]), h8(x1), acc1) : mfma(h8(x1), h8(f.q[j]), acc1);
        }
    }
}
// expand_stream2 continuing from a queue fragq_fill has requested for its first tile (ksteps == Q)
template <int Q, typename Epi>
__device__ __forceinline__ void expand_stream2_q(FragQ<Q>& f, const h16* xrow0, const h16* xrow1, int ksteps, const h16* w, int ntiles, int tile0, int tstride, int ntile, int lane, Epi&& epi) {
    const frag4* base = reinterpret_cast<const frag4*>(w) + lane;
    const size_t ws = (size_t)ntiles * 64;
    const frag4* wp = base + (size_t)tile0 * 64;
    for (int i = 0; i < ntile; ++i) {
        const frag4* wn = base + (size_t)(tile0 + (i + 1 < ntile ? i + 1 : i) * tstride) * 64;
        floatx16 acc0 = zero16(), acc1 = zero16();
        for (int b = 0; b < ksteps; b += Q) {
            const frag4* nx = b + Q < ksteps ? wp + (size_t)(b + Q) * ws : wn;
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const frag4 x0 = lds_frag(xrow0 + (b + j) * 16);
                const frag4 x1 = lds_frag(xrow1 + (b + j) * 16);
                const frag4 fr = f.q[j];
                f.q[j] = nx[(size_t)j * ws];
                acc0 = mfma(h8(fr), h8(x0), acc0);
                acc1 = mfma(h8(fr), h8(x1), acc1);
            }
        }
        epi(i, acc0, acc1);
        wp = wn;
    }
}

// A wave's T projection tiles (channel tiles tile0, tile0 + tstride, ...; both square halves each; accumulators carried by the caller) as
// ONE weight-fragment stream with K = Q k-steps per tile: every queue slot is refilled with the NEXT tile's fragment of the same k-step
// right after its use, so only the first tile pays the weight-fetch latency (the 384-channel network's full chunks: K = Q = 20).
// Tiles i >= ntile are skipped; the last tile re-requests its own fragments instead of branching.
template <int Q, int T>
__device__ __forceinline__ void project_stream2(floatx16 (&acc)[T][2], const h16* xrow0, const h16* xrow1, const h16* w, int ntiles, int tile0, int tstride, int ntile, int lane) {
    const frag4* base = reinterpret_cast<const frag4*>(w) + lane;
    const size_t ws = (size_t)ntiles * 64;
    const frag4* wp = base + (size_t)tile0 * 64;
    frag4 q[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) q[j] = wp[(size_t)j * ws];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        if (i < ntile) {
            const frag4* wn = base + (size_t)(tile0 + (i + 1 < ntile ? i + 1 : i) * tstride) * 64;
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const frag4 x0 = lds_frag(xrow0 + j * 16);
                const frag4 x1 = lds_frag(xrow1 + j * 16);
                const frag4 fr = q[j];
                q[j] = wn[(size_t)j * ws];
                acc[i][0] = mfma(h8(x0), h8(fr), acc[i][0]);
                acc[i][1] = mfma(h8(x1), h8(fr), acc[i][1]);
            }
        }
    }
}

// depthwise KxK (+bias, ReLU) for one channel and two board rows; output transposed to y2[sq][ch] with pitch ld2.
// Inputs and weights stay fp16 and go through v_dot2_f32_f16 two taps at a time (exact products, fp32 accumulation): a 3x3
// output costs 6 dot instructions instead of 9 FMAs and 5x5 15 instead of 25, and no input is converted to fp32.
template <int K>
__device__ __forceinline__ void depthwise_rows_ld(const h16* y1, h16* y2, int ld2, int ch, int g, const h16* wd, float bias) {
    constexpr int H = K / 2, NP = (K + 1) / 2, W = 8 + 2 * NP;          // taps per row in pairs; padded row width (halfs)
    half2v wp[K][NP];
#pragma unroll
    for (int dy = 0; dy < K; ++dy) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            wp[dy][j][0] = wd[dy * K + 2 * j];
            wp[dy][j][1] = 2 * j + 1 < K ? wd[dy * K + 2 * j + 1] : (h16)0.0f;
        }
    }
    h16 in[K + 1][W];                                                    // padded board rows: file f sits at index f + H
#pragma unroll
    for (int r = 0; r < K + 1; ++r) {
        const int y = 2 * g - H + r;
#pragma unroll
        for (int x = 0; x < W; ++x) in[r][x] = (h16)0.0f;
        if (y >= 0 && y < 8) {
            // one board row = 8 halfs = four dwords (the channel pitch of 66 halfs keeps rows 4-byte aligned)
            const uint32_t* row = reinterpret_cast<const uint32_t*>(y1 + y * 8);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const half2v v = __builtin_bit_cast(half2v, row[d]);
                in[r][H + 2 * d] = v[0];
                in[r][H + 2 * d + 1] = v[1];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int ff = 0; ff < 8; ++ff) {
            float s = bias;
#pragma unroll
            for (int dy = 0; dy < K; ++dy) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    half2v x;
                    x[0] = in[o + dy][ff + 2 * j];
                    x[1] = in[o + dy][ff + 2 * j + 1];
                    s = __builtin_amdgcn_fdot2(x, wp[dy][j], s, false);
                }
            }
            y2[((2 * g + o) * 8 + ff) * ld2 + ch] = (h16)fmaxf(s, 0.0f);
        }
    }
}

// One position through the whole network by one workgroup of 8 waves (the body of rise_forward_narrow and of the persistent
// evaluator rise_serve): input row `pin`, outputs to row `sIdx` of the head tensors.
#define HM_STAMP() do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0 && dbgN < 256) dbg[dbgN++] = __builtin_amdgcn_s_memtime(); } while (0)
// WT: the heads are written with write-through stores (hm_queue.hpp), the policy planes staged through LDS into 16-byte chunks.
constexpr int PRIOR_SCRATCH_HALFS = 2 * (HM_MAX_MOVES * 4 + (HM_MAX_MOVES + 8) * 4 + HM_MAX_MOVES * 4 + HM_MAX_MOVES * 4) / 2;   // PriorEpi's LDS, in halfs
// PolicyEpi (WT only): called by every thread with the staged logits `pol` ([160][64] fp16 in LDS: board A's 73 planes, then board
// B's) and `scratch` (the LDS behind them) in place of storing the logits; it ends with a workgroup barrier of its own.
struct NoPolicyEpi { static constexpr bool present = false; __device__ void operator()(const h16*, unsigned char*) const {} };
template <int CTILES, bool K5, bool WT = false, typename PolicyEpi = NoPolicyEpi>      // CTILES = C / 32 (2, 4 or 12); cin_pad must be 80
__device__ __forceinline__ void narrow_position(const NetDesc& nd, const h16* __restrict__ wh, const float* __restrict__ wf, const h16* pin, size_t sIdx,
                                                int copMax, int uHalfs, unsigned char* smem,
                                                h16* __restrict__ value, h16* __restrict__ piA, h16* __restrict__ piB, h16* __restrict__ wdl, h16* __restrict__ ml,
                                                unsigned long long* __restrict__ dbg, int& dbgN, PolicyEpi epi = PolicyEpi()) {
    // C = 64, 128 or 384: an N = C GEMM is CTILES column tiles x 2 square tiles = 2*CTILES tiles of 32x32 over 8 waves, i.e.
    // TPW tiles per wave (tile t = wave + 8i; its square half t & 1 is the wave's own).  copMax = the expanded channels kept
    // in LDS at a time: a block whose `cop` exceeds it (the 384-channel deployed net: cop up to 1152) runs its three phases
    // per chunk of copMax channels and carries the projection accumulators across the chunks in registers.
    constexpr int C = CTILES * 32, ldx = C + 8, ctiles = CTILES, TPW = (2 * CTILES + 7) / 8, NPF = CTILES > 4 ? HM_NPF_WIDE : 8, NPFP = (CTILES > 4 && HM_NPF_PROJ > 0) ? HM_NPF_PROJ : NPF;
    h16* Xs = reinterpret_cast<h16*>(smem);                             // [65][ldx]
    h16* U = Xs + 65 * ldx;                                             // union region
    h16* Ss = U;                                                        // [65][ldx] (input staging uses pitch ldi)
    h16* Y1 = U;                                                        // [chunk][66]
    h16* Y2 = U + (size_t)copMax * 66;                                  // [64][chunk + 8]
    float* Pf = reinterpret_cast<float*>(U + uHalfs);                   // per-chunk parameters: b1[chunk], b2[chunk], b3[C]  (uHalfs % 8 == 0)
    h16* Pdw = reinterpret_cast<h16*>(Pf + 2 * copMax + C);             // depthwise weights [chunk][k*k]
    float* Ev = reinterpret_cast<float*>(Pdw + (((size_t)copMax * 25 + 7) & ~(size_t)7));   // ECA / head scratch: [4][C] + [C] + 64
    // WT: policy staging [160][64] (+ the prior pipeline's scratch) behind Ss inside the union region when it has the room (narrow
    // trunks: their expansions are wide against C), else in Xs, which is dead once the policy trunk has been computed (C = 384)
    constexpr int polOff = (65 * (ldx > 88 ? ldx : 88) + 7) & ~7;
    h16* Pol = polOff + 160 * 64 + PRIOR_SCRATCH_HALFS <= uHalfs ? U + polOff : Xs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = 8 * (lane >> 5);
    const int stile = wave & 1;                                         // the square half of every tile this wave computes
    const int sqL = stile * 32 + (lane & 31);                           // the square this lane feeds as an A-row / B-column

    {
        HM_STAMP();
        // ---- input planes: NCHW [74][64] fp16 -> Ss as [sq][cin_pad] (+ zero row 64)
        const int ldi = nd.cin_pad + 8;
        {   // 16-byte stores / loads (Ss is 16-byte aligned: 65 * ldx * 2 is a multiple of 16 for C = 64 / 128 / 384; a plane row is 592 chunks of 8 halfs)
            const frag4 z = {0, 0, 0, 0};
            frag4* z16 = reinterpret_cast<frag4*>(Ss);
            for (int i = tid; i < (65 * ldi) >> 3; i += 512) z16[i] = z;
        }
        for (int i = tid; i < ldx; i += 512) Xs[64 * ldx + i] = (h16)0.0f;
        __syncthreads();
        for (int c = tid; c < HM_PLANE_VALUES / 8; c += 512) {           // chunk c = plane c / 8, squares 8 * (c % 8) ..
            const half8 v = *reinterpret_cast<const half8*>(pin + 8 * c);
            const int pl = c >> 3, sq8 = (c & 7) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) Ss[(sq8 + j) * ldi + pl] = v[j];
        }
        __syncthreads();
        HM_STAMP();   // input staged
        // ---- stem: 3x3 conv cin -> C (+bias, ReLU)
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ctile = (wave + 8 * i) >> 1;
            if (ctile < ctiles) {
                floatx16 acc = gemm_conv3<5, false>(zero16(), Ss, sqL, ldi, 64, kh, wh + nd.stem_w, ctiles, ctile, lane);
                const int co = ctile * 32 + (lane & 31);
                const float bias = wf[nd.stem_b + co];
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) Xs[(stile * 32 + drow(rg, lane)) * ldx + co] = (h16)fmaxf(acc[rg] + bias, 0.0f);
            }
        }
        __syncthreads();
        HM_STAMP();
        // ---- mobile bottleneck blocks: three phases each over (a chunk of) the `cop` expanded channels
        for (int bi = 0; bi < nd.nblocks; ++bi) {
            const BlockDesc bd = nd.blk[bi];
            const int cop = bd.cop, kk = bd.k;
            float* sb1 = Pf; float* sb2 = Pf + copMax; float* sb3 = Pf + 2 * copMax;
            for (int i = tid; i < C; i += 512) sb3[i] = wf[bd.b3 + i];
            if (bd.eca) {   // x = x * hardsigmoid(W_eca . mean_sq(x) + b)   (builder_util.py:49-80, centre tap)
                float* part = Ev;            // [4][C] partial sums, then partial dot products
                float* Mv = Ev + 4 * C;      // [C] channel means
                // work item = (quarter p of the squares / input channels, channel c): 4*C items over 512 threads
                for (int it = tid; it < 4 * C; it += 512) {
                    const int c = it % C, p = it / C;
                    float sacc = 0.0f;
                    for (int sq = 16 * p; sq < 16 * p + 16; ++sq) sacc += (float)Xs[sq * ldx + c];
                    part[p * C + c] = sacc;
                }
                __syncthreads();
                for (int c = tid; c < C; c += 512) Mv[c] = (part[c] + part[C + c] + part[2 * C + c] + part[3 * C + c]) * (1.0f / 64.0f);
                __syncthreads();
                // W_eca (dense [ci][co], 2*C*C bytes from L2) . mean: work item = (1/32 of the input channels, 8 adjacent output
                // channels) -- 16-byte loads, 32*C/8 items = three per thread at C = 384, a thread's 12 rows requested together; the
                // 32 partial sums per channel go through the union region, which nothing else uses between two blocks
                float* part32 = reinterpret_cast<float*>(U);            // [32][C]
                {
                    const h16* we = wh + bd.ecaw;                       // [ci][co]
                    constexpr int CG = C >> 3, ROWS = C >> 5;
                    for (int it = tid; it < 32 * CG; it += 512) {
                        const int cg = it % CG, p = it / CG;
                        float s[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                        const h16* wrow = we + (size_t)(p * ROWS) * C + 8 * cg;
                        half8 w8[ROWS];
#pragma unroll
                        for (int u = 0; u < ROWS; ++u) w8[u] = *reinterpret_cast<const half8*>(wrow + (size_t)u * C);
#pragma unroll
                        for (int u = 0; u < ROWS; ++u) {
                            const float m = Mv[p * ROWS + u];
#pragma unroll
                            for (int k = 0; k < 8; ++k) s[k] += (float)w8[u][k] * m;
                        }
                        float* dst = part32 + p * C + 8 * cg;
#pragma unroll
                        for (int k = 0; k < 8; ++k) dst[k] = s[k];
                    }
                }
                __syncthreads();
                for (int c = tid; c < C; c += 512) {
                    float sg = wf[bd.ecab + c];
#pragma unroll
                    for (int p = 0; p < 32; ++p) sg += part32[p * C + c];
                    Mv[c] = fminf(fmaxf(sg * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f);
                }
                __syncthreads();
                for (int i = tid; i < 32 * C; i += 512) {               // two adjacent channels per item
                    const int sq = i / (C >> 1), cc = 2 * (i - sq * (C >> 1));
                    half2v* px = reinterpret_cast<half2v*>(Xs + sq * ldx + cc);
                    half2v x2 = *px;
                    x2[0] = (h16)((float)x2[0] * Mv[cc]); x2[1] = (h16)((float)x2[1] * Mv[cc + 1]);
                    *px = x2;
                }
            }
            floatx16 pacc[TPW];                                         // projection accumulators, carried over the chunks
#pragma unroll
            for (int i = 0; i < TPW; ++i) pacc[i] = zero16();
            for (int ch0 = 0; ch0 < cop; ch0 += copMax) {
                const int chunk = min(copMax, cop - ch0), ld2 = chunk + 8;
                // parameter stage of this chunk (the previous chunk's phase 3 no longer reads Pf / Pdw: barrier at its end)
                for (int i = tid; i < chunk; i += 512) { sb1[i] = wf[bd.b1 + ch0 + i]; sb2[i] = wf[bd.b2 + ch0 + i]; }
                {   // depthwise taps of the chunk, 16 bytes per lane (offsets and counts are multiples of 8 halfs: chunk % 32 == 0)
                    const half8* src = reinterpret_cast<const half8*>(wh + bd.dw + (size_t)ch0 * kk * kk);
                    half8* dst = reinterpret_cast<half8*>(Pdw);
                    for (int i = tid; i < (chunk * kk * kk) >> 3; i += 512) dst[i] = src[i];
                }
                const int copTiles = cop >> 5, tiles = (chunk >> 5) * 2, ct0 = ch0 >> 5;
                __syncthreads();
                HM_STAMP();   // parameters staged (+ ECA)
                // phase 1 — 1x1 expand, transposed: Y1[ch][sq] = relu(W1^T x^T + b1); tiles = (chunk/32) x 2 over 8 waves
                {
                    // (one wave per channel tile for both square halves — every W1 fragment fetched once per workgroup — was
                    // measured: the expand phase went from 12 k to 33 k cycles per chunk; kept: one 32x32 tile per call)
                    const h16* brow = Xs + sqL * ldx + kh;              // tile's square half = t & 1 == stile when t = wave + 8i
                    // this wave's tiles t = wave, wave + 8, ...: channel tiles (wave >> 1) + 4i, streamed through one fragment queue
                    // (narrow trunks, K = one queue: the 384-channel variant spills with the streamed form and keeps one call per tile)
                    if constexpr ((C >> 4) == NPF) {
                        const int ntile = tiles > wave ? (tiles - wave + 7) >> 3 : 0;
                        if (ntile > 0)
                            expand_stream<NPF>(brow, C >> 4, wh + bd.w1, copTiles, ct0 + (wave >> 1), 4, ntile, lane, [&](int i, const floatx16& e) {
                                const int ct = (wave >> 1) + 4 * i;
#pragma unroll
                                for (int rg = 0; rg < 16; ++rg) {
                                    const int ch = ct * 32 + drow(rg, lane);
                                    Y1[ch * 66 + sqL] = (h16)fmaxf(e[rg] + sb1[ch], 0.0f);
                                }
                            });
                    } else {
                        for (int t = wave; t < tiles; t += 8) {
                            const int ct = t >> 1;                      // (t & 1) == (wave & 1) == stile
                            const floatx16 e = gemm_tile<true, NPF>(zero16(), brow, C >> 4, wh + bd.w1, copTiles, ct0 + ct, lane);
#pragma unroll
                            for (int rg = 0; rg < 16; ++rg) {
                                const int ch = ct * 32 + drow(rg, lane);
                                Y1[ch * 66 + sqL] = (h16)fmaxf(e[rg] + sb1[ch], 0.0f);
                            }
                        }
                    }
                }
                __syncthreads();
                HM_STAMP();   // expand done
                // phase 2 — depthwise kxk (+bias, ReLU): work item = (channel, pair of board rows).  (Requesting the projection's
                // first weight fragments before it was measured: the queue held across the depthwise arithmetic spills, 0.79 -> 1.03 ms.)
                const h16* w2c = wh + bd.w2 + (size_t)(ch0 >> 4) * ctiles * 512;
                for (int item = tid; item < chunk * 4; item += 512) {
                    const int ch = item % chunk, g = item / chunk;
                    const h16* wd = Pdw + (size_t)ch * kk * kk;
                    if (!K5 || kk == 3) depthwise_rows_ld<3>(Y1 + ch * 66, Y2, ld2, ch, g, wd, sb2[ch]);
                    else depthwise_rows_ld<5>(Y1 + ch * 66, Y2, ld2, ch, g, wd, sb2[ch]);
                }
                __syncthreads();
                HM_STAMP();   // depthwise done
                // phase 3 — 1x1 project, K = this chunk's channels: k-steps ch0/16 .. of W2
                // (streaming a wave's TPW tiles through one fragment queue, as the narrow expand phase does, was measured on the
                // 384-channel variant: 108 spilled VGPRs instead of 42; kept: one call per tile)
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int ctile = (wave + 8 * i) >> 1;
                    if (ctile < ctiles)
                        pacc[i] = gemm_tile<false, NPFP>(pacc[i], Y2 + sqL * ld2 + kh, chunk >> 4, w2c, ctiles, ctile, lane);
                }
                // (no barrier here: the next chunk's parameter stage and expand phase touch Pf / Pdw / Y1, which this phase does
                // not read, and its depthwise phase — the next writer of Y2 — starts behind the barrier after its expand phase)
            }
            // + bias + residual
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int ctile = (wave + 8 * i) >> 1;
                if (ctile < ctiles) {
                    const int co = ctile * 32 + (lane & 31);
                    const float bias = sb3[co];
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int sq = stile * 32 + drow(rg, lane);
                        Xs[sq * ldx + co] = (h16)((float)Xs[sq * ldx + co] + pacc[i][rg] + bias);
                    }
                }
            }
            __syncthreads();
            HM_STAMP();
        }
        // ---- value head: 1x1 conv C -> cv (+bias, ReLU), NCHW flatten, linear -> (wdl x3, plys)
        {
            const int cv = nd.cv;
            if (wave < 2) {
                const floatx16 e = gemm_tile<true, NPF>(zero16(), Xs + sqL * ldx + kh, C >> 4, wh + nd.v_w, 1, 0, lane);
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int ch = drow(rg, lane);
                    if (ch < cv) Y1[ch * 66 + sqL] = (h16)fmaxf(e[rg] + wf[nd.v_b + ch], 0.0f);
                }
            }
            __syncthreads();
            float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const h16* wl = wh + nd.vl_w;
            for (int i = tid; i < cv * 64; i += 512) {
                const float v = (float)Y1[(i >> 6) * 66 + (i & 63)];
#pragma unroll
                for (int o = 0; o < 4; ++o) part[o] += v * (float)wl[(size_t)o * cv * 64 + i];
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float v = part[o];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                if (lane == 0) Ev[wave * 4 + o] = v;
            }
            __syncthreads();
            if (tid == 0) {
                float lo[4];
                for (int o = 0; o < 4; ++o) {
                    float v = wf[nd.vl_b + o];
                    for (int w8 = 0; w8 < 8; ++w8) v += Ev[w8 * 4 + o];
                    lo[o] = v;
                }
                const float mx = fmaxf(lo[0], fmaxf(lo[1], lo[2]));
                const float e0 = __expf(lo[0] - mx), e1 = __expf(lo[1] - mx), e2 = __expf(lo[2] - mx);
                const float inv = 1.0f / (e0 + e1 + e2);
                const h16 hv = (h16)((e2 - e0) * inv), h0 = (h16)lo[0], h1 = (h16)lo[1], h2 = (h16)lo[2], hm = (h16)(1.0f / (1.0f + __expf(-lo[3])));
                if constexpr (WT) {
                    auto bits = [](h16 x) { return __builtin_bit_cast(uint16_t, x); };
                    hmq::store2_wt(reinterpret_cast<uint16_t*>(value + sIdx), bits(hv));
                    hmq::store2_wt(reinterpret_cast<uint16_t*>(wdl + sIdx * 3 + 0), bits(h0));
                    hmq::store2_wt(reinterpret_cast<uint16_t*>(wdl + sIdx * 3 + 1), bits(h1));
                    hmq::store2_wt(reinterpret_cast<uint16_t*>(wdl + sIdx * 3 + 2), bits(h2));
                    hmq::store2_wt(reinterpret_cast<uint16_t*>(ml + sIdx), bits(hm));
                } else {
                    value[sIdx] = hv;
                    wdl[(size_t)sIdx * 3 + 0] = h0; wdl[(size_t)sIdx * 3 + 1] = h1; wdl[(size_t)sIdx * 3 + 2] = h2;
                    ml[sIdx] = hm;
                }
            }
            __syncthreads();                                             // Y1 (inside U) is about to be overwritten by Ss
        }
        HM_STAMP();
        // ---- policy heads: shared 3x3 conv C -> C (+bias, ReLU) into Ss, then 3x3 C -> 146 (two boards)
        for (int i = tid; i < ldx; i += 512) Ss[64 * ldx + i] = (h16)0.0f;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ctile = (wave + 8 * i) >> 1;
            if (ctile < ctiles) {
                const floatx16 acc = gemm_conv3<CTILES * 2, false>(zero16(), Xs, sqL, ldx, 64, kh, wh + nd.ps_w, ctiles, ctile, lane);
                const int co = ctile * 32 + (lane & 31);
                const float bias = wf[nd.ps_b + co];
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) Ss[(stile * 32 + drow(rg, lane)) * ldx + co] = (h16)fmaxf(acc[rg] + bias, 0.0f);
            }
        }
        __syncthreads();
        HM_STAMP();
        {
            // 146 output planes padded to 160 = 5 channel tiles x 2 square tiles = 10 tiles over 8 waves
            for (int t = wave; t < 10; t += 8) {
                const int ct = t >> 1;                                  // (t & 1) == stile
                const floatx16 e = gemm_conv3<CTILES * 2, true>(zero16(), Ss, sqL, ldx, 64, kh, wh + nd.pp_w, 5, ct, lane);
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int ch = ct * 32 + drow(rg, lane);
                    if constexpr (WT) Pol[ch * 64 + sqL] = (h16)e[rg];                      // [160][64] staging behind Ss
                    else {
                        if (ch < 73) piA[(size_t)sIdx * HM_POLICY_VALUES + ch * 64 + sqL] = (h16)e[rg];
                        else if (ch < 146) piB[(size_t)sIdx * HM_POLICY_VALUES + (ch - 73) * 64 + sqL] = (h16)e[rg];
                    }
                }
            }
        }
        __syncthreads();
        if constexpr (WT && PolicyEpi::present) {
            epi(Pol, reinterpret_cast<unsigned char*>(Pol + 160 * 64));
        } else if constexpr (WT) {
            // 146 planes x 64 squares = 1168 chunks of 8 halfs: one write-through 16-byte store each
            for (int i = tid; i < 146 * 8; i += 512) {
                const int ch = i >> 3, part = i & 7;
                const hmq::u32x4q v = *reinterpret_cast<const hmq::u32x4q*>(Pol + ch * 64 + part * 8);
                h16* dst = (ch < 73 ? piA + sIdx * HM_POLICY_VALUES + ch * 64 : piB + sIdx * HM_POLICY_VALUES + (ch - 73) * 64) + part * 8;
                hmq::store16_wt(dst, v);
            }
            __syncthreads();                                             // Pol lies in U: the next position's input staging overwrites it
        }
        HM_STAMP();
    }
}

// The same position body for a workgroup of FOUR waves (256 threads, one wave per SIMD with the whole register file): wave w owns
// the channel tiles w, w + 4, ... of every GEMM for BOTH square halves (gemm_tile2 / gemm_conv3_2 / expand_stream2), so a weight
// fragment is fetched once per workgroup and multiplies M = 64 rows.  This is the evaluator role of the single-launch search
// (hm_search.hip: k_rollout), whose search role needs 256-thread workgroups with all the registers a lane can have; results are
// bit-identical to the 8-wave form (per output element the same k order).
template <int CTILES, bool K5, bool WT = false, typename PolicyEpi = NoPolicyEpi>
__device__ __forceinline__ void narrow_position4(const NetDesc& nd, const h16* __restrict__ wh, const float* __restrict__ wf, const h16* pin, size_t sIdx,
                                                int copMax, int uHalfs, unsigned char* smem,
                                                h16* __restrict__ value, h16* __restrict__ piA, h16* __restrict__ piB, h16* __restrict__ wdl, h16* __restrict__ ml,
                                                unsigned long long* __restrict__ dbg, int& dbgN, PolicyEpi epi = PolicyEpi()) {
    // C = 64, 128 or 384: an N = C GEMM is CTILES column tiles x 2 square tiles = 2*CTILES tiles of 32x32 over 8 waves, i.e.
    // TPW tiles per wave (tile t = wave + 8i; its square half t & 1 is the wave's own).  copMax = the expanded channels kept
    // in LDS at a time: a block whose `cop` exceeds it (the 384-channel deployed net: cop up to 1152) runs its three phases
    // per chunk of copMax channels and carries the projection accumulators across the chunks in registers.
    constexpr int C = CTILES * 32, ldx = C + 8, ctiles = CTILES, TPW = (CTILES + 3) / 4, NPF = CTILES > 4 ? HM_NPF_WIDE4 : 8, NPFP = (CTILES > 4 && HM_NPF_PROJ > 0) ? HM_NPF_PROJ : NPF;
    h16* Xs = reinterpret_cast<h16*>(smem);                             // [65][ldx]
    h16* U = Xs + 65 * ldx;                                             // union region
    h16* Ss = U;                                                        // [65][ldx] (input staging uses pitch ldi)
    h16* Y1 = U;                                                        // [chunk][66]
    h16* Y2 = U + (size_t)copMax * 66;                                  // [64][chunk + 8]
    float* Pf = reinterpret_cast<float*>(U + uHalfs);                   // per-chunk parameters: b1[chunk], b2[chunk], b3[C]  (uHalfs % 8 == 0)
    h16* Pdw = reinterpret_cast<h16*>(Pf + 2 * copMax + C);             // depthwise weights [chunk][k*k]
    float* Ev = reinterpret_cast<float*>(Pdw + (((size_t)copMax * 25 + 7) & ~(size_t)7));   // ECA / head scratch: [4][C] + [C] + 64
    // WT: policy staging [160][64] (+ the prior pipeline's scratch) behind Ss inside the union region when it has the room (narrow
    // trunks: their expansions are wide against C), else in Xs, which is dead once the policy trunk has been computed (C = 384)
    constexpr int polOff = (65 * (ldx > 88 ? ldx : 88) + 7) & ~7;
    h16* Pol = polOff + 160 * 64 + PRIOR_SCRATCH_HALFS <= uHalfs ? U + polOff : Xs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = 8 * (lane >> 5);
    constexpr bool PF = HM_NET_PREFETCH != 0 && CTILES <= 4;           // (the 384-channel variant spills with the extra queue: 1.07 -> 1.24 ms per position)
    const int sq0 = lane & 31;                                          // the square this lane feeds as an A-row / B-column in half 0 (half 1: + 32)

    {
        HM_STAMP();
        // ---- input planes: NCHW [74][64] fp16 -> Ss as [sq][cin_pad] (+ zero row 64)
        const int ldi = nd.cin_pad + 8;
        {   // 16-byte stores / loads (Ss is 16-byte aligned: 65 * ldx * 2 is a multiple of 16 for C = 64 / 128 / 384; a plane row is 592 chunks of 8 halfs)
            const frag4 z = {0, 0, 0, 0};
            frag4* z16 = reinterpret_cast<frag4*>(Ss);
            for (int i = tid; i < (65 * ldi) >> 3; i += 256) z16[i] = z;
        }
        for (int i = tid; i < ldx; i += 256) Xs[64 * ldx + i] = (h16)0.0f;
        __syncthreads();
        for (int c = tid; c < HM_PLANE_VALUES / 8; c += 256) {           // chunk c = plane c / 8, squares 8 * (c % 8) ..
            const half8 v = *reinterpret_cast<const half8*>(pin + 8 * c);
            const int pl = c >> 3, sq8 = (c & 7) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) Ss[(sq8 + j) * ldi + pl] = v[j];
        }
        __syncthreads();
        HM_STAMP();   // input staged
        // ---- stem: 3x3 conv cin -> C (+bias, ReLU)
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ctile = wave + 4 * i;
            if (ctile < ctiles) {
                floatx16 a0 = zero16(), a1 = zero16();
                gemm_conv3_2<5, false>(a0, a1, Ss, sq0, ldi, 64, kh, wh + nd.stem_w, ctiles, ctile, lane);
                const int co = ctile * 32 + (lane & 31);
                const float bias = wf[nd.stem_b + co];
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    Xs[drow(rg, lane) * ldx + co] = (h16)fmaxf(a0[rg] + bias, 0.0f);
                    Xs[(32 + drow(rg, lane)) * ldx + co] = (h16)fmaxf(a1[rg] + bias, 0.0f);
                }
            }
        }
        // The expand GEMM's first weight fragments (this wave's first channel tile of the block's first chunk) are requested one phase
        // ahead: here for block 0, after each projection for the chunk / block that follows (see FragQ).
        FragQ<NPF> eq;
        if (PF && nd.nblocks > 0) fragq_fill<NPF>(eq, wh + nd.blk[0].w1, nd.blk[0].cop >> 5, wave, C >> 4, lane);
        __syncthreads();
        HM_STAMP();
        // ---- mobile bottleneck blocks: three phases each over (a chunk of) the `cop` expanded channels
        for (int bi = 0; bi < nd.nblocks; ++bi) {
            const BlockDesc bd = nd.blk[bi];
            const int cop = bd.cop, kk = bd.k;
            float* sb1 = Pf; float* sb2 = Pf + copMax; float* sb3 = Pf + 2 * copMax;
            for (int i = tid; i < C; i += 256) sb3[i] = wf[bd.b3 + i];
            if (bd.eca) {   // x = x * hardsigmoid(W_eca . mean_sq(x) + b)   (builder_util.py:49-80, centre tap)
                float* part = Ev;            // [4][C] partial sums, then partial dot products
                float* Mv = Ev + 4 * C;      // [C] channel means
                // work item = (quarter p of the squares / input channels, channel c): 4*C items over the workgroup
                for (int it = tid; it < 4 * C; it += 256) {
                    const int c = it % C, p = it / C;
                    h16 xv[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) xv[j] = Xs[(16 * p + j) * ldx + c];       // sixteen LDS reads in flight, then the adds in the same order
                    float sacc = 0.0f;
#pragma unroll
                    for (int j = 0; j < 16; ++j) sacc += (float)xv[j];
                    part[p * C + c] = sacc;
                }
                __syncthreads();
                for (int c = tid; c < C; c += 256) Mv[c] = (part[c] + part[C + c] + part[2 * C + c] + part[3 * C + c]) * (1.0f / 64.0f);
                __syncthreads();
                // W_eca (dense [ci][co], 2*C*C bytes from L2) . mean: work item = (1/32 of the input channels, 8 adjacent output
                // channels) -- 16-byte loads, 32*C/8 items = three per thread at C = 384, a thread's 12 rows requested together; the
                // 32 partial sums per channel go through the union region, which nothing else uses between two blocks
                float* part32 = reinterpret_cast<float*>(U);            // [32][C]
                {
                    const h16* we = wh + bd.ecaw;                       // [ci][co]
                    constexpr int CG = C >> 3, ROWS = C >> 5;
                    // (two work items per thread and round — a thread's 2 * ROWS row loads are requested together: with four waves a thread has
                    // two or more items, and one L2 round trip serves both)
                    for (int it0 = tid; it0 < 32 * CG; it0 += (CTILES <= 4 ? 512 : 256)) {
                        const int itB = it0 + 256;
                        const bool hasB = CTILES <= 4 && itB < 32 * CG;       // (the 384-channel variant: one item per round, its 12 row loads already fill the queue)
                        const int cgA = it0 % CG, pA = it0 / CG, cgB = hasB ? itB % CG : cgA, pB = hasB ? itB / CG : pA;
                        const h16* wrowA = we + (size_t)(pA * ROWS) * C + 8 * cgA;
                        const h16* wrowB = we + (size_t)(pB * ROWS) * C + 8 * cgB;
                        half8 wA[ROWS], wB[ROWS];
#pragma unroll
                        for (int u = 0; u < ROWS; ++u) { wA[u] = *reinterpret_cast<const half8*>(wrowA + (size_t)u * C); wB[u] = *reinterpret_cast<const half8*>(wrowB + (size_t)u * C); }
                        float sA[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, sB[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int u = 0; u < ROWS; ++u) {
                            const float mA = Mv[pA * ROWS + u], mB = Mv[pB * ROWS + u];
#pragma unroll
                            for (int k = 0; k < 8; ++k) { sA[k] += (float)wA[u][k] * mA; sB[k] += (float)wB[u][k] * mB; }
                        }
                        float* dA = part32 + pA * C + 8 * cgA;
#pragma unroll
                        for (int k = 0; k < 8; ++k) dA[k] = sA[k];
                        if (hasB) {
                            float* dB = part32 + pB * C + 8 * cgB;
#pragma unroll
                            for (int k = 0; k < 8; ++k) dB[k] = sB[k];
                        }
                    }
                }
                __syncthreads();
                for (int c = tid; c < C; c += 256) {
                    float sg = wf[bd.ecab + c];
#pragma unroll
                    for (int p = 0; p < 32; ++p) sg += part32[p * C + c];
                    Mv[c] = fminf(fmaxf(sg * (1.0f / 6.0f) + 0.5f, 0.0f), 1.0f);
                }
                __syncthreads();
#pragma unroll 4
                for (int i = tid; i < 32 * C; i += 256) {               // two adjacent channels per item
                    const int sq = i / (C >> 1), cc = 2 * (i - sq * (C >> 1));
                    half2v* px = reinterpret_cast<half2v*>(Xs + sq * ldx + cc);
                    half2v x2 = *px;
                    x2[0] = (h16)((float)x2[0] * Mv[cc]); x2[1] = (h16)((float)x2[1] * Mv[cc + 1]);
                    *px = x2;
                }
            }
            floatx16 pacc[TPW][2];                                      // projection accumulators (channel tile, square half), carried over the chunks
#pragma unroll
            for (int i = 0; i < TPW; ++i) { pacc[i][0] = zero16(); pacc[i][1] = zero16(); }
            for (int ch0 = 0; ch0 < cop; ch0 += copMax) {
                const int chunk = min(copMax, cop - ch0), ld2 = chunk + 8;
                // parameter stage of this chunk (the previous chunk's phase 3 no longer reads Pf / Pdw: barrier at its end)
                {
                    for (int i = tid; i < chunk; i += 256) { sb1[i] = wf[bd.b1 + ch0 + i]; sb2[i] = wf[bd.b2 + ch0 + i]; }
                    // depthwise taps of the chunk, 16 bytes per lane (offsets and counts are multiples of 8 halfs: chunk % 32 == 0)
                    const half8* src = reinterpret_cast<const half8*>(wh + bd.dw + (size_t)ch0 * kk * kk);
                    half8* dst = reinterpret_cast<half8*>(Pdw);
                    for (int i = tid; i < (chunk * kk * kk) >> 3; i += 256) dst[i] = src[i];
                }
                const int copTiles = cop >> 5, tiles = (chunk >> 5) * 2, ct0 = ch0 >> 5;
                __syncthreads();
                HM_STAMP();   // parameters staged (+ ECA)
                // phase 1 — 1x1 expand, transposed: Y1[ch][sq] = relu(W1^T x^T + b1); chunk/32 channel tiles over 4 waves, both square halves each
                {
                    const h16* brow0 = Xs + sq0 * ldx + kh;
                    const h16* brow1 = brow0 + 32 * ldx;
                    const int ctl = chunk >> 5;                          // channel tiles of this chunk
                    auto store = [&](int ct, const floatx16& e0, const floatx16& e1) {
#pragma unroll
                        for (int rg = 0; rg < 16; ++rg) {
                            const int ch = ct * 32 + drow(rg, lane);
                            const float b = sb1[ch];
                            Y1[ch * 66 + sq0] = (h16)fmaxf(e0[rg] + b, 0.0f);
                            Y1[ch * 66 + 32 + sq0] = (h16)fmaxf(e1[rg] + b, 0.0f);
                        }
                    };
                    if constexpr ((C >> 4) == NPF) {                    // narrow trunks: K = one queue, a wave's tiles as one fragment stream
                        const int ntile = ctl > wave ? (ctl - wave + 3) >> 2 : 0;
                        if (ntile > 0 && !PF)
                            expand_stream2<NPF>(brow0, brow1, C >> 4, wh + bd.w1, copTiles, ct0 + wave, 4, ntile, lane,
                                                [&](int i, const floatx16& e0, const floatx16& e1) { store(wave + 4 * i, e0, e1); });
                        else if (ntile > 0)
                            expand_stream2_q<NPF>(eq, brow0, brow1, C >> 4, wh + bd.w1, copTiles, ct0 + wave, 4, ntile, lane,
                                                  [&](int i, const floatx16& e0, const floatx16& e1) { store(wave + 4 * i, e0, e1); });
                    } else {
                        for (int ct = wave; ct < ctl; ct += 4) {
                            floatx16 e0 = zero16(), e1 = zero16();
                            if (PF && ct == wave) gemm_tile2_q<true, NPF>(eq, e0, e1, brow0, brow1, C >> 4, wh + bd.w1, copTiles, ct0 + ct, lane);
                            else gemm_tile2<true, NPF>(e0, e1, brow0, brow1, C >> 4, wh + bd.w1, copTiles, ct0 + ct, lane);
                            store(ct, e0, e1);
                        }
                    }
                }
                __syncthreads();
                HM_STAMP();   // expand done
                // phase 2 — depthwise kxk (+bias, ReLU): work item = (channel, pair of board rows), two items per thread and round so that
                // one item's LDS reads overlap the other's arithmetic (a thread has eight items and its SIMD no second wave).  (Requesting
                // the projection's first weight fragments before this phase, possible with 512 registers per lane, was measured: no gain.)
                const h16* w2c = wh + bd.w2 + (size_t)(ch0 >> 4) * ctiles * 512;
                if (!K5 || kk == 3) {
                    // two 3x3 items per thread and round: one item's LDS reads overlap the other's arithmetic
                    for (int item = tid; item < chunk * 4; item += 512) {
                        const int itB = item + 256;
                        const bool hasB = itB < chunk * 4;
                        const int chA = item % chunk, gA = item / chunk, chB = hasB ? itB % chunk : chA, gB = hasB ? itB / chunk : gA;
                        depthwise_rows_ld<3>(Y1 + chA * 66, Y2, ld2, chA, gA, Pdw + (size_t)chA * 9, sb2[chA]);
                        if (hasB) depthwise_rows_ld<3>(Y1 + chB * 66, Y2, ld2, chB, gB, Pdw + (size_t)chB * 9, sb2[chB]);
                    }
                } else {
                    for (int item = tid; item < chunk * 4; item += 256) {
                        const int ch = item % chunk, g = item / chunk;
                        depthwise_rows_ld<5>(Y1 + ch * 66, Y2, ld2, ch, g, Pdw + (size_t)ch * 25, sb2[ch]);
                    }
                }
                __syncthreads();
                HM_STAMP();   // depthwise done
                // (Fetching the NEXT block's biases and taps into registers here, to be stored at that block's start, was measured: the
                // parameter stage shrinks by 0.5 k cycles and the projection behind the extra loads grows by 0.9 k.)
                // phase 3 — 1x1 project, K = this chunk's channels: k-steps ch0/16 .. of W2
                // (streaming a wave's TPW tiles through one fragment queue, as the narrow expand phase does, was measured on the
                // 384-channel variant: 108 spilled VGPRs instead of 42; kept: one call per tile)
                constexpr int QP = HM_NPF_PROJ4 > 0 ? HM_NPF_PROJ4 : 1;  // k-steps of a full chunk of the 384-channel network (chunk 320): its tiles as one stream
                if (HM_NPF_PROJ4 > 0 && CTILES > 4 && (chunk >> 4) == QP) {
                    const int ntile = ctiles > wave ? (ctiles - wave + 3) >> 2 : 0;
                    if (ntile > 0) project_stream2<QP, TPW>(pacc, Y2 + sq0 * ld2 + kh, Y2 + (sq0 + 32) * ld2 + kh, w2c, ctiles, wave, 4, ntile, lane);
                } else {
#pragma unroll
                    for (int i = 0; i < TPW; ++i) {
                        const int ctile = wave + 4 * i;
                        if (ctile < ctiles)
                            gemm_tile2<false, NPFP>(pacc[i][0], pacc[i][1], Y2 + sq0 * ld2 + kh, Y2 + (sq0 + 32) * ld2 + kh, chunk >> 4, w2c, ctiles, ctile, lane);
                    }
                }
                {   // the expand fragments of what comes next: this block's next chunk, or the next block's first one
                    const bool more = ch0 + copMax < cop;
                    if (!PF) {}
                    else if (more) fragq_fill<NPF>(eq, wh + bd.w1, copTiles, ((ch0 + copMax) >> 5) + wave, C >> 4, lane);
                    else if (bi + 1 < nd.nblocks) fragq_fill<NPF>(eq, wh + nd.blk[bi + 1].w1, nd.blk[bi + 1].cop >> 5, wave, C >> 4, lane);
                }
                // (no barrier here: the next chunk's parameter stage and expand phase touch Pf / Pdw / Y1, which this phase does
                // not read, and its depthwise phase — the next writer of Y2 — starts behind the barrier after its expand phase)
            }
            // + bias + residual
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int ctile = wave + 4 * i;
                if (ctile < ctiles) {
                    const int co = ctile * 32 + (lane & 31);
                    const float bias = sb3[co];
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int sq = drow(rg, lane);
                        Xs[sq * ldx + co] = (h16)((float)Xs[sq * ldx + co] + pacc[i][0][rg] + bias);
                        Xs[(sq + 32) * ldx + co] = (h16)((float)Xs[(sq + 32) * ldx + co] + pacc[i][1][rg] + bias);
                    }
                }
            }
            __syncthreads();
            HM_STAMP();
        }
        // ---- value head: 1x1 conv C -> cv (+bias, ReLU), NCHW flatten, linear -> (wdl x3, plys)
        {
            const int cv = nd.cv;
            if (wave < 2) {                                             // one square half per wave (two waves work instead of one with both halves)
                const int sqh = wave * 32 + sq0;
                const floatx16 e = gemm_tile<true, NPF>(zero16(), Xs + sqh * ldx + kh, C >> 4, wh + nd.v_w, 1, 0, lane);
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int ch = drow(rg, lane);
                    if (ch < cv) Y1[ch * 66 + sqh] = (h16)fmaxf(e[rg] + wf[nd.v_b + ch], 0.0f);
                }
            }
            __syncthreads();
            // (the partial sums are grouped as the 8-wave form groups them — thread t of 512 takes i = t, t + 512, ...; its wave's 64 lanes are
            // reduced together; the eight wave sums are added in order — so that both forms round identically: each thread here plays
            // the threads t and t + 256 of that form)
            const h16* wl = wh + nd.vl_w;
#pragma unroll
            for (int vh = 0; vh < 2; ++vh) {
                float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                for (int i = tid + 256 * vh; i < cv * 64; i += 512) {
                    const float v = (float)Y1[(i >> 6) * 66 + (i & 63)];
#pragma unroll
                    for (int o = 0; o < 4; ++o) part[o] += v * (float)wl[(size_t)o * cv * 64 + i];
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float v = part[o];
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                    if (lane == 0) Ev[(wave + 4 * vh) * 4 + o] = v;
                }
            }
            __syncthreads();
            if (tid == 0) {
                float lo[4];
                for (int o = 0; o < 4; ++o) {
                    float v = wf[nd.vl_b + o];
                    for (int w8 = 0; w8 < 8; ++w8) v += Ev[w8 * 4 + o];
                    lo[o] = v;
                }
                const float mx = fmaxf(lo[0], fmaxf(lo[1], lo[2]));
                const float e0 = __expf(lo[0] - mx), e1 = __expf(lo[1] - mx), e2 = __expf(lo[2] - mx);
                const float inv = 1.0f / (e0 + e1 + e2);
                const h16 hv = (h16)((e2 - e0) * inv), h0 = (h16)lo[0], h1 = (h16)lo[1], h2 = (h16)lo[2], hm = (h16)(1.0f / (1.0f + __expf(-lo[3])));
                if constexpr (WT) {
                    auto bits = [](h16 x) { return __builtin_bit_cast(uint16_t, x); };
                    hmq::store2_wt(reinterpret_cast<uint16_t*>(value + sIdx), bits(hv));
                    hmq::store2_wt(reinterpret_cast<uint16_t*>(wdl + sIdx * 3 + 0), bits(h0));
                    hmq::store2_wt(reinterpret_cast<uint16_t*>(wdl + sIdx * 3 + 1), bits(h1));
                    hmq::store2_wt(reinterpret_cast<uint16_t*>(wdl + sIdx * 3 + 2), bits(h2));
                    hmq::store2_wt(reinterpret_cast<uint16_t*>(ml + sIdx), bits(hm));
                } else {
                    value[sIdx] = hv;
                    wdl[(size_t)sIdx * 3 + 0] = h0; wdl[(size_t)sIdx * 3 + 1] = h1; wdl[(size_t)sIdx * 3 + 2] = h2;
                    ml[sIdx] = hm;
                }
            }
            __syncthreads();                                             // Y1 (inside U) is about to be overwritten by Ss
        }
        HM_STAMP();
        // ---- policy heads: shared 3x3 conv C -> C (+bias, ReLU) into Ss, then 3x3 C -> 146 (two boards)
        for (int i = tid; i < ldx; i += 256) Ss[64 * ldx + i] = (h16)0.0f;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int ctile = wave + 4 * i;
            if (ctile < ctiles) {
                floatx16 a0 = zero16(), a1 = zero16();
                gemm_conv3_2<CTILES * 2, false>(a0, a1, Xs, sq0, ldx, 64, kh, wh + nd.ps_w, ctiles, ctile, lane);
                const int co = ctile * 32 + (lane & 31);
                const float bias = wf[nd.ps_b + co];
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    Ss[drow(rg, lane) * ldx + co] = (h16)fmaxf(a0[rg] + bias, 0.0f);
                    Ss[(32 + drow(rg, lane)) * ldx + co] = (h16)fmaxf(a1[rg] + bias, 0.0f);
                }
            }
        }
        __syncthreads();
        HM_STAMP();
        {
            // 146 output planes padded to 160 = 5 channel tiles: tiles 0..3 with both square halves on waves 0..3, the two halves of tile 4
            // as single tiles on waves 1 and 2 (five pairs over four waves would leave one wave with twice the work)
            auto put = [&](int ch, int sq, float v) {
                if constexpr (WT) Pol[ch * 64 + sq] = (h16)v;                                   // [160][64] staging behind Ss
                else {
                    if (ch < 73) piA[(size_t)sIdx * HM_POLICY_VALUES + ch * 64 + sq] = (h16)v;
                    else if (ch < 146) piB[(size_t)sIdx * HM_POLICY_VALUES + (ch - 73) * 64 + sq] = (h16)v;
                }
            };
            {
                floatx16 e0 = zero16(), e1 = zero16();
                gemm_conv3_2<CTILES * 2, true>(e0, e1, Ss, sq0, ldx, 64, kh, wh + nd.pp_w, 5, wave, lane);
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int ch = wave * 32 + drow(rg, lane);
                    put(ch, sq0, e0[rg]); put(ch, 32 + sq0, e1[rg]);
                }
            }
            if (wave == 1 || wave == 2) {
                const int sqh = (wave - 1) * 32 + sq0;
                const floatx16 e = gemm_conv3<CTILES * 2, true>(zero16(), Ss, sqh, ldx, 64, kh, wh + nd.pp_w, 5, 4, lane);
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) put(4 * 32 + drow(rg, lane), sqh, e[rg]);
            }
        }
        __syncthreads();
        if constexpr (WT && PolicyEpi::present) {
            epi(Pol, reinterpret_cast<unsigned char*>(Pol + 160 * 64));
        } else if constexpr (WT) {
            // 146 planes x 64 squares = 1168 chunks of 8 halfs: one write-through 16-byte store each
            for (int i = tid; i < 146 * 8; i += 256) {
                const int ch = i >> 3, part = i & 7;
                const hmq::u32x4q v = *reinterpret_cast<const hmq::u32x4q*>(Pol + ch * 64 + part * 8);
                h16* dst = (ch < 73 ? piA + sIdx * HM_POLICY_VALUES + ch * 64 : piB + sIdx * HM_POLICY_VALUES + (ch - 73) * 64) + part * 8;
                hmq::store16_wt(dst, v);
            }
            __syncthreads();                                             // Pol lies in U: the next position's input staging overwrites it
        }
        HM_STAMP();
    }
}

// The prior pipeline of an evaluated leaf (hm_policy.hpp) on the logits still in LDS: wave 0 serves board A, wave 1 board B — legal
// moves in from the search workgroup's list, moves and priors in prior order out (write-through, 16 bytes per lane) to the arrays
// the search workgroup copies into its tree.  The 18.7 KB of logits per position are never written to HBM.
constexpr int PRIOR_SCRATCH_BYTES = 2 * PRIOR_SCRATCH_HALFS;
struct PriorEpi {
    static constexpr bool present = true;
    const hmq::ServeArgs* a;
    size_t slot;             // (game * 2 + buffer) * 8 + row: index of this leaf's lists
    int game;
    bool root;
    __device__ __forceinline__ void operator()(const h16* pol, unsigned char* scratch) const {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (wave < 2) {
            const int b = wave;
            unsigned char* base = scratch + (size_t)b * (PRIOR_SCRATCH_BYTES / 2);
            uint32_t* list = reinterpret_cast<uint32_t*>(base);
            float* pr = reinterpret_cast<float*>(list + HM_MAX_MOVES);           // (the list's MOVE_NONE may land in pr[0]'s place only at n = 512: lists are < 512)
            uint32_t* outM = reinterpret_cast<uint32_t*>(pr + HM_MAX_MOVES + 8);
            float* outP = reinterpret_cast<float*>(outM + HM_MAX_MOVES);
            const int cnt = a->leafCounts[slot * 2 + b];
            const int n = cnt & 0xffff, stm = (cnt >> 16) & 1;
            const uint32_t* src = a->leafMoves + (slot * 2 + b) * HM_MAX_MOVES;
            for (int i = lane; i < n; i += 64) list[i] = src[i];
            __builtin_amdgcn_wave_barrier();
            const h16* head = pol + (b ? 73 * 64 : 0);
            const float* nz = (root && a->noiseOn) ? a->noise + ((size_t)game * 2 + b) * hmp::NOISE_CAP : nullptr;
            const int nAct = hmp::board_priors_sorted(list, pr, n, stm, a->polNormal, a->polDrop,
                                                      [head](int idx) { return __builtin_bit_cast(uint16_t, head[idx]); }, nz, a->noiseEps, outM, outP);
            __builtin_amdgcn_wave_barrier();
            uint32_t* dm = a->sortedMoves + (slot * 2 + b) * HM_MAX_MOVES;
            float* dp = a->sortedPriors + (slot * 2 + b) * HM_MAX_MOVES;
            for (int c = lane; c * 4 < nAct; c += 64) {                          // 16-byte chunks (the rows are 2 KB, 16-byte aligned)
                hmq::store16_wt(dm + c * 4, *reinterpret_cast<const hmq::u32x4q*>(outM + c * 4));
                hmq::store16_wt(dp + c * 4, *reinterpret_cast<const hmq::u32x4q*>(outP + c * 4));
            }
        }
        __syncthreads();                                             // the scratch lies in the union region the next position stages into
    }
};

#undef HM_STAMP

}  // namespace hmn
