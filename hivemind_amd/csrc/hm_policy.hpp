// hm_policy.hpp — the per-board prior pipeline of a network leaf, shared by the tree kernels (hm_search.hip: expand_leaf) and the
// persistent evaluator (hm_net.hip: rise_serve, which runs it on the logits while they are still in LDS):
//   get_normalized_probability (common/utils.h:226-243; normalize_logits :101-141 with its non-finite rules)
//   -> root Dirichlet mix (search/node.h:286-315; the gamma draws are made on the host)
//   -> the rank sort of JointCandidateGenerator::initialize (environment/joint_action.h:195-278), strict total order
//      (prior desc, index asc).
// One wavefront per board; every float operation is written out (no libm, -ffp-contract=off) so that both callers — and the CPU
// restatement oracle/search.hpp — produce the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace hmp {

constexpr unsigned MT_MASK = 15u << 12, MT_PROMOTION = 3u << 12, MT_DROP = 4u << 12;   // move type field (types.h:237-263)
constexpr unsigned PT_KNIGHT = 2;
constexpr unsigned CAPTURE_BIT = 0x80000000u;                  // bit 31 of a listed move: Position::capture(m)
constexpr int NOISE_CAP = 320;                                 // > max actions per board (304 + pass)

__device__ __forceinline__ float ufirstf_(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// exp for x <= 0 from IEEE +,*,fma,rint only (identical copy: oracle/search.hpp portable_expf)
__device__ __forceinline__ float hm_expf(float x) {
    if (!(x > -87.0f)) return 0.0f;
    if (x > 0.0f) x = 0.0f;
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(-n, 0.693145751953125f, x);
    r = __builtin_fmaf(-n, 1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    p = __builtin_fmaf(p * r, r, r) + 1.0f;
    int bits = __float_as_int(p);
    bits += (int)n << 23;
    return __int_as_float(bits);
}
__device__ __forceinline__ float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31;
    uint32_t m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int s = 0; while (!(m & 0x400u)) { m <<= 1; ++s; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - s) << 23) | (m << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    return __int_as_float((int)x);
}
__device__ __forceinline__ bool finite_f(float v) { return (__float_as_int(v) & 0x7f800000) != 0x7f800000; }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(hi, fmaxf(lo, v)); }

// index of move m (capture bit ignored) in the policy head of the side `stm` (get_fast_policy_index, utils.h:183-216); -1 = none
__device__ __forceinline__ int policy_index(const int* polNormal /*[2][64][64][2]*/, const int* polDrop /*[2][64][8]*/, int stm, uint32_t m) {
    m &= ~CAPTURE_BIT;
    if (m == 0) return 0;
    if ((m & MT_MASK) == MT_DROP) return polDrop[(stm * 64 + (int)(m & 63)) * 8 + (int)((m >> 16) & 63)];
    const int f = (int)((m >> 6) & 63), to = (int)(m & 63);
    const int knight = ((m & MT_MASK) == MT_PROMOTION && ((m >> 16) & 63) == PT_KNIGHT) ? 1 : 0;
    return polNormal[((stm * 64 + f) * 64 + to) * 2 + knight];
}

// Priors of one board of a leaf.  list[0..n): the board's legal moves (bit 31 = capture; R/B under-promotions already erased),
// list[n] receives MOVE_NONE; pr: n + 1 floats of scratch; logit(idx) -> the fp16 policy logit idx of this board's head;
// noise != nullptr: the root's Dirichlet draws of this board.  Outputs, in rank order: outM[r] = move | capture bit, outP[r] = prior.
// Returns the action count n + 1.  `list`, `pr` in LDS; one wavefront; the caller brackets the call with wave barriers.
template <typename Logit>
__device__ __forceinline__ int board_priors_sorted(uint32_t* list, float* pr, int n, int stm, const int* polNormal, const int* polDrop, Logit logit,
                                                   const float* noise, float noiseEps, uint32_t* outM, float* outP) {
    const int lane = threadIdx.x & 63;
    list[n] = 0;                                               // MOVE_NONE appended last
    const int nAct = n + 1;
    __builtin_amdgcn_wave_barrier();
    if (n == 0) pr[0] = 1.0f;
    else {
        // get_normalized_probability (utils.h:226-243): gather fp16 logits through the policy tables
        float mx = -INFINITY;
        for (int i = lane; i < n + 1; i += 64) {
            const int idx = policy_index(polNormal, polDrop, stm, list[i]);
            const float lg = idx >= 0 ? h2f(logit(idx)) : -INFINITY;
            pr[i] = lg;
            if (finite_f(lg)) mx = fmaxf(mx, lg);
        }
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        mx = ufirstf_(mx);
        __builtin_amdgcn_wave_barrier();
        if (!finite_f(mx)) {                                   // normalize_logits fallback :136-141
            for (int i = lane; i < n + 1; i += 64) pr[i] = 1.0f / (float)(n + 1);
        } else {
            for (int i = lane; i < n + 1; i += 64) { const float lg = pr[i]; pr[i] = finite_f(lg) ? hm_expf(lg - mx) : 0.0f; }
            __builtin_amdgcn_wave_barrier();
            double sum = 0.0;                                  // index-order double sum, as the reference
            for (int i = 0; i < n + 1; ++i) sum += (double)pr[i];
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < n + 1; i += 64) pr[i] = (float)((double)pr[i] / sum);
        }
    }
    __builtin_amdgcn_wave_barrier();
    // root Dirichlet noise (node.h:286-315): gamma draws were made on the host
    if (noise && nAct > 1) {
        float total = 0.0f;
        for (int i = 0; i < nAct && i < NOISE_CAP; ++i) total += noise[i];
        if (total > 0.0f) {
            const float eps = clampf(noiseEps, 0.0f, 1.0f);
            for (int i = lane; i < nAct; i += 64) pr[i] = (1.0f - eps) * pr[i] + eps * noise[i < NOISE_CAP ? i : NOISE_CAP - 1] / total;
            __builtin_amdgcn_wave_barrier();
        }
    }
    // JointCandidateGenerator::initialize (joint_action.h:195-278): rank sort by (prior desc, index asc)
    for (int i = lane; i < nAct; i += 64) {
        const float pi = pr[i];
        int rank = 0;
        for (int j = 0; j < nAct; ++j) { const float pj = pr[j]; rank += (pj > pi) || (pj == pi && j < i); }
        outM[rank] = list[i];                                  // capture bit travels with the move; MOVE_NONE has none
        outP[rank] = pi;
    }
    return nAct;
}

}  // namespace hmp
