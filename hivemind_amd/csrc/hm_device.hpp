// hm_device.hpp — gfx950 device-side Bughouse primitives (bitboards in VGPRs, one position
// per lane; tables staged in LDS by the calling kernel).
//
// Behavioural contract (reference file:line, relative to engine/src):
//   legal move set AND list order  = generate<LEGAL>  Fairy-Stockfish/src/movegen.cpp:311-456
//   legality predicate             = Position::legal  Fairy-Stockfish/src/position.cpp:949-1140
//   make                           = Position::do_move position.cpp:1349-1859 + Board::make_moves
//                                    environment/board.cc:316-341 (partner-hand transfer :101-104)
//   Zobrist key                    = position.cpp:142-167, 560-607 (bit-identical)
// Design (not a port): no magic tables — sliders use hyperbola quintessence with the
// hardware bit-reverse (v_bfrev_b32), so attack generation is pure VALU and the only tables
// are 3 KB of leaper/diagonal masks + 7 KB of Zobrist keys in LDS; legality is decided from
// one pinned-piece / king-danger analysis per position instead of one attackers_to() per
// pseudo-legal move; positions are 96-byte PODs and make is copy-make.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hivemind_amd.h"

namespace hmd {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 FILE_A = 0x0101010101010101ULL;
constexpr u64 FILE_H = FILE_A << 7;
constexpr u64 RANK_1 = 0xFFULL;
constexpr u64 RANK_8 = RANK_1 << 56;

// Tables every kernel stages into LDS (block-cooperative copy from global).
struct alignas(16) AttackTab {          // 3 KB; 16-byte aligned: stage_table moves 16 bytes per lane
    u64 knight[64];
    u64 king[64];
    u64 pawn[2][64];        // pawn attacks by colour
    u64 diag[64];           // a1-h8 direction line through s, excluding s
    u64 anti[64];           // h1-a8 direction line through s, excluding s
};
struct alignas(16) ZobristTab {         // 6 KB + small
    u64 psq[2][6][64];      // [colour][P..K][sq]
    u64 in_hand[2][5][32];  // [colour][P..Q][count]
    u64 ep[8];
    u64 castle[16];
    u64 side;
    u64 time_adv;
};
struct DeviceTables {
    AttackTab att;
    ZobristTab zob;
    int pol_normal[2][64][64][2];
    int pol_drop[2][64][8];
};

__device__ __forceinline__ int lsb(u64 b) { return __builtin_ctzll(b); }
__device__ __forceinline__ int popc(u64 b) { return __popcll(b); }
__device__ __forceinline__ u64 bit(int s) { return 1ULL << s; }
__device__ __forceinline__ int pop_lsb(u64& b) { int s = lsb(b); b &= b - 1; return s; }

// Cooperative global->LDS copy of a POD table (8-byte granules).
template <typename T>
__device__ __forceinline__ void stage_table(T* lds, const T* g) {
    const u64* src = reinterpret_cast<const u64*>(g);
    u64* dst = reinterpret_cast<u64*>(lds);
    for (unsigned i = threadIdx.x; i < sizeof(T) / 8; i += blockDim.x) dst[i] = src[i];
}
// The same with 16 bytes per lane and a thread's four requests in flight before the first is stored: the 12.5 KB rules table is
// one round trip for a 256-thread block instead of six (the lockstep kernels stage it at the top of every launch).  Costs 16
// VGPRs, which the occupancy-bound kernels (perft: 2.75 -> 4.2 s with this form) cannot spare: they keep the plain copy.
// nthreads: the threads that take part (default: the whole block; k_rollout's search role runs on the first 256 threads of a block
// that may have 512)
template <typename T>
__device__ __forceinline__ void stage_table_wide(T* lds, const T* g, unsigned nthreads = blockDim.x) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    static_assert(sizeof(T) % 16 == 0 && alignof(T) >= 16, "16-byte aligned table");
    constexpr unsigned N16 = sizeof(T) / 16;
    const u32x4* src = reinterpret_cast<const u32x4*>(g);
    u32x4* dst = reinterpret_cast<u32x4*>(lds);
    for (unsigned i0 = threadIdx.x; i0 < N16; i0 += 4 * nthreads) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const unsigned i = i0 + u * nthreads; if (i < N16) v[u] = src[i]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const unsigned i = i0 + u * nthreads; if (i < N16) dst[i] = v[u]; }
    }
}

// ---- sliders: hyperbola quintessence with bit reversal -------------------------------
__device__ __forceinline__ u64 line_att(u64 occ, u64 mask_ex, int s) {
    u64 o = occ & mask_ex;
    u64 fwd = o - (bit(s) << 1);
    u64 rev = __brevll(__brevll(o) - (bit(63 - s) << 1));
    return (fwd ^ rev) & mask_ex;
}
__device__ __forceinline__ u64 rank_mask_ex(int s) { return (RANK_1 << (s & 56)) ^ bit(s); }
__device__ __forceinline__ u64 file_mask_ex(int s) { return (FILE_A << (s & 7)) ^ bit(s); }
__device__ __forceinline__ u64 rook_att(int s, u64 occ) {
    return line_att(occ, rank_mask_ex(s), s) | line_att(occ, file_mask_ex(s), s);
}
__device__ __forceinline__ u64 bishop_att(const AttackTab& t, int s, u64 occ) {
    return line_att(occ, t.diag[s], s) | line_att(occ, t.anti[s], s);
}

// Full line (incl. both squares' own bits on it) through a and b, or 0 when not aligned.
__device__ __forceinline__ u64 line_through(const AttackTab& t, int a, int b) {
    if ((a ^ b) < 8) return RANK_1 << (a & 56);          // same rank: a>>3 == b>>3
    if (((a ^ b) & 7) == 0) return FILE_A << (a & 7);
    u64 bb = bit(b);
    if (t.diag[a] & bb) return t.diag[a] | bit(a);
    if (t.anti[a] & bb) return t.anti[a] | bit(a);
    return 0;
}
// between_bb(a,b) with the reference's semantic (bitboard.h:307-320): excludes a, includes b;
// for non-aligned squares just b.
__device__ __forceinline__ u64 between_incl(const AttackTab& t, int a, int b) {
    u64 line = line_through(t, a, b);
    int lo = a < b ? a : b, hi = a < b ? b : a;
    u64 seg = (bit(hi) - 1) & ~((bit(lo) << 1) - 1);     // strictly between by index
    return (line & seg) | bit(b);
}

struct P {                 // register view of hm_pos
    u64 bt[6], bc[2], promoted, key;
    u32 hand[2];           // 5 x 6-bit counts per colour: bits 6*(pt-1)
    u32 castling, ep, stm, rule50, ply;
};
// P must stay in registers: every runtime-indexed access goes through these select helpers (a
// runtime subscript on a member array would demote the whole struct to scratch memory).
__device__ __forceinline__ u64 bc_of(const P& p, int c) { return c ? p.bc[1] : p.bc[0]; }
__device__ __forceinline__ void bc_xor(P& p, int c, u64 m) { p.bc[0] ^= c ? 0ULL : m; p.bc[1] ^= c ? m : 0ULL; }
__device__ __forceinline__ void bt_xor(P& p, int idx, u64 m) {
#pragma unroll
    for (int i = 0; i < 6; ++i) p.bt[i] ^= i == idx ? m : 0ULL;
}
__device__ __forceinline__ int hand_get(const P& p, int c, int pt) { return ((c ? p.hand[1] : p.hand[0]) >> (6 * (pt - 1))) & 63; }
__device__ __forceinline__ void hand_add(P& p, int c, int pt, int d) {
    const u32 v = (u32)d << (6 * (pt - 1));
    p.hand[0] += c ? 0u : v; p.hand[1] += c ? v : 0u;
}
__device__ __forceinline__ P pick_pos(const P* bd, int b) {      // bd[b] without a runtime subscript
    P r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.bt[i] = b ? bd[1].bt[i] : bd[0].bt[i];
    r.bc[0] = b ? bd[1].bc[0] : bd[0].bc[0]; r.bc[1] = b ? bd[1].bc[1] : bd[0].bc[1];
    r.promoted = b ? bd[1].promoted : bd[0].promoted; r.key = b ? bd[1].key : bd[0].key;
    r.hand[0] = b ? bd[1].hand[0] : bd[0].hand[0]; r.hand[1] = b ? bd[1].hand[1] : bd[0].hand[1];
    r.castling = b ? bd[1].castling : bd[0].castling; r.ep = b ? bd[1].ep : bd[0].ep; r.stm = b ? bd[1].stm : bd[0].stm;
    r.rule50 = b ? bd[1].rule50 : bd[0].rule50; r.ply = b ? bd[1].ply : bd[0].ply;
    return r;
}
__device__ __forceinline__ void put_pos(P* bd, int b, const P& x) { if (b) bd[1] = x; else bd[0] = x; }

__device__ __forceinline__ void load_pos(P& p, const hm_pos* g) {
    const u64* w = reinterpret_cast<const u64*>(g);
#pragma unroll
    for (int i = 0; i < 6; ++i) p.bt[i] = w[i];
    p.bc[0] = w[6]; p.bc[1] = w[7]; p.promoted = w[8]; p.key = w[9];
    u64 a = w[10], b = w[11];                             // 16 tail bytes
    // bytes: hand[0][0..4], hand[1][0..4], castling, ep, stm, rule50, game_ply(2)
    p.hand[0] = p.hand[1] = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) p.hand[0] |= (u32)((a >> (8 * i)) & 63) << (6 * i);
#pragma unroll
    for (int i = 0; i < 3; ++i) p.hand[1] |= (u32)((a >> (8 * (5 + i))) & 63) << (6 * i);
    p.hand[1] |= (u32)(b & 63) << 18;
    p.hand[1] |= (u32)((b >> 8) & 63) << 24;
    p.castling = (u32)(b >> 16) & 0xff;
    p.ep = (u32)(b >> 24) & 0xff;
    p.stm = (u32)(b >> 32) & 0xff;
    p.rule50 = (u32)(b >> 40) & 0xff;
    p.ply = (u32)(b >> 48) & 0xffff;
}
// Results of wave reductions / lane-0 work are equal in every lane but the compiler cannot know it; these
// hand them back as scalar values so that the control flow and arithmetic that follow stay on the scalar unit.
__device__ __forceinline__ int ufirst(int v) { return (int)__builtin_amdgcn_readfirstlane((u32)v); }
__device__ __forceinline__ float ufirstf(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ int ulane(int v, int lane) { return (int)__builtin_amdgcn_readlane((u32)v, lane); }
__device__ __forceinline__ float ulanef(float v, int lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane)); }
// Wave-wide maximum of a 64-bit key without the LDS crossbar: four DPP steps reduce each row of 16 lanes (quad swaps,
// half-row mirror, row mirror — any pairing works for a commutative reduction), the four row results are read out as
// scalars.  Arg-max users pack (order-preserving image of the score) << 32 | ~index, so the larger key is the better
// score and, among equal scores, the lower index.
template <int CTRL>
__device__ __forceinline__ u64 dpp_max_step(u64 k) {
    const u32 lo = (u32)k, hi = (u32)(k >> 32);
    const u32 olo = (u32)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    const u32 ohi = (u32)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    const u64 o = ((u64)ohi << 32) | olo;
    return o > k ? o : k;
}
__device__ __forceinline__ u64 wave_max_u64(u64 k) {
#ifdef HM_NO_DPP
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = ((u64)(u32)__shfl_xor((int)(u32)(k >> 32), off) << 32) | (u32)__shfl_xor((int)(u32)k, off);
        k = o > k ? o : k;
    }
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(k >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((u32)k);
#endif
    k = dpp_max_step<0xB1>(k);      // quad_perm [1,0,3,2]
    k = dpp_max_step<0x4E>(k);      // quad_perm [2,3,0,1]
    k = dpp_max_step<0x141>(k);     // row_half_mirror
    k = dpp_max_step<0x140>(k);     // row_mirror
    const u32 lo = (u32)k, hi = (u32)(k >> 32);
    u64 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = ((u64)(u32)__builtin_amdgcn_readlane(hi, 16 * i) << 32) | (u32)__builtin_amdgcn_readlane(lo, 16 * i);   // the builtin returns int: no sign extension

    const u64 a = r[0] > r[1] ? r[0] : r[1], b = r[2] > r[3] ? r[2] : r[3];
    return a > b ? a : b;
}
// order-preserving u32 image of a float that is not NaN (and with -0 already folded into +0): a < b <=> image(a) < image(b)
__device__ __forceinline__ u32 float_order_bits(float f) {
    const u32 b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_from_order_bits(u32 u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ void store_pos(hm_pos* g, const P& p) {
    u64* w = reinterpret_cast<u64*>(g);
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = p.bt[i];
    w[6] = p.bc[0]; w[7] = p.bc[1]; w[8] = p.promoted; w[9] = p.key;
    u64 a = 0, b = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) a |= (u64)((p.hand[0] >> (6 * i)) & 63) << (8 * i);
#pragma unroll
    for (int i = 0; i < 3; ++i) a |= (u64)((p.hand[1] >> (6 * i)) & 63) << (8 * (5 + i));
    b |= (u64)((p.hand[1] >> 18) & 63);
    b |= (u64)((p.hand[1] >> 24) & 63) << 8;
    b |= (u64)(p.castling & 0xff) << 16;
    b |= (u64)(p.ep & 0xff) << 24;
    b |= (u64)(p.stm & 0xff) << 32;
    b |= (u64)(p.rule50 > 255 ? 255 : p.rule50) << 40;
    b |= (u64)(p.ply & 0xffff) << 48;
    w[10] = a; w[11] = b;
}

__device__ __forceinline__ u64 occ_of(const P& p) { return p.bc[0] | p.bc[1]; }
__device__ __forceinline__ int piece_type_on(const P& p, int s) {   // 0 none, 1..6
    u64 b = bit(s);
    int pt = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) pt = (p.bt[i] & b) ? i + 1 : pt;
    return pt;
}

// attackers_to (position.cpp:845-855, fastAttacks)
__device__ __forceinline__ u64 attackers_to(const AttackTab& t, const P& p, int s, u64 occ, int c) {
    u64 by = bc_of(p, c);
    return ((t.pawn[c ^ 1][s] & p.bt[0]) | (t.knight[s] & p.bt[1]) | (t.king[s] & p.bt[5])
            | (rook_att(s, occ) & (p.bt[3] | p.bt[4])) | (bishop_att(t, s, occ) & (p.bt[2] | p.bt[4]))) & by;
}

// One analysis per position: checkers, pinned pieces, king-danger squares.
struct Analysis {
    u64 checkers, pinned, danger;
    int ksq;
};
__device__ __forceinline__ void analyse(const AttackTab& t, const P& p, Analysis& a) {
    const int us = p.stm, them = us ^ 1;
    const u64 occ = occ_of(p);
    const u64 kbb = p.bt[5] & bc_of(p, us);
    const int k = lsb(kbb);
    a.ksq = k;
    a.checkers = attackers_to(t, p, k, occ, them);
    // pinned: snipers on an otherwise empty board, exactly one piece (ours) between
    u64 snipers = ((rook_att(k, 0) & (p.bt[3] | p.bt[4])) | (bishop_att(t, k, 0) & (p.bt[2] | p.bt[4]))) & bc_of(p, them);
    u64 pinned = 0;
    while (snipers) {
        int s = pop_lsb(snipers);
        u64 b = between_incl(t, k, s) & ~bit(s) & occ;
        if (b && !(b & (b - 1)) && (b & bc_of(p, us))) pinned |= b;
    }
    a.pinned = pinned;
    // danger: squares attacked by `them` with our king lifted off the board
    const u64 o2 = occ ^ kbb;
    const u64 tp = p.bt[0] & bc_of(p, them);
    u64 d = them == 0 ? (((tp & ~FILE_A) << 7) | ((tp & ~FILE_H) << 9))
                      : (((tp & ~FILE_H) >> 7) | ((tp & ~FILE_A) >> 9));
    u64 b = p.bt[1] & bc_of(p, them);
    while (b) d |= t.knight[pop_lsb(b)];
    b = (p.bt[2] | p.bt[4]) & bc_of(p, them);
    while (b) d |= bishop_att(t, pop_lsb(b), o2);
    b = (p.bt[3] | p.bt[4]) & bc_of(p, them);
    while (b) d |= rook_att(pop_lsb(b), o2);
    d |= t.king[lsb(p.bt[5] & bc_of(p, them))];
    a.danger = d;
}

__device__ __forceinline__ bool ep_legal(const AttackTab& t, const P& p, int k, int from, int to) {
    const int us = p.stm;
    int capsq = to - (us == 0 ? 8 : -8);
    u64 o = (occ_of(p) ^ bit(from) ^ bit(capsq)) | bit(to);
    return !(attackers_to(t, p, k, o, us ^ 1) & o);
}

__device__ __forceinline__ u32 mk(int from, int to) { return (u32)((from << 6) | to); }
__device__ __forceinline__ u32 mk_t(u32 type, int from, int to, int pt) { return (u32)((pt << 16) | type | (from << 6) | to); }
__device__ __forceinline__ u32 mk_drop(int to, int pt) { return (u32)((pt << 22) | (pt << 16) | HM_MT_DROP | to); }

constexpr u32 ILLEGAL_FLAG = 0x80000000u;

// Ordered legal move list (generate<LEGAL>), written to out[0..n).  `out` may point to LDS
// or global memory.  Returns n.  Pseudo-legal moves are emitted in the reference's order with
// an illegal flag, then compacted with the reference's swap-with-last rule (movegen.cpp:449-453).
__device__ __forceinline__ int gen_legal_body(const AttackTab& t, const P& p, u32* out) {
    Analysis an;
    analyse(t, p, an);
    const int us = p.stm, them = us ^ 1, k = an.ksq;
    const u64 occ = occ_of(p), ours = bc_of(p, us), theirs = bc_of(p, them);
    const bool evasion = an.checkers != 0;
    int n = 0;
    auto pin_ok = [&](int from, int to) -> u32 {
        return (!(an.pinned & bit(from)) || (line_through(t, k, from) & bit(to))) ? 0u : ILLEGAL_FLAG;
    };
    if (!evasion || !(an.checkers & (an.checkers - 1))) {
        const u64 target = evasion ? between_incl(t, k, lsb(an.checkers)) : ~ours;
        // ---- pawns
        {
            const int up = us == 0 ? 8 : -8;
            const u64 r7 = us == 0 ? (RANK_1 << 48) : (RANK_1 << 8);
            const u64 r3 = us == 0 ? (RANK_1 << 16) : (RANK_1 << 40);
            const u64 empty = ~occ;
            const u64 enemies = evasion ? an.checkers : theirs;
            const u64 pawns = p.bt[0] & ours;
            const u64 on7 = pawns & r7, not7 = pawns & ~r7;
            const int dUR = us == 0 ? 9 : -9, dUL = us == 0 ? 7 : -7;
            auto shUp = [&](u64 b) { return us == 0 ? b << 8 : b >> 8; };
            auto shUR = [&](u64 b) { return us == 0 ? (b & ~FILE_H) << 9 : (b & ~FILE_A) >> 9; };
            auto shUL = [&](u64 b) { return us == 0 ? (b & ~FILE_A) << 7 : (b & ~FILE_H) >> 7; };
            u64 b1 = shUp(not7) & empty;
            u64 b2 = shUp(b1 & r3) & empty;
            if (evasion) { b1 &= target; b2 &= target; }
            while (b1) { int to = pop_lsb(b1); out[n++] = mk(to - up, to) | pin_ok(to - up, to); }
            while (b2) { int to = pop_lsb(b2); out[n++] = mk(to - 2 * up, to) | pin_ok(to - 2 * up, to); }
            if (on7) {
                u64 p1 = shUR(on7) & enemies, p2 = shUL(on7) & enemies, p3 = shUp(on7) & empty;
                if (evasion) p3 &= target;
                while (p1) {
                    int to = pop_lsb(p1); u32 f = pin_ok(to - dUR, to);
                    for (int pt = 5; pt >= 2; --pt) out[n++] = mk_t(HM_MT_PROMOTION, to - dUR, to, pt) | f;
                }
                while (p2) {
                    int to = pop_lsb(p2); u32 f = pin_ok(to - dUL, to);
                    for (int pt = 5; pt >= 2; --pt) out[n++] = mk_t(HM_MT_PROMOTION, to - dUL, to, pt) | f;
                }
                while (p3) {
                    int to = pop_lsb(p3); u32 f = pin_ok(to - up, to);
                    for (int pt = 5; pt >= 2; --pt) out[n++] = mk_t(HM_MT_PROMOTION, to - up, to, pt) | f;
                }
            }
            u64 c1 = shUR(not7) & enemies, c2 = shUL(not7) & enemies;
            while (c1) { int to = pop_lsb(c1); out[n++] = mk(to - dUR, to) | pin_ok(to - dUR, to); }
            while (c2) { int to = pop_lsb(c2); out[n++] = mk(to - dUL, to) | pin_ok(to - dUL, to); }
            if (p.ep < 64 && !(evasion && (target & bit((int)p.ep + up)))) {
                u64 e = not7 & t.pawn[them][p.ep];
                while (e) {
                    int from = pop_lsb(e);
                    out[n++] = mk_t(HM_MT_EN_PASSANT, from, p.ep, 0) | (ep_legal(t, p, k, from, p.ep) ? 0u : ILLEGAL_FLAG);
                }
            }
        }
        // ---- knights, bishops, rooks, queens
        {
            u64 bb = p.bt[1] & ours;
            while (bb) {
                int from = pop_lsb(bb);
                u64 b = t.knight[from] & target;
                u32 f = (an.pinned & bit(from)) ? ILLEGAL_FLAG : 0u;     // a pinned knight never moves on its line
                while (b) out[n++] = mk(from, pop_lsb(b)) | f;
            }
            bb = p.bt[2] & ours;
            while (bb) {
                int from = pop_lsb(bb);
                u64 b = bishop_att(t, from, occ) & target;
                while (b) { int to = pop_lsb(b); out[n++] = mk(from, to) | pin_ok(from, to); }
            }
            bb = p.bt[3] & ours;
            while (bb) {
                int from = pop_lsb(bb);
                u64 b = rook_att(from, occ) & target;
                while (b) { int to = pop_lsb(b); out[n++] = mk(from, to) | pin_ok(from, to); }
            }
            bb = p.bt[4] & ours;
            while (bb) {
                int from = pop_lsb(bb);
                u64 b = (rook_att(from, occ) | bishop_att(t, from, occ)) & target;
                while (b) { int to = pop_lsb(b); out[n++] = mk(from, to) | pin_ok(from, to); }
            }
        }
        // ---- drops (real ones are always legal; in EVASIONS the reference also emits virtual
        // drops for empty pockets onto check squares and strips them later: emitted flagged)
        {
            const u64 b0 = target & ~occ;
            const int ek = lsb(p.bt[5] & theirs);
            for (int pt = 1; pt <= 5; ++pt) {
                u64 b = pt == 1 ? (b0 & ~(RANK_1 | RANK_8)) : b0;
                if (hand_get(p, us, pt) > 0) {
                    while (b) out[n++] = mk_drop(pop_lsb(b), pt);
                } else if (evasion) {
                    u64 cs = pt == 1 ? t.pawn[them][ek]
                           : pt == 2 ? t.knight[ek]
                           : pt == 3 ? bishop_att(t, ek, occ)
                           : pt == 4 ? rook_att(ek, occ)
                                     : (rook_att(ek, occ) | bishop_att(t, ek, occ));
                    b &= cs;
                    while (b) out[n++] = mk_drop(pop_lsb(b), pt) | ILLEGAL_FLAG;
                }
            }
        }
    }
    // ---- king
    {
        u64 b = t.king[k] & ~ours;
        while (b) { int to = pop_lsb(b); out[n++] = mk(k, to) | ((an.danger & bit(to)) ? ILLEGAL_FLAG : 0u); }
        if (!evasion) {
            const u32 rights = us == 0 ? (p.castling & 3) : ((p.castling >> 2) & 3);
            const int base = us == 0 ? 0 : 56;
            if ((rights & 1) && !(occ & (bit(base + 5) | bit(base + 6))))
                out[n++] = mk_t(HM_MT_CASTLING, k, base + 7, 0) | ((an.danger & (bit(base + 5) | bit(base + 6))) ? ILLEGAL_FLAG : 0u);
            if ((rights & 2) && !(occ & (bit(base + 1) | bit(base + 2) | bit(base + 3))))
                out[n++] = mk_t(HM_MT_CASTLING, k, base, 0) | ((an.danger & (bit(base + 2) | bit(base + 3))) ? ILLEGAL_FLAG : 0u);
        }
    }
    // ---- generate<LEGAL> compaction: illegal entries are overwritten by the last element
    int cur = 0;
    while (cur != n) {
        u32 m = out[cur];
        if (m & ILLEGAL_FLAG) out[cur] = out[--n];
        else ++cur;
    }
    return n;
}
// The generator is a real call (not inlined into its many call sites).  `p` is taken by value — a by-reference
// argument would pin the caller's position in scratch memory for its whole lifetime — and the attack tables
// (always staged in LDS) and, for gen_legal, the output list travel as LDS pointers so that the body compiles
// to ds_read/ds_write instead of flat accesses.  gen_legal_to is the form for lists in global memory.
typedef const __attribute__((address_space(3))) AttackTab* LdsAttackTab;
typedef __attribute__((address_space(3))) u32* LdsList;
__device__ __attribute__((noinline)) int gen_legal_lds(LdsAttackTab t, const P p, LdsList out) {
    return gen_legal_body(*(const AttackTab*)t, p, (u32*)out);
}
__device__ __attribute__((noinline)) int gen_legal_glb(LdsAttackTab t, const P p, u32* out) {
    return gen_legal_body(*(const AttackTab*)t, p, out);
}
__device__ __forceinline__ int gen_legal(const AttackTab& t, const P& p, u32* out) {          // out in LDS
    return gen_legal_lds((LdsAttackTab)&t, p, (LdsList)out);
}
__device__ __forceinline__ int gen_legal_to(const AttackTab& t, const P& p, u32* out) {       // out anywhere
    return gen_legal_glb((LdsAttackTab)&t, p, out);
}

// Wave-cooperative generate<LEGAL>: all 64 lanes of a wave call this with the SAME position; the list (identical
// to gen_legal's, order included) lands in `out` (LDS).  Each emission group of the reference order is a
// bitboard whose set bits map to list slots by rank: pawn / drop / king groups take one lane per target square,
// the piece groups one lane per from-square with a wave prefix sum over the per-piece target counts; the
// swap-with-last removal of the illegal entries is replayed hole by hole with ballots.
__device__ __forceinline__ int gen_legal_wave_body(const AttackTab& t, const P& p, u32* out) {
    const int lane = threadIdx.x & 63;
    Analysis an;
    analyse(t, p, an);
    const int us = p.stm, them = us ^ 1, k = an.ksq;
    const u64 occ = occ_of(p), ours = bc_of(p, us), theirs = bc_of(p, them);
    const bool evasion = an.checkers != 0;
    const u64 lbit = bit(lane), below = lbit - 1;
    int n = 0;
    auto pin_ok = [&](int from, int to) -> u32 {
        return (!(an.pinned & bit(from)) || (line_through(t, k, from) & bit(to))) ? 0u : ILLEGAL_FLAG;
    };
    // one list slot (or `mult` consecutive slots) per set bit of B, in LSB order; this lane serves bit `lane`
    auto slot_of = [&](u64 B, int mult) -> int { return (B & lbit) ? n + mult * popc(B & below) : -1; };
    if (!evasion || !(an.checkers & (an.checkers - 1))) {
        const u64 target = evasion ? between_incl(t, k, lsb(an.checkers)) : ~ours;
        {   // ---- pawns
            const int up = us == 0 ? 8 : -8;
            const u64 r7 = us == 0 ? (RANK_1 << 48) : (RANK_1 << 8);
            const u64 r3 = us == 0 ? (RANK_1 << 16) : (RANK_1 << 40);
            const u64 empty = ~occ;
            const u64 enemies = evasion ? an.checkers : theirs;
            const u64 pawns = p.bt[0] & ours;
            const u64 on7 = pawns & r7, not7 = pawns & ~r7;
            const int dUR = us == 0 ? 9 : -9, dUL = us == 0 ? 7 : -7;
            auto shUp = [&](u64 b) { return us == 0 ? b << 8 : b >> 8; };
            auto shUR = [&](u64 b) { return us == 0 ? (b & ~FILE_H) << 9 : (b & ~FILE_A) >> 9; };
            auto shUL = [&](u64 b) { return us == 0 ? (b & ~FILE_A) << 7 : (b & ~FILE_H) >> 7; };
            u64 b1 = shUp(not7) & empty;
            u64 b2 = shUp(b1 & r3) & empty;
            if (evasion) { b1 &= target; b2 &= target; }
            int sl = slot_of(b1, 1);
            if (sl >= 0) out[sl] = mk(lane - up, lane) | pin_ok(lane - up, lane);
            n += popc(b1);
            sl = slot_of(b2, 1);
            if (sl >= 0) out[sl] = mk(lane - 2 * up, lane) | pin_ok(lane - 2 * up, lane);
            n += popc(b2);
            if (on7) {
                u64 p1 = shUR(on7) & enemies, p2 = shUL(on7) & enemies, p3 = shUp(on7) & empty;
                if (evasion) p3 &= target;
                const u64 pb[3] = {p1, p2, p3};
                const int pd[3] = {dUR, dUL, up};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    sl = slot_of(pb[q], 4);
                    if (sl >= 0) {
                        const int from = lane - pd[q];
                        const u32 f = pin_ok(from, lane);
#pragma unroll
                        for (int pt = 5; pt >= 2; --pt) out[sl + (5 - pt)] = mk_t(HM_MT_PROMOTION, from, lane, pt) | f;
                    }
                    n += 4 * popc(pb[q]);
                }
            }
            const u64 c1 = shUR(not7) & enemies, c2 = shUL(not7) & enemies;
            sl = slot_of(c1, 1);
            if (sl >= 0) out[sl] = mk(lane - dUR, lane) | pin_ok(lane - dUR, lane);
            n += popc(c1);
            sl = slot_of(c2, 1);
            if (sl >= 0) out[sl] = mk(lane - dUL, lane) | pin_ok(lane - dUL, lane);
            n += popc(c2);
            if (p.ep < 64 && !(evasion && (target & bit((int)p.ep + up)))) {
                const u64 e = not7 & t.pawn[them][p.ep];         // from-squares, LSB order
                sl = slot_of(e, 1);
                if (sl >= 0) out[sl] = mk_t(HM_MT_EN_PASSANT, lane, p.ep, 0) | (ep_legal(t, p, k, lane, p.ep) ? 0u : ILLEGAL_FLAG);
                n += popc(e);
            }
        }
        // ---- knights, bishops, rooks, queens: lane = from-square, slots by wave prefix sum of the target counts
#pragma unroll
        for (int ty = 1; ty <= 4; ++ty) {
            const u64 bb = p.bt[ty] & ours;
            if (!bb) continue;                                      // uniform
            u64 tg = 0;
            if (bb & lbit) {
                tg = ty == 1 ? t.knight[lane]
                   : ty == 2 ? bishop_att(t, lane, occ)
                   : ty == 3 ? rook_att(lane, occ)
                             : (rook_att(lane, occ) | bishop_att(t, lane, occ));
                tg &= target;
            }
            const int cnt = popc(tg);
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
            int sl2 = n + incl - cnt;
            const u32 kf = (ty == 1 && (an.pinned & lbit)) ? ILLEGAL_FLAG : 0u;   // a pinned knight never moves on its line
            while (tg) {
                const int to = pop_lsb(tg);
                out[sl2++] = mk(lane, to) | (ty == 1 ? kf : pin_ok(lane, to));
            }
            n += ulane(incl, 63);
        }
        {   // ---- drops (virtual drops for empty pockets in evasions are emitted flagged, as the reference does)
            const u64 b0 = target & ~occ;
            const int ek = lsb(p.bt[5] & theirs);
#pragma unroll
            for (int pt = 1; pt <= 5; ++pt) {
                u64 b = pt == 1 ? (b0 & ~(RANK_1 | RANK_8)) : b0;
                u32 fl = 0;
                if (hand_get(p, us, pt) == 0) {
                    if (!evasion) continue;                         // uniform
                    const u64 cs = pt == 1 ? t.pawn[them][ek]
                                 : pt == 2 ? t.knight[ek]
                                 : pt == 3 ? bishop_att(t, ek, occ)
                                 : pt == 4 ? rook_att(ek, occ)
                                           : (rook_att(ek, occ) | bishop_att(t, ek, occ));
                    b &= cs;
                    fl = ILLEGAL_FLAG;
                }
                const int sl = slot_of(b, 1);
                if (sl >= 0) out[sl] = mk_drop(lane, pt) | fl;
                n += popc(b);
            }
        }
    }
    {   // ---- king
        const u64 b = t.king[k] & ~ours;
        const int sl = slot_of(b, 1);
        if (sl >= 0) out[sl] = mk(k, lane) | ((an.danger & lbit) ? ILLEGAL_FLAG : 0u);
        n += popc(b);
        if (!evasion) {
            const u32 rights = us == 0 ? (p.castling & 3) : ((p.castling >> 2) & 3);
            const int base = us == 0 ? 0 : 56;
            if ((rights & 1) && !(occ & (bit(base + 5) | bit(base + 6))))
                out[n++] = mk_t(HM_MT_CASTLING, k, base + 7, 0) | ((an.danger & (bit(base + 5) | bit(base + 6))) ? ILLEGAL_FLAG : 0u);
            if ((rights & 2) && !(occ & (bit(base + 1) | bit(base + 2) | bit(base + 3))))
                out[n++] = mk_t(HM_MT_CASTLING, k, base, 0) | ((an.danger & (bit(base + 2) | bit(base + 3))) ? ILLEGAL_FLAG : 0u);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- generate<LEGAL> compaction (movegen.cpp:449-453): every illegal entry is overwritten by the last element;
    // replayed as "first illegal at or after cur" (ballot) <- "last legal before n" (illegal tail entries drop out)
    int cur = 0;
    while (cur < n) {
        int hole = -1;
        for (int base = cur & ~63; base < n && hole < 0; base += 64) {
            const int i = base + lane;
            const bool bad = i >= cur && i < n && (out[i] & ILLEGAL_FLAG);
            const u64 m = __ballot(bad);
            if (m) hole = base + __builtin_ctzll(m);
        }
        if (hole < 0) break;
        --n;
        while (n > hole && (out[n] & ILLEGAL_FLAG)) --n;            // uniform LDS reads
        if (n > hole) { const u32 mv = out[n]; __builtin_amdgcn_wave_barrier(); out[hole] = mv; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        cur = hole + 1;
    }
    return n;
}

// (a real call, like gen_legal: position by value, tables and list as LDS pointers)
__device__ __attribute__((noinline)) int gen_legal_wave_lds(LdsAttackTab t, const P p, LdsList out) {
    return gen_legal_wave_body(*(const AttackTab*)t, p, (u32*)out);
}
__device__ __forceinline__ int gen_legal_wave(const AttackTab& t, const P& p, u32* out) {
    return gen_legal_wave_lds((LdsAttackTab)&t, p, (LdsList)out);
}

// Legal move COUNT only (bulk counting at perft leaves): same set as gen_legal, no list.
__device__ inline int count_legal(const AttackTab& t, const P& p) {
    Analysis an;
    analyse(t, p, an);
    const int us = p.stm, them = us ^ 1, k = an.ksq;
    const u64 occ = occ_of(p), ours = bc_of(p, us), theirs = bc_of(p, them);
    const bool evasion = an.checkers != 0;
    int n = popc(t.king[k] & ~ours & ~an.danger);
    if (evasion && (an.checkers & (an.checkers - 1))) return n;
    const u64 target = evasion ? between_incl(t, k, lsb(an.checkers)) : ~ours;
    const u64 free_ = ~an.pinned;
    // pawns: unpinned set-wise, pinned one by one
    {
        const u64 r7 = us == 0 ? (RANK_1 << 48) : (RANK_1 << 8);
        const u64 r3 = us == 0 ? (RANK_1 << 16) : (RANK_1 << 40);
        const u64 r8 = us == 0 ? RANK_8 : RANK_1;
        const u64 empty = ~occ;
        const u64 enemies = evasion ? an.checkers : theirs;
        const u64 pawns = p.bt[0] & ours;
        auto shUp = [&](u64 b) { return us == 0 ? b << 8 : b >> 8; };
        auto shUR = [&](u64 b) { return us == 0 ? (b & ~FILE_H) << 9 : (b & ~FILE_A) >> 9; };
        auto shUL = [&](u64 b) { return us == 0 ? (b & ~FILE_A) << 7 : (b & ~FILE_H) >> 7; };
        const u64 fp = pawns & free_;
        u64 s1 = shUp(fp) & empty;
        u64 s2 = shUp(s1 & r3) & empty;
        u64 cr = shUR(fp) & enemies, cl = shUL(fp) & enemies;
        if (evasion) { s1 &= target; s2 &= target; }
        // promotions count 4x: targets on the last rank
        n += popc(s1 & ~r8) + 4 * popc(s1 & r8) + popc(s2);
        n += popc(cr & ~r8) + 4 * popc(cr & r8) + popc(cl & ~r8) + 4 * popc(cl & r8);
        (void)r7;
        u64 pp = pawns & an.pinned;
        while (pp) {                       // pinned pawns: moves must stay on the pin line
            int from = pop_lsb(pp);
            u64 line = line_through(t, k, from);
            u64 f = bit(from);
            u64 a1 = shUp(f) & empty;
            u64 a2 = shUp(a1 & r3) & empty;
            u64 c = (shUR(f) | shUL(f)) & enemies;
            if (evasion) { a1 &= target; a2 &= target; }
            u64 all = (a1 | a2 | c) & line;
            n += popc(all & ~r8) + 4 * popc(all & r8);
        }
        if (p.ep < 64) {
            const int up = us == 0 ? 8 : -8;
            if (!(evasion && (target & bit((int)p.ep + up)))) {
                u64 e = pawns & ~r7 & t.pawn[them][p.ep];
                while (e) n += ep_legal(t, p, k, pop_lsb(e), p.ep) ? 1 : 0;
            }
        }
    }
    u64 bb = p.bt[1] & ours & free_;
    while (bb) n += popc(t.knight[pop_lsb(bb)] & target);
    bb = (p.bt[2] | p.bt[4]) & ours;
    while (bb) {
        int from = pop_lsb(bb);
        u64 a = bishop_att(t, from, occ) & target;
        if (an.pinned & bit(from)) a &= line_through(t, k, from);
        n += popc(a);
    }
    bb = (p.bt[3] | p.bt[4]) & ours;
    while (bb) {
        int from = pop_lsb(bb);
        u64 a = rook_att(from, occ) & target;
        if (an.pinned & bit(from)) a &= line_through(t, k, from);
        n += popc(a);
    }
    {
        const u64 b0 = target & ~occ;
        int kinds = 0;
        for (int pt = 2; pt <= 5; ++pt) kinds += hand_get(p, us, pt) > 0;
        n += kinds * popc(b0);
        if (hand_get(p, us, 1) > 0) n += popc(b0 & ~(RANK_1 | RANK_8));
    }
    if (!evasion) {
        const u32 rights = us == 0 ? (p.castling & 3) : ((p.castling >> 2) & 3);
        const int base = us == 0 ? 0 : 56;
        if ((rights & 1) && !(occ & (bit(base + 5) | bit(base + 6))) && !(an.danger & (bit(base + 5) | bit(base + 6)))) ++n;
        if ((rights & 2) && !(occ & (bit(base + 1) | bit(base + 2) | bit(base + 3))) && !(an.danger & (bit(base + 2) | bit(base + 3)))) ++n;
    }
    return n;
}

// MoveList<LEGAL>::size() > 0, which is all Board::is_checkmate asks of the lists (board.cc:169-208).  Most positions answer
// from a few bitboard tests: out of check, a drop (never exposes the own king) or any move of an unpinned non-king piece
// other than en passant is legal, so the first one found settles it; only positions in check, or with nothing but king /
// pinned-piece / en-passant moves, take the full count.
__device__ inline bool has_legal_move(const AttackTab& t, const P& p) {
    const int us = p.stm, them = us ^ 1;
    const u64 occ = occ_of(p), ours = bc_of(p, us), theirs = bc_of(p, them);
    const u64 kbb = p.bt[5] & ours;
    const int k = lsb(kbb);
    if (attackers_to(t, p, k, occ, them) == 0) {
        const u64 empty = ~occ;
        const u32 hand = us ? p.hand[1] : p.hand[0];
        if ((hand >> 6) != 0 && empty) return true;                                   // N, B, R or Q in hand
        if ((hand & 63u) != 0 && (empty & ~(RANK_1 | RANK_8))) return true;           // pawn in hand
        // pinned pieces (as analyse())
        u64 snipers = ((rook_att(k, 0) & (p.bt[3] | p.bt[4])) | (bishop_att(t, k, 0) & (p.bt[2] | p.bt[4]))) & theirs;
        u64 pinned = 0;
        while (snipers) {
            const int s = pop_lsb(snipers);
            const u64 b = between_incl(t, k, s) & ~bit(s) & occ;
            if (b && !(b & (b - 1)) && (b & ours)) pinned |= b;
        }
        const u64 free_ = ours & ~pinned & ~kbb;
        const u64 fp = p.bt[0] & free_;
        const u64 up = us == 0 ? fp << 8 : fp >> 8;
        if (up & empty) return true;                                                  // pawn push (promotions included)
        const u64 cr = us == 0 ? (fp & ~FILE_H) << 9 : (fp & ~FILE_A) >> 9;
        const u64 cl = us == 0 ? (fp & ~FILE_A) << 7 : (fp & ~FILE_H) >> 7;
        if ((cr | cl) & theirs) return true;                                          // pawn capture
        u64 bb = p.bt[1] & free_;
        while (bb) if (t.knight[pop_lsb(bb)] & ~ours) return true;
        bb = (p.bt[2] | p.bt[4]) & free_;
        while (bb) if (bishop_att(t, pop_lsb(bb), occ) & ~ours) return true;
        bb = (p.bt[3] | p.bt[4]) & free_;
        while (bb) if (rook_att(pop_lsb(bb), occ) & ~ours) return true;
    }
    return count_legal(t, p) > 0;
}

// do_move.  Returns pieceToHand as (colour<<3 | pt) or 0 (position.cpp:1459).
__device__ inline int do_move(const AttackTab& t, const ZobristTab& z, P& p, u32 m) {
    const int us = p.stm, them = us ^ 1;
    const int from = (m >> 6) & 63, to = m & 63;
    const u32 mt = m & (15u << 12);
    u64 k = p.key ^ z.side;
    ++p.ply; ++p.rule50;
    int toHand = 0;
    const u64 fb = bit(from), tb = bit(to);
    int pt;                                   // moving piece type 1..6
    if (mt == HM_MT_DROP) pt = (m >> 16) & 63;
    else pt = piece_type_on(p, from);
    if (mt == HM_MT_CASTLING) {
        const bool ks = to > from;
        const int base = us == 0 ? 0 : 56;
        const int rto = base + (ks ? 5 : 3), kto = base + (ks ? 6 : 2);
        p.bt[5] ^= fb | bit(kto);
        p.bt[3] ^= tb | bit(rto);
        bc_xor(p, us, fb | bit(kto) | tb | bit(rto));
        k ^= z.psq[us][3][to] ^ z.psq[us][3][rto] ^ z.psq[us][5][from] ^ z.psq[us][5][kto];
    } else {
        int cap = 0, capsq = to;
        if (mt == HM_MT_EN_PASSANT) { cap = 1; capsq = to - (us == 0 ? 8 : -8); }
        else if (mt != HM_MT_DROP) cap = piece_type_on(p, to);
        if (cap) {
            const u64 cb = bit(capsq);
            const bool capProm = (p.promoted & cb) != 0;
            bt_xor(p, cap - 1, cb); bc_xor(p, them, cb); p.promoted &= ~cb;
            toHand = (them << 3) | (capProm ? 1 : cap);
            k ^= z.psq[them][cap - 1][capsq];
            p.rule50 = 0;
        }
        if (mt == HM_MT_DROP) {
            int h = hand_get(p, us, pt);
            k ^= z.psq[us][pt - 1][to] ^ z.in_hand[us][pt - 1][h - 1] ^ z.in_hand[us][pt - 1][h];
            bt_xor(p, pt - 1, tb); bc_xor(p, us, tb);
            hand_add(p, us, pt, -1);
        } else {
            k ^= z.psq[us][pt - 1][from] ^ z.psq[us][pt - 1][to];
            bt_xor(p, pt - 1, fb | tb); bc_xor(p, us, fb | tb);
            if (p.promoted & fb) p.promoted ^= fb | tb;
        }
    }
    if (p.ep < 64) { k ^= z.ep[p.ep & 7]; p.ep = 64; }
    if (mt != HM_MT_DROP && p.castling) {
        // castlingRightsMask[from] | castlingRightsMask[to]
        auto cm = [](int s) -> u32 {
            return s == 0 ? 2u : s == 7 ? 1u : s == 4 ? 3u : s == 56 ? 8u : s == 63 ? 4u : s == 60 ? 12u : 0u;
        };
        u32 mask = cm(from) | cm(to);
        if (mask) {
            k ^= z.castle[p.castling];
            p.castling &= ~mask;
            k ^= z.castle[p.castling];
        }
    }
    if (pt == 1) {
        const int push = us == 0 ? 8 : -8;
        int d = to - from; d = d < 0 ? -d : d;
        if (mt != HM_MT_DROP && d == 16 && (t.pawn[us][to - push] & p.bt[0] & bc_of(p, them))) {
            p.ep = (u32)(to - push);
            k ^= z.ep[p.ep & 7];
        } else if (mt == HM_MT_PROMOTION) {
            const int pr = (m >> 16) & 63;
            p.bt[0] ^= tb; bt_xor(p, pr - 1, tb); p.promoted |= tb;
            k ^= z.psq[us][0][to] ^ z.psq[us][pr - 1][to];
        }
        p.rule50 = 0;
    }
    p.key = k;
    p.stm = (u32)them;
    return toHand;
}
__device__ __forceinline__ void add_to_hand(const ZobristTab& z, P& p, int pc) {   // position.cpp:47-51
    const int c = pc >> 3, pt = pc & 7;
    const int h = hand_get(p, c, pt);
    p.key ^= z.in_hand[c][pt - 1][h] ^ z.in_hand[c][pt - 1][h + 1];
    hand_add(p, c, pt, 1);
}
// Board::make_moves (board.cc:316-341): A then B, captured piece to the partner board's hand.
__device__ __forceinline__ void make_joint(const AttackTab& t, const ZobristTab& z, P& A, P& B, u32 ma, u32 mb) {
    if (ma) { int h = do_move(t, z, A, ma); if (h) add_to_hand(z, B, h); }
    if (mb) { int h = do_move(t, z, B, mb); if (h) add_to_hand(z, A, h); }
}

// Position::key() (position.h:1156-1159)
__device__ __forceinline__ u64 pos_key(const P& p) {
    return p.rule50 < 14 ? p.key : p.key ^ ((u64)((p.rule50 - 14) / 8) * 6364136223846793005ULL + 1442695040888963407ULL);
}

}  // namespace hmd
