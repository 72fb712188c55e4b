// hm_rules_device.hpp — device-side Bughouse team rules on top of hm_device.hpp.
//
// Reference behaviour (engine/src):
//   Board::is_checkmate                       environment/board.cc:169-208
//   Board::can_partner_provide_blocking_piece environment/board.cc:214-314
//   Board::is_draw / repetition_count         environment/board.h:326-332,411-453
//   Board::hash_key / history_key / mix_hash  environment/board.h:44-60,126-138
//   classify_terminal_position, has_unavoidable_waiting_board_mate, immediate_mates_on_board
//                                             search/searchthread.cc:21-139
//   find_immediate_root_mate                  search/agent.cc:136-238
// Design: a joint position is two register-resident `P`s; "push" is copy-make, so the
// reference's push/pop pairs become value copies.  Repetition identity is an integer key with
// the equivalence classes of Board::board_only_key (placement incl. promoted markers, side,
// castling, ep) instead of std::hash of a FEN substring.  Functions here are executed by one
// lane (or redundantly by all lanes of a wave with identical inputs); move lists go to caller
// provided scratch (LDS).
#pragma once
#include "hm_device.hpp"
#include "hm_prof.hpp"

namespace hmd {

struct alignas(16) RulesTab {
    AttackTab att;
    ZobristTab zob;
    u64 z_promoted[64];     // repetition-key marks for promoted pieces ('~' in the reference's FEN key)
    u64 in_hand_const;
    uint16_t pocket_f16[64];   // count/16 as fp16 (planes.cc:118), host-rounded
    uint16_t r50_f16[64];      // min(r50,50)/50 as fp16 (planes.cc:202-205)
};

typedef const __attribute__((address_space(3))) RulesTab* LdsRulesTab;   // real calls take the LDS-staged tables as LDS pointers

constexpr u64 HISTORY_HASH_SEED = 0xcbf29ce484222325ULL;
__device__ __forceinline__ u64 mix_hash(u64 key, u64 value) {   // board.h:126-131
    value += 0x9e3779b97f4a7c15ULL;
    value = (value ^ (value >> 30)) * 0xbf58476d1ce4e5b9ULL;
    value = (value ^ (value >> 27)) * 0x94d049bb133111ebULL;
    return key ^ (value ^ (value >> 31));
}

// repetition key: st->key with the in-hand terms removed and promoted marks added
__device__ inline u64 rep_key(const RulesTab& t, const P& p) {
    u64 k = p.key ^ t.in_hand_const;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int pt = 1; pt <= 5; ++pt) k ^= t.zob.in_hand[c][pt - 1][hand_get(p, c, pt) & 31];
    u64 b = p.promoted;
    while (b) k ^= t.z_promoted[pop_lsb(b)];
    return k;
}

// Per-board history view: keys[0..len) are the repetition keys of every position reached on
// that board (game history followed by the current search path, current position last);
// prefix = positionHistoryPrefixes.back().
struct Hist {
    const u64* keys;
    int len;
    u64 prefix;
};
__device__ __forceinline__ u64 history_key(const Hist& h) { return mix_hash(h.prefix, (u64)h.len); }   // board.h:133-138
__device__ inline u64 board_hash_key(const P& A, const P& B, const Hist& hA, const Hist& hB, bool adv, u64 timeAdvKey) {   // board.h:44-60
    const u64 k0 = mix_hash(pos_key(A), (u64)A.rule50);
    const u64 k1 = mix_hash(pos_key(B), (u64)B.rule50);
    u64 combined = k0 ^ (k1 + 0x9e3779b97f4a7c15ULL + (k0 << 6) + (k0 >> 2));
    u64 rc = history_key(hA);
    rc ^= history_key(hB) + 0x9e3779b97f4a7c15ULL + (rc << 6) + (rc >> 2);
    combined ^= rc + 0x9e3779b97f4a7c15ULL + (combined << 6) + (combined >> 2);
    return adv ? (combined ^ timeAdvKey) : combined;
}
// occurrences of the current (last) key in the whole history, current included (board.h:326-332).
// Wave-cooperative: call with all 64 lanes active and identical arguments; lanes stride the keys.
__device__ inline int repetition_count(const Hist& h) {
    if (h.len <= 0) return 0;
    const u64 cur = h.keys[h.len - 1];
    int c = 0;
    for (int i = threadIdx.x & 63; i < h.len; i += 64) c += h.keys[i] == cur;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    return ufirst(c);
}
__device__ inline bool is_draw_on_board(const P& p, const Hist& h, int ply) {   // board.h:423-453
    if (p.rule50 >= 100) return true;
    const int threshold = ply > 0 ? 1 : 2;
    const u64 cur = h.keys[h.len - 1];
    int c = 0;
    for (int i = threadIdx.x & 63; i + 1 < h.len; i += 64) c += h.keys[i] == cur;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    return ufirst(c) >= threshold;
}

__device__ __forceinline__ u64 checkers_of(const AttackTab& t, const P& p) {
    return attackers_to(t, p, lsb(p.bt[5] & bc_of(p, p.stm)), occ_of(p), p.stm ^ 1);
}
__device__ __forceinline__ bool is_capture(const P& p, u32 m) {   // position.h:1212-1216
    const u32 mt = m & (15u << 12);
    return mt == HM_MT_EN_PASSANT || (mt != HM_MT_CASTLING && mt != HM_MT_DROP && (occ_of(p) & bit(m & 63)));
}
__device__ inline bool gives_check(const RulesTab& t, const P& p, u32 m) {
    P c = p;
    do_move(t.att, t.zob, c, m);
    return checkers_of(t.att, c) != 0;
}

// has_useful_capture lambda of board.cc:262-290 over one partner position
__device__ inline bool has_useful_capture(const RulesTab& t, const P& q, int partnerSide, bool pawnValid, u32* scratch) {
    const int n = gen_legal(t.att, q, scratch);
    for (int i = 0; i < n; ++i) {
        const u32 m = scratch[i];
        const u32 mt = m & (15u << 12);
        const int to = m & 63;
        int cap;
        if (mt == HM_MT_EN_PASSANT) cap = 1;
        else if (mt == HM_MT_DROP) cap = 0;
        else cap = piece_type_on(q, to);              // incl. castling's own rook on `to` (reference quirk)
        if (!cap) continue;
        const int ct = (q.promoted & bit(to)) ? 1 : cap;
        if (ct == 1) { if (pawnValid) return true; }
        else return true;
    }
    (void)partnerSide;
    return false;
}

// board.cc:214-314.  bd[0], bd[1] = BOARD_A, BOARD_B.  scratch: 2 lists of HM_MAX_MOVES.
__device__ __attribute__((noinline)) bool can_partner_blocking_scan(LdsRulesTab tl, const P bdA, const P bdB, int boardInCheck, int checkedSide, bool adv, LdsList scratchl) {
    const RulesTab& t = *(const RulesTab*)tl;
    u32* scratch = (u32*)scratchl;
    // by-value copies: callers keep their positions in registers
    const P p = boardInCheck ? bdB : bdA, partner = boardInCheck ? bdA : bdB;
    const int partnerSide = checkedSide ^ 1;
    const bool partnerTurn = (int)partner.stm == partnerSide;
    if (!partnerTurn && !adv) return false;
    const int k = lsb(p.bt[5] & bc_of(p, checkedSide));
    const u64 chk = checkers_of(t.att, p);
    if (chk & (chk - 1)) return false;
    const int csq = lsb(chk);
    const u64 blocking = between_incl(t.att, k, csq);
    const u64 avail = blocking & ~occ_of(p);
    if (!avail) return false;
    const bool pawnValid = (avail & ~(RANK_1 | RANK_8)) != 0;
    if (partnerTurn) return has_useful_capture(t, partner, partnerSide, pawnValid, scratch);
    // time advantage: every opponent reply must leave the partner an immediate useful capture
    u32* replies = scratch;
    u32* inner = scratch + HM_MAX_MOVES;
    const int n = gen_legal(t.att, partner, replies);
    if (!n) return false;
    for (int i = 0; i < n; ++i) {
        P f = partner;
        do_move(t.att, t.zob, f, replies[i]);          // hands of the other board are irrelevant here
        if (!has_useful_capture(t, f, partnerSide, pawnValid, inner)) return false;
    }
    return true;
}

// (a real call: reached only for a board that is in check without a legal move)
__device__ __forceinline__ bool can_partner_provide_blocking_piece(const RulesTab& t, const P& bdA, const P& bdB, int boardInCheck, int checkedSide, bool adv, u32* scratch) {
    return can_partner_blocking_scan((LdsRulesTab)&t, bdA, bdB, boardInCheck, checkedSide, adv, (LdsList)scratch);
}

// "Has a legal move" of both boards as 0 / 1 counts (is_checkmate_c only tests its counts against zero), computed by lanes
// 0 and 1 concurrently (wave-uniform call).
__device__ inline void legal_counts(const RulesTab& t, const P* bd, int& cntA, int& cntB) {
    const int lane = threadIdx.x & 63;
    int c = 0;
    if (lane < 2) c = has_legal_move(t.att, pick_pos(bd, lane)) ? 1 : 0;
    cntA = ulane(c, 0); cntB = ulane(c, 1);
}

// board.cc:169-208 with the per-board MoveList<LEGAL>::size() values supplied.  scratch: 2 lists.
__device__ inline bool is_checkmate_c(const RulesTab& t, const P* bd, int side, bool adv, int cntA, int cntB, u32* scratch) {
    const bool onA = (int)bd[0].stm == side, onB = (int)bd[1].stm == (side ^ 1);
    if (onA) {
        if (cntA == 0 && checkers_of(t.att, bd[0]) && !can_partner_provide_blocking_piece(t, bd[0], bd[1], 0, side, adv, scratch)) return true;
    }
    if (onB) {
        if (cntB == 0 && checkers_of(t.att, bd[1]) && !can_partner_provide_blocking_piece(t, bd[0], bd[1], 1, side ^ 1, adv, scratch)) return true;
    }
    if (onA || onB) {
        const bool movesA = onA && cntA > 0, movesB = onB && cntB > 0;
        if (!movesA && !movesB && (!adv || (onA && onB))) return true;
    }
    return false;
}

// board.cc:169-208.  scratch: 2 lists.  Safe in lane-divergent code (no cross-lane operations).  A real call.
__device__ __attribute__((noinline)) bool is_checkmate_scan(LdsRulesTab tl, const P bdA, const P bdB, int side, bool adv, LdsList scratchl) {
    const RulesTab& t = *(const RulesTab*)tl;
    u32* scratch = (u32*)scratchl;
    const P bd[2] = {bdA, bdB};
    const bool onA = (int)bd[0].stm == side, onB = (int)bd[1].stm == (side ^ 1);
    int cntA = -1, cntB = -1;
    if (onA) {
        cntA = has_legal_move(t.att, bd[0]) ? 1 : 0;
        if (cntA == 0 && checkers_of(t.att, bd[0]) && !can_partner_provide_blocking_piece(t, bd[0], bd[1], 0, side, adv, scratch)) return true;
    }
    if (onB) {
        cntB = has_legal_move(t.att, bd[1]) ? 1 : 0;
        if (cntB == 0 && checkers_of(t.att, bd[1]) && !can_partner_provide_blocking_piece(t, bd[0], bd[1], 1, side ^ 1, adv, scratch)) return true;
    }
    if (onA || onB) {
        const bool movesA = onA && cntA > 0, movesB = onB && cntB > 0;
        if (!movesA && !movesB && (!adv || (onA && onB))) return true;
    }
    return false;
}

__device__ __forceinline__ bool is_checkmate(const RulesTab& t, const P* bd, int side, bool adv, u32* scratch) {
    return is_checkmate_scan((LdsRulesTab)&t, bd[0], bd[1], side, adv, (LdsList)scratch);
}

__device__ __forceinline__ bool is_double_sit_legal(bool adv, bool aOn, bool bOn) { return adv && (aOn != bOn); }          // joint_action.h:14-18
__device__ __forceinline__ bool is_single_pass_legal(bool adv, bool aOn, bool bOn, bool partnerCapture) {                   // joint_action.h:26-33
    return adv || !(aOn && bOn) || partnerCapture;
}

// Joint position with both histories, as the search sees it.
struct JBoard {
    P bd[2];
    u64* hist[2];       // writable key arrays (game history + path)
    int hlen[2];
    u64 prefix[2];
};
__device__ __forceinline__ Hist hist_of(const JBoard& j, int b) {   // no runtime subscripts: JBoard stays in registers
    return b ? Hist{j.hist[1], j.hlen[1], j.prefix[1]} : Hist{j.hist[0], j.hlen[0], j.prefix[0]};
}
__device__ inline bool jb_is_draw(const JBoard& j, int ply) {
    return is_draw_on_board(j.bd[0], hist_of(j, 0), ply) || is_draw_on_board(j.bd[1], hist_of(j, 1), ply);
}
// Board::push_move incl. history (board.cc:98-108); only `lane0` writes the history array.
__device__ __forceinline__ void jb_push(const RulesTab& t, JBoard& j, int b, u32 m, bool writer) {
    const int h = do_move(t.att, t.zob, j.bd[b], m);
    if (h) add_to_hand(t.zob, j.bd[1 - b], h);
    const u64 k = rep_key(t, j.bd[b]);
    if (writer) j.hist[b][j.hlen[b]] = k;
    j.hlen[b]++;
    j.prefix[b] = mix_hash(j.prefix[b], k);
}
__device__ __forceinline__ void jb_make(const RulesTab& t, JBoard& j, u32 ma, u32 mb, bool writer) {   // board.cc:316-341
    if (ma) jb_push(t, j, 0, ma, writer);
    if (mb) jb_push(t, j, 1, mb, writer);
}

// searchthread.cc:21-39; returns number of mating moves written to `out` (wave-uniform call).
// scratch: 3 lists.  Lanes test the moves in parallel for "gives check and leaves the victim a board
// without legal moves" (necessary for Board::is_checkmate); survivors are verified in list order.
__device__ inline int immediate_mates_on_board(const RulesTab& t, const JBoard& j, int b, int victimTeam, bool victimAdv, u32* out, u32* scratch) {
    const int lane = threadIdx.x & 63;
    u32* list = scratch;
    int n = 0;
    const P src = pick_pos(j.bd, b), other = pick_pos(j.bd, 1 - b);
    n = gen_legal_wave(t.att, src, list);              // wave-uniform call
    __builtin_amdgcn_wave_barrier();
    TRACE_EV(13, n, b);
    int k = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        bool cand = false;
        if (i < n) {
            P moved = src, partner = other;
            const int h = do_move(t.att, t.zob, moved, list[i]);
            if (checkers_of(t.att, moved)) {
                if (h) add_to_hand(t.zob, partner, h);
                // victim's on-turn boards: board A plays victimTeam's colour, board B the opposite colour
                const bool vMoved = (int)moved.stm == (b ? victimTeam ^ 1 : victimTeam);
                const bool vPartner = (int)partner.stm == (b ? victimTeam : victimTeam ^ 1);
                cand = (vMoved && count_legal(t.att, moved) == 0) || (vPartner && count_legal(t.att, partner) == 0);
            }
        }
        u64 mask = __ballot(cand);
        TRACE_EV(14, __popcll(mask), base);
        while (mask) {
            const int q = __builtin_ctzll(mask);
            mask &= mask - 1;
            const u32 m = list[base + q];
            P moved = src, partner = other;
            const int h = do_move(t.att, t.zob, moved, m);
            if (h) add_to_hand(t.zob, partner, h);
            P nb[2];
            put_pos(nb, b, moved); put_pos(nb, 1 - b, partner);
            if (is_checkmate(t, nb, victimTeam, victimAdv, scratch + HM_MAX_MOVES)) out[k++] = m;
        }
    }
    return k;
}

// searchthread.cc:41-97.  scratch: 6 lists.  `j` is used read-only; the repetition check after a
// reply needs the history with that reply's key appended, handled on the fly.
// A real call with the joint position passed by value: the rule is heavy (lane-parallel mate scan, nested
// checkmate tests) and inlining it into the traversal kernel drives that kernel to the 512-VGPR limit.
// (tables and scratch lists travel as LDS pointers so the body keeps ds_read/ds_write accesses).
__device__ __attribute__((noinline)) bool waiting_board_mate_scan(LdsRulesTab tl, const JBoard j, int team, bool adv, int searchPly, LdsList scratchl) {
    const RulesTab& t = *(const RulesTab*)tl;
    u32* scratch = (u32*)scratchl;
    const bool aOn = (int)j.bd[0].stm == team;
    const int active = aOn ? 0 : 1, waiting = 1 - active;
    u32* mating = scratch;
    const int nm = immediate_mates_on_board(t, j, waiting, team, adv, mating, scratch + HM_MAX_MOVES);
    TRACE_EV(10, nm, waiting);
    if (!nm) return false;
    u32* replies = scratch + HM_MAX_MOVES;
    const P actP = pick_pos(j.bd, active), waitP = pick_pos(j.bd, waiting);
    int nr = gen_legal_wave(t.att, actP, replies);
    if (adv) replies[nr++] = 0;
    TRACE_EV(11, nr, adv);
    if (!nr) return false;
    u32* tmp = scratch + 2 * HM_MAX_MOVES;                    // 4 lists left
    for (int r = 0; r < nr; ++r) {
        const u32 reply = replies[r];
        P na = actP, nw = waitP;                       // boards after the reply: active / waiting
        bool drawAfter;
        if (reply) {
            const int h = do_move(t.att, t.zob, na, reply);
            if (h) add_to_hand(t.zob, nw, h);
            // is_draw(searchPly+1) on the pushed position: active board's history gains one key
            const u64 k = rep_key(t, na);
            bool d = na.rule50 >= 100;
            if (!d) {   // threshold 1 (ply > 0): any earlier occurrence
                const Hist h0 = hist_of(j, active);
                bool f = false;
                for (int i = threadIdx.x & 63; i < h0.len; i += 64) f |= h0.keys[i] == k;
                d = __ballot(f) != 0ULL;
            }
            drawAfter = d || is_draw_on_board(nw, hist_of(j, waiting), searchPly + 1);
        } else {
            drawAfter = jb_is_draw(j, searchPly + 1);
        }
        bool persists = false;
        P nb[2];
        put_pos(nb, active, na); put_pos(nb, waiting, nw);
        if (!is_checkmate(t, nb, team ^ 1, !adv, tmp) && !drawAfter) {
            for (int i = 0; i < nm && !persists; ++i) {
                const u32 mm = mating[i];
                // is_legal_move(waiting, mm)
                const int nl = gen_legal_wave(t.att, nw, tmp);
                bool legal = false;
                for (int q = 0; q < nl; ++q) legal |= tmp[q] == mm;
                if (!legal) continue;
                P ma = na, mw = nw;
                const int h = do_move(t.att, t.zob, mw, mm);
                if (h) add_to_hand(t.zob, ma, h);
                P nb2[2];
                put_pos(nb2, active, ma); put_pos(nb2, waiting, mw);
                persists = is_checkmate(t, nb2, team, adv, tmp);
            }
        }
        TRACE_EV(12, r, (persists ? 1 : 0) | (drawAfter ? 2 : 0));
        if (!persists) return false;
    }
    return true;
}

__device__ __forceinline__ bool has_unavoidable_waiting_board_mate(const RulesTab& t, const JBoard& j, int team, bool adv, int searchPly, u32* scratch) {
    const bool aOn = (int)j.bd[0].stm == team, bOn = (int)j.bd[1].stm == (team ^ 1);
    if (aOn == bOn) return false;                     // the rule needs exactly one board on turn
    return waiting_board_mate_scan((LdsRulesTab)&t, j, team, adv, searchPly, (LdsList)scratch);
}

// searchthread.cc:101-139.  Returns 0 NONE, 1 WIN, 2 LOSS, 3 DRAW (== NodeType numbering).
__device__ __forceinline__ int classify_terminal_position(const RulesTab& t, const JBoard& j, int teamToPlay, int rootTeam, bool rootAdv, int searchPly, int* endInPly, u32* scratch) {
    *endInPly = 0;
    const bool adv = teamToPlay == rootTeam ? rootAdv : !rootAdv;
    int cntA, cntB;
    PROF_T(t0);
    legal_counts(t, j.bd, cntA, cntB);               // each board's count serves exactly one of the two tests
    PROF_ADD(15, t0);
    PROF_T(t1);
    if (is_checkmate_c(t, j.bd, teamToPlay ^ 1, !adv, cntA, cntB, scratch)) { *endInPly = 1; return 1; }
    if (is_checkmate_c(t, j.bd, teamToPlay, adv, cntA, cntB, scratch)) { *endInPly = 1; return 2; }
    PROF_ADD(16, t1);
    PROF_T(t2);
    const bool drawn = jb_is_draw(j, searchPly);
    PROF_ADD(17, t2);
    if (drawn) return 3;
    PROF_T(t3);
    const bool wmate = searchPly > 0 && has_unavoidable_waiting_board_mate(t, j, teamToPlay, adv, searchPly, scratch);
    PROF_ADD(18, t3);
    if (wmate) { *endInPly = 3; return 2; }
    return 0;
}

}  // namespace hmd
