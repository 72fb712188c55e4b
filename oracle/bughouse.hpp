// oracle/bughouse.hpp — CPU restatement of the reference's Bughouse rules.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under hivemind_amd/ may include, link or
// call this file; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg use it, and only as the checker.
//
// What it restates (paths relative to /root/reference/engine/src):
//   * Fairy-Stockfish Position for the "bughouse" variant (variant.cpp:43-51):
//     generate<LEGAL> incl. list order (movegen.cpp:311-456), legal()
//     (position.cpp:949-1140), do_move (position.cpp:1349-1859), Zobrist keys
//     (position.cpp:142-167, 560-607), FEN set() (position.cpp:232-470).
//   * Board (environment/board.h, board.cc): partner-hand transfer, joint
//     make/unmake, is_checkmate, can_partner_provide_blocking_piece, is_draw,
//     repetition history, hash_key.
//   * perft (tools/benchmark.cc:59-76).
// Pinning: validated move-for-move against the reference's own sources built
// into oracle/_ref/libhmref.so (oracle/difftest.cc, tests/test_oracle_golden.py)
// and against the reference's gtest known answers (tests/golden/*.json).
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cstdint>
#include <cstring>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/hivemind_amd.h"

namespace hmo {

typedef uint64_t BB;
typedef uint32_t Move;

enum : int { WHITE = 0, BLACK = 1 };
enum : int { NO_PT = 0, PAWN = 1, KNIGHT = 2, BISHOP = 3, ROOK = 4, QUEEN = 5, KING = 6 };
enum : uint32_t { NORMAL = 0, EN_PASSANT = 1u << 12, CASTLING = 2u << 12, PROMOTION = 3u << 12, DROP = 4u << 12 };
enum : int { WHITE_OO = 1, WHITE_OOO = 2, BLACK_OO = 4, BLACK_OOO = 8 };
constexpr int SQ_NONE = 64;
constexpr Move MOVE_NONE = 0;

inline int to_sq(Move m) { return m & 63; }
inline int from_sq(Move m) { return (m >> 6) & 63; }
inline uint32_t type_of(Move m) { return m & (15u << 12); }
inline int promo_type(Move m) { return (m >> 16) & 63; }      // also dropped piece type
inline int in_hand_type(Move m) { return (m >> 22) & 63; }
inline Move make_move(int from, int to) { return Move((from << 6) | to); }
inline Move make_typed(uint32_t t, int from, int to, int pt = 0) { return Move((pt << 16) | t | (from << 6) | to); }
inline Move make_drop(int to, int pt) { return Move((pt << 22) | (pt << 16) | DROP | to); }   // types.h:777-779

inline int lsb(BB b) { return __builtin_ctzll(b); }
inline int msb(BB b) { return 63 - __builtin_clzll(b); }
inline int popcnt(BB b) { return __builtin_popcountll(b); }
inline int pop_lsb(BB& b) { int s = lsb(b); b &= b - 1; return s; }
inline BB sq_bb(int s) { return 1ULL << s; }
inline bool more_than_one(BB b) { return b & (b - 1); }

constexpr BB FileA = 0x0101010101010101ULL, FileH = FileA << 7;
constexpr BB Rank1 = 0xFFULL, Rank8 = Rank1 << 56;
inline BB rank_bb(int r) { return Rank1 << (8 * r); }
inline BB file_bb(int f) { return FileA << f; }

// ---------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------
struct Tables {
    BB knight[64], king[64], pawnAtt[2][64];
    BB ray[8][64];           // N,S,E,W,NE,NW,SE,SW
    BB between[64][64];      // FSF semantic: excludes s1, includes s2 (bitboard.h:307-320)
    int castleMask[64];      // castlingRightsMask
    // Zobrist (position.cpp:142-167)
    uint64_t psq[2][7][64];  // [colour][P..K(6)][sq]
    uint64_t zEp[8], zCastle[16], zSide;
    uint64_t inHand[2][6][64];   // [colour][P..Q][count]
    uint64_t inHandConst;        // XOR of inHand[pc][0] over every piece type that never has a hand count
    uint64_t zPromoted[64];      // oracle-own keys for the '~' marker in repetition identity
    uint64_t timeAdvantage;      // environment/zobrist.cc:16-28
    // policy tables (common/globals.cc:50-104)
    int polNormal[2][64][64][2];
    int polDrop[2][64][8];

    Tables() { init_attacks(); init_zobrist(); init_policy(); }

    static bool on_board(int f, int r) { return f >= 0 && f < 8 && r >= 0 && r < 8; }

    void init_attacks() {
        static const int kn[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
        static const int dirs[8][2] = {{0, 1}, {0, -1}, {1, 0}, {-1, 0}, {1, 1}, {-1, 1}, {1, -1}, {-1, -1}};  // (df,dr)
        for (int s = 0; s < 64; ++s) {
            int f = s & 7, r = s >> 3;
            knight[s] = king[s] = pawnAtt[0][s] = pawnAtt[1][s] = 0;
            for (auto& d : kn) if (on_board(f + d[0], r + d[1])) knight[s] |= sq_bb((r + d[1]) * 8 + f + d[0]);
            for (auto& d : dirs) if (on_board(f + d[0], r + d[1])) king[s] |= sq_bb((r + d[1]) * 8 + f + d[0]);
            for (int df : {-1, 1}) {
                if (on_board(f + df, r + 1)) pawnAtt[WHITE][s] |= sq_bb((r + 1) * 8 + f + df);
                if (on_board(f + df, r - 1)) pawnAtt[BLACK][s] |= sq_bb((r - 1) * 8 + f + df);
            }
            for (int d = 0; d < 8; ++d) {
                BB b = 0;
                int ff = f + dirs[d][0], rr = r + dirs[d][1];
                while (on_board(ff, rr)) { b |= sq_bb(rr * 8 + ff); ff += dirs[d][0]; rr += dirs[d][1]; }
                ray[d][s] = b;
            }
        }
        for (int a = 0; a < 64; ++a)
            for (int b = 0; b < 64; ++b) {
                between[a][b] = sq_bb(b);
                for (int d = 0; d < 8; ++d)
                    if (ray[d][a] & sq_bb(b)) {
                        // squares strictly between, plus b
                        BB seg = ray[d][a] & ~ray[d][b];
                        between[a][b] = seg;  // includes b, excludes a
                    }
            }
        std::memset(castleMask, 0, sizeof castleMask);
        castleMask[0] = WHITE_OOO; castleMask[7] = WHITE_OO; castleMask[4] = WHITE_OO | WHITE_OOO;
        castleMask[56] = BLACK_OOO; castleMask[63] = BLACK_OO; castleMask[60] = BLACK_OO | BLACK_OOO;
    }

    // misc.h:146-165 xorshift64*
    struct PRNG {
        uint64_t s;
        explicit PRNG(uint64_t seed) : s(seed) {}
        uint64_t rand64() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 2685821657736338717ULL; }
    };

    void init_zobrist() {
        PRNG rng(1070372);
        // psq: for c in {W,B}, pt = PAWN..KING(=63 in FSF), s = 0..63
        for (int c = 0; c < 2; ++c)
            for (int pt = 1; pt <= 63; ++pt)
                for (int s = 0; s < 64; ++s) {
                    uint64_t v = rng.rand64();
                    if (pt <= 5) psq[c][pt][s] = v;
                    else if (pt == 63) psq[c][KING][s] = v;
                }
        for (int f = 0; f < 8; ++f) zEp[f] = rng.rand64();
        for (int cr = 0; cr < 16; ++cr) zCastle[cr] = rng.rand64();
        zSide = rng.rand64();
        (void)rng.rand64();                       // noPawns
        for (int i = 0; i < 2 * 11; ++i) (void)rng.rand64();  // checks[2][CHECKS_NB]
        inHandConst = 0;
        for (int c = 0; c < 2; ++c)
            for (int pt = 1; pt <= 63; ++pt)
                for (int n = 0; n < 64; ++n) {
                    uint64_t v = rng.rand64();
                    if (pt <= 5) inHand[c][pt][n] = v;
                    else if (n == 0) inHandConst ^= v;   // set_state xors inHand[pc][0] for every pt (position.cpp:598-606)
                }
        // oracle-own keys (not in the reference): promoted-marker identity for repetition keys
        uint64_t x = 0x9e3779b97f4a7c15ULL;
        for (int s = 0; s < 64; ++s) {
            x += 0x9e3779b97f4a7c15ULL;
            uint64_t z = x;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
            zPromoted[s] = z ^ (z >> 31);
        }
        // environment/zobrist.cc: mt19937_64(1070372): ply[1024] then timeAdvantage
        std::mt19937_64 mt(1070372);
        for (int i = 0; i < 1024; ++i) (void)mt();
        timeAdvantage = mt();
    }

    // Label table restated from constants.h:24 (UCI_MOVES) by construction, first
    // occurrence wins (globals.cc:51-56).  Channel layout: SURVEY.md Appendix A.
    static std::string sqname(int s) { std::string r; r += char('a' + (s & 7)); r += char('1' + (s >> 3)); return r; }
    void init_policy() {
        std::vector<std::string> labels(HM_POLICY_VALUES, "illegal");
        for (int s = 0; s < 64; ++s) labels[s] = "pass";
        const char dropc[5] = {'P', 'N', 'R', 'B', 'Q'};   // NB: R before B
        for (int k = 0; k < 5; ++k)
            for (int s = 0; s < 64; ++s) {
                int r = s >> 3;
                if (dropc[k] == 'P' && (r == 0 || r == 7)) continue;
                labels[(1 + k) * 64 + s] = std::string(1, dropc[k]) + "@" + sqname(s);
            }
        static const int qd[8][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}};  // N,NE,E,SE,S,SW,W,NW
        for (int d = 0; d < 8; ++d)
            for (int dist = 1; dist <= 7; ++dist)
                for (int s = 0; s < 64; ++s) {
                    int f = (s & 7) + qd[d][0] * dist, r = (s >> 3) + qd[d][1] * dist;
                    if (on_board(f, r)) labels[(6 + d * 7 + dist - 1) * 64 + s] = sqname(s) + sqname(r * 8 + f);
                }
        static const int kn[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
        for (int k = 0; k < 8; ++k)
            for (int s = 0; s < 64; ++s) {
                int f = (s & 7) + kn[k][0], r = (s >> 3) + kn[k][1];
                if (on_board(f, r)) labels[(62 + k) * 64 + s] = sqname(s) + sqname(r * 8 + f);
            }
        static const int up[3] = {-1, 0, 1};
        for (int k = 0; k < 3; ++k)
            for (int s = 48; s < 56; ++s) {
                int f = (s & 7) + up[k];
                if (on_board(f, 7)) labels[(70 + k) * 64 + s] = sqname(s) + sqname(56 + f) + "n";
            }
        policy_labels = labels;
        build_policy_tables(labels);
    }
    std::vector<std::string> policy_labels;

    static std::string mirror_uci(const std::string& u) {   // globals.cc:31-48
        if (u == "pass") return u;
        std::string r = u;
        if (u.size() >= 4 && u[1] == '@') { r[3] = char('0' + (9 - (u[3] - '0'))); return r; }
        if (u.size() >= 4) { r[1] = char('0' + (9 - (u[1] - '0'))); r[3] = char('0' + (9 - (u[3] - '0'))); }
        return r;
    }
    void build_policy_tables(const std::vector<std::string>& labels) {
        std::vector<std::pair<std::string, int>> idx;   // first occurrence wins
        auto find = [&](const std::string& s) -> int {
            for (auto& p : idx) if (p.first == s) return p.second;
            return -1;
        };
        // build a hash-free but fast map: sort unique
        std::vector<std::pair<std::string, int>> tmp;
        for (int i = 0; i < (int)labels.size(); ++i) tmp.emplace_back(labels[i], i);
        std::stable_sort(tmp.begin(), tmp.end(), [](auto& a, auto& b) { return a.first < b.first; });
        for (auto& p : tmp) if (idx.empty() || idx.back().first != p.first) idx.push_back(p);
        auto bfind = [&](const std::string& s) -> int {
            auto it = std::lower_bound(idx.begin(), idx.end(), s, [](auto& a, const std::string& k) { return a.first < k; });
            return (it != idx.end() && it->first == s) ? it->second : -1;
        };
        (void)find;
        const char ptc[8] = {' ', 'P', 'N', 'B', 'R', 'Q', 'K', ' '};
        for (int c = 0; c < 2; ++c) {
            for (int f = 0; f < 64; ++f)
                for (int t = 0; t < 64; ++t) {
                    std::string u = sqname(f) + sqname(t);
                    polNormal[c][f][t][0] = bfind(c ? mirror_uci(u) : u);
                    std::string un = u + "n";
                    polNormal[c][f][t][1] = bfind(c ? mirror_uci(un) : un);
                }
            for (int t = 0; t < 64; ++t)
                for (int pt = 0; pt < 8; ++pt) {
                    polDrop[c][t][pt] = -1;
                    if (pt >= 1 && pt <= 5) {
                        std::string u = std::string(1, ptc[pt]) + "@" + sqname(t);
                        polDrop[c][t][pt] = bfind(c ? mirror_uci(u) : u);
                    }
                }
        }
    }
};

inline const Tables& T() { static Tables t; return t; }

// sliding attacks by classical rays
inline BB ray_att(int d, int s, BB occ) {
    const Tables& t = T();
    BB a = t.ray[d][s];
    BB bl = a & occ;
    if (bl) {
        // dirs with increasing square index: N(0), E(2), NE(4), NW(5)
        int b = (d == 0 || d == 2 || d == 4 || d == 5) ? lsb(bl) : msb(bl);
        a ^= t.ray[d][b];
    }
    return a;
}
inline BB rook_att(int s, BB occ) { return ray_att(0, s, occ) | ray_att(1, s, occ) | ray_att(2, s, occ) | ray_att(3, s, occ); }
inline BB bishop_att(int s, BB occ) { return ray_att(4, s, occ) | ray_att(5, s, occ) | ray_att(6, s, occ) | ray_att(7, s, occ); }
inline BB piece_att(int pt, int s, BB occ) {
    switch (pt) {
        case KNIGHT: return T().knight[s];
        case BISHOP: return bishop_att(s, occ);
        case ROOK: return rook_att(s, occ);
        case QUEEN: return rook_att(s, occ) | bishop_att(s, occ);
        case KING: return T().king[s];
    }
    return 0;
}

// get_fast_policy_index (common/utils.h:184-216)
inline int policy_index(Move m, int stm) {
    if (m == MOVE_NONE) return 0;
    const Tables& t = T();
    if (type_of(m) == DROP) {
        int pt = promo_type(m);
        if (pt >= 1 && pt <= 5) return t.polDrop[stm][to_sq(m)][pt];
        return -1;
    }
    int f = from_sq(m), to = to_sq(m);
    if (type_of(m) == PROMOTION) {
        int pt = promo_type(m);
        if (pt == KNIGHT) return t.polNormal[stm][f][to][1];
        if (pt == QUEEN) return t.polNormal[stm][f][to][0];
        return -1;
    }
    return t.polNormal[stm][f][to][0];
}

// ---------------------------------------------------------------------------
// Position (one board)
// ---------------------------------------------------------------------------
struct Pos {
    uint8_t board[64];     // 0 or (colour<<3)|pt
    BB byType[7];
    BB byColor[2];
    BB promoted;
    int hand[2][6];
    int castling, ep, stm, rule50, gamePly;
    uint64_t key;          // StateInfo::key
    BB checkers;

    BB pieces() const { return byColor[0] | byColor[1]; }
    BB pieces(int c, int pt) const { return byColor[c] & byType[pt]; }
    int ksq(int c) const { return lsb(pieces(c, KING)); }
    static int pc_color(int pc) { return pc >> 3; }
    static int pc_type(int pc) { return pc & 7; }
    static int make_pc(int c, int pt) { return (c << 3) | pt; }

    void clear() { std::memset(this, 0, sizeof *this); ep = SQ_NONE; }

    void put(int c, int pt, int s, bool prom = false) {
        board[s] = (uint8_t)make_pc(c, pt);
        byType[pt] |= sq_bb(s); byColor[c] |= sq_bb(s);
        if (prom) promoted |= sq_bb(s);
    }
    void remove(int s) {
        int pc = board[s];
        byType[pc_type(pc)] ^= sq_bb(s); byColor[pc_color(pc)] ^= sq_bb(s);
        board[s] = 0; promoted &= ~sq_bb(s);
    }
    void move_piece(int from, int to) {   // position.h:1259-1275
        int pc = board[from];
        BB ft = sq_bb(from) ^ sq_bb(to);
        byType[pc_type(pc)] ^= ft; byColor[pc_color(pc)] ^= ft;
        board[from] = 0; board[to] = (uint8_t)pc;
        if (promoted & sq_bb(from)) promoted ^= ft;
    }

    // attackers_to (position.cpp:845-855, fastAttacks path)
    BB attackers_to(int s, BB occ, int c) const {
        const Tables& t = T();
        return (t.pawnAtt[c ^ 1][s] & pieces(c, PAWN))
             | (t.knight[s] & pieces(c, KNIGHT))
             | (rook_att(s, occ) & byColor[c] & (byType[ROOK] | byType[QUEEN]))
             | (bishop_att(s, occ) & byColor[c] & (byType[BISHOP] | byType[QUEEN]))
             | (t.king[s] & pieces(c, KING));
    }

    // set_state (position.cpp:560-607)
    void compute_key() {
        const Tables& t = T();
        uint64_t k = 0;
        for (BB b = pieces(); b;) { int s = pop_lsb(b); k ^= t.psq[pc_color(board[s])][pc_type(board[s])][s]; }
        if (ep != SQ_NONE) k ^= t.zEp[ep & 7];
        if (stm == BLACK) k ^= t.zSide;
        k ^= t.zCastle[castling];
        for (int c = 0; c < 2; ++c)
            for (int pt = PAWN; pt <= QUEEN; ++pt) k ^= t.inHand[c][pt][hand[c][pt]];
        k ^= t.inHandConst;
        key = k;
    }
    void compute_checkers() { checkers = byType[KING] & byColor[stm] ? attackers_to(ksq(stm), pieces(), stm ^ 1) : 0; }

    // Position::key() (position.h:1156-1159)
    uint64_t pos_key() const {
        return rule50 < 14 ? key : key ^ (uint64_t((rule50 - 14) / 8) * 6364136223846793005ULL + 1442695040888963407ULL);
    }

    // Repetition identity: same equivalence classes as Board::board_only_key
    // (board.h:68-89): placement incl. '~' markers, side to move, castling, ep square.
    uint64_t rep_key() const {
        const Tables& t = T();
        uint64_t k = 0;
        for (BB b = pieces(); b;) { int s = pop_lsb(b); k ^= t.psq[pc_color(board[s])][pc_type(board[s])][s]; }
        for (BB b = promoted; b;) k ^= t.zPromoted[pop_lsb(b)];
        if (ep != SQ_NONE) k ^= t.zEp[ep & 7];
        if (stm == BLACK) k ^= t.zSide;
        k ^= t.zCastle[castling];
        return k;
    }

    // Position::fen(false, true) (position.cpp:637-768), bughouse subset: placement with '~' after promoted pieces, [hand] (white
    // then black, queen down to pawn), side, castling, ep square, rule50, fullmove number
    std::string fen() const {
        static const char pcs[] = " PNBRQK";
        std::string s;
        for (int r = 7; r >= 0; --r) {
            int empty = 0;
            for (int f = 0; f < 8; ++f) {
                const int sq = r * 8 + f, pc = board[sq];
                if (!pc) { ++empty; continue; }
                if (empty) { s += (char)('0' + empty); empty = 0; }
                const char ch = pcs[pc_type(pc)];
                s += pc_color(pc) == BLACK ? (char)tolower((unsigned char)ch) : ch;
                if (promoted & sq_bb(sq)) s += '~';
            }
            if (empty) s += (char)('0' + empty);
            if (r > 0) s += '/';
        }
        s += '[';
        for (int c = WHITE; c <= BLACK; ++c)
            for (int pt = QUEEN; pt >= PAWN; --pt) s += std::string((size_t)hand[c][pt], c == BLACK ? (char)tolower((unsigned char)pcs[pt]) : pcs[pt]);
        s += ']';
        s += stm == WHITE ? " w " : " b ";
        if (castling & WHITE_OO) s += 'K';
        if (castling & WHITE_OOO) s += 'Q';
        if (castling & BLACK_OO) s += 'k';
        if (castling & BLACK_OOO) s += 'q';
        if (!castling) s += '-';
        if (ep != SQ_NONE) { s += ' '; s += (char)('a' + (ep & 7)); s += (char)('1' + (ep >> 3)); s += ' '; }
        else s += " - ";
        s += std::to_string(rule50) + " " + std::to_string(1 + (gamePly - (stm == BLACK ? 1 : 0)) / 2);
        return s;
    }

    // FEN (position.cpp:232-470), bughouse subset: placement[hand] stm castling ep rule50 fullmove
    void set_fen(const std::string& fen) {
        clear();
        std::istringstream ss(fen);
        std::string placement, stmS, cast = "-", epS = "-";
        ss >> placement >> stmS;
        int s = 56, slashes = 0;
        size_t i = 0;
        static const std::string pcs = " PNBRQK";
        for (; i < placement.size(); ++i) {
            char ch = placement[i];
            if (ch == '[') break;
            if (isdigit((unsigned char)ch)) s += ch - '0';
            else if (ch == '/') {
                if (++slashes == 8) { ++i; break; }       // pieces in hand after an additional slash (position.cpp:290-296)
                s -= 16;
            }
            else {
                size_t idx = pcs.find((char)toupper((unsigned char)ch));
                if (idx == std::string::npos || idx == 0) continue;
                bool prom = (i + 1 < placement.size() && placement[i + 1] == '~');
                put(islower((unsigned char)ch) ? BLACK : WHITE, (int)idx, s, prom);
                if (prom) ++i;
                ++s;
            }
        }
        for (; i < placement.size(); ++i) {
            char ch = placement[i];
            size_t idx = pcs.find((char)toupper((unsigned char)ch));
            if (ch == '[' || ch == ']' || idx == std::string::npos || idx == 0 || idx == 6) continue;
            hand[islower((unsigned char)ch) ? BLACK : WHITE][idx]++;
        }
        stm = (stmS == "b") ? BLACK : WHITE;
        // optional castling / ep
        ss >> std::ws;
        int r50 = 0, full = 1;
        if (!isdigit(ss.peek())) {
            ss >> cast >> epS;
        }
        ss >> r50 >> full;
        for (char ch : cast) {
            int c = islower((unsigned char)ch) ? BLACK : WHITE;
            char up = (char)toupper((unsigned char)ch);
            int rook = make_pc(c, ROOK);
            int rsq = -1;
            if (up == 'K') { for (rsq = c ? 63 : 7; rsq >= (c ? 56 : 0) && board[rsq] != rook; --rsq) {} }
            else if (up == 'Q') { for (rsq = c ? 56 : 0; rsq <= (c ? 63 : 7) && board[rsq] != rook; ++rsq) {} }
            else continue;
            if (rsq < (c ? 56 : 0) || rsq > (c ? 63 : 7)) continue;
            // standard chess only: king on e-file, rook on a/h
            int kfile = 4;
            if (board[(c ? 56 : 0) + kfile] != make_pc(c, KING)) continue;
            if (rsq == (c ? 63 : 7)) castling |= c ? BLACK_OO : WHITE_OO;
            else if (rsq == (c ? 56 : 0)) castling |= c ? BLACK_OOO : WHITE_OOO;
        }
        if (epS.size() == 2 && epS[0] >= 'a' && epS[0] <= 'h' && epS[1] >= '1' && epS[1] <= '8') {
            int e = (epS[1] - '1') * 8 + (epS[0] - 'a');
            const Tables& t = T();
            int push_us = stm == WHITE ? 8 : -8;
            bool ok = (t.pawnAtt[stm ^ 1][e] & pieces(stm, PAWN))
                   && (pieces(stm ^ 1, PAWN) & sq_bb(e - push_us))
                   && !(pieces() & (sq_bb(e) | sq_bb(e + push_us)));
            ep = ok ? e : SQ_NONE;
        }
        rule50 = r50;
        gamePly = std::max(2 * (full - 1), 0) + (stm == BLACK);
        compute_key();
        compute_checkers();
    }

    // -------------------------------------------------------------------
    // move generation in the reference's order
    // -------------------------------------------------------------------
    BB check_squares(int pt) const {   // set_check_info (position.cpp:521-552): attacks_bb(~stm, pt, ksq(~stm), pieces())
        int k = ksq(stm ^ 1);
        if (pt == PAWN) return T().pawnAtt[stm ^ 1][k];
        return piece_att(pt, k, pieces());
    }

    int gen_pseudo(Move* list) const {
        const Tables& t = T();
        const int us = stm, them = stm ^ 1;
        const int k = ksq(us);
        const bool evasion = checkers != 0;
        const BB occ = pieces();
        Move* m = list;
        BB target = 0;
        if (!evasion || !more_than_one(checkers)) {
            target = evasion ? t.between[k][lsb(checkers)] : ~byColor[us];
            // ---- pawns (movegen.cpp:102-246)
            {
                const int up = us == WHITE ? 8 : -8;
                const BB r7 = us == WHITE ? rank_bb(6) : rank_bb(1);
                const BB r3 = us == WHITE ? rank_bb(2) : rank_bb(5);
                auto shiftUp = [&](BB b) { return us == WHITE ? b << 8 : b >> 8; };
                auto shiftUR = [&](BB b) { return us == WHITE ? (b & ~FileH) << 9 : (b & ~FileA) >> 9; };  // NE / SW
                auto shiftUL = [&](BB b) { return us == WHITE ? (b & ~FileA) << 7 : (b & ~FileH) >> 7; };  // NW / SE
                const int dUR = us == WHITE ? 9 : -9, dUL = us == WHITE ? 7 : -7;
                const BB empty = ~occ;
                const BB enemies = evasion ? checkers : byColor[them];
                const BB on7 = pieces(us, PAWN) & r7, not7 = pieces(us, PAWN) & ~r7;
                BB b1 = shiftUp(not7) & empty;
                BB b2 = shiftUp(b1 & r3) & empty;
                if (evasion) { b1 &= target; b2 &= target; }
                while (b1) { int to = pop_lsb(b1); *m++ = make_move(to - up, to); }
                while (b2) { int to = pop_lsb(b2); *m++ = make_move(to - 2 * up, to); }
                if (on7) {
                    BB p1 = shiftUR(on7) & enemies, p2 = shiftUL(on7) & enemies, p3 = shiftUp(on7) & empty;
                    if (evasion) p3 &= target;
                    auto promos = [&](int from, int to) {   // Q,R,B,N (variant.h:54 std::greater)
                        for (int pt : {QUEEN, ROOK, BISHOP, KNIGHT}) *m++ = make_typed(PROMOTION, from, to, pt);
                    };
                    while (p1) { int to = pop_lsb(p1); promos(to - dUR, to); }
                    while (p2) { int to = pop_lsb(p2); promos(to - dUL, to); }
                    while (p3) { int to = pop_lsb(p3); promos(to - up, to); }
                }
                BB c1 = shiftUR(not7) & enemies, c2 = shiftUL(not7) & enemies;
                while (c1) { int to = pop_lsb(c1); *m++ = make_move(to - dUR, to); }
                while (c2) { int to = pop_lsb(c2); *m++ = make_move(to - dUL, to); }
                if (ep != SQ_NONE) {
                    // "An en passant capture cannot resolve a discovered check" (movegen.cpp:232-234);
                    // in FSF this `return`s from generate_pawn_moves only.
                    if (!(evasion && (target & sq_bb(ep + up)))) {
                        BB e = not7 & t.pawnAtt[them][ep];
                        while (e) *m++ = make_typed(EN_PASSANT, pop_lsb(e), ep);
                    }
                }
            }
            // ---- knights, bishops, rooks, queens (movegen.cpp:249-308)
            for (int pt = KNIGHT; pt <= QUEEN; ++pt) {
                BB bb = pieces(us, pt);
                while (bb) {
                    int from = pop_lsb(bb);
                    BB b = piece_att(pt, from, occ) & target;
                    while (b) *m++ = make_move(from, pop_lsb(b));
                }
            }
            // ---- drops (movegen.cpp:75-100, 345-347), incl. the virtual drops FSF
            // emits in EVASIONS and later strips (they perturb the final order)
            {
                BB b0 = target & ~byColor[them] & ~occ;
                for (int pt = PAWN; pt <= QUEEN; ++pt) {
                    BB b = b0;
                    if (pt == PAWN) b &= ~(Rank1 | Rank8);
                    if (hand[us][pt] > 0) {
                        while (b) *m++ = make_drop(pop_lsb(b), pt);
                    } else if (evasion) {
                        b &= check_squares(pt);
                        while (b) *m++ = make_drop(pop_lsb(b), pt);   // virtual
                    }
                }
            }
        }
        // ---- king (movegen.cpp:383-401)
        {
            BB b = t.king[k] & (evasion ? ~byColor[us] : target);
            while (b) *m++ = make_move(k, pop_lsb(b));
            if (!evasion && (castling & (us == WHITE ? 3 : 12))) {
                for (int side = 0; side < 2; ++side) {
                    int cr = us == WHITE ? (side == 0 ? WHITE_OO : WHITE_OOO) : (side == 0 ? BLACK_OO : BLACK_OOO);
                    if (!(castling & cr)) continue;
                    int rsq = (us == WHITE ? 0 : 56) + (side == 0 ? 7 : 0);
                    BB path = side == 0 ? (sq_bb(rsq - 1) | sq_bb(rsq - 2)) : (sq_bb(rsq + 1) | sq_bb(rsq + 2) | sq_bb(rsq + 3));
                    if (path & occ) continue;
                    *m++ = make_typed(CASTLING, k, rsq);
                }
            }
        }
        return int(m - list);
    }

    bool is_virtual_drop(Move m) const { return type_of(m) == DROP && hand[stm][in_hand_type(m)] <= 0; }

    // legal() (position.cpp:949-1140) for this variant
    bool legal(Move m) const {
        const int us = stm, them = stm ^ 1;
        const int from = from_sq(m), to = to_sq(m);
        const BB occ = pieces();
        const int k = ksq(us);
        if (type_of(m) == EN_PASSANT) {
            int capsq = to - (us == WHITE ? 8 : -8);
            BB o = (occ ^ sq_bb(from) ^ sq_bb(capsq)) | sq_bb(to);
            return !(attackers_to(k, o, them) & o);
        }
        if (type_of(m) == CASTLING) {
            int kto = (us == WHITE ? 0 : 56) + (to > from ? 6 : 2);
            int step = kto > from ? -1 : 1;
            for (int s = kto; s != from; s += step)
                if (attackers_to(s, occ, them)) return false;
            return !attackers_to(kto, occ ^ sq_bb(to), them);
        }
        BB o = (type_of(m) != DROP ? occ ^ sq_bb(from) : occ) | sq_bb(to);
        if (type_of(m) != DROP && pc_type(board[from]) == KING) return !attackers_to(to, o, them);
        return !(attackers_to(k, o, them) & ~sq_bb(to));
    }

    int gen_legal(Move* list) const {   // generate<LEGAL> (movegen.cpp:439-456)
        int n = gen_pseudo(list);
        int cur = 0;
        while (cur != n) {
            if (!legal(list[cur]) || is_virtual_drop(list[cur])) list[cur] = list[--n];
            else ++cur;
        }
        return n;
    }
    std::vector<Move> legal_moves() const {
        Move buf[1024];
        int n = gen_legal(buf);
        return std::vector<Move>(buf, buf + n);
    }

    // do_move (position.cpp:1349-1859); returns pieceToHand as (colour<<3|pt) or 0
    int do_move(Move m) {
        const Tables& t = T();
        uint64_t k = key ^ t.zSide;
        ++gamePly; ++rule50;
        const int us = stm, them = stm ^ 1;
        const int from = from_sq(m), to = to_sq(m);
        const uint32_t mt = type_of(m);
        int pc = mt == DROP ? make_pc(us, promo_type(m)) : board[from];
        int captured = mt == EN_PASSANT ? make_pc(them, PAWN) : (mt == DROP ? 0 : board[to]);
        int toHand = 0;
        if (mt == CASTLING) {
            bool kingSide = to > from;
            int rfrom = to, rto = (us == WHITE ? 0 : 56) + (kingSide ? 5 : 3), kto = (us == WHITE ? 0 : 56) + (kingSide ? 6 : 2);
            remove(from); remove(rfrom);
            put(us, KING, kto); put(us, ROOK, rto);
            k ^= t.psq[us][ROOK][rfrom] ^ t.psq[us][ROOK][rto];
            k ^= t.psq[us][KING][from] ^ t.psq[us][KING][kto];
            captured = 0;
        }
        if (captured) {
            int capsq = to;
            if (mt == EN_PASSANT) capsq -= (us == WHITE ? 8 : -8);
            bool capProm = promoted & sq_bb(capsq);
            remove(capsq);
            toHand = capProm ? make_pc(them, PAWN) : captured;
            k ^= t.psq[them][pc_type(captured)][capsq];
            rule50 = 0;
        }
        if (mt == DROP) {
            int pt = in_hand_type(m);
            k ^= t.psq[us][pc_type(pc)][to] ^ t.inHand[us][pt][hand[us][pt] - 1] ^ t.inHand[us][pt][hand[us][pt]];
        } else if (mt != CASTLING) {
            k ^= t.psq[us][pc_type(pc)][from] ^ t.psq[us][pc_type(pc)][to];
        }
        if (ep != SQ_NONE) { k ^= t.zEp[ep & 7]; ep = SQ_NONE; }
        if (mt != DROP && castling && (t.castleMask[from] | t.castleMask[to])) {
            k ^= t.zCastle[castling];
            castling &= ~(t.castleMask[from] | t.castleMask[to]);
            k ^= t.zCastle[castling];
        }
        if (mt == DROP) {
            put(us, pc_type(pc), to);
            hand[us][in_hand_type(m)]--;
        } else if (mt != CASTLING) {
            move_piece(from, to);
        }
        if (pc_type(pc) == PAWN) {
            const int push = us == WHITE ? 8 : -8;
            if (mt != DROP && std::abs(to - from) == 16 && (t.pawnAtt[us][to - push] & pieces(them, PAWN))) {
                ep = to - push;
                k ^= t.zEp[ep & 7];
            } else if (mt == PROMOTION) {
                int pt = promo_type(m);
                remove(to);
                put(us, pt, to, true);
                k ^= t.psq[us][PAWN][to] ^ t.psq[us][pt][to];
            }
            rule50 = 0;
        }
        key = k;
        stm = them;
        compute_checkers();
        return toHand;
    }

    // add/remove_from_hand_with_key (position.cpp:47-57)
    void add_to_hand(int c, int pt) {
        const Tables& t = T();
        key ^= t.inHand[c][pt][hand[c][pt]] ^ t.inHand[c][pt][hand[c][pt] + 1];
        hand[c][pt]++;
    }
    void remove_from_hand(int c, int pt) {
        const Tables& t = T();
        key ^= t.inHand[c][pt][hand[c][pt]] ^ t.inHand[c][pt][hand[c][pt] - 1];
        hand[c][pt]--;
    }

    bool is_capture(Move m) const {   // position.h:1212-1216
        return (board[to_sq(m)] && type_of(m) != CASTLING && type_of(m) != DROP) || type_of(m) == EN_PASSANT;
    }
    bool gives_check(Move m) const {   // semantic of position.cpp:1248-1343: is the opponent in check afterwards
        Pos c = *this;
        c.do_move(m);
        return c.checkers != 0;
    }

    void to_compact(hm_pos* o) const {
        for (int i = 0; i < 6; ++i) o->by_type[i] = byType[i + 1];
        o->by_color[0] = byColor[0]; o->by_color[1] = byColor[1];
        o->promoted = promoted; o->key = key;
        for (int c = 0; c < 2; ++c) for (int pt = 1; pt <= 5; ++pt) o->hand[c][pt - 1] = (uint8_t)hand[c][pt];
        o->castling = (uint8_t)castling; o->ep = (uint8_t)ep; o->stm = (uint8_t)stm;
        o->rule50 = (uint8_t)std::min(rule50, 255); o->game_ply = (uint16_t)gamePly;
    }
    void from_compact(const hm_pos* o) {
        clear();
        for (int pt = 1; pt <= 6; ++pt)
            for (BB b = o->by_type[pt - 1]; b;) {
                int s = pop_lsb(b);
                int c = (o->by_color[1] >> s) & 1;
                put(c, pt, s, (o->promoted >> s) & 1);
            }
        for (int c = 0; c < 2; ++c) for (int pt = 1; pt <= 5; ++pt) hand[c][pt] = o->hand[c][pt - 1];
        castling = o->castling; ep = o->ep; stm = o->stm; rule50 = o->rule50; gamePly = o->game_ply;
        key = o->key;
        compute_checkers();
    }
};

static const char* const START_FEN = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1";

// ---------------------------------------------------------------------------
// Board (two positions) — environment/board.h, board.cc
// ---------------------------------------------------------------------------
struct Board {
    Pos pos[2];
    struct Undo { Pos before; int toHand; Move move; };
    std::vector<Undo> states[2];
    std::vector<uint64_t> positionHistory[2], positionHistoryPrefixes[2];
    std::vector<Move> moveHistory[2];

    static constexpr uint64_t HISTORY_HASH_SEED = 0xcbf29ce484222325ULL;
    static uint64_t mix_hash(uint64_t key, uint64_t value) {   // board.h:126-131
        value += 0x9e3779b97f4a7c15ULL;
        value = (value ^ (value >> 30)) * 0xbf58476d1ce4e5b9ULL;
        value = (value ^ (value >> 27)) * 0x94d049bb133111ebULL;
        return key ^ (value ^ (value >> 31));
    }

    Board() { set_fen(0, START_FEN); set_fen(1, START_FEN); }

    void set_fen(int b, const std::string& fen) {   // board.h:183-191
        pos[b].set_fen(fen);
        states[b].clear();
        positionHistory[b].clear(); positionHistoryPrefixes[b].clear(); moveHistory[b].clear();
        record_position(b);
    }
    void set(const std::string& fen) {   // board.cc:27-49 "fenA | fenB"
        size_t bar = fen.find('|');
        auto trim = [](std::string s) {
            size_t a = s.find_first_not_of(" \n\r\t\f\v"), e = s.find_last_not_of(" \n\r\t\f\v");
            return a == std::string::npos ? std::string() : s.substr(a, e - a + 1);
        };
        set_fen(0, trim(fen.substr(0, bar)));
        set_fen(1, trim(bar == std::string::npos ? std::string() : fen.substr(bar + 1)));
    }
    void record_position(int b) {   // board.h:95-102
        uint64_t k = pos[b].rep_key();
        positionHistory[b].push_back(k);
        uint64_t prefix = positionHistoryPrefixes[b].empty() ? HISTORY_HASH_SEED : positionHistoryPrefixes[b].back();
        positionHistoryPrefixes[b].push_back(mix_hash(prefix, k));
    }
    void unrecord_position(int b) {
        if (!positionHistory[b].empty()) { positionHistory[b].pop_back(); positionHistoryPrefixes[b].pop_back(); }
    }
    uint64_t history_key(int b) const {   // board.h:133-138
        uint64_t prefix = positionHistoryPrefixes[b].empty() ? HISTORY_HASH_SEED : positionHistoryPrefixes[b].back();
        return mix_hash(prefix, positionHistory[b].size());
    }
    uint64_t hash_key(bool adv = false) const {   // board.h:44-60
        uint64_t k0 = mix_hash(pos[0].pos_key(), (uint64_t)pos[0].rule50);
        uint64_t k1 = mix_hash(pos[1].pos_key(), (uint64_t)pos[1].rule50);
        uint64_t combined = k0 ^ (k1 + 0x9e3779b97f4a7c15ULL + (k0 << 6) + (k0 >> 2));
        uint64_t rc = history_key(0);
        rc ^= history_key(1) + 0x9e3779b97f4a7c15ULL + (rc << 6) + (rc >> 2);
        combined ^= rc + 0x9e3779b97f4a7c15ULL + (combined << 6) + (combined >> 2);
        return adv ? (combined ^ T().timeAdvantage) : combined;
    }

    void apply(int b, Move m) {
        Undo u; u.before = pos[b]; u.move = m;
        u.toHand = pos[b].do_move(m);
        states[b].push_back(u);
        if (u.toHand) pos[1 - b].add_to_hand(Pos::pc_color(u.toHand), Pos::pc_type(u.toHand));   // board.cc:101-104
        record_position(b);
        moveHistory[b].push_back(m);
    }
    void revert(int b) {
        Undo& u = states[b].back();
        if (u.toHand) pos[1 - b].remove_from_hand(Pos::pc_color(u.toHand), Pos::pc_type(u.toHand));
        pos[b] = u.before;
        states[b].pop_back();
        unrecord_position(b);
        if (!moveHistory[b].empty()) moveHistory[b].pop_back();
    }
    void push_move(int b, Move m) { apply(b, m); }   // board.cc:98-108
    void pop_move(int b) { revert(b); }              // board.cc:117-130

    bool is_legal_move(int b, Move m) const {        // board.cc:110-113
        if (m == MOVE_NONE) return true;
        Move buf[1024];
        int n = pos[b].gen_legal(buf);
        return std::find(buf, buf + n, m) != buf + n;
    }
    std::vector<Move> legal_moves(int b) const { return pos[b].legal_moves(); }

    void make_moves(Move a, Move b) {   // board.cc:316-341
        if (!is_legal_move(0, a) || !is_legal_move(1, b)) throw std::logic_error("illegal joint action");
        if (a != MOVE_NONE) apply(0, a);
        if (b != MOVE_NONE) apply(1, b);
    }
    void unmake_moves(Move a, Move b) {   // board.cc:343-358
        if (b != MOVE_NONE) revert(1);
        if (a != MOVE_NONE) revert(0);
    }

    Move last_move(int b) const { return moveHistory[b].empty() ? MOVE_NONE : moveHistory[b].back(); }
    int repetition_count(int b) const {   // board.h:326-332
        uint64_t k = pos[b].rep_key();
        return (int)std::count(positionHistory[b].begin(), positionHistory[b].end(), k);
    }
    bool is_draw_on_board(int b, int ply = 0) const {   // board.h:423-453
        if (pos[b].rule50 >= 100) return true;
        uint64_t k = pos[b].rep_key();
        int threshold = ply > 0 ? 1 : 2, cnt = 0;
        const auto& h = positionHistory[b];
        for (size_t i = 0; i + 1 < h.size(); ++i)
            if (h[i] == k && ++cnt >= threshold) return true;
        return false;
    }
    bool is_draw(int ply = 0) const { return is_draw_on_board(0, ply) || is_draw_on_board(1, ply); }

    // board.cc:214-314
    bool can_partner_provide_blocking_piece(int boardInCheck, int checkedSide, bool adv) {
        int pb = 1 - boardInCheck;
        int partnerSide = checkedSide ^ 1;
        bool partnerTurn = pos[pb].stm == partnerSide;
        if (!partnerTurn && !adv) return false;
        const Pos& p = pos[boardInCheck];
        int k = p.ksq(checkedSide);
        BB checkers = p.checkers;
        if (more_than_one(checkers)) return false;
        int csq = lsb(checkers);
        BB blocking = T().between[k][csq];
        // reference: `if (!blocking_squares) return false;` — between_bb always contains the checker
        // square, so this never fires; the empty-square filter below does the work.
        if (!blocking) return false;
        BB avail = blocking & ~p.pieces();
        if (!avail) return false;
        BB pawnValid = avail & ~(Rank1 | Rank8);
        auto useful = [&](const Board& cand) {
            const Pos& q = cand.pos[pb];
            Move buf[1024];
            int n = q.gen_legal(buf);
            for (int i = 0; i < n; ++i) {
                Move m = buf[i];
                int to = to_sq(m);
                int captured = type_of(m) == EN_PASSANT ? Pos::make_pc(partnerSide ^ 1, PAWN)
                             : (type_of(m) == DROP ? 0 : q.board[to]);
                // NB reference uses piece_on(to) for every non-ep move, incl. castling (own rook on `to`)
                if (type_of(m) == CASTLING) captured = q.board[to];
                if (!captured) continue;
                int ct = (q.promoted & sq_bb(to)) ? PAWN : Pos::pc_type(captured);
                if (ct == PAWN) { if (pawnValid) return true; }
                else return true;
            }
            return false;
        };
        if (partnerTurn) return useful(*this);
        if (adv) {
            Move buf[1024];
            int n = pos[pb].gen_legal(buf);
            if (!n) return false;
            for (int i = 0; i < n; ++i) {
                Board fut(*this);
                fut.push_move(pb, buf[i]);
                if (!useful(fut)) return false;
            }
            return true;
        }
        return false;
    }

    // board.cc:169-208
    bool is_checkmate(int side, bool adv = false) {
        Move buf[1024];
        if (pos[0].stm == side && pos[0].checkers) {
            if (!pos[0].gen_legal(buf) && !can_partner_provide_blocking_piece(0, side, adv)) return true;
        }
        if (pos[1].stm == (side ^ 1) && pos[1].checkers) {
            if (!pos[1].gen_legal(buf) && !can_partner_provide_blocking_piece(1, side ^ 1, adv)) return true;
        }
        bool onA = pos[0].stm == side, onB = pos[1].stm == (side ^ 1);
        if (onA || onB) {
            bool movesA = onA && pos[0].gen_legal(buf) > 0;
            bool movesB = onB && pos[1].gen_legal(buf) > 0;
            if (!movesA && !movesB && (!adv || (onA && onB))) return true;
        }
        return false;
    }

    void to_compact(hm_board* o, int team, bool adv) const {
        std::memset(o, 0, sizeof *o);
        for (int b = 0; b < 2; ++b) {
            pos[b].to_compact(&o->pos[b]);
            o->last_move[b] = last_move(b);
            o->rep_count[b] = (uint8_t)std::min(repetition_count(b), 3);
        }
        o->team = (uint8_t)team; o->time_adv = adv ? 1 : 0;
    }
};

// perft (tools/benchmark.cc:59-76)
inline uint64_t perft(Board& board, int depth) {
    if (depth == 0) return 1;
    auto a = board.legal_moves(0), b = board.legal_moves(1);
    if (depth == 1) return (uint64_t)a.size() * b.size();
    uint64_t nodes = 0;
    for (Move ma : a)
        for (Move mb : b) {
            board.make_moves(ma, mb);
            nodes += perft(board, depth - 1);
            board.unmake_moves(ma, mb);
        }
    return nodes;
}

// Same count, without the (redundant for perft) legality re-check of make_moves: used as
// the timed CPU "port" baseline so the port is not handicapped by board.cc:317.
inline uint64_t perft_fast(const Pos& A, const Pos& B, int depth) {
    Move ma[1024], mb[1024];
    int na = A.gen_legal(ma), nb = B.gen_legal(mb);
    if (depth == 1) return (uint64_t)na * nb;
    uint64_t nodes = 0;
    for (int i = 0; i < na; ++i) {
        Pos a2 = A;
        int ha = a2.do_move(ma[i]);
        for (int j = 0; j < nb; ++j) {
            Pos a3 = a2, b2 = B;
            if (ha) b2.add_to_hand(Pos::pc_color(ha), Pos::pc_type(ha));
            int hb = b2.do_move(mb[j]);
            if (hb) a3.add_to_hand(Pos::pc_color(hb), Pos::pc_type(hb));
            nodes += perft_fast(a3, b2, depth - 1);
        }
    }
    return nodes;
}

// ---------------------------------------------------------------------------
// board_to_planes (environment/planes.cc:96-265) from the compact state
// ---------------------------------------------------------------------------
inline BB flip_vertical(BB b) { return __builtin_bswap64(b); }

// Emits the 74 planes as (mask, value) pairs: element s of plane p = mask bit s ? value : 0.
// Every plane of the reference encoder has that shape (bitboard planes: value 1; constant planes:
// mask = all ones).
struct PlaneDesc { BB mask; float value; };
inline void plane_descs(const hm_board& bd, PlaneDesc out[HM_NB_PLANES]) {
    const int team = bd.team;
    int p = 0;
    for (int b = 0; b < 2; ++b) {
        const hm_pos& q = bd.pos[b];
        const bool flip = (b == 0) ? team == BLACK : team == WHITE;            // planes.cc:90-93
        const int first = (b == 0) ? team : team ^ 1, second = first ^ 1;      // planes.cc:99
        auto orient = [&](BB x) { return flip ? flip_vertical(x) : x; };
        for (int c : {first, second})
            for (int pt = 0; pt < 6; ++pt) out[p++] = {orient(q.by_type[pt] & q.by_color[c]), 1.0f};
        for (int c : {first, second})
            for (int pt = 0; pt < 5; ++pt) out[p++] = {~0ULL, (float)q.hand[c][pt] / 16.0f};
        for (int c : {first, second}) out[p++] = {orient(q.promoted & q.by_color[c]), 1.0f};
        out[p++] = {orient(q.ep < 64 ? sq_bb(q.ep) : 0), 1.0f};
        out[p++] = {~0ULL, q.stm == first ? 1.0f : 0.0f};                       // planes.cc:153-159
        out[p++] = {~0ULL, 1.0f};
        // castling (planes.cc:161-176): "own" colour = first
        const int oo[2] = {WHITE_OO, BLACK_OO}, ooo[2] = {WHITE_OOO, BLACK_OOO};
        out[p++] = {~0ULL, (q.castling & oo[first]) ? 1.0f : 0.0f};
        out[p++] = {~0ULL, (q.castling & ooo[first]) ? 1.0f : 0.0f};
        out[p++] = {~0ULL, (q.castling & oo[second]) ? 1.0f : 0.0f};
        out[p++] = {~0ULL, (q.castling & ooo[second]) ? 1.0f : 0.0f};
        out[p++] = {~0ULL, bd.time_adv ? 1.0f : 0.0f};
        // last move (planes.cc:182-198)
        Move lm = bd.last_move[b];
        BB fromBB = 0, toBB = 0;
        if (lm != MOVE_NONE) {
            if (type_of(lm) != DROP) fromBB = sq_bb(flip ? from_sq(lm) ^ 56 : from_sq(lm));
            toBB = sq_bb(flip ? to_sq(lm) ^ 56 : to_sq(lm));
        }
        out[p++] = {fromBB, 1.0f};
        out[p++] = {toBB, 1.0f};
        out[p++] = {~0ULL, (float)std::min<int>(q.rule50, 50) / 50.0f};        // planes.cc:202-205
        out[p++] = {~0ULL, bd.rep_count[b] >= 2 ? 1.0f : 0.0f};
        out[p++] = {~0ULL, bd.rep_count[b] >= 3 ? 1.0f : 0.0f};
    }
}

inline void planes_f32(const hm_board& bd, float* out) {
    PlaneDesc d[HM_NB_PLANES];
    plane_descs(bd, d);
    for (int p = 0; p < HM_NB_PLANES; ++p)
        for (int s = 0; s < 64; ++s) out[p * 64 + s] = ((d[p].mask >> s) & 1) ? d[p].value : 0.0f;
}

// IEEE binary16 round-to-nearest-even of a float (== __float2half_rn, planes.cc:22-27)
inline uint16_t f32_to_f16_rn(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        int shift = 14 - e;
        uint32_t h = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = (uint32_t)(e << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(sign | h);
}
inline void planes_f16(const hm_board& bd, uint16_t* out) {
    PlaneDesc d[HM_NB_PLANES];
    plane_descs(bd, d);
    for (int p = 0; p < HM_NB_PLANES; ++p) {
        uint16_t v = f32_to_f16_rn(d[p].value);
        for (int s = 0; s < 64; ++s) out[p * 64 + s] = ((d[p].mask >> s) & 1) ? v : 0;
    }
}
// selfplay.cc:464-476 encode_planes: u8 = clamp(lround(v*255), 0, 255)
inline void planes_u8(const hm_board& bd, uint8_t* out) {
    PlaneDesc d[HM_NB_PLANES];
    plane_descs(bd, d);
    for (int p = 0; p < HM_NB_PLANES; ++p) {
        long q = std::lround(d[p].value * 255.0f);
        uint8_t v = (uint8_t)std::min<long>(255, std::max<long>(0, q));
        for (int s = 0; s < 64; ++s) out[p * 64 + s] = ((d[p].mask >> s) & 1) ? v : 0;
    }
}

}  // namespace hmo
