// hm_host.hpp — host-side table construction for the device library.
// Replaces the reference start-up sequence (main.cc:75-81): Bitboards::init
// (Fairy-Stockfish/src/bitboard.cpp:297-358; no magics needed here), Position::init
// (position.cpp:142-167, Zobrist stream order), environment/zobrist.cc:16-28 and
// init_policy_index (common/globals.cc:50-104).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "hm_device.hpp"

struct PlaneConsts {
    // value of count/16 and min(r50,50)/50 in each output dtype, computed on the host with the
    // reference's float expressions (planes.cc:118, 202-205; selfplay.cc:464-476 for u8)
    uint32_t pocket[3][64];
    uint32_t r50[3][64];
    uint32_t one[3];
};

struct HostTables {
    hmd::DeviceTables dev;
    PlaneConsts plane_consts;
    uint64_t in_hand_const = 0;   // XOR of inHand[pc][0] over piece types that never hold a count
    bool built = false;
};

inline uint16_t hm_f32_to_f16_rn(float f) {   // IEEE binary16, round-to-nearest-even
    uint32_t x; std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ex = (x >> 23) & 0xff;
    uint32_t m = x & 0x7fffffu;
    if (ex == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0));
    int e = (int)ex - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        int shift = 14 - e;
        uint32_t h = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (m >> 13), rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(sign | h);
}

inline void build_host_tables(HostTables& h) {
    using namespace hmd;
    if (h.built) return;
    std::memset(&h.dev, 0, sizeof h.dev);
    auto on = [](int f, int r) { return f >= 0 && f < 8 && r >= 0 && r < 8; };
    static const int kn[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
    for (int s = 0; s < 64; ++s) {
        const int f = s & 7, r = s >> 3;
        for (auto& d : kn) if (on(f + d[0], r + d[1])) h.dev.att.knight[s] |= 1ULL << ((r + d[1]) * 8 + f + d[0]);
        for (int df = -1; df <= 1; ++df)
            for (int dr = -1; dr <= 1; ++dr)
                if ((df || dr) && on(f + df, r + dr)) h.dev.att.king[s] |= 1ULL << ((r + dr) * 8 + f + df);
        for (int df = -1; df <= 1; df += 2) {
            if (on(f + df, r + 1)) h.dev.att.pawn[0][s] |= 1ULL << ((r + 1) * 8 + f + df);
            if (on(f + df, r - 1)) h.dev.att.pawn[1][s] |= 1ULL << ((r - 1) * 8 + f + df);
        }
        for (int t = 0; t < 64; ++t) {
            if (t == s) continue;
            const int tf = t & 7, tr = t >> 3;
            if (tf - f == tr - r) h.dev.att.diag[s] |= 1ULL << t;
            if (tf - f == -(tr - r)) h.dev.att.anti[s] |= 1ULL << t;
        }
    }
    // Zobrist: xorshift64* stream, seed 1070372, drawn in the reference's loop order
    // (psq for piece types 1..63 although only P..Q and KING=63 exist).
    uint64_t st = 1070372;
    auto rnd = [&]() { st ^= st >> 12; st ^= st << 25; st ^= st >> 27; return st * 2685821657736338717ULL; };
    for (int c = 0; c < 2; ++c)
        for (int pt = 1; pt <= 63; ++pt)
            for (int s = 0; s < 64; ++s) {
                uint64_t v = rnd();
                if (pt <= 5) h.dev.zob.psq[c][pt - 1][s] = v;
                else if (pt == 63) h.dev.zob.psq[c][5][s] = v;
            }
    for (int f = 0; f < 8; ++f) h.dev.zob.ep[f] = rnd();
    for (int cr = 0; cr < 16; ++cr) h.dev.zob.castle[cr] = rnd();
    h.dev.zob.side = rnd();
    (void)rnd();                                   // noPawns
    for (int i = 0; i < 22; ++i) (void)rnd();      // checks[2][11]
    h.in_hand_const = 0;
    for (int c = 0; c < 2; ++c)
        for (int pt = 1; pt <= 63; ++pt)
            for (int n = 0; n < 64; ++n) {
                uint64_t v = rnd();
                if (pt <= 5) { if (n < 32) h.dev.zob.in_hand[c][pt - 1][n] = v; }
                else if (n == 0) h.in_hand_const ^= v;
            }
    {
        std::mt19937_64 mt(1070372);               // environment/zobrist.cc
        for (int i = 0; i < 1024; ++i) (void)mt();
        h.dev.zob.time_adv = mt();
    }
    // policy labels (constants.h:24) built by construction; first occurrence wins
    {
        auto sqn = [](int s) { std::string r; r += char('a' + (s & 7)); r += char('1' + (s >> 3)); return r; };
        std::vector<std::string> lab(HM_POLICY_VALUES, "illegal");
        for (int s = 0; s < 64; ++s) lab[s] = "pass";
        const char dropc[5] = {'P', 'N', 'R', 'B', 'Q'};
        for (int k = 0; k < 5; ++k)
            for (int s = 0; s < 64; ++s) {
                if (k == 0 && ((s >> 3) == 0 || (s >> 3) == 7)) continue;
                lab[(1 + k) * 64 + s] = std::string(1, dropc[k]) + "@" + sqn(s);
            }
        static const int qd[8][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}};
        for (int d = 0; d < 8; ++d)
            for (int dist = 1; dist <= 7; ++dist)
                for (int s = 0; s < 64; ++s) {
                    int f = (s & 7) + qd[d][0] * dist, r = (s >> 3) + qd[d][1] * dist;
                    if (on(f, r)) lab[(6 + d * 7 + dist - 1) * 64 + s] = sqn(s) + sqn(r * 8 + f);
                }
        for (int k = 0; k < 8; ++k)
            for (int s = 0; s < 64; ++s) {
                int f = (s & 7) + kn[k][0], r = (s >> 3) + kn[k][1];
                if (on(f, r)) lab[(62 + k) * 64 + s] = sqn(s) + sqn(r * 8 + f);
            }
        for (int k = 0; k < 3; ++k)
            for (int s = 48; s < 56; ++s) {
                int f = (s & 7) + (k - 1);
                if (on(f, 7)) lab[(70 + k) * 64 + s] = sqn(s) + sqn(56 + f) + "n";
            }
        std::map<std::string, int> idx;
        for (int i = 0; i < (int)lab.size(); ++i) idx.emplace(lab[i], i);   // emplace keeps the first
        auto mirror = [](std::string u) {
            if (u == "pass") return u;
            if (u.size() >= 4 && u[1] == '@') { u[3] = char('0' + (9 - (u[3] - '0'))); return u; }
            if (u.size() >= 4) { u[1] = char('0' + (9 - (u[1] - '0'))); u[3] = char('0' + (9 - (u[3] - '0'))); }
            return u;
        };
        auto find = [&](const std::string& s) { auto it = idx.find(s); return it == idx.end() ? -1 : it->second; };
        const char ptc[8] = {' ', 'P', 'N', 'B', 'R', 'Q', 'K', ' '};
        for (int c = 0; c < 2; ++c) {
            for (int f = 0; f < 64; ++f)
                for (int t = 0; t < 64; ++t) {
                    std::string u = sqn(f) + sqn(t);
                    h.dev.pol_normal[c][f][t][0] = find(c ? mirror(u) : u);
                    h.dev.pol_normal[c][f][t][1] = find(c ? mirror(u + "n") : u + "n");
                }
            for (int t = 0; t < 64; ++t)
                for (int pt = 0; pt < 8; ++pt) {
                    h.dev.pol_drop[c][t][pt] = -1;
                    if (pt >= 1 && pt <= 5) {
                        std::string u = std::string(1, ptc[pt]) + "@" + sqn(t);
                        h.dev.pol_drop[c][t][pt] = find(c ? mirror(u) : u);
                    }
                }
        }
    }
    // plane scalar values per dtype
    for (int k = 0; k < 64; ++k) {
        float pv = (float)k / 16.0f;
        float rv = (float)(k > 50 ? 50 : k) / 50.0f;
        uint32_t pb, rb; std::memcpy(&pb, &pv, 4); std::memcpy(&rb, &rv, 4);
        h.plane_consts.pocket[HM_DT_F16][k] = hm_f32_to_f16_rn(pv);
        h.plane_consts.r50[HM_DT_F16][k] = hm_f32_to_f16_rn(rv);
        h.plane_consts.pocket[HM_DT_F32][k] = pb;
        h.plane_consts.r50[HM_DT_F32][k] = rb;
        auto q = [](float v) { long r = std::lround(v * 255.0f); return (uint32_t)(r < 0 ? 0 : r > 255 ? 255 : r); };
        h.plane_consts.pocket[HM_DT_U8][k] = q(pv);
        h.plane_consts.r50[HM_DT_U8][k] = q(rv);
    }
    {
        float one = 1.0f; uint32_t ob; std::memcpy(&ob, &one, 4);
        h.plane_consts.one[HM_DT_F16] = 0x3C00u;
        h.plane_consts.one[HM_DT_F32] = ob;
        h.plane_consts.one[HM_DT_U8] = 255u;
    }
    h.built = true;
}

// Dual start position (Board::Board(), board.cc:52-69) as compact state incl. Zobrist key
// (set_state, position.cpp:560-607).
inline uint64_t host_compute_key(const HostTables& h, const hm_pos& p) {
    uint64_t k = 0;
    for (int pt = 0; pt < 6; ++pt)
        for (uint64_t b = p.by_type[pt]; b; b &= b - 1) {
            int s = __builtin_ctzll(b);
            int c = (p.by_color[1] >> s) & 1;
            k ^= h.dev.zob.psq[c][pt][s];
        }
    if (p.ep < 64) k ^= h.dev.zob.ep[p.ep & 7];
    if (p.stm) k ^= h.dev.zob.side;
    k ^= h.dev.zob.castle[p.castling & 15];
    for (int c = 0; c < 2; ++c)
        for (int pt = 0; pt < 5; ++pt) k ^= h.dev.zob.in_hand[c][pt][p.hand[c][pt] & 31];
    return k ^ h.in_hand_const;
}

inline void startpos(const HostTables& h, hm_board* out) {
    std::memset(out, 0, sizeof *out);
    hm_pos p;
    std::memset(&p, 0, sizeof p);
    p.by_type[0] = 0x00FF00000000FF00ULL;
    p.by_type[1] = 0x4200000000000042ULL;
    p.by_type[2] = 0x2400000000000024ULL;
    p.by_type[3] = 0x8100000000000081ULL;
    p.by_type[4] = 0x0800000000000008ULL;
    p.by_type[5] = 0x1000000000000010ULL;
    p.by_color[0] = 0x000000000000FFFFULL;
    p.by_color[1] = 0xFFFF000000000000ULL;
    p.castling = 15; p.ep = 64; p.stm = 0; p.rule50 = 0; p.game_ply = 0;
    p.key = host_compute_key(h, p);
    out->pos[0] = p; out->pos[1] = p;
    out->rep_count[0] = out->rep_count[1] = 1;
}

inline int host_policy_index(const HostTables& h, uint32_t m, int stm) {   // utils.h:184-216
    if (m == 0) return 0;
    const uint32_t mt = m & (15u << 12);
    const int to = m & 63, from = (m >> 6) & 63, pt = (m >> 16) & 63;
    if (mt == HM_MT_DROP) return (pt >= 1 && pt <= 5) ? h.dev.pol_drop[stm][to][pt] : -1;
    if (mt == HM_MT_PROMOTION) {
        if (pt == HM_KNIGHT) return h.dev.pol_normal[stm][from][to][1];
        if (pt == HM_QUEEN) return h.dev.pol_normal[stm][from][to][0];
        return -1;
    }
    return h.dev.pol_normal[stm][from][to][0];
}
