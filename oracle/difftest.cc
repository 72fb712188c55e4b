// oracle/difftest.cc — differential fuzzer: oracle restatement vs the reference build.
// TEST INFRASTRUCTURE ONLY (runs in the build container, where oracle/_ref exists).
//
// Plays seeded random two-board games, and at every ply compares, between
// oracle/bughouse.hpp and oracle/_ref/libhmref.so (the reference's own sources):
//   legal move LISTS (order included) on both boards, compact state (bitboards, hands,
//   castling, ep, rule50, ply, Zobrist key bit-for-bit), last move, repetition count,
//   is_checkmate(side, adv) for all 4 combinations, is_draw(0/1), gives_check/is_capture
//   per move, hash_key equivalence classes, and push/pop + make/unmake symmetry.
// usage: difftest [games] [seed] [max_plies]
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>

#include "bughouse.hpp"

extern "C" {
void ref_init();
void* ref_board_new();
void ref_board_free(void*);
int ref_legal_moves(void*, int, uint32_t*);
void ref_push(void*, int, uint32_t);
void ref_pop(void*, int);
int ref_make_moves(void*, uint32_t, uint32_t);
void ref_unmake_moves(void*, uint32_t, uint32_t);
int ref_is_checkmate(void*, int, int);
int ref_is_draw(void*, int);
uint64_t ref_hash_key(void*, int);
uint64_t ref_board_only_key(void*, int);
uint64_t ref_pos_key(void*, int);
int ref_gives_check(void*, int, uint32_t);
int ref_is_capture(void*, int, uint32_t);
void ref_compact(void*, int, int, hm_board*);
long long ref_perft(void*, int);
uint64_t ref_time_advantage_key();
void ref_policy_tables(int*, int*);
}

static long long g_checks = 0;
#define CHECK(cond, ...)                                                  \
    do {                                                                  \
        ++g_checks;                                                       \
        if (!(cond)) {                                                    \
            std::fprintf(stderr, "MISMATCH %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);                            \
            std::fprintf(stderr, "\n");                                   \
            std::exit(1);                                                 \
        }                                                                 \
    } while (0)

int main(int argc, char** argv) {
    int games = argc > 1 ? std::atoi(argv[1]) : 200;
    uint64_t seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 42;
    int maxPlies = argc > 3 ? std::atoi(argv[3]) : 160;
    ref_init();
    CHECK(ref_time_advantage_key() == hmo::T().timeAdvantage, "timeAdvantage key");
    {
        static int rn[2 * 64 * 64 * 2], rd[2 * 64 * 8];
        ref_policy_tables(rn, rd);
        CHECK(!std::memcmp(rn, hmo::T().polNormal, sizeof rn), "POLICY_TABLE_NORMAL");
        CHECK(!std::memcmp(rd, hmo::T().polDrop, sizeof rd), "POLICY_TABLE_DROP");
    }
    std::mt19937_64 rng(seed);
    std::map<uint64_t, uint64_t> ref2ora, ora2ref;      // hash_key class bijection
    std::map<uint64_t, uint64_t> rk2ora, ora2rk;        // board_only_key class bijection
    long long plies = 0, maxMoves = 0, mates = 0, draws = 0;
    for (int g = 0; g < games; ++g) {
        void* R = ref_board_new();
        hmo::Board O;
        for (int ply = 0; ply < maxPlies; ++ply) {
            uint32_t rm[2][1024];
            int rn[2];
            std::vector<hmo::Move> om[2];
            for (int b = 0; b < 2; ++b) {
                rn[b] = ref_legal_moves(R, b, rm[b]);
                om[b] = O.legal_moves(b);
                CHECK((int)om[b].size() == rn[b], "game %d ply %d board %d: move count %zu vs %d", g, ply, b, om[b].size(), rn[b]);
                for (int i = 0; i < rn[b]; ++i)
                    CHECK(om[b][i] == rm[b][i], "game %d ply %d board %d: move[%d] %x vs %x", g, ply, b, i, om[b][i], rm[b][i]);
                if (rn[b] > maxMoves) maxMoves = rn[b];
                for (int i = 0; i < rn[b]; ++i) {
                    CHECK(O.pos[b].gives_check(om[b][i]) == (ref_gives_check(R, b, rm[b][i]) != 0), "gives_check %x", rm[b][i]);
                    CHECK(O.pos[b].is_capture(om[b][i]) == (ref_is_capture(R, b, rm[b][i]) != 0), "is_capture %x", rm[b][i]);
                }
            }
            int team = (int)(rng() & 1), adv = (int)(rng() & 1);
            hm_board rc, oc;
            ref_compact(R, team, adv, &rc);
            O.to_compact(&oc, team, adv);
            CHECK(!std::memcmp(&rc, &oc, sizeof rc), "game %d ply %d: compact state differs", g, ply);
            for (int b = 0; b < 2; ++b) {
                CHECK(ref_pos_key(R, b) == O.pos[b].pos_key(), "pos key()");
                uint64_t rk = ref_board_only_key(R, b), ok = O.pos[b].rep_key();
                auto it = rk2ora.find(rk);
                if (it == rk2ora.end()) rk2ora[rk] = ok; else CHECK(it->second == ok, "rep key class split (ref same, oracle differs)");
                auto it2 = ora2rk.find(ok);
                if (it2 == ora2rk.end()) ora2rk[ok] = rk; else CHECK(it2->second == rk, "rep key class merge (oracle same, ref differs)");
            }
            for (int a = 0; a < 2; ++a) {
                uint64_t rh = ref_hash_key(R, a), oh = O.hash_key(a);
                auto it = ref2ora.find(rh);
                if (it == ref2ora.end()) ref2ora[rh] = oh; else CHECK(it->second == oh, "hash_key class split");
                auto it2 = ora2ref.find(oh);
                if (it2 == ora2ref.end()) ora2ref[oh] = rh; else CHECK(it2->second == rh, "hash_key class merge");
                for (int side = 0; side < 2; ++side) {
                    int rmate = ref_is_checkmate(R, side, a);
                    int omate = O.is_checkmate(side, a);
                    CHECK(rmate == omate, "game %d ply %d: is_checkmate(side=%d, adv=%d) %d vs %d", g, ply, side, a, omate, rmate);
                    mates += rmate;
                }
            }
            for (int p = 0; p < 2; ++p) {
                int rd = ref_is_draw(R, p), od = O.is_draw(p);
                CHECK(rd == od, "is_draw(%d)", p);
                draws += rd;
            }
            // make/unmake symmetry on a random joint action
            if (rn[0] && rn[1]) {
                uint32_t a = rm[0][rng() % rn[0]], b = rm[1][rng() % rn[1]];
                hm_board before; O.to_compact(&before, 0, 0);
                uint64_t hb = O.hash_key(false);
                CHECK(ref_make_moves(R, a, b) == 0, "ref make_moves");
                O.make_moves(a, b);
                hm_board r2, o2; ref_compact(R, 0, 0, &r2); O.to_compact(&o2, 0, 0);
                CHECK(!std::memcmp(&r2, &o2, sizeof r2), "state after make_moves");
                ref_unmake_moves(R, a, b); O.unmake_moves(a, b);
                hm_board after; O.to_compact(&after, 0, 0);
                CHECK(!std::memcmp(&before, &after, sizeof after) && hb == O.hash_key(false), "unmake_moves restore");
            }
            // advance: random board, random legal move (SURVEY §8d generator)
            int b = (int)(rng() & 1);
            if (!rn[b]) b ^= 1;
            if (!rn[b]) break;
            // bias toward captures/drops a little so pockets fill up
            uint32_t mv = rm[b][rng() % rn[b]];
            for (int tries = 0; tries < 2; ++tries) {
                uint32_t alt = rm[b][rng() % rn[b]];
                if (ref_is_capture(R, b, alt)) { mv = alt; break; }
            }
            ref_push(R, b, mv);
            O.push_move(b, mv);
            ++plies;
        }
        ref_board_free(R);
    }
    // perft 1..3 (benchmark.cc:59-76)
    {
        void* R = ref_board_new();
        hmo::Board O;
        for (int d = 1; d <= 3; ++d) {
            long long r = ref_perft(R, d);
            uint64_t o = hmo::perft(O, d);
            uint64_t f = hmo::perft_fast(O.pos[0], O.pos[1], d);
            CHECK((uint64_t)r == o && o == f, "perft(%d): ref %lld oracle %llu fast %llu", d, r, (unsigned long long)o, (unsigned long long)f);
            std::printf("perft(%d) = %lld\n", d, r);
        }
        ref_board_free(R);
    }
    std::printf("OK games=%d plies=%lld checks=%lld max_moves=%lld mate_flags=%lld draw_flags=%lld hash_classes=%zu rep_classes=%zu\n",
                games, plies, g_checks, maxMoves, mates, draws, ref2ora.size(), rk2ora.size());
    return 0;
}
