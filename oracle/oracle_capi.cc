// oracle/oracle_capi.cc — C entry points of the CPU oracle for ctypes (tests, smoke,
// bench cpu_baseline).  TEST INFRASTRUCTURE ONLY: nothing under hivemind_amd/ links this.
#include <chrono>
#include <cstring>
#include <string>

#include "bughouse.hpp"
#if __has_include("search.hpp")
#include "search.hpp"
#endif

using namespace hmo;

static std::string uci_of(const Pos& p, Move m) {   // Fairy-Stockfish stubs.cpp:20-58
    if (m == MOVE_NONE) return "pass";              // Board::uci_move (board.h:340-343)
    auto sq = [](int s) { std::string r; r += char('a' + (s & 7)); r += char('1' + (s >> 3)); return r; };
    int from = from_sq(m), to = to_sq(m);
    if (type_of(m) == CASTLING) to = (from & 56) + (to > from ? 6 : 2);
    std::string s;
    if (type_of(m) == DROP) { s += " PNBRQK"[promo_type(m)]; s += '@'; }
    else s += sq(from);
    s += sq(to);
    if (type_of(m) == PROMOTION) s += " pnbrqk"[promo_type(m)];
    (void)p;
    return s;
}

extern "C" {

void* ora_board_new() { return new Board(); }
void* ora_board_clone(void* h) { return new Board(*static_cast<Board*>(h)); }
void ora_board_free(void* h) { delete static_cast<Board*>(h); }
void ora_board_set(void* h, const char* fen) { static_cast<Board*>(h)->set(fen); }
int ora_fen(void* h, int b, char* buf, int cap) {
    const std::string s = static_cast<Board*>(h)->pos[b].fen();
    if ((int)s.size() + 1 > cap) return -1;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}
void ora_board_set_fen(void* h, int b, const char* fen) { static_cast<Board*>(h)->set_fen(b, fen); }
// positions from a compact state; history restarts here (like Board::set_fen)
void ora_board_from_compact(void* h, const hm_board* c) {
    Board& bd = *static_cast<Board*>(h);
    for (int b = 0; b < 2; ++b) {
        bd.pos[b].from_compact(&c->pos[b]);
        bd.states[b].clear();
        bd.positionHistory[b].clear(); bd.positionHistoryPrefixes[b].clear(); bd.moveHistory[b].clear();
        bd.record_position(b);
    }
}
int ora_legal_moves(void* h, int b, uint32_t* out) {
    return static_cast<Board*>(h)->pos[b].gen_legal(out);
}
void ora_push(void* h, int b, uint32_t m) { static_cast<Board*>(h)->push_move(b, m); }
void ora_pop(void* h, int b) { static_cast<Board*>(h)->pop_move(b); }
int ora_make_moves(void* h, uint32_t a, uint32_t b) {
    try { static_cast<Board*>(h)->make_moves(a, b); } catch (const std::logic_error&) { return -1; }
    return 0;
}
void ora_unmake_moves(void* h, uint32_t a, uint32_t b) { static_cast<Board*>(h)->unmake_moves(a, b); }
int ora_is_checkmate(void* h, int side, int adv) { return static_cast<Board*>(h)->is_checkmate(side, adv != 0); }
int ora_is_draw(void* h, int ply) { return static_cast<Board*>(h)->is_draw(ply); }
uint64_t ora_hash_key(void* h, int adv) { return static_cast<Board*>(h)->hash_key(adv != 0); }
uint64_t ora_rep_key(void* h, int b) { return static_cast<Board*>(h)->pos[b].rep_key(); }
uint64_t ora_pos_key(void* h, int b) { return static_cast<Board*>(h)->pos[b].pos_key(); }
int ora_repetition_count(void* h, int b) { return static_cast<Board*>(h)->repetition_count(b); }
int ora_in_check(void* h, int b) { return static_cast<Board*>(h)->pos[b].checkers != 0; }
int ora_gives_check(void* h, int b, uint32_t m) { return m != 0 && static_cast<Board*>(h)->pos[b].gives_check(m); }
int ora_is_capture(void* h, int b, uint32_t m) { return static_cast<Board*>(h)->pos[b].is_capture(m); }
void ora_compact(void* h, int team, int adv, hm_board* o) { static_cast<Board*>(h)->to_compact(o, team, adv != 0); }
int ora_uci(void* h, int b, uint32_t m, char* buf, int cap) {
    std::string s = uci_of(static_cast<Board*>(h)->pos[b], m);
    std::strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0;
    return (int)s.size();
}
uint64_t ora_perft(void* h, int depth) { return perft(*static_cast<Board*>(h), depth); }
uint64_t ora_perft_fast(void* h, int depth) {
    Board& b = *static_cast<Board*>(h);
    return perft_fast(b.pos[0], b.pos[1], depth);
}
uint64_t ora_perft_single(void* h, int b, int depth) {
    struct R { static uint64_t go(const Pos& p, int d) {
        Move mv[1024]; int n = p.gen_legal(mv);
        if (d == 1) return n;
        uint64_t t = 0;
        for (int i = 0; i < n; ++i) { Pos c = p; c.do_move(mv[i]); t += go(c, d - 1); }
        return t; } };
    return R::go(static_cast<Board*>(h)->pos[b], depth);
}

// compact-state entry points (no handle): what the GPU kernels are diffed against
int ora_legal_moves_pos(const hm_pos* p, uint32_t* out) { Pos q; q.from_compact(p); return q.gen_legal(out); }
void ora_planes(const hm_board* boards, size_t n, int dtype, void* out) {
    for (size_t i = 0; i < n; ++i) {
        if (dtype == HM_DT_F16) planes_f16(boards[i], static_cast<uint16_t*>(out) + i * HM_PLANE_VALUES);
        else if (dtype == HM_DT_F32) planes_f32(boards[i], static_cast<float*>(out) + i * HM_PLANE_VALUES);
        else planes_u8(boards[i], static_cast<uint8_t*>(out) + i * HM_PLANE_VALUES);
    }
}
// Board::make_moves on compact states (history scalars: last_move updated, rep_count left to caller)
void ora_make_moves_compact(const hm_board* in, uint32_t a, uint32_t b, hm_board* out) {
    Pos A, B; A.from_compact(&in->pos[0]); B.from_compact(&in->pos[1]);
    *out = *in;
    if (a) { int h = A.do_move(a); if (h) B.add_to_hand(Pos::pc_color(h), Pos::pc_type(h)); out->last_move[0] = a; }
    if (b) { int h = B.do_move(b); if (h) A.add_to_hand(Pos::pc_color(h), Pos::pc_type(h)); out->last_move[1] = b; }
    A.to_compact(&out->pos[0]); B.to_compact(&out->pos[1]);
}
int ora_policy_index(uint32_t m, int stm) { return policy_index(m, stm); }
void ora_policy_tables(int* normal, int* drop) {
    std::memcpy(normal, T().polNormal, sizeof(T().polNormal));
    std::memcpy(drop, T().polDrop, sizeof(T().polDrop));
}
int ora_policy_label(int idx, char* buf, int cap) {
    const std::string& s = T().policy_labels[idx];
    std::strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0;
    return (int)s.size();
}

// SURVEY §8(d) synthetic workload: seeded random playouts (uniform board, uniform legal move,
// <= maxPlies per game, restart on dead end); team/adv random bits.  Counter-based splitmix64
// RNG so the device generator (hivemind_amd/csrc) can be checked against it draw for draw.
static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
size_t ora_random_positions(uint64_t seed, size_t n, int maxPlies, hm_board* out) {
    size_t produced = 0;
    uint64_t game = 0;
    while (produced < n) {
        Board bd;
        uint64_t ctr = 0;
        auto rnd = [&]() { return splitmix64(seed ^ splitmix64(game * 0x10001ULL + (ctr++))); };
        for (int ply = 0; ply < maxPlies && produced < n; ++ply) {
            uint64_t r = rnd();
            bd.to_compact(&out[produced++], (int)(r & 1), (int)((r >> 1) & 1));
            int b = (int)((r >> 2) & 1);
            Move mv[1024];
            int nm = bd.pos[b].gen_legal(mv);
            if (!nm) { b ^= 1; nm = bd.pos[b].gen_legal(mv); }
            if (!nm) break;
            bd.push_move(b, mv[(r >> 8) % (uint64_t)nm]);
        }
        ++game;
    }
    return produced;
}

// ---- search (oracle/search.hpp) ------------------------------------------------------------
typedef void (*ora_eval_cb)(const uint16_t* planes, int n, uint16_t* value, uint16_t* piA, uint16_t* piB,
                            uint16_t* wdl, uint16_t* ml);
void* ora_search_new(int tie_mode, int exp_mode) {
    Search* s = new Search();
    s->cfg.tie_mode = tie_mode; s->cfg.exp_mode = exp_mode;
    s->evaluator = hash_evaluator;
    return s;
}
void ora_search_free(void* s) { delete static_cast<Search*>(s); }
void ora_search_set_callback(void* sp, ora_eval_cb cb) {
    Search* s = static_cast<Search*>(sp);
    if (!cb) { s->evaluator = hash_evaluator; return; }
    s->evaluator = [cb](const uint16_t* planes, int n, EvalOutputs& out) {
        out.value.assign(n, 0); out.piA.assign((size_t)n * HM_POLICY_VALUES, 0); out.piB.assign((size_t)n * HM_POLICY_VALUES, 0);
        out.wdl.assign((size_t)n * 3, 0); out.movesLeft.assign(n, 0);
        cb(planes, n, out.value.data(), out.piA.data(), out.piB.data(), out.wdl.data(), out.movesLeft.data());
    };
}
void ora_search_set_noise(void* sp, float alpha, float eps, uint64_t seed) {
    Search* s = static_cast<Search*>(sp);
    s->cfg.rootDirichletAlpha = alpha; s->cfg.rootDirichletEpsilon = eps; s->cfg.rootNoiseSeed = seed;
}
void ora_search_set_batch_size(void* sp, int b) { static_cast<Search*>(sp)->cfg.batchSize = b; }   // Engine::getBatchSize() (searchthread.cc:663)
void ora_search_set_transpositions(void* sp, int on) { static_cast<Search*>(sp)->cfg.enableTranspositions = on != 0; }
int ora_search_run(void* sp, void* board, int team, int adv, int targetNodes) {
    Search* s = static_cast<Search*>(sp);
    s->evalTrace.clear(); s->sameBatchCollisions = s->reservationCollisions = s->evalCalls = s->evalRows = 0;
    return s->run(*static_cast<Board*>(board), team, adv != 0, targetNodes) ? 1 : 0;
}
int ora_search_edges(void* sp, uint32_t* moveA, uint32_t* moveB, int* visits, float* q, float* prior, int cap) {
    auto e = static_cast<Search*>(sp)->root_edge_stats();
    int n = (int)std::min<size_t>(e.size(), (size_t)cap);
    for (int i = 0; i < n; ++i) { moveA[i] = e[i].moveA; moveB[i] = e[i].moveB; visits[i] = e[i].visits; q[i] = e[i].q; prior[i] = e[i].prior; }
    return (int)e.size();
}
float ora_search_root_q(void* sp) { return static_cast<Search*>(sp)->root_q(); }
int ora_search_best_move(void* sp) { return static_cast<Search*>(sp)->best_move_index(); }
// pv_lines: out_idx / out_type / out_end / out_len [multi_pv], out_q [multi_pv], out_moves [multi_pv][max_depth][2]; returns the line count
int ora_search_pv_lines(void* sp, int multi_pv, int max_depth, int* out_idx, int* out_type, int* out_end, int* out_len, float* out_q, uint32_t* out_moves) {
    const auto lines = static_cast<Search*>(sp)->pv_lines(multi_pv, max_depth);
    for (size_t l = 0; l < lines.size(); ++l) {
        out_idx[l] = lines[l].childIdx; out_type[l] = lines[l].childType; out_end[l] = lines[l].childEndInPly; out_q[l] = lines[l].q;
        out_len[l] = (int)lines[l].moves.size() / 2;
        for (size_t i = 0; i < lines[l].moves.size(); ++i) out_moves[l * (size_t)max_depth * 2 + i] = lines[l].moves[i];
    }
    return (int)lines.size();
}
void ora_search_set_tree_reuse(void* sp, int on) { Search* s = static_cast<Search*>(sp); s->enableTreeReuse = on != 0; if (!on) s->reset_search_state(); }
void ora_search_reset(void* sp) { static_cast<Search*>(sp)->reset_search_state(); }
int ora_search_reused_visits(void* sp) { return static_cast<Search*>(sp)->reusedVisits; }
// the candidates Agent::store_next_root_candidates kept after the last run (agent.cc:1373-1451): [0] = the selected child, then every
// reply generated below it.  out[i*6..]: joint move A, B of the reply (0, 0 for the selected child), visits, node type, endInPly, is it the
// selected child's own best move
int ora_search_retained(void* sp, int* out, int cap) {
    Search* s = static_cast<Search*>(sp);
    const auto& c = s->nextRootCandidates;
    if (c.empty() || !c[0].node) return 0;
    const std::shared_ptr<Node>& own = c[0].node;
    const int best = own->isExpanded && !own->children.empty() ? own->get_best_move_idx_with_q_weight(s->cfg.qVetoDelta, s->cfg.qValueWeight) : -1;
    int n = 0;
    for (size_t i = 0; i < c.size() && n < cap; ++i, ++n) {
        int* o = out + n * 6;
        o[0] = o[1] = 0; o[5] = 0;
        if (i > 0) {
            size_t k = 0, seen = 0;                      // the (i-1)-th non-null child of the selected child
            for (; k < own->children.size(); ++k) if (own->children[k] && seen++ == i - 1) break;
            o[0] = (int)own->gen.generated[k].moveA; o[1] = (int)own->gen.generated[k].moveB; o[5] = (int)k == best ? 1 : 0;
        }
        o[2] = c[i].node ? c[i].node->visits : -1; o[3] = c[i].node ? (int)c[i].node->nodeType : -1; o[4] = c[i].node ? c[i].node->endInPly : 0;
    }
    return n;
}
void ora_search_info(void* sp, int* out /*8*/) {
    Search* s = static_cast<Search*>(sp);
    out[0] = s->nodesSearched; out[1] = s->evalRows; out[2] = s->evalCalls; out[3] = s->sameBatchCollisions;
    out[4] = s->reservationCollisions; out[5] = s->nodeCounter; out[6] = s->root ? (int)s->root->nodeType : -1;
    out[7] = s->root ? s->root->visits : 0;
}
// lookups that found their position in the transposition table (insertOrGet, transposition_table.h:83-103)
int ora_search_tt_hits(void* sp) { return static_cast<Search*>(sp)->ttHits; }
int ora_search_trace(void* sp, uint64_t* out, int cap) {
    Search* s = static_cast<Search*>(sp);
    int n = (int)std::min<size_t>(s->evalTrace.size(), (size_t)cap);
    for (int i = 0; i < n; ++i) out[i] = s->evalTrace[i];
    return (int)s->evalTrace.size();
}
// classify_terminal_position (searchthread.cc:101-139): outcome | endInPly << 8
int ora_classify(void* h, int teamToPlay, int rootTeam, int rootAdv, int searchPly) {
    int e = 0;
    const int to = (int)classify_terminal_position(*static_cast<Board*>(h), teamToPlay, rootTeam, rootAdv != 0, searchPly, &e);
    return to | (e << 8);
}
int ora_search_ctx_trace(void* sp, uint64_t* out, int cap) {
    Search* s = static_cast<Search*>(sp);
    int n = (int)std::min<size_t>(s->ctxTrace.size(), (size_t)cap);
    for (int i = 0; i < n; ++i) out[i] = s->ctxTrace[i];
    return (int)s->ctxTrace.size();
}
void ora_hash_evaluator(const uint16_t* planes, int n, uint16_t* value, uint16_t* piA, uint16_t* piB, uint16_t* wdl, uint16_t* ml) {
    EvalOutputs o;
    hash_evaluator(planes, n, o);
    std::memcpy(value, o.value.data(), 2 * (size_t)n); std::memcpy(piA, o.piA.data(), 2 * o.piA.size());
    std::memcpy(piB, o.piB.data(), 2 * o.piB.size()); std::memcpy(wdl, o.wdl.data(), 2 * o.wdl.size());
    std::memcpy(ml, o.movesLeft.data(), 2 * (size_t)n);
}
// JointCandidateGenerator driven to exhaustion: returns the number of candidates, writes (moveA, moveB, jointPrior)
int ora_gen_enumerate(const uint32_t* a, int nA, const uint32_t* b, int nB, const float* pa, const float* pb, int adv, int aOn, int bOn,
                      const uint8_t* ca, const uint8_t* cb, int tie_mode, uint32_t* outA, uint32_t* outB, float* outP, int cap) {
    CandidateGenerator g;
    g.initialize(std::vector<Move>(a, a + nA), std::vector<Move>(b, b + nB), std::vector<float>(pa, pa + nA), std::vector<float>(pb, pb + nB),
                 adv != 0, aOn != 0, bOn != 0, ca ? std::vector<uint8_t>(ca, ca + nA) : std::vector<uint8_t>(),
                 cb ? std::vector<uint8_t>(cb, cb + nB) : std::vector<uint8_t>(), tie_mode);
    int n = 0;
    while (g.hasNext()) {
        Candidate c = g.getNext();
        if (n < cap) { outA[n] = c.moveA; outB[n] = c.moveB; outP[n] = c.jointPrior; }
        ++n;
    }
    return n;
}
float ora_joint_prior(uint32_t mA, float pA, uint32_t mB, float pB, int aOn, int bOn, int adv, int aCan, int bCan, int capA, int capB) {
    JointActionRules r{aOn != 0, bOn != 0, adv != 0, aCan != 0, bCan != 0};
    return Candidate(mA, pA, 0, mB, pB, 0, r, capA != 0, capB != 0).jointPrior;
}
int ora_is_double_sit_legal(int adv, int aOn, int bOn) { return is_double_sit_legal(adv != 0, aOn != 0, bOn != 0); }
int ora_is_single_pass_legal(int adv, int aOn, int bOn, int cap) { return is_single_pass_legal(adv != 0, aOn != 0, bOn != 0, cap != 0); }
int ora_pw_allowed_children(int visits, int isRoot) { return get_allowed_children(visits, isRoot ? 4.0f : 2.0f, 0.4f); }
float ora_get_cpuct(float v) { return get_cpuct(v, 2.5f, 19652.0f); }
float ora_portable_expf(float x) { return portable_expf(x); }

// timing helper of tools/cpu_baseline.py: `reps` calls of Board::legal_moves(b) on this position; returns the moves counted
long long ora_time_legal_moves(void* h, int b, int reps) {
    Board& board = *static_cast<Board*>(h);
    long long n = 0;
    for (int r = 0; r < reps; ++r) n += (long long)board.legal_moves(b).size();
    return n;
}
// timing helpers for bench.py's cpu_baseline leg
double ora_time_planes(const hm_board* boards, size_t n, int dtype, void* out, int reps) {
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) ora_planes(boards, n, dtype, out);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"

#if __has_include("search.hpp")
#include "oracle_lab.cc"
#if __has_include("selfplay.hpp")
#include "oracle_selfplay.cc"
#endif
#if __has_include("tournament.hpp")
#include "oracle_tournament.cc"
#endif
#endif
