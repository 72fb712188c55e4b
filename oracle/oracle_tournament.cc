// oracle/oracle_tournament.cc — C entry points of the tournament restatement (oracle/tournament.hpp).
// TEST INFRASTRUCTURE ONLY; included by oracle_capi.cc.
#include "tournament.hpp"

extern "C" {

// the salted hash evaluator, for callers that need the same two deterministic "networks" outside the oracle
void ora_hash_evaluator_salted(const uint16_t* planes, int n, uint64_t salt, uint16_t* value, uint16_t* piA, uint16_t* piB, uint16_t* wdl, uint16_t* ml) {
    EvalOutputs out;
    hash_evaluator_salted(planes, n, salt, out);
    std::memcpy(value, out.value.data(), 2 * (size_t)n);
    std::memcpy(piA, out.piA.data(), 2 * (size_t)n * HM_POLICY_VALUES);
    std::memcpy(piB, out.piB.data(), 2 * (size_t)n * HM_POLICY_VALUES);
    std::memcpy(wdl, out.wdl.data(), 2 * (size_t)n * 3);
    std::memcpy(ml, out.movesLeft.data(), 2 * (size_t)n);
}

// cfg: the product's own hm_tournament_config (include/hivemind_amd.h), so tests hand both sides one struct
void* ora_tournament_new(const hm_tournament_config* c, int tie_mode, int exp_mode, uint64_t saltContender, uint64_t saltBaseline) {
    Tournament* t = new Tournament();
    t->cfg.games = c->games; t->cfg.nodes = c->nodes; t->cfg.moveTimeMs = c->move_time_ms;
    t->cfg.contenderBatchSize = c->contender_batch_size; t->cfg.baselineBatchSize = c->baseline_batch_size;
    t->cfg.maxMacroPlies = c->max_macro_plies; t->cfg.dirichletAlpha = c->dirichlet_alpha; t->cfg.dirichletEpsilon = c->dirichlet_epsilon;
    t->cfg.contenderPwCoefficient = c->contender_pw_coefficient; t->cfg.baselinePwCoefficient = c->baseline_pw_coefficient;
    t->cfg.seed = c->seed;
    t->search.cfg.tie_mode = tie_mode; t->search.cfg.exp_mode = exp_mode;
    t->contender = [saltContender](const uint16_t* p, int n, EvalOutputs& o) { hash_evaluator_salted(p, n, saltContender, o); };
    t->baseline = [saltBaseline](const uint16_t* p, int n, EvalOutputs& o) { hash_evaluator_salted(p, n, saltBaseline, o); };
    return t;
}
void ora_tournament_free(void* h) { delete static_cast<Tournament*>(h); }
// returns 0, or -1 with the exception text in err
int ora_tournament_run(void* h, char* err, int cap) {
    try { static_cast<Tournament*>(h)->run(); }
    catch (const std::exception& e) { if (err && cap > 0) std::snprintf(err, (size_t)cap, "%s", e.what()); return -1; }
    return 0;
}
static int64_t copy_text(const std::string& s, char* out, int64_t cap) {
    if ((int64_t)s.size() + 1 > cap) return -(int64_t)s.size() - 1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int64_t)s.size();
}
int64_t ora_tournament_summary(void* h, const char* contender, const char* baseline, char* out, int64_t cap) {
    return copy_text(static_cast<Tournament*>(h)->summary(contender, baseline), out, cap);
}
int64_t ora_tournament_pgn(void* h, const char* contender, const char* baseline, char* out, int64_t cap) {
    return copy_text(static_cast<Tournament*>(h)->pgn(contender, baseline), out, cap);
}
// statistics of a hand-made result (the reference's known answers, engine/tests/test_tournament.cc:20-72):
// out[0] score, [1] has elo, [2] elo, [3] has score CI, [4..5] score CI, [6] has elo CI, [7..8] elo CI, [9] paired method
void ora_tournament_stats(uint64_t contenderWins, uint64_t baselineWins, uint64_t draws, const double* pairs, int nPairs, double* out) {
    TournamentResult r;
    r.contenderWins = contenderWins; r.baselineWins = baselineWins; r.draws = draws;
    r.pairScores.assign(pairs, pairs + nPairs);
    out[0] = r.contenderScore();
    const auto elo = r.contenderElo();
    out[1] = elo ? 1 : 0; out[2] = elo ? *elo : 0;
    const auto si = r.scoreConfidence95();
    out[3] = si ? 1 : 0; out[4] = si ? si->first : 0; out[5] = si ? si->second : 0;
    const auto ei = r.eloConfidence95();
    out[6] = ei ? 1 : 0; out[7] = ei ? ei->first : 0; out[8] = ei ? ei->second : 0;
    out[9] = r.confidenceMethod() == "paired-opening normal approximation" ? 1 : 0;
}
int ora_move_uci(uint32_t m, char* out, int cap) {
    const std::string s = board_uci_move(m);
    std::snprintf(out, (size_t)cap, "%s", s.c_str());
    return (int)s.size();
}

}  // extern "C"
