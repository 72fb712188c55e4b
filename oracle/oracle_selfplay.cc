// oracle/oracle_selfplay.cc — C entry points of the self-play loop restatement (oracle/selfplay.hpp).
// TEST INFRASTRUCTURE ONLY; included by oracle_capi.cc.
#include "selfplay.hpp"

extern "C" {

// cfg: the product's own hm_selfplay_config (include/hivemind_amd.h), so tests hand both sides one struct
void* ora_selfplay_new(const hm_selfplay_config* c, int tie_mode, int exp_mode) {
    SelfPlay* s = new SelfPlay();
    s->cfg.games = c->games; s->cfg.nodes = c->nodes; s->cfg.maxMacroPlies = c->max_macro_plies; s->cfg.chunkSamples = c->chunk_samples;
    s->cfg.rawPolicyMeanMacroPlies = c->raw_policy_mean_macro_plies; s->cfg.rawPolicyMaxMacroPlies = c->raw_policy_max_macro_plies;
    s->cfg.rawPolicyHighTemperatureProbability = c->raw_policy_high_temperature_probability;
    s->cfg.mctsTemperature = c->mcts_temperature; s->cfg.mctsTemperatureDecay = c->mcts_temperature_decay; s->cfg.mctsTemperaturePlies = c->mcts_temperature_plies;
    s->cfg.resignThreshold = c->resign_threshold; s->cfg.resignConsecutivePlies = c->resign_consecutive_plies; s->cfg.resignDisableFraction = c->resign_disable_fraction;
    s->cfg.nodeRandomFactor = c->node_random_factor; s->cfg.dirichletAlpha = c->dirichlet_alpha; s->cfg.dirichletEpsilon = c->dirichlet_epsilon;
    s->cfg.seed = c->seed;
    s->search.cfg.tie_mode = tie_mode; s->search.cfg.exp_mode = exp_mode;
    s->search.evaluator = hash_evaluator;
    return s;
}
void ora_selfplay_free(void* h) { delete static_cast<SelfPlay*>(h); }
void ora_selfplay_set_callback(void* h, ora_eval_cb cb) {
    SelfPlay* s = static_cast<SelfPlay*>(h);
    if (!cb) { s->search.evaluator = hash_evaluator; return; }
    s->search.evaluator = [cb](const uint16_t* planes, int n, EvalOutputs& out) {
        out.value.assign(n, 0); out.piA.assign((size_t)n * HM_POLICY_VALUES, 0); out.piB.assign((size_t)n * HM_POLICY_VALUES, 0);
        out.wdl.assign((size_t)n * 3, 0); out.movesLeft.assign(n, 0);
        cb(planes, n, out.value.data(), out.piA.data(), out.piB.data(), out.wdl.data(), out.movesLeft.data());
    };
}
// Plays game `gameIndex` of run `seed`; returns the serialized record bytes (caller buffer, cap bytes).
// info[8]: samples, raw plies, winner, termination, total nodes, actions played, -, -
static thread_local GameResult g_lastGame;
int64_t ora_selfplay_game(void* h, uint64_t gameIndex, uint8_t* out, uint64_t cap, int64_t* info) {
    SelfPlay* s = static_cast<SelfPlay*>(h);
    try { g_lastGame = s->play_game(s->cfg.seed, gameIndex); }
    catch (const std::exception& e) { std::fprintf(stderr, "ora_selfplay_game: %s\n", e.what()); return -1; }
    const GameResult& r = g_lastGame;
    if (info) { info[0] = (int64_t)r.samples; info[1] = (int64_t)r.rawPlies; info[2] = r.winner; info[3] = r.termination; info[4] = (int64_t)r.totalNodes; info[5] = (int64_t)r.movesA.size(); info[6] = info[7] = 0; }
    if (r.records.size() > cap) return -(int64_t)r.records.size();
    if (!r.records.empty()) std::memcpy(out, r.records.data(), r.records.size());
    return (int64_t)r.records.size();
}
// joint actions of the last game played on this thread: (moveA, moveB, raw flag) per macro-ply
int ora_selfplay_last_actions(uint32_t* a, uint32_t* b, uint8_t* raw, int cap) {
    const GameResult& r = g_lastGame;
    const int n = (int)std::min<size_t>(r.movesA.size(), (size_t)cap);
    for (int i = 0; i < n; ++i) { a[i] = r.movesA[(size_t)i]; b[i] = r.movesB[(size_t)i]; raw[i] = r.rawFlags[(size_t)i]; }
    return (int)r.movesA.size();
}

}  // extern "C"
