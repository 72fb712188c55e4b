// oracle/selfplay.hpp — CPU restatement of the reference's self-play game loop.
//
// TEST INFRASTRUCTURE ONLY (see bughouse.hpp).  Follows tools/selfplay.cc line by line:
//   mix_seed :160-165, apply_temperature :167-185, sample_initialization_length :187-200,
//   sample_raw_policy_temperature :202-216, randomized_node_budget :218-224, mcts_temperature :226-232,
//   prepare_raw_policy / sample_raw_policy_action :276-376, action_leads_to_terminal :378-390,
//   marginal_policy :400-427, select_action :429-462, encode_planes :464-476, run_selfplay :558-748,
//   TrainingSample :40-53 and the sample layout of ChunkWriter::flush :126-142.
// on top of oracle::Search (one search thread = the reference's only deterministic configuration).
//
// PARITY STATUS: tools/selfplay.cc cannot be built in this image (nn/engine.h -> TensorRT), and the
// reference holds no test or fixture for run_selfplay, so this file is pinned only through its parts:
// the search it drives (tests/golden/search_cases.json), the Board rules (reference build), the record
// layout (the reference reader's struct strings) — the loop itself is "parity unpinned".
//
// One documented deviation, shared with the product: every game draws from its own
// std::mt19937_64 seeded mix_seed(runId, gameIndex) instead of one engine threaded through the
// sequential game loop, so a game's records do not depend on which games were played before it.
#pragma once
#include <map>
#include <numeric>

#include "search.hpp"

namespace hmo {

struct SelfPlayConfig {   // tools/selfplay.h:10-31
    uint64_t games = 1, nodes = 800, maxMacroPlies = 400, chunkSamples = 16384;
    double rawPolicyMeanMacroPlies = 8.0;
    uint64_t rawPolicyMaxMacroPlies = 30;
    double rawPolicyHighTemperatureProbability = 0.05;
    double mctsTemperature = 1.0, mctsTemperatureDecay = 0.93;
    uint64_t mctsTemperaturePlies = 20;
    float resignThreshold = -0.90f;
    uint64_t resignConsecutivePlies = 3;
    double resignDisableFraction = 0.10;
    double nodeRandomFactor = 0.05;
    float dirichletAlpha = 0.3f, dirichletEpsilon = 0.25f;
    uint64_t seed = 0;
};

inline uint64_t sp_mix_seed(uint64_t seed, uint64_t value) {   // :160-165
    value += 0x9e3779b97f4a7c15ULL;
    value = (value ^ (value >> 30)) * 0xbf58476d1ce4e5b9ULL;
    value = (value ^ (value >> 27)) * 0x94d049bb133111ebULL;
    return seed ^ (value ^ (value >> 31));
}
inline void sp_apply_temperature(std::vector<float>& p, double temperature) {   // :167-185
    if (p.empty() || temperature <= 0.0) throw std::invalid_argument("Policy temperature must be positive");
    const double exponent = 1.0 / temperature;
    double total = 0.0;
    for (float& v : p) { v = static_cast<float>(std::pow(std::max(0.0f, v), exponent)); total += v; }
    if (!std::isfinite(total) || total <= 0.0) { std::fill(p.begin(), p.end(), 1.0f / static_cast<float>(p.size())); return; }
    for (float& v : p) v = static_cast<float>(v / total);
}

struct SparsePolicyEntry { uint16_t index; float probability; };
struct TrainingSample {   // :40-53
    uint64_t gameId = 0;
    uint32_t nodes = 0;
    uint16_t macroPly = 0, movesLeft = 0;
    uint8_t team = 0, hasTimeAdvantage = 0;
    int8_t outcome = 0;
    uint8_t wdl = 1;
    float rootQ = 0.0f;
    std::array<uint8_t, HM_PLANE_VALUES> planes{};
    std::vector<SparsePolicyEntry> policyA, policyB;
};
template <typename T>
inline void sp_put(std::vector<uint8_t>& out, const T& v) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
    out.insert(out.end(), p, p + sizeof(T));
}
inline void sp_serialize(std::vector<uint8_t>& out, const TrainingSample& s) {   // ChunkWriter::flush body :126-142
    sp_put(out, s.gameId); sp_put(out, s.nodes); sp_put(out, s.macroPly); sp_put(out, s.movesLeft);
    sp_put(out, s.team); sp_put(out, s.hasTimeAdvantage); sp_put(out, s.outcome); sp_put(out, s.wdl); sp_put(out, s.rootQ);
    out.insert(out.end(), s.planes.begin(), s.planes.end());
    for (const auto* pol : {&s.policyA, &s.policyB}) {
        sp_put(out, static_cast<uint16_t>(pol->size()));
        for (const SparsePolicyEntry& e : *pol) { sp_put(out, e.index); sp_put(out, e.probability); }
    }
}

struct GameResult {
    std::vector<uint8_t> records;     // serialized samples of this game, in macro-ply order
    uint64_t samples = 0, rawPlies = 0, totalNodes = 0;
    int winner = -1;                  // -1 none, 0 white team, 1 black team (selfplay.cc:593,609)
    int termination = 0;              // 0 macro-ply limit, 1 checkmate, 2 draw, 3 resignation, 4 no legal action
    std::vector<uint32_t> movesA, movesB;   // the joint actions played (diagnostics: lets a test replay the game)
    std::vector<uint8_t> rawFlags;          // 1 = raw-policy ply
};

class SelfPlay {
public:
    SelfPlayConfig cfg;
    Search search;                    // cfg.tie_mode / exp_mode / evaluator set by the caller

    static std::array<uint8_t, HM_PLANE_VALUES> encode_planes(Board& board, int team, bool adv) {   // :464-476
        hm_board cb;
        board.to_compact(&cb, team, adv);
        std::array<uint8_t, HM_PLANE_VALUES> out;
        planes_u8(cb, out.data());
        return out;
    }
    static bool action_leads_to_terminal(const Board& board, Move a, Move b, int team, bool adv) {   // :378-390
        Board future(board);
        future.make_moves(a, b);
        return future.is_checkmate(team ^ 1, !adv) || future.is_checkmate(team, adv) || future.is_draw();
    }
    static std::vector<SparsePolicyEntry> marginal_policy(const Board& board, int b, const std::vector<RootEdge>& edges) {   // :400-427
        std::map<uint16_t, uint64_t> visitsByMove;
        uint64_t total = 0;
        for (const RootEdge& e : edges) {
            if (e.visits <= 0) continue;
            const Move m = b == 0 ? e.moveA : e.moveB;
            const int idx = policy_index(m, board.pos[b].stm);
            if (idx < 0) throw std::runtime_error("Move is absent from policy map");
            visitsByMove[static_cast<uint16_t>(idx)] += static_cast<uint64_t>(e.visits);
            total += static_cast<uint64_t>(e.visits);
        }
        if (total == 0) throw std::runtime_error("Search returned no visited root edges");
        std::vector<SparsePolicyEntry> pol;
        for (const auto& kv : visitsByMove) pol.push_back({kv.first, static_cast<float>(kv.second) / static_cast<float>(total)});
        return pol;
    }
    static size_t select_action(const std::vector<RootEdge>& edges, double temperature, std::mt19937_64& rng) {   // :429-462 (returns the edge index)
        if (edges.empty()) throw std::runtime_error("Cannot select from an empty root");
        auto byVisits = [](const RootEdge& l, const RootEdge& r) { return l.visits < r.visits; };
        if (temperature <= 1e-6) return (size_t)(std::max_element(edges.begin(), edges.end(), byVisits) - edges.begin());
        std::vector<double> w;
        const int maxVisits = std::max_element(edges.begin(), edges.end(), byVisits)->visits;
        for (const RootEdge& e : edges)
            w.push_back(e.visits > 0 && maxVisits > 0 ? std::exp((std::log(static_cast<double>(e.visits)) - std::log(static_cast<double>(maxVisits))) / temperature) : 0.0);
        if (std::accumulate(w.begin(), w.end(), 0.0) <= 0.0) return 0;
        std::discrete_distribution<size_t> d(w.begin(), w.end());
        return d(rng);
    }

    // prepare_raw_policy (:276-300) on the evaluator's fp16 policy row
    void prepare_raw_policy(Board& board, int b, bool onTurn, const uint16_t* policy, double temperature, std::vector<Move>& actions, std::vector<float>& probs) {
        if (onTurn) {
            actions = board.legal_moves(b);
            actions.erase(std::remove_if(actions.begin(), actions.end(), [](Move m) { return !is_policy_move_representable(m); }), actions.end());
        }
        if (actions.empty()) { actions.push_back(MOVE_NONE); probs.push_back(1.0f); return; }
        actions.push_back(MOVE_NONE);
        probs = get_normalized_probability(policy, actions, board.pos[b].stm, search.cfg.exp_mode);
        sp_apply_temperature(probs, temperature);
    }
    // sample_raw_policy_action (:301-376)
    void sample_raw_policy_action(Board& board, int team, bool adv, double temperature, std::mt19937_64& rng, Move& outA, Move& outB) {
        hm_board cb;
        board.to_compact(&cb, team, adv);
        std::vector<uint16_t> obs(HM_PLANE_VALUES);
        planes_f16(cb, obs.data());                                   // RawPolicyEvaluator::evaluate :245-260 (one position)
        EvalOutputs out;
        search.evaluator(obs.data(), 1, out);
        const bool aOn = board.pos[0].stm == team, bOn = board.pos[1].stm == (team ^ 1);
        std::vector<Move> aA, aB;
        std::vector<float> pA, pB;
        prepare_raw_policy(board, 0, aOn, out.piA.data(), temperature, aA, pA);
        prepare_raw_policy(board, 1, bOn, out.piB.data(), temperature, aB, pB);
        auto caps = [&](const std::vector<Move>& v, int b) {
            std::vector<uint8_t> c;
            for (Move m : v) c.push_back(m != MOVE_NONE && board.pos[b].is_capture(m) ? 1 : 0);
            return c;
        };
        const std::vector<uint8_t> cA = caps(aA, 0), cB = caps(aB, 1);
        JointActionRules rules;
        rules.aOnTurn = aOn; rules.bOnTurn = bOn; rules.adv = adv;
        rules.aCanMove = aOn && aA.size() > 1; rules.bCanMove = bOn && aB.size() > 1;
        auto make = [&](size_t iA, size_t iB) { return Candidate(aA[iA], pA[iA], iA, aB[iB], pB[iB], iB, rules, cA[iA] != 0, cB[iB] != 0); };
        std::discrete_distribution<size_t> sampleA(pA.begin(), pA.end());
        std::discrete_distribution<size_t> sampleB(pB.begin(), pB.end());
        // :351 draws both samples inside one argument list, whose evaluation order C++ leaves unspecified (g++ draws B
        // first, clang A first); the product and this restatement fix A then B.
        const size_t sA = sampleA(rng);
        const size_t sB = sampleB(rng);
        Candidate cand = make(sA, sB);
        if (cand.jointPrior >= 0.0f) { outA = cand.moveA; outB = cand.moveB; return; }
        std::vector<std::pair<size_t, size_t>> legal;
        std::vector<double> w;
        for (size_t iA = 0; iA < aA.size(); ++iA)
            for (size_t iB = 0; iB < aB.size(); ++iB) {
                if (make(iA, iB).jointPrior < 0.0f) continue;
                legal.emplace_back(iA, iB);
                w.push_back(static_cast<double>(pA[iA]) * static_cast<double>(pB[iB]));
            }
        if (std::accumulate(w.begin(), w.end(), 0.0) <= 0.0) throw std::runtime_error("Raw policy produced no legal joint action");
        std::discrete_distribution<size_t> ls(w.begin(), w.end());
        const auto pr = legal[ls(rng)];
        cand = make(pr.first, pr.second);
        outA = cand.moveA; outB = cand.moveB;
    }

    GameResult play_game(uint64_t runId, uint64_t gameIndex) {   // body of the game loop, run_selfplay :587-744
        GameResult res;
        Board board;
        std::mt19937_64 rng(sp_mix_seed(runId, gameIndex));
        const int startingTeam = static_cast<int>(rng() & 1ULL);
        int team = startingTeam == 0 ? WHITE : BLACK;
        bool adv = false;
        std::vector<TrainingSample> samples;
        size_t initLength = 0;                                        // sample_initialization_length :187-200
        if (cfg.rawPolicyMeanMacroPlies > 0.0 && cfg.rawPolicyMaxMacroPlies != 0) {
            std::exponential_distribution<double> d(1.0 / cfg.rawPolicyMeanMacroPlies);
            size_t len = static_cast<size_t>(std::llround(d(rng)));
            if (len > cfg.rawPolicyMaxMacroPlies) {
                std::uniform_int_distribution<size_t> clipped(0, cfg.rawPolicyMaxMacroPlies);
                len = clipped(rng);
            }
            initLength = len;
        }
        bool rawActive = initLength > 0;
        const bool canResign = cfg.resignThreshold < 0.0f
            && (cfg.resignDisableFraction <= 0.0 || std::uniform_real_distribution<double>(0.0, 1.0)(rng) >= cfg.resignDisableFraction);
        std::array<size_t, 2> resignPlies{{0, 0}};
        int winner = -1, termination = 0;
        for (size_t macroPly = 0; macroPly < cfg.maxMacroPlies; ++macroPly) {
            if (board.is_checkmate(team, adv)) { winner = team == WHITE ? 1 : 0; termination = 1; break; }
            if (board.is_draw()) { termination = 2; break; }
            if (rawActive && macroPly < initLength) {
                std::uniform_real_distribution<double> unit(0.0, 1.0);                // sample_raw_policy_temperature :202-216
                double temperature = 1.0;
                if (!(unit(rng) >= cfg.rawPolicyHighTemperatureProbability)) {
                    const double choice = unit(rng);
                    temperature = choice < 0.75 ? 2.0 : (choice < 0.95 ? 5.0 : 10.0);
                }
                Move a = MOVE_NONE, b = MOVE_NONE;
                sample_raw_policy_action(board, team, adv, temperature, rng, a, b);
                if (!action_leads_to_terminal(board, a, b, team, adv)) {
                    if (a != MOVE_NONE) board.push_move(0, a);
                    if (b != MOVE_NONE) board.push_move(1, b);
                    res.movesA.push_back(a); res.movesB.push_back(b); res.rawFlags.push_back(1);
                    res.rawPlies++;
                    team ^= 1; adv = !adv;
                    continue;
                }
                rawActive = false;
            }
            TrainingSample sample;
            sample.gameId = gameIndex;
            sample.macroPly = static_cast<uint16_t>(std::min<size_t>(macroPly, 65535));
            sample.team = team == WHITE ? 0 : 1;
            sample.hasTimeAdvantage = adv ? 1 : 0;
            sample.planes = encode_planes(board, team, adv);
            std::uniform_real_distribution<double> jitter(-cfg.nodeRandomFactor, cfg.nodeRandomFactor);   // randomized_node_budget :218-224
            const size_t targetNodes = std::max<size_t>(1, static_cast<size_t>(std::llround(static_cast<double>(cfg.nodes) * (1.0 + jitter(rng)))));
            search.cfg.rootDirichletAlpha = cfg.dirichletAlpha;
            search.cfg.rootDirichletEpsilon = cfg.dirichletEpsilon;
            search.cfg.rootNoiseSeed = sp_mix_seed(runId, gameIndex * cfg.maxMacroPlies + macroPly);
            const bool ok = search.run(board, team, adv, (int)targetNodes);
            const std::vector<RootEdge> edges = ok ? search.root_edge_stats() : std::vector<RootEdge>();
            if (edges.empty()) { winner = team == WHITE ? 1 : 0; termination = 4; break; }
            uint64_t actual = 0;
            for (const RootEdge& e : edges) actual += static_cast<uint64_t>(std::max(0, e.visits));
            sample.nodes = static_cast<uint32_t>(std::min<uint64_t>(actual, 0xffffffffu));
            const float rootQ = search.root_q();
            sample.rootQ = rootQ;
            sample.policyA = marginal_policy(board, 0, edges);
            sample.policyB = marginal_policy(board, 1, edges);
            samples.push_back(std::move(sample));
            const size_t teamIdx = team == WHITE ? 0 : 1;
            if (canResign) {
                if (rootQ <= cfg.resignThreshold) {
                    if (++resignPlies[teamIdx] >= cfg.resignConsecutivePlies) { winner = team == WHITE ? 1 : 0; termination = 3; break; }
                } else resignPlies[teamIdx] = 0;
            }
            const double temperature = (cfg.mctsTemperaturePlies > 0 && macroPly >= cfg.mctsTemperaturePlies) ? 0.0   // mcts_temperature :226-232
                : cfg.mctsTemperature * std::pow(cfg.mctsTemperatureDecay, static_cast<double>(macroPly / 2));
            const RootEdge& pick = edges[select_action(edges, temperature, rng)];
            res.movesA.push_back(pick.moveA); res.movesB.push_back(pick.moveB); res.rawFlags.push_back(0);
            if (pick.moveA == MOVE_NONE && pick.moveB == MOVE_NONE) { team ^= 1; adv = !adv; continue; }
            if (pick.moveA != MOVE_NONE) board.push_move(0, pick.moveA);
            if (pick.moveB != MOVE_NONE) board.push_move(1, pick.moveB);
            team ^= 1; adv = !adv;
        }
        if (winner < 0 && board.is_checkmate(team, adv)) { winner = team == WHITE ? 1 : 0; termination = 1; }
        else if (winner < 0 && board.is_draw()) termination = 2;
        for (size_t i = 0; i < samples.size(); ++i) {
            TrainingSample& s = samples[i];
            s.outcome = winner < 0 ? 0 : (s.team == winner ? 1 : -1);
            s.wdl = static_cast<uint8_t>(s.outcome + 1);
            s.movesLeft = static_cast<uint16_t>(std::min<size_t>(samples.size() - i, 65535));
            sp_serialize(res.records, s);
            res.totalNodes += s.nodes;
        }
        res.samples = samples.size();
        res.winner = winner; res.termination = termination;
        return res;
    }
};

}  // namespace hmo
