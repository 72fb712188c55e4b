// oracle/search.hpp — CPU restatement of the reference's joint-action MCGS (single search thread).
//
// TEST INFRASTRUCTURE ONLY (see bughouse.hpp).  PARITY STATUS: the reference's search/*.cc and
// tools/selfplay.cc cannot be built in this image (they include nn/engine.h -> TensorRT and
// <cuda_fp16.h> -> <nv/target>, both absent), so this restatement is pinned by the reference's own
// known answers instead: 47 gtest cases of engine/tests/test_move_gen.cc (backup perspective, solver
// propagation, classify_terminal_position incl. the waiting-board mate, collision cancel, softmax
// robustness, moves-left discount, progressive-widening gates, virtual-loss / pending-evaluation
// diversion, dynamic FPU, transposition edges, best-move rule, Q/solver at depth) transcribed as
// data in tests/golden/search_cases.json and replayed through oracle/oracle_lab.cc
// (tests/test_oracle_search_cases.py; every reference TEST is either a case or listed with the
// reason it is out of scope), plus the PW schedule / cpuct / sit rules from the reference build
// (tests/golden/pw_schedule.json, tests/test_oracle_search.py).  What stays unpinned: a whole
// multi-hundred-node search has no reference-held expected visit vector in the tree.
// Every function cites the reference lines it follows (paths relative to engine/src).
//
// Schedule restated: Agent(1) — one SearchThread, batch size B=8, double-buffered lookahead
// (searchthread.cc:661-708), node budget loop (agent.cc:331-341), finish_pending_iteration.
//
// Two deliberate, documented knobs (both default to the reference's behaviour):
//   tie_mode  0: std::sort / std::priority_queue exactly as joint_action.h (order of equal
//                priors is then libstdc++-defined);  1: strict total order (prior desc, index
//                asc) — implementation independent; what the GPU engine implements.
//   exp_mode  0: std::exp (glibc);  1: portable_expf below (bit-identical on host and gfx950).
#pragma once
#include <cmath>
#include <functional>
#include <memory>
#include <queue>
#include <unordered_map>
#include <unordered_set>

#include "bughouse.hpp"

namespace hmo {

// ---- search_params.h:26-273 -------------------------------------------------------------
struct SearchConfig {
    int batchSize = 8;
    float cpuctInit = 2.5f, cpuctBase = 19652.0f;
    bool enableTranspositions = true;
    float drawContempt = 0.0f;
    bool enableDynamicFpu = true;
    float fpuReduction = 1.0f;
    bool enableWdlEval = true;
    float wdlValueWeight = 0.25f;
    float movesLeftDiscount = 0.005f;
    float pwCoefficient = 2.0f, rootPwCoefficient = 4.0f, pwExponent = 0.4f;
    float qValueWeight = 1.0f, qVetoDelta = 0.4f;
    float rootDirichletAlpha = 0.0f, rootDirichletEpsilon = 0.0f;
    uint64_t rootNoiseSeed = 0;
    int tie_mode = 0;
    int exp_mode = 0;
};
constexpr float Q_INIT = -1.0f;

inline float get_cpuct(float totalVisits, float init, float base) {   // search_params.h:307-309
    return std::log((totalVisits + base + 1.0f) / base) + init;
}
inline int get_allowed_children(int visitCount, float coefficient, float exponent) {   // :311-317
    if (visitCount <= 0) return 1;
    return static_cast<int>(std::ceil(coefficient * std::pow(static_cast<float>(visitCount), exponent)));
}

// exp for x <= 0 built from IEEE +,*,fma,rint only: the same source compiled for gfx950 gives
// the same bits (hivemind_amd/csrc/hm_search_device.hpp keeps an identical copy).
inline float portable_expf(float x) {
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
    int32_t bits; std::memcpy(&bits, &p, 4);
    bits += (int32_t)n << 23;                      // n in [-126, 0], p in [0.5, 2): stays normal
    float out; std::memcpy(&out, &bits, 4);
    return out;
}

inline float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int s = 0; while (!(m & 0x400u)) { m <<= 1; ++s; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - s) << 23) | (m << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f; std::memcpy(&f, &x, 4);
    return f;
}

// ---- joint_action.h ---------------------------------------------------------------------
inline bool is_double_sit_legal(bool adv, bool aOnTurn, bool bOnTurn) { return adv && (aOnTurn != bOnTurn); }   // :14-18
inline bool is_single_pass_legal(bool adv, bool aOnTurn, bool bOnTurn, bool partnerCapture) {                  // :26-33
    return adv || !(aOnTurn && bOnTurn) || partnerCapture;
}
struct JointActionRules { bool aOnTurn = false, bOnTurn = false, adv = false, aCanMove = false, bCanMove = false; };

struct Candidate {   // JointActionCandidate :64-112
    Move moveA = 0, moveB = 0;
    float priorA = 0, priorB = 0, jointPrior = 0;
    size_t idxA = 0, idxB = 0;
    Candidate() {}
    Candidate(Move mA, float pA, size_t iA, Move mB, float pB, size_t iB, const JointActionRules& r, bool capA, bool capB)
        : moveA(mA), moveB(mB), priorA(pA), priorB(pB), idxA(iA), idxB(iB) {
        const bool sitsA = mA == MOVE_NONE, sitsB = mB == MOVE_NONE;
        bool invalid = false;
        if (sitsA && sitsB) invalid = !is_double_sit_legal(r.adv, r.aOnTurn, r.bOnTurn);
        else if (sitsA && r.aCanMove) invalid = !is_single_pass_legal(r.adv, r.aOnTurn, r.bOnTurn, capB);
        else if (sitsB && r.bCanMove) invalid = !is_single_pass_legal(r.adv, r.aOnTurn, r.bOnTurn, capA);
        jointPrior = invalid ? -1.0f : pA * pB;
    }
};
struct CandLess {
    int tie_mode;
    bool operator()(const Candidate& a, const Candidate& b) const {     // max-heap on expansionPriority :108-111
        if (a.jointPrior != b.jointPrior) return a.jointPrior < b.jointPrior;
        if (!tie_mode) return false;
        if (a.idxA != b.idxA) return a.idxA > b.idxA;                      // smaller index = higher priority
        return a.idxB > b.idxB;
    }
};

class CandidateGenerator {   // JointCandidateGenerator :126-359
public:
    std::vector<Move> actA, actB;
    std::vector<float> priA, priB;
    std::vector<uint8_t> capA, capB;
    std::priority_queue<Candidate, std::vector<Candidate>, CandLess> heap{CandLess{0}};
    std::unordered_set<uint64_t> visited;
    std::vector<Candidate> generated;
    JointActionRules rules;

    void push(size_t iA, size_t iB) {   // pushCandidate :146-177
        if (iA >= actA.size() || iB >= actB.size()) return;
        uint64_t key = ((uint64_t)iA << 32) | iB;
        if (!visited.insert(key).second) return;
        Candidate c(actA[iA], priA[iA], iA, actB[iB], priB[iB], iB, rules, capA[iA] != 0, capB[iB] != 0);
        if (c.jointPrior >= 0.0f) heap.push(c);
        else { push(iA + 1, iB); push(iA, iB + 1); }
    }
    void initialize(const std::vector<Move>& a, const std::vector<Move>& b, const std::vector<float>& pa,
                    const std::vector<float>& pb, bool adv, bool aOnTurn, bool bOnTurn,
                    const std::vector<uint8_t>& ca, const std::vector<uint8_t>& cb, int tie_mode) {   // :195-278
        actA.clear(); actB.clear(); priA.clear(); priB.clear(); capA.clear(); capB.clear();
        heap = std::priority_queue<Candidate, std::vector<Candidate>, CandLess>(CandLess{tie_mode});
        visited.clear(); generated.clear();
        rules = JointActionRules();
        rules.aOnTurn = aOnTurn; rules.bOnTurn = bOnTurn; rules.adv = adv;
        auto hasReal = [](const std::vector<Move>& v) { return std::any_of(v.begin(), v.end(), [](Move m) { return m != MOVE_NONE; }); };
        rules.aCanMove = aOnTurn && hasReal(a);
        rules.bCanMove = bOnTurn && hasReal(b);
        if (a.empty() || b.empty()) return;
        auto order = [&](const std::vector<float>& p) {
            std::vector<size_t> idx(p.size());
            for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
            if (tie_mode) std::sort(idx.begin(), idx.end(), [&p](size_t i, size_t j) { return p[i] != p[j] ? p[i] > p[j] : i < j; });
            else std::sort(idx.begin(), idx.end(), [&p](size_t i, size_t j) { return p[i] > p[j]; });
            return idx;
        };
        for (size_t i : order(pa)) { actA.push_back(a[i]); priA.push_back(pa[i]); capA.push_back(i < ca.size() ? ca[i] : 0); }
        for (size_t i : order(pb)) { actB.push_back(b[i]); priB.push_back(pb[i]); capB.push_back(i < cb.size() ? cb[i] : 0); }
        push(0, 0);
        if (heap.empty()) { push(1, 0); push(0, 1); }
    }
    bool hasNext() const { return !heap.empty(); }
    Candidate peekNext() const { return heap.empty() ? Candidate() : heap.top(); }   // :305-310
    Candidate getNext() {   // :312-328
        if (heap.empty()) return Candidate();
        Candidate best = heap.top();
        heap.pop();
        push(best.idxA + 1, best.idxB);
        push(best.idxA, best.idxB + 1);
        generated.push_back(best);
        return best;
    }
};

// ---- node.h / node.cc ---------------------------------------------------------------------
enum class NodeType : uint8_t { UNSOLVED = 0, WIN = 1, LOSS = 2, DRAW = 3 };

struct Node {
    std::vector<std::shared_ptr<Node>> children;
    CandidateGenerator gen;
    int expandedCount = 0;
    std::vector<float> qValues, childValueSum, childPriors;
    std::vector<int> childVisits, virtualLoss;
    int virtualVisitSum = 0;
    float valueSum = 0.0f;
    int depth = 0, visits = 0;
    bool evaluationPending = false, isExpanded = false;
    int team;
    uint64_t hash = 0;
    NodeType nodeType = NodeType::UNSOLVED;
    std::vector<NodeType> childNodeTypes;
    int unsolvedChildCount = 0, endInPly = 0;
    int id = 0;   // creation index (diagnostics only)

    explicit Node(int t, uint64_t h = 0) : team(t), hash(h) {}

    void update_and_remove_virtual_loss(size_t i, float v) {   // node.h:104-121
        virtualLoss[i]--; virtualVisitSum--; childVisits[i]++;
        if (childVisits[i] == 1) { childValueSum[i] = v; qValues[i] = v; }
        else { childValueSum[i] += v; qValues[i] = childValueSum[i] / static_cast<float>(childVisits[i]); }
        valueSum += v; visits++;
    }
    void update(size_t i, float v) {   // node.h:85-102
        childVisits[i]++;
        if (childVisits[i] == 1) { childValueSum[i] = v; qValues[i] = v; }
        else { childValueSum[i] += v; qValues[i] = childValueSum[i] / static_cast<float>(childVisits[i]); }
        valueSum += v; visits++;
    }
    void update_terminal(float v) { valueSum += v; visits++; }   // :123-127
    bool try_reserve() { if (evaluationPending) return false; evaluationPending = true; return true; }   // :384-388
    void release() { evaluationPending = false; }
    float Q() const {   // :433-447
        if (nodeType == NodeType::WIN) return 1.0f;
        if (nodeType == NodeType::LOSS) return -1.0f;
        if (nodeType == NodeType::DRAW) return 0.0f;
        return visits > 0 ? valueSum / static_cast<float>(visits) : valueSum;
    }
    void mark_as_win(int ply) { nodeType = NodeType::WIN; valueSum = 1.0f * (visits + 1); endInPly = ply; }     // :487-492
    void mark_as_loss(int ply) { nodeType = NodeType::LOSS; valueSum = -1.0f * (visits + 1); endInPly = ply; }  // :497-502
    void mark_as_draw(int ply) { nodeType = NodeType::DRAW; endInPly = ply; }                                   // :507-511

    bool should_expand_new_child(const SearchConfig& c) const {   // :151-175
        const bool allLose = !children.empty() && std::all_of(children.begin(), children.end(),
            [](const std::shared_ptr<Node>& ch) { return ch && ch->nodeType == NodeType::WIN; });
        if (gen.hasNext() && allLose) return true;
        for (size_t i = 0; i < childVisits.size(); ++i)
            if (childVisits[i] + virtualLoss[i] == 0) return false;
        const float coef = depth == 0 ? c.rootPwCoefficient : c.pwCoefficient;
        return gen.hasNext() && expandedCount < get_allowed_children(visits + virtualVisitSum, coef, c.pwExponent);
    }
    // expand_next_joint_child (:199-262): existingNode = a transposition-table node to reuse (edge seeded with one
    // pseudo-visit and Q = -existing.Q()), reserveForSelection = take the evaluation reservation and one virtual loss
    std::shared_ptr<Node> expand_next_joint_child(std::shared_ptr<Node> existingNode, uint64_t positionHash, Candidate& outAction,
                                                  int* outIdx, bool reserveForSelection, bool* outReserved, int& nodeCounter) {
        if (outReserved) *outReserved = false;
        if (!gen.hasNext()) return nullptr;
        Candidate cand = gen.getNext();
        outAction = cand;
        std::shared_ptr<Node> child;
        float childQ;
        if (existingNode) { child = existingNode; childQ = -existingNode->Q(); }
        else {
            child = std::make_shared<Node>(team ^ 1, positionHash);
            child->id = nodeCounter++;
            child->depth = depth + 1;
            childQ = Q_INIT;
        }
        if (reserveForSelection) {
            if (!child->try_reserve()) return nullptr;
            if (outReserved) *outReserved = true;
        }
        childValueSum.push_back(childQ); childPriors.push_back(cand.jointPrior);
        childVisits.push_back(existingNode ? 1 : 0);
        virtualLoss.push_back(reserveForSelection ? 1 : 0);
        if (reserveForSelection) virtualVisitSum++;
        children.push_back(child); qValues.push_back(childQ);
        expandedCount++;
        if (outIdx) *outIdx = expandedCount - 1;
        return child;
    }
    void apply_virtual_loss(int i, int amount = 1) { if (i >= 0 && (size_t)i < virtualLoss.size()) { virtualLoss[i] += amount; virtualVisitSum += amount; } }    // :417-423
    void remove_virtual_loss(int i, int amount = 1) { if (i >= 0 && (size_t)i < virtualLoss.size()) { virtualLoss[i] -= amount; virtualVisitSum -= amount; } }   // :425-431
    void set_value(float v) { valueSum = v; }                                                                    // :403-406
    float get_child_q(int i) const { return i >= 0 && (size_t)i < qValues.size() ? qValues[i] : 0.0f; }           // :638-644
    void replace_child(int i, const std::shared_ptr<Node>& c) { if (i >= 0 && (size_t)i < children.size()) children[i] = c; }   // :373-378

    // get_best_move_idx_with_q_weight (:656-754): solver-aware final move rule with Q-veto and Q-weighting
    int get_best_move_idx_with_q_weight(float qVetoDelta, float qValueWeight) const {
        if (childVisits.empty() || qValues.empty()) return -1;
        if (nodeType == NodeType::WIN) {
            int bestIdx = -1, shortest = INT32_MAX;
            for (size_t i = 0; i < childNodeTypes.size(); ++i)
                if (childNodeTypes[i] == NodeType::LOSS && children[i] && children[i]->endInPly < shortest) { shortest = children[i]->endInPly; bestIdx = (int)i; }
            if (bestIdx >= 0) return bestIdx;
        }
        if (nodeType == NodeType::LOSS) {
            int bestIdx = 0, longest = 0;
            for (size_t i = 0; i < children.size(); ++i)
                if (children[i] && children[i]->endInPly > longest) { longest = children[i]->endInPly; bestIdx = (int)i; }
            return bestIdx;
        }
        const bool hasNonLosing = std::any_of(children.begin(), children.end(), [](const std::shared_ptr<Node>& c) { return c && c->nodeType != NodeType::WIN; });
        auto eligible = [&](size_t i) { return !hasNonLosing || !children[i] || children[i]->nodeType != NodeType::WIN; };
        size_t first = 0;
        while (first < childVisits.size() && !eligible(first)) ++first;
        if (first == childVisits.size()) return -1;
        int bestVisitIdx = (int)first, maxVisits = childVisits[first], secondVisitIdx = -1;
        for (size_t i = first + 1; i < childVisits.size(); ++i) {
            if (!eligible(i)) continue;
            if (childVisits[i] > maxVisits) { secondVisitIdx = bestVisitIdx; maxVisits = childVisits[i]; bestVisitIdx = (int)i; }
            else if (secondVisitIdx < 0 || childVisits[i] > childVisits[secondVisitIdx]) secondVisitIdx = (int)i;
        }
        int bestQIdx = (int)first;
        float bestQ = qValues[first];
        for (size_t i = first + 1; i < qValues.size(); ++i) {
            if (!eligible(i)) continue;
            if (qValues[i] > bestQ) { bestQ = qValues[i]; bestQIdx = (int)i; }
        }
        if (qVetoDelta > 0.0f && bestQIdx != bestVisitIdx)
            if (qValues[bestQIdx] > qValues[bestVisitIdx] + qVetoDelta && childVisits[bestQIdx] > 1) return bestQIdx;
        if (qValueWeight > 0.0f && secondVisitIdx >= 0 && qValues[secondVisitIdx] > qValues[bestVisitIdx]) {
            const float qDifference = qValues[secondVisitIdx] - qValues[bestVisitIdx];
            const float adjusted = childVisits[secondVisitIdx] + qDifference * qValueWeight * childVisits[bestVisitIdx];
            if (adjusted > childVisits[bestVisitIdx]) return secondVisitIdx;
        }
        return bestVisitIdx;
    }
    void init_child_node_types() {   // :531-541
        if (childNodeTypes.size() < children.size()) {
            size_t old = childNodeTypes.size();
            childNodeTypes.resize(children.size(), NodeType::UNSOLVED);
            unsolvedChildCount += (int)(children.size() - old);
        }
    }
    bool update_child_node_type(int idx, NodeType ct) {   // :549-613
        if (nodeType != NodeType::UNSOLVED) return false;
        if (idx < 0 || (size_t)idx >= childNodeTypes.size()) return false;
        if (childNodeTypes[idx] != NodeType::UNSOLVED) return false;
        childNodeTypes[idx] = ct;
        unsolvedChildCount--;
        if (ct == NodeType::LOSS) {
            nodeType = NodeType::WIN;
            if (children[idx]) endInPly = children[idx]->endInPly + 1;
            return true;
        }
        if (unsolvedChildCount == 0 && isExpanded && !gen.hasNext()) {
            bool allWins = true, hasDrawn = false;
            int longest = 0;
            for (size_t i = 0; i < childNodeTypes.size(); ++i) {
                if (childNodeTypes[i] != NodeType::WIN) allWins = false;
                if (childNodeTypes[i] == NodeType::DRAW) hasDrawn = true;
                if (children[i] && children[i]->endInPly > longest) longest = children[i]->endInPly;
            }
            if (allWins) { nodeType = NodeType::LOSS; endInPly = longest + 1; return true; }
            if (hasDrawn) { nodeType = NodeType::DRAW; return true; }
        }
        return false;
    }
};

// ---- evaluator seam (nn/engine.h:43-81): fp16 planes in, fp16 heads out --------------------
struct EvalOutputs {
    std::vector<uint16_t> value, piA, piB, wdl, movesLeft;   // [n], [n*4672], [n*4672], [n*3], [n]
};
typedef std::function<void(const uint16_t* planes, int n, EvalOutputs& out)> Evaluator;

// Deterministic stand-in network for parity tests: FNV-1a over the 4736 fp16
// input words -> splitmix64 stream -> heads quantised to 1e-3 and rounded to fp16.
// salt != 0 gives a second, unrelated deterministic "network" (tournaments need two)
inline void hash_evaluator_salted(const uint16_t* planes, int n, uint64_t salt, EvalOutputs& out) {
    out.value.assign(n, 0); out.piA.assign((size_t)n * HM_POLICY_VALUES, 0); out.piB.assign((size_t)n * HM_POLICY_VALUES, 0);
    out.wdl.assign((size_t)n * 3, 0); out.movesLeft.assign(n, 0);
    for (int i = 0; i < n; ++i) {
        uint64_t h = 0xcbf29ce484222325ULL ^ salt;
        const uint16_t* p = planes + (size_t)i * HM_PLANE_VALUES;
        for (int k = 0; k < HM_PLANE_VALUES; ++k) { h ^= p[k]; h *= 0x100000001b3ULL; }
        uint64_t s = h;
        auto next = [&]() {
            s += 0x9e3779b97f4a7c15ULL;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
            return z ^ (z >> 31);
        };
        auto q = [&](int lo, int hi) {   // integer thousandths in [lo, hi)
            return (float)(lo + (int)(next() % (uint64_t)(hi - lo))) * 0.001f;
        };
        out.value[i] = f32_to_f16_rn(q(-900, 901));
        for (int k = 0; k < 3; ++k) out.wdl[(size_t)i * 3 + k] = f32_to_f16_rn(q(-2000, 2001));
        out.movesLeft[i] = f32_to_f16_rn(q(0, 1001));
        for (int k = 0; k < HM_POLICY_VALUES; ++k) {
            uint64_t r = next();
            out.piA[(size_t)i * HM_POLICY_VALUES + k] = f32_to_f16_rn((float)((int)(r % 8001) - 4000) * 0.001f);
            out.piB[(size_t)i * HM_POLICY_VALUES + k] = f32_to_f16_rn((float)((int)((r >> 32) % 8001) - 4000) * 0.001f);
        }
    }
}

inline void hash_evaluator(const uint16_t* planes, int n, EvalOutputs& out) { hash_evaluator_salted(planes, n, 0, out); }

// ---- common/utils.h:127-167,226-243 -------------------------------------------------------
inline std::vector<float> normalize_logits(const std::vector<float>& logits, int exp_mode) {
    std::vector<float> prob(logits.size(), 0.0f);
    float maxLogit = -INFINITY;
    for (float l : logits) if (std::isfinite(l)) maxLogit = std::max(maxLogit, l);
    if (!std::isfinite(maxLogit)) {
        if (!prob.empty()) std::fill(prob.begin(), prob.end(), 1.0f / prob.size());
        return prob;
    }
    double sum = 0.0;
    for (size_t i = 0; i < logits.size(); ++i)
        if (std::isfinite(logits[i])) {
            prob[i] = exp_mode ? portable_expf(logits[i] - maxLogit) : std::exp(logits[i] - maxLogit);
            sum += prob[i];
        }
    if (!std::isfinite(sum) || sum <= 0.0) {
        size_t fc = std::count_if(logits.begin(), logits.end(), [](float v) { return std::isfinite(v); });
        if (fc > 0) for (size_t i = 0; i < logits.size(); ++i) prob[i] = std::isfinite(logits[i]) ? 1.0f / fc : 0.0f;
        return prob;
    }
    for (float& p : prob) p = static_cast<float>(p / sum);
    return prob;
}
inline std::vector<float> get_normalized_probability(const uint16_t* policy, const std::vector<Move>& actions, int stm, int exp_mode) {
    std::vector<float> logits(actions.size());
    for (size_t i = 0; i < actions.size(); ++i) {
        int idx = policy_index(actions[i], stm);
        logits[i] = idx >= 0 ? f16_to_f32(policy[idx]) : -INFINITY;
    }
    return normalize_logits(logits, exp_mode);
}
inline bool is_policy_move_representable(Move m) {   // utils.h:169-182
    if (m == MOVE_NONE) return true;
    if (type_of(m) == PROMOTION) { int pt = promo_type(m); if (pt == ROOK || pt == BISHOP) return false; }
    return true;
}

// ---- searchthread.cc:21-139 ----------------------------------------------------------------
enum class TerminalOutcome : uint8_t { NONE, WIN, LOSS, DRAW };

inline std::vector<Move> immediate_mates_on_board(Board& board, int b, int victimTeam, bool victimAdv) {   // :21-39
    std::vector<Move> mates;
    for (Move m : board.legal_moves(b)) {
        if (!board.pos[b].gives_check(m)) continue;
        board.push_move(b, m);
        bool mate = board.is_checkmate(victimTeam, victimAdv);
        board.pop_move(b);
        if (mate) mates.push_back(m);
    }
    return mates;
}
inline bool has_unavoidable_waiting_board_mate(Board& board, int team, bool adv, int searchPly) {   // :41-97
    const bool aOn = board.pos[0].stm == team, bOn = board.pos[1].stm == (team ^ 1);
    if (aOn == bOn) return false;
    const int active = aOn ? 0 : 1, waiting = 1 - active;
    std::vector<Move> mating = immediate_mates_on_board(board, waiting, team, adv);
    if (mating.empty()) return false;
    std::vector<Move> replies = board.legal_moves(active);
    if (adv) replies.push_back(MOVE_NONE);
    if (replies.empty()) return false;
    for (Move reply : replies) {
        if (reply != MOVE_NONE) board.push_move(active, reply);
        bool persists = false;
        if (!board.is_checkmate(team ^ 1, !adv) && !board.is_draw(searchPly + 1)) {
            for (Move mm : mating) {
                if (!board.is_legal_move(waiting, mm)) continue;
                board.push_move(waiting, mm);
                persists = board.is_checkmate(team, adv);
                board.pop_move(waiting);
                if (persists) break;
            }
        }
        if (reply != MOVE_NONE) board.pop_move(active);
        if (!persists) return false;
    }
    return true;
}
inline TerminalOutcome classify_terminal_position(Board& board, int teamToPlay, int rootTeam, bool rootAdv, int searchPly, int* endInPly) {   // :101-139
    *endInPly = 0;
    const bool adv = teamToPlay == rootTeam ? rootAdv : !rootAdv;
    if (board.is_checkmate(teamToPlay ^ 1, !adv)) { *endInPly = 1; return TerminalOutcome::WIN; }
    if (board.is_checkmate(teamToPlay, adv)) { *endInPly = 1; return TerminalOutcome::LOSS; }
    if (board.is_draw(searchPly)) return TerminalOutcome::DRAW;
    if (searchPly > 0 && has_unavoidable_waiting_board_mate(board, teamToPlay, adv, searchPly)) { *endInPly = 3; return TerminalOutcome::LOSS; }
    return TerminalOutcome::NONE;
}

// ---- SearchThread + Agent(1) ---------------------------------------------------------------
struct TrajectoryEntry { std::shared_ptr<Node> node; Candidate action; int selectedChildIdx; };
struct LeafContext {
    std::shared_ptr<Node> leaf;
    std::vector<TrajectoryEntry> trajectory;
    int teamToPlay = 0;
    bool sitPlaneActive = false, isTerminal = false, hasReservation = false;
    float terminalValue = 0.0f;
    uint64_t leafHash = 0;
};
struct RootEdge { Move moveA, moveB; int visits; float q, prior; };

class Search {
public:
    SearchConfig cfg;
    Evaluator evaluator;
    std::shared_ptr<Node> root;
    std::unordered_map<uint64_t, std::shared_ptr<Node>> tt;
    int nodesSearched = 0, sameBatchCollisions = 0, reservationCollisions = 0, evalCalls = 0, evalRows = 0, nodeCounter = 0, ttHits = 0;
    // optional trace of every evaluated leaf hash, for step-by-step diffing against the GPU engine
    std::vector<uint64_t> evalTrace;
    // optional per-attempt event log of collect_batch: (collect# << 32) | (code << 24) | (path length << 8) | outcome
    std::vector<uint64_t> ctxTrace;
    uint64_t collectSeq = 0;
    void ev(int code, int len, int outcome) { ctxTrace.push_back((collectSeq << 32) | ((uint64_t)code << 24) | ((uint64_t)(len & 0xffff) << 8) | (uint64_t)(outcome & 0xff)); }

    struct Batch {
        std::vector<uint16_t> obs;
        std::vector<LeafContext> contexts;
        int validInferenceCount = 0;
    };
    Batch batches[2];
    int pendingBatchIndex = -1;
    std::vector<TrajectoryEntry> trajectory;
    int rootTeam = 0;

    void unmake_trajectory(Board& board) {
        for (auto it = trajectory.rbegin(); it != trajectory.rend(); ++it)
            if (it->action.moveA != MOVE_NONE || it->action.moveB != MOVE_NONE) board.unmake_moves(it->action.moveA, it->action.moveB);
    }
    void cancel_virtual_losses(const std::vector<TrajectoryEntry>& tr) {   // searchthread.cc:241-247
        for (const auto& e : tr)
            if (e.selectedChildIdx >= 0) { e.node->virtualLoss[e.selectedChildIdx]--; e.node->virtualVisitSum--; }
    }

    void backup(std::vector<TrajectoryEntry>& tr, float v) {   // :197-239
        NodeType childType = tr.empty() ? NodeType::UNSOLVED : tr.back().node->nodeType;
        if (childType == NodeType::WIN) v = 1.0f;
        else if (childType == NodeType::LOSS) v = -1.0f;
        else if (childType == NodeType::DRAW) v = tr.back().node->team == tr.front().node->team ? -cfg.drawContempt : cfg.drawContempt;
        for (auto it = tr.rbegin(); it != tr.rend(); ++it) {
            Node* node = it->node.get();
            int idx = it->selectedChildIdx;
            if (idx >= 0) {
                node->update_and_remove_virtual_loss(idx, v);
                if (childType != NodeType::UNSOLVED) {
                    node->init_child_node_types();
                    node->update_child_node_type(idx, childType);
                    childType = node->nodeType;
                } else childType = NodeType::UNSOLVED;
            } else node->update_terminal(v);
            v = -v;
        }
    }

    struct Selection { std::shared_ptr<Node> child; int idx = -1; bool reserved = false; std::shared_ptr<Node> pending; };
    Selection select_child_and_apply_virtual_loss(Node& n) {   // node.cc:6-119
        size_t numExpanded = (size_t)n.expandedCount;
        if (numExpanded == 0 || n.children.empty()) return {};
        int visits = n.visits + n.virtualVisitSum;
        const float sqrtVisits = std::sqrt(static_cast<float>(visits));
        const float c = get_cpuct(static_cast<float>(visits), cfg.cpuctInit, cfg.cpuctBase);
        const float explorationBase = c * sqrtVisits;
        const size_t limit = std::min(numExpanded, n.children.size());
        float visitedPolicySum = 0.0f;
        if (cfg.enableDynamicFpu && visits > 0)
            for (size_t i = 0; i < limit; ++i)
                if (n.childVisits[i] + n.virtualLoss[i] > 0) visitedPolicySum += n.childPriors[i];
        const float parentQ = visits > 0 ? (n.valueSum / static_cast<float>(visits)) : 0.0f;
        const float fpuQ = cfg.enableDynamicFpu && visits > 0
            ? std::min(1.0f, std::max(-1.0f, parentQ - cfg.fpuReduction * std::sqrt(std::max(0.0f, visitedPolicySum))))
            : Q_INIT;
        const bool hasNonLosing = n.nodeType == NodeType::UNSOLVED
            && std::any_of(n.children.begin(), n.children.begin() + limit, [](const std::shared_ptr<Node>& ch) { return ch && ch->nodeType != NodeType::WIN; });
        std::vector<uint8_t> unavailable;
        std::shared_ptr<Node> pending;
        while (true) {
            float bestScore = -INFINITY;
            std::shared_ptr<Node> best;
            int sel = -1;
            for (size_t i = 0; i < limit; ++i) {
                if (!n.children[i] || (!unavailable.empty() && unavailable[i])) continue;
                if (hasNonLosing && n.children[i]->nodeType == NodeType::WIN) continue;
                const int vl = n.virtualLoss[i];
                const uint32_t ni = (uint32_t)n.childVisits[i], ne = ni + (uint32_t)vl;
                float q;
                if (ne == 0) q = fpuQ;
                else if (vl == 0) q = n.qValues[i];
                else q = (n.childValueSum[i] - static_cast<float>(vl)) / static_cast<float>(ne);   // VIRTUAL_LOSS style
                const float u = explorationBase * n.childPriors[i] / (1.0f + static_cast<float>(ne));
                const float score = q + u;
                if (score > bestScore) { bestScore = score; best = n.children[i]; sel = (int)i; }
            }
            if (sel < 0) return {nullptr, -1, false, pending};
            bool reserved = false;
            if (!best->isExpanded && best->nodeType == NodeType::UNSOLVED) {
                if (!best->try_reserve()) {
                    pending = best;
                    if (unavailable.empty()) unavailable.resize(limit, 0);
                    unavailable[sel] = 1;
                    continue;
                }
                reserved = true;
            }
            n.virtualLoss[sel]++; n.virtualVisitSum++;
            return {best, sel, reserved, nullptr};
        }
    }

    struct Canon { bool expanded = false; std::shared_ptr<Node> pending; };
    Canon canonicalize_child(Board& board, Node* parent, int idx, const Candidate& action, std::shared_ptr<Node>& child, bool& reserved, bool rootAdv) {   // :741-806
        if (!cfg.enableTranspositions) return {child->isExpanded, nullptr};
        if (child->hash != 0) return {child->isExpanded, nullptr};
        const bool childAdv = child->team == rootTeam ? rootAdv : !rootAdv;
        const uint64_t h = board.hash_key(childAdv);
        child->hash = h;
        std::shared_ptr<Node> canonical;
        auto it = tt.find(h);
        if (it != tt.end()) { canonical = it->second; ttHits++; } else { tt.emplace(h, child); canonical = child; }   // insertOrGet transposition_table.h:83-103
        const bool isAncestor = std::any_of(trajectory.begin(), trajectory.end(), [&](const TrajectoryEntry& e) { return e.node.get() == canonical.get(); });
        const bool teamMismatch = canonical->team != child->team;
        if (canonical == child || isAncestor || teamMismatch) return {child->isExpanded, nullptr};
        if (reserved) { child->release(); reserved = false; }
        if (idx >= 0 && (size_t)idx < parent->children.size()) parent->children[idx] = canonical;
        child = canonical;
        if (child->isExpanded) return {true, nullptr};
        if (child->nodeType != NodeType::UNSOLVED) return {false, nullptr};
        if (!child->try_reserve()) {
            parent->virtualLoss[idx]--; parent->virtualVisitSum--;
            board.unmake_moves(action.moveA, action.moveB);
            return {false, child};
        }
        reserved = true;
        return {false, nullptr};
    }

    struct LeafSel { std::shared_ptr<Node> leaf; bool reserved = false; std::shared_ptr<Node> pending; };
    LeafSel select_and_expand(Board& board, bool rootAdv) {   // :818-916
        std::shared_ptr<Node> cur = root, next;
        int childIdx;
        bool reserved = false;
        trajectory.push_back({cur, Candidate(), -1});
        while (true) {
            if (cur->nodeType != NodeType::UNSOLVED) break;
            if (!cur->isExpanded) {
                if (!reserved) {
                    if (!cur->try_reserve()) return {nullptr, false, cur};
                    reserved = true;
                }
                break;
            }
            if (cur->should_expand_new_child(cfg)) {
                Candidate act;
                bool childReserved = false;
                next = cur->expand_next_joint_child(nullptr, 0, act, &childIdx, true, &childReserved, nodeCounter);
                if (next) {
                    board.make_moves(act.moveA, act.moveB);
                    Canon cr = canonicalize_child(board, cur.get(), childIdx, act, next, childReserved, rootAdv);
                    if (cr.pending) return {nullptr, false, cr.pending};
                    trajectory.back().selectedChildIdx = childIdx;
                    trajectory.push_back({next, act, -1});
                    if (cr.expanded) { cur = next; reserved = false; continue; }
                    return {next, childReserved, nullptr};
                }
            }
            Selection s = select_child_and_apply_virtual_loss(*cur);
            if (!s.child || s.idx < 0) return {nullptr, false, s.pending};
            next = s.child; childIdx = s.idx; reserved = s.reserved;
            Candidate act = cur->gen.generated[childIdx];
            board.make_moves(act.moveA, act.moveB);
            Canon cr = canonicalize_child(board, cur.get(), childIdx, act, next, reserved, rootAdv);
            if (cr.pending) return {nullptr, false, cr.pending};
            trajectory.back().selectedChildIdx = childIdx;
            trajectory.push_back({next, act, -1});
            cur = next;
        }
        return {cur, reserved, nullptr};
    }

    void collect_batch(Batch& batch, Board& board, bool rootAdv) {   // :255-442
        const int B = cfg.batchSize;
        batch.contexts.clear();
        batch.obs.assign((size_t)B * HM_PLANE_VALUES, 0);
        batch.validInferenceCount = 0;
        const int maxAttempts = B * 2;
        int attempts = 0;
        collectSeq++;
        while ((int)batch.contexts.size() < B && attempts < maxAttempts) {
            attempts++;
            LeafContext ctx;
            trajectory.clear();
            LeafSel sel = select_and_expand(board, rootAdv);
            if (!sel.leaf) {
                ev(1, (int)trajectory.size(), 0);
                reservationCollisions++;
                cancel_virtual_losses(trajectory);
                unmake_trajectory(board);
                continue;
            }
            bool collision = std::any_of(batch.contexts.begin(), batch.contexts.end(), [&](const LeafContext& p) { return p.leaf == sel.leaf; });
            if (collision) {
                ev(2, (int)trajectory.size(), 0);
                sameBatchCollisions++;
                if (sel.reserved) sel.leaf->release();
                cancel_virtual_losses(trajectory);
                unmake_trajectory(board);
                continue;
            }
            ctx.trajectory = trajectory;
            ctx.leaf = sel.leaf;
            ctx.hasReservation = sel.reserved;
            ctx.teamToPlay = sel.leaf->team;
            ctx.sitPlaneActive = (ctx.teamToPlay == rootTeam) == rootAdv;
            int searchPly = (int)trajectory.size() - 1;
            NodeType solved = ctx.leaf->nodeType;
            auto drawValue = [&]() { return ctx.teamToPlay == rootTeam ? -cfg.drawContempt : cfg.drawContempt; };
            if (solved != NodeType::UNSOLVED) {
                ev(3, (int)trajectory.size(), (int)solved);
                ctx.isTerminal = true;
                ctx.terminalValue = solved == NodeType::WIN ? 1.0f : solved == NodeType::LOSS ? -1.0f : drawValue();
                batch.contexts.push_back(std::move(ctx));
                unmake_trajectory(board);
                continue;
            }
            int endInPly = 0;
            TerminalOutcome to = classify_terminal_position(board, ctx.teamToPlay, rootTeam, rootAdv, searchPly, &endInPly);
            if (to != TerminalOutcome::NONE) {
                ev(4, (int)trajectory.size(), (int)to | (endInPly << 4));
                ctx.isTerminal = true;
                if (to == TerminalOutcome::WIN) { ctx.terminalValue = 1.0f; ctx.leaf->mark_as_win(endInPly); }
                else if (to == TerminalOutcome::LOSS) { ctx.terminalValue = -1.0f; ctx.leaf->mark_as_loss(endInPly); }
                else { ctx.terminalValue = drawValue(); ctx.leaf->mark_as_draw(1); }
                batch.contexts.push_back(std::move(ctx));
                unmake_trajectory(board);
                continue;
            }
            ctx.isTerminal = false;
            if (!ctx.hasReservation) {
                ev(5, (int)trajectory.size(), 0);
                reservationCollisions++;
                cancel_virtual_losses(trajectory);
                unmake_trajectory(board);
                continue;
            }
            bool leafAdv = ctx.teamToPlay == rootTeam ? rootAdv : !rootAdv;
            ctx.leafHash = board.hash_key(leafAdv);
            hm_board cb;
            board.to_compact(&cb, ctx.teamToPlay, ctx.sitPlaneActive);
            planes_f16(cb, batch.obs.data() + (size_t)batch.validInferenceCount * HM_PLANE_VALUES);
            ev(6, (int)trajectory.size(), 0);
            batch.validInferenceCount++;
            batch.contexts.push_back(std::move(ctx));
            unmake_trajectory(board);
        }
    }

    bool try_init_and_expand(Node& leaf, const std::vector<Move>& aA, const std::vector<Move>& aB, const std::vector<float>& pA,
                             const std::vector<float>& pB, bool adv, bool aOn, bool bOn, const std::vector<uint8_t>& cA, const std::vector<uint8_t>& cB) {   // node.h:269-342
        if (leaf.isExpanded) return false;
        std::vector<float> rA = pA, rB = pB;
        if (leaf.depth == 0 && cfg.rootDirichletAlpha > 0.0f && cfg.rootDirichletEpsilon > 0.0f) {
            auto noise = [&](std::vector<float>& pri, uint64_t salt) {
                if (pri.size() <= 1) return;
                std::mt19937_64 eng(cfg.rootNoiseSeed ^ leaf.hash ^ salt);
                std::gamma_distribution<float> gamma(cfg.rootDirichletAlpha, 1.0f);
                std::vector<float> nz(pri.size());
                float total = 0.0f;
                for (float& s : nz) { s = gamma(eng); total += s; }
                if (total <= 0.0f) return;
                const float eps = std::min(1.0f, std::max(0.0f, cfg.rootDirichletEpsilon));
                for (size_t i = 0; i < pri.size(); ++i) pri[i] = (1.0f - eps) * pri[i] + eps * nz[i] / total;
            };
            noise(rA, 0x9e3779b97f4a7c15ULL);
            noise(rB, 0xbf58476d1ce4e5b9ULL);
        }
        leaf.gen.initialize(aA, aB, rA, rB, adv, aOn, bOn, cA, cB, cfg.tie_mode);
        leaf.expandedCount = 0;
        if (leaf.gen.hasNext()) {
            Candidate c = leaf.gen.getNext();
            auto child = std::make_shared<Node>(leaf.team ^ 1);
            child->id = nodeCounter++;
            child->depth = leaf.depth + 1;
            leaf.childValueSum.push_back(Q_INIT); leaf.childPriors.push_back(c.jointPrior);
            leaf.childVisits.push_back(0); leaf.virtualLoss.push_back(0);
            leaf.children.push_back(child); leaf.qValues.push_back(Q_INIT);
            leaf.expandedCount++;
            leaf.isExpanded = true;
            return true;
        }
        return false;
    }

    float shape_value(uint16_t valueH, const uint16_t* wdl, uint16_t mlH) {   // searchthread.cc:569-619
        const float bv = f16_to_f32(valueH);
        float scalar = std::isfinite(bv) ? std::min(1.0f, std::max(-1.0f, bv)) : 0.0f;
        float nv = scalar;
        if (cfg.enableWdlEval && wdl) {
            const float l = f16_to_f32(wdl[0]), d = f16_to_f32(wdl[1]), w = f16_to_f32(wdl[2]);
            if (std::isfinite(l) && std::isfinite(d) && std::isfinite(w)) {
                const float mx = std::max(l, std::max(d, w));
                auto ex = [&](float x) { return cfg.exp_mode ? portable_expf(x) : std::exp(x); };
                const float el = ex(l - mx), ed = ex(d - mx), ew = ex(w - mx);
                const float sum = el + ed + ew;
                if (std::isfinite(sum) && sum > 0.0f) {
                    const float pl = el / sum, pd = ed / sum, pw = ew / sum;
                    const float wv = pw - pl - cfg.drawContempt * pd;
                    const float ww = std::min(1.0f, std::max(0.0f, cfg.wdlValueWeight));
                    nv = (1.0f - ww) * scalar + ww * wv;
                }
            }
        }
        if (cfg.movesLeftDiscount > 0.0f) {
            const float np = std::min(1.0f, std::max(0.0f, f16_to_f32(mlH)));
            const float disc = std::min(1.0f, std::max(0.0f, cfg.movesLeftDiscount));
            nv *= 1.0f - disc * np;
        }
        return std::min(1.0f, std::max(-1.0f, nv));
    }

    void process_batch(Batch& batch, Board& board, bool rootAdv, const EvalOutputs* outs) {   // :444-639
        int inf = 0;
        for (auto& ctx : batch.contexts) {
            if (ctx.isTerminal) {
                if (ctx.hasReservation) ctx.leaf->release();
                backup(ctx.trajectory, ctx.terminalValue);
                continue;
            }
            if (ctx.leaf->nodeType != NodeType::UNSOLVED) {
                ctx.leaf->release();
                backup(ctx.trajectory, 0.0f);
                inf++;
                continue;
            }
            for (const auto& e : ctx.trajectory)
                if (e.action.moveA != MOVE_NONE || e.action.moveB != MOVE_NONE) board.make_moves(e.action.moveA, e.action.moveB);
            const uint16_t* piA = outs->piA.data() + (size_t)inf * HM_POLICY_VALUES;
            const uint16_t* piB = outs->piB.data() + (size_t)inf * HM_POLICY_VALUES;
            const bool leafAdv = ctx.teamToPlay == rootTeam ? rootAdv : !rootAdv;
            std::vector<Move> aA, aB;
            const bool aOn = board.pos[0].stm == ctx.teamToPlay, bOn = board.pos[1].stm == (ctx.teamToPlay ^ 1);
            auto filtered = [&](int b) {
                std::vector<Move> v = board.legal_moves(b);
                v.erase(std::remove_if(v.begin(), v.end(), [](Move m) { return !is_policy_move_representable(m); }), v.end());
                return v;
            };
            if (aOn) aA = filtered(0);
            if (bOn) aB = filtered(1);
            std::vector<float> pA, pB;
            if (aA.empty()) { aA.push_back(MOVE_NONE); pA.push_back(1.0f); }
            else { aA.push_back(MOVE_NONE); pA = get_normalized_probability(piA, aA, board.pos[0].stm, cfg.exp_mode); }
            if (aB.empty()) { aB.push_back(MOVE_NONE); pB.push_back(1.0f); }
            else { aB.push_back(MOVE_NONE); pB = get_normalized_probability(piB, aB, board.pos[1].stm, cfg.exp_mode); }
            auto caps = [&](const std::vector<Move>& v, int b) {
                std::vector<uint8_t> c;
                for (Move m : v) c.push_back(m != MOVE_NONE && board.pos[b].is_capture(m) ? 1 : 0);
                return c;
            };
            const std::vector<uint8_t> cA = caps(aA, 0), cB = caps(aB, 1);
            if (ctx.leafHash != 0) ctx.leaf->hash = ctx.leafHash;       // expand_leaf_node :929-946
            try_init_and_expand(*ctx.leaf, aA, aB, pA, pB, leafAdv, aOn, bOn, cA, cB);
            ctx.leaf->release();
            const float nv = shape_value(outs->value[inf], outs->wdl.data() + (size_t)inf * 3, outs->movesLeft[inf]);
            backup(ctx.trajectory, nv);
            for (auto it = ctx.trajectory.rbegin(); it != ctx.trajectory.rend(); ++it)
                if (it->action.moveA != MOVE_NONE || it->action.moveB != MOVE_NONE) board.unmake_moves(it->action.moveA, it->action.moveB);
            inf++;
        }
        nodesSearched += (int)batch.contexts.size();
        batch.contexts.clear();
        batch.validInferenceCount = 0;
    }
    void abort_batch(Batch& batch) {   // :641-659
        int done = 0;
        for (auto& ctx : batch.contexts) {
            if (ctx.hasReservation) ctx.leaf->release();
            if (ctx.isTerminal) { backup(ctx.trajectory, ctx.terminalValue); done++; }
            else cancel_virtual_losses(ctx.trajectory);
        }
        nodesSearched += done;
        batch.contexts.clear();
        batch.validInferenceCount = 0;
    }
    void run_eval(Batch& b, EvalOutputs& out) {
        evaluator(b.obs.data(), b.validInferenceCount, out);
        evalCalls++; evalRows += b.validInferenceCount;
        for (auto& c : b.contexts) if (!c.isTerminal) evalTrace.push_back(c.leafHash);
    }
    void run_iteration(Board& board, bool rootAdv) {   // :661-708
        if (pendingBatchIndex < 0) {
            Batch& b0 = batches[0];
            collect_batch(b0, board, rootAdv);
            if (b0.validInferenceCount == 0) { process_batch(b0, board, rootAdv, nullptr); return; }
            pendingBatchIndex = 0;
        }
        const int done = pendingBatchIndex, look = 1 - done;
        collect_batch(batches[look], board, rootAdv);
        EvalOutputs out;
        run_eval(batches[done], out);
        pendingBatchIndex = -1;
        process_batch(batches[done], board, rootAdv, &out);
        if (batches[look].validInferenceCount == 0) { process_batch(batches[look], board, rootAdv, nullptr); return; }
        pendingBatchIndex = look;
    }
    void finish_pending_iteration(Board& board, bool rootAdv) {   // :710-726
        if (pendingBatchIndex < 0) return;
        const int done = pendingBatchIndex;
        pendingBatchIndex = -1;
        EvalOutputs out;
        run_eval(batches[done], out);
        process_batch(batches[done], board, rootAdv, &out);
    }
    void discard_pending_iteration() {   // :728-739
        if (pendingBatchIndex < 0) return;
        const int done = pendingBatchIndex;
        pendingBatchIndex = -1;
        abort_batch(batches[done]);
    }

    // find_immediate_root_mate (agent.cc:136-238)
    bool find_immediate_root_mate(Board& board, int team, bool adv, Candidate& out) {
        const bool aOn = board.pos[0].stm == team, bOn = board.pos[1].stm == (team ^ 1);
        std::vector<Move> aA, aB;
        if (aOn) aA = board.legal_moves(0);
        if (bOn) aB = board.legal_moves(1);
        const JointActionRules rules{aOn, bOn, adv, !aA.empty(), !aB.empty()};
        const bool aChk = board.pos[0].checkers != 0, bChk = board.pos[1].checkers != 0;
        auto part = [&](int b, std::vector<Move>& v) { std::stable_partition(v.begin(), v.end(), [&](Move m) { return board.pos[b].gives_check(m); }); };
        if (aOn) part(0, aA);
        if (bOn) part(1, aB);
        if (aOn)
            for (size_t i = 0; i < aA.size(); ++i) {
                const Move m = aA[i];
                if (!aChk && !bChk && !board.pos[0].gives_check(m)) continue;
                const bool cap = board.pos[0].is_capture(m);
                if (!bOn || is_single_pass_legal(adv, aOn, bOn, cap)) {
                    board.push_move(0, m);
                    const bool mate = board.is_checkmate(team ^ 1, !adv);
                    board.pop_move(0);
                    if (mate) { out = Candidate(m, 1.0f, i, MOVE_NONE, 1.0f, 0, rules, cap, false); return true; }
                }
            }
        if (bOn)
            for (size_t i = 0; i < aB.size(); ++i) {
                const Move m = aB[i];
                if (!aChk && !bChk && !board.pos[1].gives_check(m)) continue;
                const bool cap = board.pos[1].is_capture(m);
                if (!aOn || is_single_pass_legal(adv, aOn, bOn, cap)) {
                    board.push_move(1, m);
                    const bool mate = board.is_checkmate(team ^ 1, !adv);
                    board.pop_move(1);
                    if (mate) { out = Candidate(MOVE_NONE, 1.0f, 0, m, 1.0f, i, rules, false, cap); return true; }
                }
            }
        if (aOn && bOn)
            for (size_t i = 0; i < aA.size(); ++i) {
                const Move mA = aA[i];
                const bool chkA = board.pos[0].gives_check(mA), capA = board.pos[0].is_capture(mA);
                for (size_t j = 0; j < aB.size(); ++j) {
                    const Move mB = aB[j];
                    if (!aChk && !bChk && !chkA && !board.pos[1].gives_check(mB)) continue;
                    const bool capB = board.pos[1].is_capture(mB);
                    board.make_moves(mA, mB);
                    const bool mate = board.is_checkmate(team ^ 1, !adv);
                    board.unmake_moves(mA, mB);
                    if (mate) { out = Candidate(mA, 1.0f, i, mB, 1.0f, j, rules, capA, capB); return true; }
                }
            }
        return false;
    }

    // Agent::run_search node-budget path (agent.cc:421-558, 331-352, 808-839). Returns false when the
    // position is terminal / has no action ("bestmove (none)").
    // Tree reuse between searches (Agent::try_reuse_tree / store_next_root_candidates, agent.cc:1345-1451; ENABLE_TREE_REUSE,
    // search_params.h:194).  Off by default: self-play and tournaments call Agent::reset_search_state before every search
    // (selfplay.cc:653, tournament.cc:383); the UCI front end keeps the candidates from one `go` to the next.
    struct Retained { std::shared_ptr<Node> node; uint64_t positionHash; std::string signature; };
    std::vector<Retained> nextRootCandidates;
    bool enableTreeReuse = false;
    int reusedVisits = -1;                       // visits of the recovered root ("info string Tree reuse: N visits recovered"), -1 = fresh root
    static std::string board_signature(Board& b) { return b.pos[0].fen() + "|" + b.pos[1].fen(); }
    void reset_search_state() { root.reset(); nextRootCandidates.clear(); tt.clear(); }   // agent.cc:403-412
    std::shared_ptr<Node> try_reuse_tree(uint64_t positionHash, int team, const std::string& signature) {
        std::shared_ptr<Node> reused;
        for (const Retained& c : nextRootCandidates)
            if (c.node && c.positionHash == positionHash && c.node->team == team && !c.signature.empty() && c.signature == signature) { reused = c.node; break; }
        nextRootCandidates.clear();
        return reused;
    }
    void store_next_root_candidates(Board& board, bool adv) {
        nextRootCandidates.clear();
        if (!root || !root->isExpanded) return;
        if (root->children.empty() || root->childVisits.empty()) return;
        int bestIdx = root->get_best_move_idx_with_q_weight(cfg.qVetoDelta, cfg.qValueWeight);
        if (bestIdx < 0) {
            int maxVisits = 0;
            for (size_t i = 0; i < root->childVisits.size(); ++i) if (root->childVisits[i] > maxVisits) { maxVisits = root->childVisits[i]; bestIdx = (int)i; }
        }
        if (bestIdx < 0 || (size_t)bestIdx >= root->children.size()) return;
        const Candidate own = root->gen.generated[bestIdx];
        Board ownNext(board);
        ownNext.make_moves(own.moveA, own.moveB);
        const std::shared_ptr<Node>& ownNextRoot = root->children[bestIdx];
        nextRootCandidates.push_back({ownNextRoot, ownNext.hash_key(!adv), board_signature(ownNext)});
        if (!ownNextRoot || !ownNextRoot->isExpanded || ownNextRoot->children.empty()) return;
        Board reply(ownNext);
        for (size_t i = 0; i < ownNextRoot->children.size(); ++i) {
            if (!ownNextRoot->children[i]) continue;
            const Candidate r = ownNextRoot->gen.generated[i];
            reply.make_moves(r.moveA, r.moveB);
            nextRootCandidates.push_back({ownNextRoot->children[i], reply.hash_key(adv), board_signature(reply)});
            reply.unmake_moves(r.moveA, r.moveB);
        }
    }

    bool run(Board& board, int team, bool adv, int targetNodes) {
        if (!enableTreeReuse) nextRootCandidates.clear();
        reusedVisits = -1;
        root.reset(); tt.clear();
        nodesSearched = 0; pendingBatchIndex = -1; rootTeam = team;
        const bool aOn = board.pos[0].stm == team, bOn = board.pos[1].stm == (team ^ 1);
        const bool canWait = is_double_sit_legal(adv, aOn, bOn);
        if (board.is_checkmate(team ^ 1, !adv) || board.is_checkmate(team, adv) || board.is_draw()) return false;
        {   // Board::legal_moves(side, adv).empty() && !canWait (agent.cc:448)
            bool any = false;
            if (!board.is_checkmate(team, adv)) {
                if (aOn && !board.legal_moves(0).empty()) any = true;
                if (bOn && !board.legal_moves(1).empty()) any = true;
            }
            if (!any && !canWait) return false;
        }
        Candidate mateAct;
        if (find_immediate_root_mate(board, team, adv, mateAct)) {   // agent.cc:455-499
            root = std::make_shared<Node>(team, board.hash_key(adv));
            root->id = nodeCounter++;
            SearchConfig saved = cfg;
            cfg.rootDirichletAlpha = 0.0f;
            try_init_and_expand(*root, {mateAct.moveA}, {mateAct.moveB}, {1.0f}, {1.0f}, adv, aOn, bOn,
                                {(uint8_t)(mateAct.moveA != MOVE_NONE && board.pos[0].is_capture(mateAct.moveA))},
                                {(uint8_t)(mateAct.moveB != MOVE_NONE && board.pos[1].is_capture(mateAct.moveB))});
            cfg = saved;
            if (!root->children.empty() && root->children[0]) {
                root->children[0]->mark_as_loss(0);
                root->init_child_node_types();
                root->update_child_node_type(0, NodeType::LOSS);
            }
            root->update(0, 1.0f);
            root->mark_as_win(1);
            if (enableTreeReuse) store_next_root_candidates(board, adv);
            return true;
        }
        const uint64_t positionHash = board.hash_key(adv);
        std::shared_ptr<Node> reused = enableTreeReuse ? try_reuse_tree(positionHash, team, board_signature(board)) : nullptr;
        if (reused) {                                   // agent.cc:514-525
            root = reused;
            root->hash = positionHash;
            root->depth = 0;
            reusedVisits = root->visits;
        } else {
            root = std::make_shared<Node>(team, positionHash);
            root->id = nodeCounter++;
        }
        if (cfg.enableTranspositions) tt.emplace(root->hash, root);
        while (nodesSearched < targetNodes) {
            if (root->nodeType != NodeType::UNSOLVED) break;
            run_iteration(board, adv);
        }
        if (root->nodeType != NodeType::UNSOLVED) discard_pending_iteration();
        else finish_pending_iteration(board, adv);
        if (enableTreeReuse) store_next_root_candidates(board, adv);
        return true;
    }

    std::vector<RootEdge> root_edge_stats() const {   // agent.cc:1004-1017
        std::vector<RootEdge> out;
        if (!root || !root->isExpanded) return out;
        size_t n = std::min(root->childVisits.size(), root->gen.generated.size());
        for (size_t i = 0; i < n; ++i)
            out.push_back({root->gen.generated[i].moveA, root->gen.generated[i].moveB, root->childVisits[i], root->qValues[i], root->childPriors[i]});
        return out;
    }
    float root_q() const { return root ? root->Q() : 0.0f; }
    // Final `info ... pv` lines of Agent::run_search (agent.cc:917-965): root children ordered by visit count (std::sort with the
    // reference's comparator), the solver-aware best move pulled to the front, up to multiPV lines; each line's PV follows
    // extract_pv_from_child (agent.cc:1218-1290): the root edge, then get_best_move_idx_with_q_weight (most-visited fallback)
    // through expanded nodes, maxDepth joint actions at most.
    struct PvLine { int childIdx, childType, childEndInPly; float q; std::vector<uint32_t> moves; /* moveA, moveB per depth */ };
    std::vector<PvLine> pv_lines(int multiPV, int maxDepth = 20) const {
        std::vector<PvLine> out;
        if (!root || !root->isExpanded) return out;
        const size_t numChildren = std::min(root->childVisits.size(), root->children.size());
        std::vector<size_t> order(numChildren);
        for (size_t i = 0; i < numChildren; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return root->childVisits[a] > root->childVisits[b]; });
        const int solverIdx = root->get_best_move_idx_with_q_weight(cfg.qVetoDelta, cfg.qValueWeight);
        if (solverIdx >= 0) {
            auto it = std::find(order.begin(), order.end(), (size_t)solverIdx);
            if (it != order.end() && it != order.begin()) { order.erase(it); order.insert(order.begin(), (size_t)solverIdx); }
        }
        const int numPVs = std::min(multiPV, (int)numChildren);
        for (int k = 0; k < numPVs; ++k) {
            const size_t ci = order[k];
            PvLine line;
            line.childIdx = (int)ci; line.q = root->qValues[ci];
            const Node* cur = root->children[ci].get();
            line.childType = cur ? (int)cur->nodeType : 0; line.childEndInPly = cur ? cur->endInPly : 0;
            line.moves.push_back(root->gen.generated[ci].moveA); line.moves.push_back(root->gen.generated[ci].moveB);
            for (int depth = 1; depth < maxDepth; ++depth) {
                if (!cur || !cur->isExpanded || cur->children.empty() || cur->childVisits.empty()) break;
                int best = cur->get_best_move_idx_with_q_weight(cfg.qVetoDelta, cfg.qValueWeight);
                if (best < 0) {
                    best = 0;
                    int maxVisits = 0;
                    for (size_t i = 0; i < cur->children.size() && i < cur->childVisits.size(); ++i)
                        if (cur->childVisits[i] > maxVisits) { maxVisits = cur->childVisits[i]; best = (int)i; }
                }
                line.moves.push_back(cur->gen.generated[best].moveA); line.moves.push_back(cur->gen.generated[best].moveB);
                cur = cur->children[best].get();
            }
            out.push_back(std::move(line));
        }
        return out;
    }
    // index of the joint action Agent::run_search returns (agent.cc:859-889)
    int best_move_index() const {
        if (!root || !root->isExpanded) return -1;
        if (root->childVisits.empty() || root->children.empty()) return -1;
        const size_t n = std::min(root->childVisits.size(), root->children.size());
        int best = root->get_best_move_idx_with_q_weight(cfg.qVetoDelta, cfg.qValueWeight);
        if (best < 0) {
            int maxVisits = 0;
            for (size_t i = 0; i < n; ++i) if (root->childVisits[i] > maxVisits) { maxVisits = root->childVisits[i]; best = (int)i; }
        }
        if ((size_t)best >= root->gen.generated.size()) best = 0;
        return best;
    }
};

}  // namespace hmo
