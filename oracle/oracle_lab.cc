// oracle/oracle_lab.cc — scripted access to the search restatement's Node / SearchThread pieces so the
// reference's own gtest known answers (engine/tests/test_move_gen.cc:280-291, 446-1296) can be replayed
// step by step (tests/golden/search_cases.json, tests/test_oracle_search_cases.py).
// TEST INFRASTRUCTURE ONLY; included by oracle_capi.cc.
#include <map>

namespace {

struct Lab {
    Search search;
    std::vector<std::shared_ptr<Node>> nodes;          // registry: id -> node
    int id_of(const std::shared_ptr<Node>& n) {
        if (!n) return -1;
        for (size_t i = 0; i < nodes.size(); ++i) if (nodes[i] == n) return (int)i;
        nodes.push_back(n);
        return (int)nodes.size() - 1;
    }
    Node& at(int id) { return *nodes.at((size_t)id); }
};
Lab& L(void* h) { return *static_cast<Lab*>(h); }

}  // namespace

extern "C" {

void* ora_lab_new(int tie_mode, int exp_mode) {
    Lab* l = new Lab();
    l->search.cfg.tie_mode = tie_mode; l->search.cfg.exp_mode = exp_mode;
    l->search.evaluator = hash_evaluator;
    return l;
}
void ora_lab_free(void* h) { delete static_cast<Lab*>(h); }
// SearchParams::RuntimeConfig fields by name (search_params.h:275-295)
int ora_lab_config(void* h, const char* key, double v) {
    SearchConfig& c = L(h).search.cfg;
    const std::string k = key;
    if (k == "pwCoefficient") c.pwCoefficient = (float)v;
    else if (k == "rootPwCoefficient") c.rootPwCoefficient = (float)v;
    else if (k == "pwExponent") c.pwExponent = (float)v;
    else if (k == "enableDynamicFpu") c.enableDynamicFpu = v != 0;
    else if (k == "fpuReduction") c.fpuReduction = (float)v;
    else if (k == "enableTranspositions") c.enableTranspositions = v != 0;
    else if (k == "drawContempt") c.drawContempt = (float)v;
    else if (k == "movesLeftDiscount") c.movesLeftDiscount = (float)v;
    else if (k == "enableWdlEval") c.enableWdlEval = v != 0;
    else if (k == "wdlValueWeight") c.wdlValueWeight = (float)v;
    else if (k == "cpuctInit") c.cpuctInit = (float)v;
    else if (k == "cpuctBase") c.cpuctBase = (float)v;
    else return -1;
    return 0;
}
double ora_lab_config_get(void* h, const char* key) {
    const SearchConfig& c = L(h).search.cfg;
    const std::string k = key;
    if (k == "pwCoefficient") return c.pwCoefficient;
    if (k == "rootPwCoefficient") return c.rootPwCoefficient;
    if (k == "pwExponent") return c.pwExponent;
    if (k == "enableDynamicFpu") return c.enableDynamicFpu;
    if (k == "fpuReduction") return c.fpuReduction;
    if (k == "enableTranspositions") return c.enableTranspositions;
    if (k == "drawContempt") return c.drawContempt;
    if (k == "movesLeftDiscount") return c.movesLeftDiscount;
    if (k == "wdlValueWeight") return c.wdlValueWeight;
    if (k == "cpuctInit") return c.cpuctInit;
    if (k == "cpuctBase") return c.cpuctBase;
    if (k == "qVetoDelta") return c.qVetoDelta;
    if (k == "qValueWeight") return c.qValueWeight;
    if (k == "batchSize") return c.batchSize;
    return NAN;
}

int ora_lab_node_new(void* h, int team, uint64_t hash) {
    auto n = std::make_shared<Node>(team, hash);
    n->id = L(h).search.nodeCounter++;
    return L(h).id_of(n);
}
int ora_lab_init_expand(void* h, int id, const uint32_t* a, int nA, const uint32_t* b, int nB, const float* pa, const float* pb,
                        int adv, int aOn, int bOn, const uint8_t* ca, const uint8_t* cb) {
    return L(h).search.try_init_and_expand(L(h).at(id), std::vector<Move>(a, a + nA), std::vector<Move>(b, b + nB), std::vector<float>(pa, pa + nA),
                                           std::vector<float>(pb, pb + nB), adv != 0, aOn != 0, bOn != 0,
                                           ca ? std::vector<uint8_t>(ca, ca + nA) : std::vector<uint8_t>(),
                                           cb ? std::vector<uint8_t>(cb, cb + nB) : std::vector<uint8_t>()) ? 1 : 0;
}
void ora_lab_update(void* h, int id, int idx, float v) { L(h).at(id).update((size_t)idx, v); }
void ora_lab_update_terminal(void* h, int id, float v) { L(h).at(id).update_terminal(v); }
void ora_lab_apply_vl(void* h, int id, int idx) { L(h).at(id).apply_virtual_loss(idx); }
void ora_lab_remove_vl(void* h, int id, int idx) { L(h).at(id).remove_virtual_loss(idx); }
void ora_lab_mark(void* h, int id, int type, int ply) {
    Node& n = L(h).at(id);
    if (type == 1) n.mark_as_win(ply); else if (type == 2) n.mark_as_loss(ply); else n.mark_as_draw(ply);
}
void ora_lab_set_value(void* h, int id, float v) { L(h).at(id).set_value(v); }
void ora_lab_set_depth(void* h, int id, int d) { L(h).at(id).depth = d; }
int ora_lab_child(void* h, int id, int idx) {
    Node& n = L(h).at(id);
    if (idx < 0 || (size_t)idx >= n.children.size()) return -1;
    return L(h).id_of(n.children[(size_t)idx]);
}
void ora_lab_replace_child(void* h, int id, int idx, int childId) { L(h).at(id).replace_child(idx, L(h).nodes.at((size_t)childId)); }
// Node::expand_next_joint_child (node.h:199-262); existing < 0: new node.  Returns child id or -1.
int ora_lab_expand_next(void* h, int id, int existing, uint64_t hash, int reserve, int* outIdx, int* outReserved, uint32_t* moveA, uint32_t* moveB) {
    Lab& l = L(h);
    Candidate act;
    int idx = -1;
    bool reserved = false;
    auto child = l.at(id).expand_next_joint_child(existing >= 0 ? l.nodes.at((size_t)existing) : nullptr, hash, act, &idx, reserve != 0, &reserved, l.search.nodeCounter);
    if (outIdx) *outIdx = idx;
    if (outReserved) *outReserved = reserved;
    if (moveA) *moveA = act.moveA;
    if (moveB) *moveB = act.moveB;
    return l.id_of(child);
}
int ora_lab_should_expand(void* h, int id) { return L(h).at(id).should_expand_new_child(L(h).search.cfg) ? 1 : 0; }
int ora_lab_has_unexpanded(void* h, int id) { return L(h).at(id).gen.hasNext() ? 1 : 0; }
void ora_lab_peek_next(void* h, int id, uint32_t* moveA, uint32_t* moveB, float* prior) {
    const Candidate c = L(h).at(id).gen.peekNext();
    *moveA = c.moveA; *moveB = c.moveB; *prior = c.jointPrior;
}
void ora_lab_joint_action(void* h, int id, int idx, uint32_t* moveA, uint32_t* moveB, float* priorA, float* priorB) {
    const Node& n = L(h).at(id);
    Candidate c;
    if (idx >= 0 && (size_t)idx < n.gen.generated.size()) c = n.gen.generated[(size_t)idx];
    *moveA = c.moveA; *moveB = c.moveB; *priorA = c.priorA; *priorB = c.priorB;
}
// Node::select_child_and_apply_virtual_loss (node.cc:6-119)
int ora_lab_select(void* h, int id, int* idx, int* reserved, int* pending) {
    Lab& l = L(h);
    Search::Selection s = l.search.select_child_and_apply_virtual_loss(l.at(id));
    *idx = s.idx; *reserved = s.reserved; *pending = l.id_of(s.pending);
    return l.id_of(s.child);
}
int ora_lab_reserve(void* h, int id) { return L(h).at(id).try_reserve() ? 1 : 0; }
void ora_lab_release(void* h, int id) { L(h).at(id).release(); }
void ora_lab_init_types(void* h, int id) { L(h).at(id).init_child_node_types(); }
int ora_lab_update_type(void* h, int id, int idx, int type) { return L(h).at(id).update_child_node_type(idx, (NodeType)type) ? 1 : 0; }
static std::vector<TrajectoryEntry> lab_traj(Lab& l, const int* ids, const int* idxs, int n) {
    std::vector<TrajectoryEntry> tr;
    for (int i = 0; i < n; ++i) tr.push_back({l.nodes.at((size_t)ids[i]), Candidate(), idxs[i]});
    return tr;
}
void ora_lab_backup(void* h, const int* ids, const int* idxs, int n, float v) {   // SearchThread::backup searchthread.cc:197-239
    auto tr = lab_traj(L(h), ids, idxs, n);
    L(h).search.backup(tr, v);
}
void ora_lab_cancel_vl(void* h, const int* ids, const int* idxs, int n) {         // :241-247
    L(h).search.cancel_virtual_losses(lab_traj(L(h), ids, idxs, n));
}
int ora_lab_best_move(void* h, int id, float qVeto, float qWeight) { return L(h).at(id).get_best_move_idx_with_q_weight(qVeto, qWeight); }
// field: 0 Q(), 1 visits, 2 nodeType, 3 endInPly, 4 children.size(), 5 team, 6 child_q[idx], 7 child_visits[idx], 8 virtualLoss[idx],
//        9 isExpanded, 10 evaluationPending, 11 expandedCount, 12 childPriors[idx], 13 valueSum, 14 depth, 15 virtualVisitSum
double ora_lab_get(void* h, int id, int field, int idx) {
    const Node& n = L(h).at(id);
    auto ok = [&](size_t sz) { return idx >= 0 && (size_t)idx < sz; };
    switch (field) {
        case 0: return n.Q();
        case 1: return n.visits;
        case 2: return (int)n.nodeType;
        case 3: return n.endInPly;
        case 4: return (double)n.children.size();
        case 5: return n.team;
        case 6: return n.get_child_q(idx);
        case 7: return ok(n.childVisits.size()) ? n.childVisits[(size_t)idx] : NAN;
        case 8: return ok(n.virtualLoss.size()) ? n.virtualLoss[(size_t)idx] : NAN;
        case 9: return n.isExpanded;
        case 10: return n.evaluationPending;
        case 11: return n.expandedCount;
        case 12: return ok(n.childPriors.size()) ? n.childPriors[(size_t)idx] : NAN;
        case 13: return n.valueSum;
        case 14: return n.depth;
        case 15: return n.virtualVisitSum;
    }
    return NAN;
}
// SearchThread::set_root_node / set_transposition_table + TranspositionTable::insertOrGet
void ora_lab_set_root(void* h, int id) { Lab& l = L(h); l.search.root = l.nodes.at((size_t)id); l.search.rootTeam = l.search.root->team; }
int ora_lab_tt_insert_or_get(void* h, uint64_t hash, int id) {
    Lab& l = L(h);
    auto it = l.search.tt.find(hash);
    if (it != l.search.tt.end()) { l.search.ttHits++; return l.id_of(it->second); }
    l.search.tt.emplace(hash, l.nodes.at((size_t)id));
    return id;
}
int ora_lab_tt_hits(void* h) { return L(h).search.ttHits; }
// SearchThread::select_and_expand (searchthread.cc:818-916): leaf id or -1
int ora_lab_select_and_expand(void* h, void* board, int rootAdv, int* reserved, int* pending, int* trajLen) {
    Lab& l = L(h);
    l.search.trajectory.clear();
    Search::LeafSel s = l.search.select_and_expand(*static_cast<Board*>(board), rootAdv != 0);
    *reserved = s.reserved; *pending = l.id_of(s.pending);
    if (trajLen) *trajLen = (int)l.search.trajectory.size();
    return l.id_of(s.leaf);
}
// Agent::store_next_root_candidates / try_reuse_tree (agent.cc:1345-1451) on the lab's root and board
void ora_lab_store_candidates(void* h, void* board, int adv) { L(h).search.store_next_root_candidates(*static_cast<Board*>(board), adv != 0); }
int ora_lab_retained_count(void* h) { return (int)L(h).search.nextRootCandidates.size(); }
int ora_lab_try_reuse(void* h, void* board, int adv, int team) {
    Lab& l = L(h);
    Board& b = *static_cast<Board*>(board);
    return l.id_of(l.search.try_reuse_tree(b.hash_key(adv != 0), team, Search::board_signature(b)));
}
// shape_value (searchthread.cc:569-619) on f32 heads rounded to fp16 like the engine's outputs
float ora_lab_shape_value(void* h, float value, const float* wdl, float movesLeft) {
    uint16_t w[3] = {0, 0, 0};
    if (wdl) for (int i = 0; i < 3; ++i) w[i] = f32_to_f16_rn(wdl[i]);
    return L(h).search.shape_value(f32_to_f16_rn(value), wdl ? w : nullptr, f32_to_f16_rn(movesLeft));
}

// ---- policy helpers (common/utils.h:127-167, 226-243) -----------------------------------------
void ora_normalize_logits(const float* logits, int n, int exp_mode, float* out) {
    auto p = normalize_logits(std::vector<float>(logits, logits + n), exp_mode);
    std::memcpy(out, p.data(), sizeof(float) * (size_t)n);
}
// get_normalized_probability for an f32 policy (f16 == 0) or an fp16 one (f16 == 1: `policy` holds uint16)
void ora_normalized_probability(const void* policy, int f16, const uint32_t* actions, int n, int stm, int exp_mode, float* out) {
    std::vector<Move> act(actions, actions + n);
    std::vector<float> p;
    if (f16) p = get_normalized_probability(static_cast<const uint16_t*>(policy), act, stm, exp_mode);
    else {
        std::vector<float> logits((size_t)n);
        const float* pol = static_cast<const float*>(policy);
        for (int i = 0; i < n; ++i) { const int idx = policy_index(act[(size_t)i], stm); logits[(size_t)i] = idx >= 0 ? pol[idx] : -INFINITY; }
        p = normalize_logits(logits, exp_mode);
    }
    std::memcpy(out, p.data(), sizeof(float) * (size_t)n);
}
int ora_is_policy_move_representable(uint32_t m) { return is_policy_move_representable(m) ? 1 : 0; }
int ora_policy_index_of_label(const char* label) {
    const auto& labels = T().policy_labels;
    for (size_t i = 0; i < labels.size(); ++i) if (labels[i] == label) return (int)i;
    return -1;
}
uint16_t ora_f32_to_f16(float f) { return f32_to_f16_rn(f); }
float ora_get_cpuct3(float visits, float init, float base) { return get_cpuct(visits, init, base); }
int ora_allowed_children3(int visits, float coef, float exponent) { return get_allowed_children(visits, coef, exponent); }

// ---- Board extras used by the scripted cases --------------------------------------------------
void ora_board_record_position(void* h, int b) { static_cast<Board*>(h)->record_position(b); }
void ora_board_add_to_hand(void* h, int b, int color, int pt) { static_cast<Board*>(h)->pos[b].add_to_hand(color, pt); }
int ora_board_count_in_hand(void* h, int b, int color, int pt) { return static_cast<Board*>(h)->pos[b].hand[color][pt]; }
int ora_board_rule50(void* h, int b) { return static_cast<Board*>(h)->pos[b].rule50; }
int ora_board_stm(void* h, int b) { return static_cast<Board*>(h)->pos[b].stm; }
int ora_board_history_len(void* h, int b) { return (int)static_cast<Board*>(h)->positionHistory[b].size(); }
int ora_board_prefix_len(void* h, int b) { return (int)static_cast<Board*>(h)->positionHistoryPrefixes[b].size(); }
uint32_t ora_board_last_move(void* h, int b) { return static_cast<Board*>(h)->last_move(b); }
int ora_board_is_legal_move(void* h, int b, uint32_t m) { return static_cast<Board*>(h)->is_legal_move(b, m) ? 1 : 0; }
// Stockfish::UCI::to_move semantic: the legal move whose UCI string matches (0 if none)
uint32_t ora_board_uci_to_move(void* h, int b, const char* uci) {
    Board& bd = *static_cast<Board*>(h);
    for (Move m : bd.legal_moves(b)) if (uci_of(bd.pos[b], m) == uci) return m;
    return 0;
}

}  // extern "C"
