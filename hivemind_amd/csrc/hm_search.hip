// hm_search.hip — the lockstep kernels (k_collect / k_process), the per-ply kernels (k_begin, k_root_stats, k_apply, ...) and the host
// side of the hm_sp_* C ABI.  The device functions of the search live in hm_search_device.hpp, the single-launch search
// (search workgroups + evaluator workgroups in one kernel) in hm_rollout.hip.
#include "hm_search_device.hpp"

namespace hms {

// Two waves per game: wave 0 walks the tree; wave 1 (another SIMD of the same CU) turns the hm_board images wave 0
// posts in LDS into fp16 planes, so the 9.4 KB plane writes overlap the next descent instead of extending it.
//
// The traversal is a chain of dependent reads (node -> edges -> child nodes -> ...), each an L2 round trip of several
// hundred cycles for a lone wave.  One block per CU leaves its 160 KB of LDS idle, so the launch first copies the game's
// whole node pool (64 B per node: 88 KB at nodes = 400) and its Game record into LDS with wide coalesced loads, walks the
// tree there, and writes both back at the end; edges, generator blocks and the transposition table stay in HBM/L2.
// Searches whose pool does not fit (prm.ldsNodes == 0) walk the pool in place.
// MIRROR (the node pool in LDS for the launch): a template parameter for the reason given at k_search
template <bool MIRROR>
__global__ __launch_bounds__(COLLECT_THREADS) void k_collect(Pools pl, Params prm, uint16_t* planesNext, int* rowsNext, int* activeCount) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *activeCount = 0;     // k_process of this iteration re-counts
    if (threadIdx.x == 0) {
        LegClock* ck = pl.clk;
        if (blockIdx.x == 0) {                                     // the previous k_process has finished: fold its interval
            const u64 pe = ck->pEnd, ps = ck->pStart;
            if (pe) { ck->sumP += pe - ps; ck->cntP++; }
            ck->pStart = ~0ULL; ck->pEnd = 0;
        }
        atomicMin(&ck->cStart, (u64)__builtin_amdgcn_s_memrealtime());
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char s_nodes[];
    __shared__ RulesTab s_rt;
    __shared__ WaveLds L;
    __shared__ __attribute__((aligned(16))) Game s_game;
    constexpr int TABN = 512;                                      // visits beyond this read the tables in HBM
    __shared__ float s_cpuct[TABN];
    __shared__ uint16_t s_pwRoot[TABN], s_pwNode[TABN];
    __shared__ u32 s_dirty[LDS_DIRTY_BITS / 32];                   // one bit per mirrored node (hm_sp_create_ex admits the mirror only for pools this small)
    static_assert(sizeof(Game) % 4 == 0 && sizeof(Node) == 64, "LDS mirrors are copied in 4 / 16 byte words");
    PROF_INIT();
    PROF_T(ta);
#ifdef HM_SEARCH_PROF
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
    G s = make_view(pl, prm, blockIdx.x);
    Game* const gGame = s.g;
    Node* const gNodes = s.nodes;
    for (unsigned i = threadIdx.x; i < sizeof(Game) / 4; i += COLLECT_THREADS) reinterpret_cast<u32*>(&s_game)[i] = reinterpret_cast<const u32*>(gGame)[i];
    stage_table_wide(&s_rt, pl.rules);
    {
        const bool alt = gGame->pwSel != 0;                      // this game's progressive-widening profile
        const int* pwr = alt ? pl.pwRootAlt : pl.pwRoot;
        const int* pwn = alt ? pl.pwNodeAlt : pl.pwNode;
        for (int i = threadIdx.x; i < TABN; i += COLLECT_THREADS) { s_cpuct[i] = pl.cpuctTab[i]; s_pwRoot[i] = (uint16_t)min(pwr[i], 65535); s_pwNode[i] = (uint16_t)min(pwn[i], 65535); }
    }
    if (threadIdx.x == 0) { L.posted = 0; L.done = 0; L.listWords = 0; L.servedCnt = 0; L.servedCntB = 0; L.postCount = 0; L.reqSeq = 0; L.typeSeq = 0; L.ackSeq = 0; L.createSeq = 0; L.svcStop = 0; L.svcValid = 0; L.reqResult = 0; L.gq.reqSeq = 0; L.gq.ackSeq = 0; }
    if (threadIdx.x < LDS_DIRTY_BITS / 32) s_dirty[threadIdx.x] = 0;
    __syncthreads();
    const bool searching = s_game.status == ST_SEARCHING;
    const int startCount = s_game.nodeCount;                       // nodes with a higher id are created by this launch
    constexpr bool mirror = MIRROR;
    if (mirror) {
        const int words = searching ? s_game.nodeCount * 4 : 0;    // uint4 words (an idle / finished game touches no node)
        const uint4* src = reinterpret_cast<const uint4*>(gNodes);
        uint4* dst = reinterpret_cast<uint4*>(s_nodes);
        for (int i = threadIdx.x; i < words; i += COLLECT_THREADS) dst[i] = src[i];
        s.nodes = reinterpret_cast<Node*>(s_nodes); s.ldsTree = true;
        s.dirty = s_dirty;
        __syncthreads();
    }
    s.g = &s_game;
    s.ldsCpuct = s_cpuct; s.ldsPwRoot = s_pwRoot; s.ldsPwNode = s_pwNode; s.tabN = TABN;
    PROF_ADD(9, ta);
    const int wave = threadIdx.x >> 6;
    s.ackSeq = &L.ackSeq; s.typeSeq = &L.typeSeq; s.createSeq = &L.createSeq; s.createFast = &L.createRes.fast;
    s.gq = &L.gq; s.genAckSeq = &L.gq.ackSeq;
    if (wave == 0) {
        const int rows = collect_step(s, s_rt, L, planesNext, blockIdx.x);
        if (threadIdx.x == 0) {
            s_game.nodesVisited += s.nv; s_game.edgesScanned += s.es;
            if (rowsNext) rowsNext[blockIdx.x] = rows;             // batch size of this game for the evaluator
            __hip_atomic_store(&L.svcStop, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        PROF_ADD(10, ta);
#ifdef HM_SEARCH_PROF
        if (blockIdx.x == 0 && threadIdx.x == 0) { s_prof[30] += __builtin_amdgcn_s_memtime() - ta; s_prof[31] += __builtin_amdgcn_s_memrealtime() - rt0_; }
        if (threadIdx.x == 0 && blockIdx.x < 64 && searching && g_colLaunch < PROF_LAUNCHES) g_colDur[g_colLaunch][blockIdx.x] = (unsigned int)(__builtin_amdgcn_s_memtime() - ta);
#endif
    } else collect_helper_role(s, s_rt, L, pl, s_game, blockIdx.x, planesNext, wave);
    PROF_T(tdr);
    __syncthreads();
    PROF_ADD(25, tdr);
    PROF_T(twb);
    if (searching) {                                               // an idle / finished game was not touched
        if (threadIdx.x == 0) s_game.listWords += L.listWords;
        if (mirror) {
            const int words = s_game.nodeCount * 4;
            const uint4* src = reinterpret_cast<const uint4*>(s_nodes);
            uint4* dst = reinterpret_cast<uint4*>(gNodes);
            for (int i = threadIdx.x; i < words; i += COLLECT_THREADS) {
                const int id = i >> 2;                             // only what this launch touched or created goes back to HBM
                if (id >= startCount || ((s_dirty[id >> 5] >> (id & 31)) & 1u)) dst[i] = src[i];
            }
        }
        __syncthreads();
        for (unsigned i = threadIdx.x; i < sizeof(Game) / 4; i += COLLECT_THREADS) reinterpret_cast<u32*>(gGame)[i] = reinterpret_cast<const u32*>(&s_game)[i];
    }
    PROF_ADD(26, twb);
    if (threadIdx.x == 0) atomicMax(&pl.clk->cEnd, (u64)__builtin_amdgcn_s_memrealtime());
    PROF_FLUSH();
}

__global__ __launch_bounds__(64 * (BATCH + 1)) void k_process(Pools pl, Params prm, NetOut out, int* activeCount) {
    __shared__ RulesTab s_rt;
    __shared__ ExpLds s_exp[BATCH];
    PROF_INIT();
    if (threadIdx.x == 0) {
        LegClock* ck = pl.clk;
        if (blockIdx.x == 0) {                                     // k_collect and the forward(s) of this iteration have finished
            const u64 ce = ck->cEnd, cs = ck->cStart, ne = ck->nEnd, ns = ck->nStart;
            if (ce) { ck->sumC += ce - cs; ck->cntC++; }
            if (ne) { ck->sumN += ne - ns; ck->cntN++; }
            ck->cStart = ~0ULL; ck->cEnd = 0; ck->nStart = ~0ULL; ck->nEnd = 0;
        }
        atomicMin(&ck->pStart, (u64)__builtin_amdgcn_s_memrealtime());
    }
    stage_table_wide(&s_rt, pl.rules);
    __syncthreads();
    G s = make_view(pl, prm, blockIdx.x);
    process_step(s, s_rt, s_exp, out, blockIdx.x, activeCount);
    if (threadIdx.x == 0) atomicMax(&pl.clk->pEnd, (u64)__builtin_amdgcn_s_memrealtime());
#ifdef HM_SEARCH_PROF
    if (blockIdx.x == 0 && threadIdx.x == 0) g_colLaunch++;
#endif
    PROF_FLUSH();
}

// Final move rule of Agent::run_search (agent.cc:859-889): Node::get_best_move_idx_with_q_weight (node.h:656-754,
// solver-aware, Q-veto, Q-weighting), the most-visited fallback and the index clamp.  One lane; the root has few edges.
__device__ inline int best_move_index(const G& s, const Node& r, float qVetoDelta, float qValueWeight) {
    if (!(r.flags & F_EXPANDED) || r.expanded <= 0) return -1;
    const Edge* e = edges_of(s, r);
    const int n = r.expanded;
    int best = -1;
    bool decided = false;
    if (r.type == T_WIN) {
        int shortest = 0x7fffffff;
        for (int i = 0; i < r.cntTypes; ++i)
            if (e[i].ctype == T_LOSS && s.nodes[e[i].child].endInPly < shortest) { shortest = s.nodes[e[i].child].endInPly; best = i; }
        decided = best >= 0;
    }
    if (!decided && r.type == T_LOSS) {
        best = 0;
        int longest = 0;
        for (int i = 0; i < n; ++i) { const int ep = s.nodes[e[i].child].endInPly; if (ep > longest) { longest = ep; best = i; } }
        decided = true;
    }
    if (!decided) {
        bool hasNonLosing = false;
        for (int i = 0; i < n; ++i) hasNonLosing |= s.nodes[e[i].child].type != T_WIN;
        auto eligible = [&](int i) { return !hasNonLosing || s.nodes[e[i].child].type != T_WIN; };
        int first = 0;
        while (first < n && !eligible(first)) ++first;
        if (first == n) best = -1;
        else {
            int bestVisitIdx = first, maxVisits = e[first].visits, secondVisitIdx = -1;
            for (int i = first + 1; i < n; ++i) {
                if (!eligible(i)) continue;
                if (e[i].visits > maxVisits) { secondVisitIdx = bestVisitIdx; maxVisits = e[i].visits; bestVisitIdx = i; }
                else if (secondVisitIdx < 0 || e[i].visits > e[secondVisitIdx].visits) secondVisitIdx = i;
            }
            int bestQIdx = first;
            float bestQ = e[first].q;
            for (int i = first + 1; i < n; ++i) { if (!eligible(i)) continue; if (e[i].q > bestQ) { bestQ = e[i].q; bestQIdx = i; } }
            best = bestVisitIdx;
            bool done = false;
            if (qVetoDelta > 0.0f && bestQIdx != bestVisitIdx && e[bestQIdx].q > e[bestVisitIdx].q + qVetoDelta && e[bestQIdx].visits > 1) { best = bestQIdx; done = true; }
            if (!done && qValueWeight > 0.0f && secondVisitIdx >= 0 && e[secondVisitIdx].q > e[bestVisitIdx].q) {
                const float qDifference = e[secondVisitIdx].q - e[bestVisitIdx].q;
                const float adjusted = (float)e[secondVisitIdx].visits + qDifference * qValueWeight * (float)e[bestVisitIdx].visits;
                if (adjusted > (float)e[bestVisitIdx].visits) best = secondVisitIdx;
            }
        }
    }
    if (best < 0) {                                       // agent.cc:872-880
        int maxVisits = 0;
        for (int i = 0; i < n; ++i) if (e[i].visits > maxVisits) { maxVisits = e[i].visits; best = i; }
    }
    if (best < 0 || best >= n) best = 0;                  // agent.cc:882-886
    return best;
}

// Agent::store_next_root_candidates + try_reuse_tree (agent.cc:1345-1451) in one step, at the start of the next search: the
// candidates are the previous root's selected child (final-move rule) and every reply generated below it, in that order; the
// first whose hash, side to play and position equal the new root's is taken.  The reference compares hash and a FEN signature of
// both boards; here the cached NodePos (both positions incl. pockets, castling, ep, clocks, and the history chain) is compared
// word for word, which is the same condition.  A candidate that was generated but never reached has no position record and is
// not taken (the reference would adopt its empty node, which searches exactly like a fresh root).
__device__ inline int find_reusable_root(const G& s, int prevRoot, u64 hash, int team) {
    const Node& r = s.nodes[prevRoot];
    const int b = best_move_index(s, r, s.prm->qVetoDelta, s.prm->qValueWeight);
    if (b < 0) return -1;
    auto matches = [&](int id) {
        const Node& nd = s.nodes[id];
        if (nd.hash != hash || (int)nd.team != team || !nd.posOff) return false;
        const NodePos* np = nodepos_of(s, nd);
        const u64* a = reinterpret_cast<const u64*>(np->pos);
        const u64* c = reinterpret_cast<const u64*>(s.g->pos);
        bool same = np->hlen[0] == s.g->hlen[0] && np->hlen[1] == s.g->hlen[1] && np->prefix[0] == s.g->prefix[0] && np->prefix[1] == s.g->prefix[1];
        for (int i = 0; i < (int)(2 * sizeof(hm_pos) / 8); ++i) same = same && a[i] == c[i];
        return same;
    };
    const int c = edges_of(s, r)[b].child;
    if (matches(c)) return c;
    const Node& cn = s.nodes[c];
    if (!(cn.flags & F_EXPANDED)) return -1;
    const Edge* e = edges_of(s, cn);
    for (int i = 0; i < cn.expanded; ++i) if (matches(e[i].child)) return e[i].child;
    return -1;
}

// Agent::run_search prologue (agent.cc:421-558): early outs, 1-ply root mate scan, root + TT setup.
// rootSig (pinned host memory, or nullptr): [2 g] = root hash, [2 g + 1] = (legal moves + pass) per board, published with system
// scope the moment they are known, so that the host can make the game's Dirichlet draws while this kernel goes on with the root
// mate scan; the host presets [2 g + 1] to ~0 ("not yet") and every path through the kernel stores it exactly once.
__device__ __forceinline__ void root_signal(u64* rootSig, int g, u64 hash, u64 counts) {
    if (!rootSig || (threadIdx.x & 63) != 0) return;
    __hip_atomic_store(&rootSig[2 * g], hash, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&rootSig[2 * g + 1], counts, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(64) void k_begin(Pools pl, Params prm, const int* targetNodes, const u64* noiseSeeds, float alpha, float eps, const uint8_t* searchMask, u64* rootHashOut, u64* rootSig) {
    __shared__ RulesTab s_rt;
    __shared__ WaveLds L;
    stage_table(&s_rt, pl.rules);
    __syncthreads();
    const int g = blockIdx.x, lane = threadIdx.x & 63;
    G s = make_view(pl, prm, g);
    if ((searchMask && !searchMask[g]) || !s.g->live) { s.g->status = ST_IDLE; s.g->root = -1; rootHashOut[2 * g] = 0; rootHashOut[2 * g + 1] = 0; root_signal(rootSig, g, 0, 0); return; }
    const RulesTab& rt = s_rt;
    for (int i = lane; i < prm.ttCap; i += 64) s.ttVals[i] = -1;
    wave_fence();
    Game& gm = *s.g;
    // The pool itself is reset below, once it is known that the previous search's tree is not carried over (tree reuse).
    const int prevRoot = gm.reuseMode ? gm.lastRootP1 - 1 : -1;
    gm.reusedVisits = -1;
    gm.ttCount = 0; gm.ttHits = 0; gm.nodesSearched = 0; gm.pending = -1;
    gm.ctxCount[0] = gm.ctxCount[1] = gm.validCount[0] = gm.validCount[1] = 0;
    gm.sameBatchCollisions = gm.reservationCollisions = gm.evalRows = gm.overflow = gm.maxDepth = 0;
    gm.nodesVisited = gm.edgesScanned = 0; gm.fresh = 0; gm.listWords = 0;
    gm.targetNodes = targetNodes[g]; gm.noiseSeed = noiseSeeds ? noiseSeeds[g] : 0; gm.alpha = alpha; gm.eps = eps;
    gm.root = -1;
    Path p;
    path_reset(s, p);
    const int team = gm.team;
    const bool adv = gm.adv != 0;
    u32* scratch = &L.lists[0][0];
    const bool aOn = (int)p.jb.bd[0].stm == team, bOn = (int)p.jb.bd[1].stm == (team ^ 1);
    const bool canWait = is_double_sit_legal(adv, aOn, bOn);
    const u64 rootHash = board_hash_key(p.jb.bd[0], p.jb.bd[1], hist_of(p.jb, 0), hist_of(p.jb, 1), adv, rt.zob.time_adv);
    gm.rootHash = rootHash;
    rootHashOut[2 * g] = rootHash;
    rootHashOut[2 * g + 1] = 0;              // (legal moves + pass) per board, for the host's Dirichlet draws
    const bool mateUs = is_checkmate(rt, p.jb.bd, team, adv, scratch);
    if (is_checkmate(rt, p.jb.bd, team ^ 1, !adv, scratch) || mateUs || jb_is_draw(p.jb, 0)) { gm.status = ST_NOACTION; root_signal(rootSig, g, rootHash, 0); return; }
    const int cA = aOn ? count_legal(rt.att, p.jb.bd[0]) : 0, cB = bOn ? count_legal(rt.att, p.jb.bd[1]) : 0;
    if (cA + cB == 0 && !canWait) { gm.status = ST_NOACTION; root_signal(rootSig, g, rootHash, 0); return; }
    // ---- find_immediate_root_mate (agent.cc:136-238).  The reference walks the candidates
    // sequentially; here every candidate is pre-filtered lane-parallel with a necessary condition
    // for Board::is_checkmate (some on-turn board of the victim has no legal move), and only the
    // survivors are verified, in the reference's order, with the full rule.
    {
        u32* la = L.lists[6];
        u32* lb = L.lists[7];
        u32* flA = L.lists[2];
        u32* flB = L.lists[3];
        u32* ord = L.lists[4];           // low 16: A order, high 16: B order (stable_partition by gives_check)
        int nA = 0, nB = 0;
        if (aOn) nA = gen_legal_wave(rt.att, p.jb.bd[0], la);
        if (bOn) nB = gen_legal_wave(rt.att, p.jb.bd[1], lb);
        rootHashOut[2 * g + 1] = (u64)(u32)(nA + 1) | ((u64)(u32)(nB + 1) << 32);
        root_signal(rootSig, g, rootHash, (u64)(u32)(nA + 1) | ((u64)(u32)(nB + 1) << 32));
        __builtin_amdgcn_wave_barrier();
        const bool aChk = checkers_of(rt.att, p.jb.bd[0]) != 0, bChk = checkers_of(rt.att, p.jb.bd[1]) != 0;
        for (int i = lane; i < nA; i += 64) flA[i] = (gives_check(rt, p.jb.bd[0], la[i]) ? 1u : 0u) | (is_capture(p.jb.bd[0], la[i]) ? 2u : 0u);
        for (int i = lane; i < nB; i += 64) flB[i] = (gives_check(rt, p.jb.bd[1], lb[i]) ? 1u : 0u) | (is_capture(p.jb.bd[1], lb[i]) ? 2u : 0u);
        __builtin_amdgcn_wave_barrier();
        {
            // stable_partition by "gives check" (agent.cc:150-158), lane-parallel: checking moves first, each group in list order
            auto partition = [&](const u32* fl, int n, bool high) {
                int nChk = 0;
                for (int c0 = 0; c0 < n; c0 += 64) nChk += __popcll(__ballot(c0 + lane < n && (fl[c0 + lane] & 1)));
                int pc = 0, pn = nChk;
                const u64 below = (1ULL << lane) - 1ULL;
                for (int c0 = 0; c0 < n; c0 += 64) {
                    const int i = c0 + lane;
                    const bool in = i < n, chk = in && (fl[i] & 1);
                    const u64 mC = __ballot(chk), mN = __ballot(in && !chk);
                    if (in) {
                        const int pos = chk ? pc + __popcll(mC & below) : pn + __popcll(mN & below);
                        ord[pos] = high ? (ord[pos] & 0xffffu) | ((u32)i << 16) : (u32)i;
                    }
                    pc += __popcll(mC); pn += __popcll(mN);
                }
                __builtin_amdgcn_wave_barrier();
            };
            partition(flA, nA, false);
            for (int i = nA + lane; i < nB; i += 64) ord[i] = 0;
            __builtin_amdgcn_wave_barrier();
            partition(flB, nB, true);
        }
        __builtin_amdgcn_wave_barrier();
        const int victim = team ^ 1;
        auto no_move_board = [&](const P* nb) {   // necessary condition for is_checkmate(victim, .)
            const bool vA = (int)nb[0].stm == victim, vB = (int)nb[1].stm == (victim ^ 1);
            return (vA && !has_legal_move(rt.att, nb[0])) || (vB && !has_legal_move(rt.att, nb[1]));
        };
        bool found = false;
        u32 fa = 0, fb = 0;
        // phases 1/2: a move on one board, pass on the other; phase 3: a move on both boards
        for (int phase = 0; phase < 3 && !found; ++phase) {
            const bool on = phase == 0 ? aOn : phase == 1 ? bOn : (aOn && bOn);
            if (!on) continue;
            const int total = phase == 0 ? nA : phase == 1 ? nB : nA * nB;
            // Two passes per window of 1024 candidates: the cheap admission test (gives check / a board already in check, sit
            // rules) first, its survivors compacted, then the expensive joint make + "victim has a board without a legal move"
            // on densely packed lanes — few candidates pass the first test, and a wave pays for its slowest lane.
            u32* cand = L.helperLists[0];                                  // 1024 entries (helperLists[0..1] are contiguous)
            auto admit = [&](int t) {
                if (phase == 0) {
                    const int i = (int)(ord[t] & 0xffffu);
                    return (aChk || bChk || (flA[i] & 1)) && (!bOn || is_single_pass_legal(adv, aOn, bOn, (flA[i] & 2) != 0));
                }
                if (phase == 1) {
                    const int i = (int)(ord[t] >> 16);
                    return (aChk || bChk || (flB[i] & 1)) && (!aOn || is_single_pass_legal(adv, aOn, bOn, (flB[i] & 2) != 0));
                }
                const int i = (int)(ord[t / nB] & 0xffffu), j = (int)(ord[t % nB] >> 16);
                return aChk || bChk || (flA[i] & 1) || (flB[j] & 1);
            };
            for (int w0 = 0; w0 < total && !found; w0 += 1024) {
                int nc = 0;
                for (int t0 = w0; t0 < min(total, w0 + 1024); t0 += 64) {
                    const int t = t0 + lane;
                    const bool ok = t < total && admit(t);
                    const u64 m = __ballot(ok);
                    if (ok) cand[nc + __popcll(m & ((1ULL << lane) - 1ULL))] = (u32)t;      // ascending in t
                    nc += __popcll(m);
                }
                __builtin_amdgcn_wave_barrier();
                auto moves_of = [&](int t, u32& mA, u32& mB) {
                    mA = 0; mB = 0;
                    if (phase == 0) mA = la[ord[t] & 0xffffu];
                    else if (phase == 1) mB = lb[ord[t] >> 16];
                    else { mA = la[ord[t / nB] & 0xffffu]; mB = lb[ord[t % nB] >> 16]; }
                };
                for (int c = lane; c < nc; c += 64) {
                    const int t = (int)cand[c];
                    u32 mA, mB;
                    moves_of(t, mA, mB);
                    P nb[2] = {p.jb.bd[0], p.jb.bd[1]};
                    make_joint(rt.att, rt.zob, nb[0], nb[1], mA, mB);
                    if (no_move_board(nb)) cand[c] = (u32)t | 0x80000000u;                 // survivor, marked in place: no list to overflow
                }
                __builtin_amdgcn_wave_barrier();
                // verify this window's survivors in candidate order (the list is ascending) with the full rule
                for (int c = 0; c < nc && !found; ++c) {
                    const u32 v = cand[c];
                    if (!(v >> 31)) continue;
                    u32 mA, mB;
                    moves_of((int)(v & 0x7fffffffu), mA, mB);
                    P nb[2] = {p.jb.bd[0], p.jb.bd[1]};
                    make_joint(rt.att, rt.zob, nb[0], nb[1], mA, mB);
                    if (is_checkmate(rt, nb, victim, !adv, scratch)) { found = true; fa = mA; fb = mB; }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (found) {   // agent.cc:458-499: trivial proven tree
            gm.nodeCount = 0; gm.arenaTop = 1;
            const int root = node_alloc(s, team, 0);
            const int child = node_alloc(s, team ^ 1, 1);
            Node rn = s.nodes[root];
            rn.hash = rootHash;
            Edge* slot = edge_append(s, rn);
            *slot = Edge{child, 1.0f, 1.0f, 1.0f, 1, 0, fa, fb, 0, 0, T_LOSS, 0, 0, 0};
            rn.expanded = 1; rn.flags |= F_EXPANDED; rn.cntTypes = 1; rn.unsolved = 0;
            Node& cn = s.nodes[child];
            cn.type = T_LOSS; cn.valueSum = -1.0f; cn.endInPly = 0;
            rn.visits = 1;
            rn.type = T_WIN; rn.valueSum = 1.0f * (float)(rn.visits + 1); rn.endInPly = 1;
            s.nodes[root] = rn;
            gm.root = root;
            gm.lastRootP1 = root + 1;
            gm.status = ST_DONE;
            return;
        }
    }
    // Tree reuse (agent.cc:507-525): the previous search's selected child or one of the replies below it becomes the root, with
    // everything searched beneath it, when it is this very position (hash, side, both boards, history) and the pool still has
    // room for the node budget behind the nodes it already holds.
    int root = (prevRoot >= 0 && prevRoot < gm.nodeCount) ? find_reusable_root(s, prevRoot, rootHash, team) : -1;
    if (root >= 0) {
        const int fitNodes = (prm.nodeCap - 64 - gm.nodeCount) / 3 - 2 * BATCH;
        const long long fitArena = ((long long)prm.arenaCap - (long long)gm.arenaTop - 8192) / 768 - 2 * BATCH;
        const int fit = (int)(fitArena < (long long)fitNodes ? fitArena : (long long)fitNodes);
        if (fit >= gm.targetNodes) {}
        else if (gm.reuseMode == 2 && fit >= gm.targetNodes / 2) gm.targetNodes = fit;
        else root = -1;
    }
    if (root >= 0) {
        Node& rn = s.nodes[root];
        rn.hash = rootHash; rn.depth = 0;
        gm.reusedVisits = rn.visits;
    } else {
        gm.nodeCount = 0; gm.arenaTop = 1;
        root = node_alloc(s, team, 0);
        if (root < 0) { gm.status = ST_ERROR; return; }
        s.nodes[root].hash = rootHash;
        path_reset(s, p);                                          // the mate scan left p.jb untouched, but be explicit
        if (!path_store(s, s_rt, p, root)) { gm.status = ST_ERROR; return; }   // the root's position record (every descent starts from it)
    }
    gm.root = root;
    gm.lastRootP1 = root + 1;
    if (prm.enableTranspositions) tt_insert_or_get(s, rootHash, root);
    gm.status = ST_SEARCHING;
}

// root_edge_stats / root_q (agent.cc:1004-1024): out[g][0] = edge count, then per edge (moveA, moveB, visits).
struct RootOut { int* counts; u32* moveA; u32* moveB; int* visits; float* q; float* prior; float* rootQ; int* info; int maxEdges; };
__global__ __launch_bounds__(64) void k_root_stats(Pools pl, Params prm, RootOut o) {
    const int g = blockIdx.x, lane = threadIdx.x & 63;
    G s = make_view(pl, prm, g);
    Game& gm = *s.g;
    int n = 0;
    float rq = 0.0f;
    if (gm.root >= 0) {
        const Node& r = s.nodes[gm.root];
        rq = r.type == T_WIN ? 1.0f : r.type == T_LOSS ? -1.0f : r.type == T_DRAW ? 0.0f : (r.visits > 0 ? r.valueSum / (float)r.visits : r.valueSum);
        if (r.flags & F_EXPANDED) {
            n = r.expanded < o.maxEdges ? r.expanded : o.maxEdges;
            const Edge* e = edges_of(s, r);
            for (int i = lane; i < n; i += 64) {
                const size_t k = (size_t)g * o.maxEdges + i;
                o.moveA[k] = e[i].moveA; o.moveB[k] = e[i].moveB; o.visits[k] = e[i].visits; o.q[k] = e[i].q; o.prior[k] = e[i].prior;
            }
        }
    }
    if (lane == 0) {
        o.counts[g] = n; o.rootQ[g] = rq;
        int* inf = o.info + (size_t)g * HM_SP_INFO_INTS;
        inf[12] = gm.root >= 0 ? best_move_index(s, s.nodes[gm.root], prm.qVetoDelta, prm.qValueWeight) : -1;
        inf[13] = gm.listWords; inf[14] = inf[15] = 0;
        inf[18] = gm.ttHits; inf[19] = gm.ttCount;
        if (inf[12] >= 0 && gm.root >= 0 && (s.nodes[gm.root].flags & F_EXPANDED) && inf[12] < s.nodes[gm.root].expanded) {
            const Node& bc = s.nodes[edges_of(s, s.nodes[gm.root])[inf[12]].child];    // for format_uci_score (agent.cc:48-78)
            inf[14] = bc.type; inf[15] = bc.endInPly;
        }
        inf[0] = gm.status; inf[1] = gm.nodesSearched; inf[2] = gm.evalRows; inf[3] = gm.sameBatchCollisions; inf[4] = gm.reservationCollisions;
        inf[5] = gm.nodeCount; inf[6] = gm.root >= 0 ? s.nodes[gm.root].type : -1; inf[7] = gm.root >= 0 ? s.nodes[gm.root].visits : 0;
        inf[8] = gm.overflow; inf[9] = gm.maxDepth; inf[10] = gm.nodesVisited; inf[11] = gm.edgesScanned;
        inf[16] = gm.reusedVisits; inf[17] = gm.targetNodes;
    }
}

// Principal variations (Agent::extract_pv_from_child, agent.cc:1218-1290): line l starts with root edge childIdx[l] and then follows
// the final-move rule (best_move_index; its fallbacks never fire on a node with children) through expanded nodes, maxDepth joint
// actions at most.  One lane per line; out: moves[l][depth][2], lens[l], and the first child's solver type / end-in-ply for
// format_uci_score (agent.cc:48-78).
__global__ void k_pv(Pools pl, Params prm, int game, int nLines, const int* childIdx, int maxDepth, u32* moves, int* lens, int* ctype, int* cend) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nLines) return;
    G s = make_view(pl, prm, game);
    lens[l] = 0; ctype[l] = 0; cend[l] = 0;
    if (s.g->root < 0) return;
    const Node& r = s.nodes[s.g->root];
    const int ci = childIdx[l];
    if (!(r.flags & F_EXPANDED) || ci < 0 || ci >= r.expanded) return;
    u32* out = moves + (size_t)l * maxDepth * 2;
    const Edge& e0 = edges_of(s, r)[ci];
    out[0] = e0.moveA; out[1] = e0.moveB;
    int cur = e0.child, len = 1;
    ctype[l] = s.nodes[cur].type; cend[l] = s.nodes[cur].endInPly;
    for (; len < maxDepth; ++len) {
        const Node& n = s.nodes[cur];
        if (!(n.flags & F_EXPANDED) || n.expanded <= 0) break;
        const int b = best_move_index(s, n, prm.qVetoDelta, prm.qValueWeight);
        if (b < 0) break;
        const Edge& e = edges_of(s, n)[b];
        out[2 * len] = e.moveA; out[2 * len + 1] = e.moveB;
        cur = e.child;
    }
    lens[l] = len;
}

// Board::push_move of the chosen joint action + team / time-advantage flip (selfplay.cc:694-716)
__global__ __launch_bounds__(64) void k_apply(Pools pl, Params prm, const u32* moveA, const u32* moveB, const uint8_t* mask, int* err) {
    __shared__ RulesTab s_rt;
    stage_table(&s_rt, pl.rules);
    __syncthreads();
    const int g = blockIdx.x;
    if (mask && !mask[g]) return;
    G s = make_view(pl, prm, g);
    if (!s.g->live) return;
    Path p;
    path_reset(s, p);
    const u32 ma = moveA[g], mb = moveB[g];
    if (p.jb.hlen[0] + 2 >= prm.histGame || p.jb.hlen[1] + 2 >= prm.histGame) { if (threadIdx.x == 0) atomicOr(err, 1); return; }   // history pool full: nothing applied
    jb_make(s_rt, p.jb, ma, mb, true);
    store_pos(&s.g->pos[0], p.jb.bd[0]);
    store_pos(&s.g->pos[1], p.jb.bd[1]);
    if (ma) s.g->lastMove[0] = ma;
    if (mb) s.g->lastMove[1] = mb;
    s.g->hlen[0] = p.jb.hlen[0]; s.g->hlen[1] = p.jb.hlen[1];
    s.g->prefix[0] = p.jb.prefix[0]; s.g->prefix[1] = p.jb.prefix[1];
    s.g->team ^= 1; s.g->adv ^= 1;
}

// side to act of every game slot, without touching positions or history (the UCI front end replays single-board moves with
// hm_sp_apply, which flips the side per call, and then sets Team / Mode as the options say: uci.cc:283-296)
__global__ void k_set_side(Pools pl, int n, const uint8_t* team, const uint8_t* adv) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) { pl.games[g].team = team[g] ? 1 : 0; pl.games[g].adv = adv[g] ? 1 : 0; }
}
// tree reuse policy of every game slot (mode[g]: 0 off, 1 reuse when the node budget fits, 2 shrink the budget to fit); reset != 0
// also forgets the previous search's tree (Agent::reset_search_state, agent.cc:403-412)
__global__ void k_set_reuse(Pools pl, int n, const uint8_t* mode, int reset) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) { if (mode) pl.games[g].reuseMode = mode[g]; if (reset) pl.games[g].lastRootP1 = 0; }
}
// ends the search of the masked games at their next collect (Agent::set_is_running(false), agent.h: the UCI `stop` / movetime path)
__global__ void k_stop(Pools pl, int n, const uint8_t* mask) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n && (!mask || mask[g])) pl.games[g].targetNodes = 0;
}

// (re)start games from a compact board (history restarts, like Board::set)
__global__ __launch_bounds__(64) void k_set_games(Pools pl, Params prm, const hm_board* boards, const uint8_t* mask) {
    __shared__ RulesTab s_rt;
    stage_table(&s_rt, pl.rules);
    __syncthreads();
    const int g = blockIdx.x;
    if (mask && !mask[g]) return;
    G s = make_view(pl, prm, g);
    Game& gm = *s.g;
    gm.pos[0] = boards[g].pos[0]; gm.pos[1] = boards[g].pos[1];
    gm.lastMove[0] = gm.lastMove[1] = 0;            // Board::set clears moveHistory (board.cc:38,48)
    gm.team = boards[g].team; gm.adv = boards[g].time_adv;
    for (int b = 0; b < 2; ++b) {
        P q;
        load_pos(q, &gm.pos[b]);
        const u64 k = rep_key(s_rt, q);
        s.hist[b][0] = k;
        gm.hlen[b] = 1;
        gm.prefix[b] = mix_hash(HISTORY_HASH_SEED, k);
    }
    gm.status = ST_IDLE; gm.root = -1; gm.pending = -1; gm.overflow = 0; gm.live = 1; gm.pwSel = 0;
}

__global__ void k_set_batch(Pools pl, int n, const uint8_t* b) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) pl.games[g].batch = b ? (int)b[g] : 0;
}
__global__ void k_set_pw_sel(Pools pl, int n, const uint8_t* sel) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) pl.games[g].pwSel = sel[g] ? 1 : 0;
}

// game state export: hm_board (for record planes / host bookkeeping) + terminal flags
// flags[g]: bit0 is_checkmate(team, adv), bit1 is_draw()  (selfplay.cc:608-616)
__global__ __launch_bounds__(64) void k_game_state(Pools pl, Params prm, hm_board* out, int* flags) {
    __shared__ RulesTab s_rt;
    __shared__ u32 scratch[2 * HM_MAX_MOVES];
    stage_table(&s_rt, pl.rules);
    __syncthreads();
    const int g = blockIdx.x, lane = threadIdx.x & 63;
    G s = make_view(pl, prm, g);
    if (!s.g->live) {                                          // empty slot: no position to query
        if (lane == 0) { hm_board z{}; out[g] = z; flags[g] = 0; }
        return;
    }
    Path p;
    path_reset(s, p);
    const bool mate = is_checkmate(s_rt, p.jb.bd, s.g->team, s.g->adv != 0, scratch);
    const bool draw = jb_is_draw(p.jb, 0);
    const int rcA = repetition_count(hist_of(p.jb, 0)), rcB = repetition_count(hist_of(p.jb, 1));
    if (lane == 0) {
        hm_board hb;
        hb.pos[0] = s.g->pos[0]; hb.pos[1] = s.g->pos[1];
        hb.last_move[0] = s.g->lastMove[0]; hb.last_move[1] = s.g->lastMove[1];
        hb.rep_count[0] = (uint8_t)(rcA > 3 ? 3 : rcA); hb.rep_count[1] = (uint8_t)(rcB > 3 ? 3 : rcB);
        hb.team = (uint8_t)s.g->team; hb.time_adv = (uint8_t)s.g->adv; hb.reserved = 0;
        out[g] = hb;
        flags[g] = (mate ? 1 : 0) | (draw ? 2 : 0);
    }
}

// Raw-policy opening support (selfplay.cc:277-376): per game, actions (+pass) and probabilities of
// both boards from the net's policy heads; plus action_leads_to_terminal for a proposed action.
struct RawOut { u32* moves; float* probs; int* counts; uint8_t* caps; uint8_t* onTurn; };   // [g][2][HM_MAX_MOVES]
// allMoves != 0: the UCI `policy` listing (uci.cc:306-393) -- every legal move, rook / bishop promotions kept with logit -inf
// (get_fast_policy_index returns -1 for them, utils.h:183-216) instead of being dropped as the search does.
__global__ __launch_bounds__(64) void k_raw_policy(Pools pl, Params prm, const uint16_t* piA, const uint16_t* piB, RawOut o, int allMoves) {
    __shared__ RulesTab s_rt;
    __shared__ WaveLds L;
    stage_table(&s_rt, pl.rules);
    __syncthreads();
    const int g = blockIdx.x, lane = threadIdx.x & 63;
    G s = make_view(pl, prm, g);
    if (!s.g->live) { if (lane == 0) { o.counts[g * 2] = o.counts[g * 2 + 1] = 0; o.onTurn[g * 2] = o.onTurn[g * 2 + 1] = 0; } return; }
    // reuse expand_leaf's prior computation through a throw-away context: replicate its first half
    P bd[2];
    load_pos(bd[0], &s.g->pos[0]);
    load_pos(bd[1], &s.g->pos[1]);
    const int team = s.g->team;
    const bool on[2] = {(int)bd[0].stm == team, (int)bd[1].stm == (team ^ 1)};
    for (int b = 0; b < 2; ++b) {
        u32* list = L.lists[b];
        float* pr = L.exp.priors[b];
        int n = 0;
        if (on[b]) {
            n = gen_legal(s_rt.att, bd[b], list);
            int k = 0;
            for (int i = 0; i < n; ++i) {
                const u32 m = list[i];
                const bool bad = (m & (15u << 12)) == HM_MT_PROMOTION && (((m >> 16) & 63) == HM_ROOK || ((m >> 16) & 63) == HM_BISHOP);
                if (!bad || allMoves) list[k++] = m;
            }
            n = k;
        }
        list[n] = 0;
        const uint16_t* pol = (b == 0 ? piA : piB) + (size_t)g * HM_POLICY_VALUES;
        const int stm = (int)bd[b].stm;
        if (n == 0) pr[0] = 1.0f;
        else {
            float mx = -INFINITY;
            for (int i = 0; i < n + 1; ++i) {
                const u32 m = list[i];
                int idx;
                if (m == 0) idx = 0;
                else if ((m & (15u << 12)) == HM_MT_DROP) idx = pl.polDrop[(stm * 64 + (m & 63)) * 8 + ((m >> 16) & 63)];
                else if ((m & (15u << 12)) == HM_MT_PROMOTION && (((m >> 16) & 63) == HM_ROOK || ((m >> 16) & 63) == HM_BISHOP)) idx = -1;
                else {
                    const int knight = ((m & (15u << 12)) == HM_MT_PROMOTION && ((m >> 16) & 63) == HM_KNIGHT) ? 1 : 0;
                    idx = pl.polNormal[((stm * 64 + ((m >> 6) & 63)) * 64 + (m & 63)) * 2 + knight];
                }
                const float lg = idx >= 0 ? h2f(pol[idx]) : -INFINITY;
                pr[i] = lg;
                if (finite_f(lg)) mx = fmaxf(mx, lg);
            }
            if (!finite_f(mx)) { for (int i = 0; i < n + 1; ++i) pr[i] = 1.0f / (float)(n + 1); }
            else {
                double sum = 0.0;
                for (int i = 0; i < n + 1; ++i) { const float lg = pr[i]; pr[i] = finite_f(lg) ? hm_expf(lg - mx) : 0.0f; sum += (double)pr[i]; }
                for (int i = 0; i < n + 1; ++i) pr[i] = (float)((double)pr[i] / sum);
            }
        }
        const size_t base = ((size_t)g * 2 + b) * HM_MAX_MOVES;
        for (int i = lane; i < n + 1; i += 64) {
            o.moves[base + i] = list[i]; o.probs[base + i] = pr[i];
            o.caps[base + i] = (list[i] != 0 && is_capture(bd[b], list[i])) ? 1 : 0;
        }
        if (lane == 0) { o.counts[g * 2 + b] = n + 1; o.onTurn[g * 2 + b] = on[b]; }
        __builtin_amdgcn_wave_barrier();
    }
}
// action_leads_to_terminal (selfplay.cc:378-390)
__global__ __launch_bounds__(64) void k_action_terminal(Pools pl, Params prm, const u32* moveA, const u32* moveB, int* out) {
    __shared__ RulesTab s_rt;
    __shared__ u32 scratch[2 * HM_MAX_MOVES];
    stage_table(&s_rt, pl.rules);
    __syncthreads();
    const int g = blockIdx.x, lane = threadIdx.x & 63;
    G s = make_view(pl, prm, g);
    if (!s.g->live) { if (lane == 0) out[g] = 0; return; }
    Path p;
    path_reset(s, p);
    const int team = s.g->team;
    const bool adv = s.g->adv != 0;
    // the future position's keys go just past the game history (scratch region of the pool)
    jb_make(s_rt, p.jb, moveA[g], moveB[g], true);
    wave_fence();
    const bool t = is_checkmate(s_rt, p.jb.bd, team ^ 1, !adv, scratch) || is_checkmate(s_rt, p.jb.bd, team, adv, scratch) || jb_is_draw(p.jb, 0);
    if (lane == 0) out[g] = t ? 1 : 0;
}

// test hook: classify_terminal_position (searchthread.cc:101-139) and Board::is_draw / repetition_count on each game's
// CURRENT position with its real game history.  args[g] = {teamToPlay, rootTeam, rootAdv, searchPly};
// out[g*4] = outcome | endInPly << 8, is_draw(searchPly), repetition_count(A), repetition_count(B)
__global__ __launch_bounds__(64) void k_classify(Pools pl, Params prm, const int* args, int* out) {
    __shared__ RulesTab s_rt;
    __shared__ u32 scratch[6 * HM_MAX_MOVES];
    stage_table(&s_rt, pl.rules);
    __syncthreads();
    const int g = blockIdx.x, lane = threadIdx.x & 63;
    G s = make_view(pl, prm, g);
    if (!s.g->live) { if (lane == 0) { out[g * 4] = out[g * 4 + 1] = out[g * 4 + 2] = out[g * 4 + 3] = 0; } return; }
    Path p;
    path_reset(s, p);
    const int* a = args + g * 4;
    int e = 0;
    const int to = classify_terminal_position(s_rt, p.jb, a[0], a[1], a[2] != 0, a[3], &e, scratch);
    const bool draw = jb_is_draw(p.jb, a[3]);
    const int rcA = repetition_count(hist_of(p.jb, 0)), rcB = repetition_count(hist_of(p.jb, 1));
    if (lane == 0) { out[g * 4] = to | (e << 8); out[g * 4 + 1] = draw ? 1 : 0; out[g * 4 + 2] = rcA; out[g * 4 + 3] = rcB; }
}

// test hook: Board queries on compact boards without history (tests/test_gpu_rules.py)
__global__ __launch_bounds__(64) void k_rules_probe(const RulesTab* rules, const hm_board* boards, int n, int* out /*n*8*/, u64* keys /*n*4*/) {
    __shared__ RulesTab s_rt;
    __shared__ u32 scratch[6 * HM_MAX_MOVES];
    __shared__ u64 hk[2][4];
    stage_table(&s_rt, rules);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        JBoard j;
        load_pos(j.bd[0], &boards[i].pos[0]);
        load_pos(j.bd[1], &boards[i].pos[1]);
        for (int b = 0; b < 2; ++b) {
            const u64 k = rep_key(s_rt, j.bd[b]);
            hk[b][0] = k;
            j.hist[b] = hk[b]; j.hlen[b] = 1; j.prefix[b] = mix_hash(HISTORY_HASH_SEED, k);
        }
        __builtin_amdgcn_wave_barrier();
        int r[8];
        r[0] = is_checkmate(s_rt, j.bd, 0, false, scratch);
        r[1] = is_checkmate(s_rt, j.bd, 0, true, scratch);
        r[2] = is_checkmate(s_rt, j.bd, 1, false, scratch);
        r[3] = is_checkmate(s_rt, j.bd, 1, true, scratch);
        r[4] = checkers_of(s_rt.att, j.bd[0]) != 0;
        r[5] = checkers_of(s_rt.att, j.bd[1]) != 0;
        int e0 = 0, e1 = 0;
        r[6] = classify_terminal_position(s_rt, j, boards[i].team, boards[i].team, boards[i].time_adv != 0, 1, &e0, scratch) | (e0 << 8);
        r[7] = classify_terminal_position(s_rt, j, boards[i].team ^ 1, boards[i].team, boards[i].time_adv != 0, 1, &e1, scratch) | (e1 << 8);
        if (lane == 0) {
            for (int q = 0; q < 8; ++q) out[(size_t)i * 8 + q] = r[q];
            keys[(size_t)i * 4 + 0] = board_hash_key(j.bd[0], j.bd[1], hist_of(j, 0), hist_of(j, 1), false, s_rt.zob.time_adv);
            keys[(size_t)i * 4 + 1] = board_hash_key(j.bd[0], j.bd[1], hist_of(j, 0), hist_of(j, 1), true, s_rt.zob.time_adv);
            keys[(size_t)i * 4 + 2] = hk[0][0];
            keys[(size_t)i * 4 + 3] = hk[1][0];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace hms

struct hm_sp;
extern "C" int hm_sp_active_on(hm_sp* sp, int* pinned_out, hipStream_t stream);

// =========================================================================================
// host side
// =========================================================================================
using namespace hms;

struct hm_sp {
    Params prm;
    Pools pl;
    int nGames;
    int maxEdges;
    std::vector<void*> allocs;
    int* d_pv = nullptr;                 // hm_sp_pv_lines scratch (grown on demand)
    size_t pvInts = 0;
    // device outputs
    RootOut ro;
    u64* d_rootHash;
    u64* h_rootSig = nullptr;              // pinned, device-visible: k_begin publishes root hash / action counts per game here (root_signal)
    float* h_noise = nullptr;              // pinned staging of the Dirichlet draws
    int* d_active;
    int* d_target;
    u64* d_seed;
    uint8_t* d_mask;
    u32 *d_moveA, *d_moveB;
    int* d_applyErr;
    hm_board* d_boards;
    int* d_flags;
    RawOut raw;
    int* d_term;
    float alpha = 0.0f, eps = 0.0f;
    std::vector<u64> h_rootHash;
    float pwExponent = 0.4f, altPw = -1.0f, altRootPw = -1.0f;
    int *d_pwRootAlt = nullptr, *d_pwNodeAlt = nullptr;
    size_t ldsMirrorMax = 0;               // LDS left beside k_collect's static LDS (the largest node mirror a CU admits)
    // persistent search (hm_sp_search): [SrvQueue | done counters of every (game, buffer)] — one block, zeroed before every search
    unsigned char* d_queue = nullptr;
    size_t queueBytes = 0;
    int searchLdsNodes = 0;                // the node pool fits in LDS behind the search role's fixed LDS (hm_search_device.hpp: SearchLds)
    int numCUs = 0;
    hipEvent_t evFork = nullptr, evJoin = nullptr, evT0 = nullptr, evT1 = nullptr;
    hipStream_t sTree = nullptr;           // (unused since the single launch runs on the null stream)
    bool applyPending = false;             // hm_sp_apply_deferred has run: the next hm_sp_game_state checks its error flag
    bool anyReuse = false;                 // some slot keeps its tree between searches (hm_sp_set_tree_reuse): an abandoned search cannot be repeated then
    unsigned lastQueueError = 0;           // SrvQueue::error of the last hm_sp_search (4: the two roles were not resident together, 5: stalled)
    std::vector<std::pair<const void*, size_t>> ldsSet;   // k_rollout instantiations whose dynamic-LDS limit has been raised, and to what
    bool lastBeginMasked = false;          // hm_sp_begin_search was given a mask (hm_sp_begin_again relaunches with the same inputs)
    int lastBeginActive = 0;               // slots that mask admitted (an upper bound of the searching game workgroups)
    unsigned* h_qinit = nullptr;           // pinned: {producers, error, consumers} as uploaded after the memset
    // Per-ply traffic with the host goes through two pinned staging blocks and contiguous device blocks: one copy per call
    // and direction instead of one per array (a pageable hipMemcpy costs tens of microseconds before the first byte moves).
    unsigned char* h_stage = nullptr;      // pinned
    size_t stageBytes = 0;
    unsigned char* d_in = nullptr;         // [target G*4 | seed G*8 | mask G | moveA G*4 | moveB G*4 | apply error flag 4] (+ padding)
    unsigned char* d_out = nullptr;        // RootOut arrays, then boards / flags of hm_sp_game_state
    size_t inBytes = 0, outRootBytes = 0;  // size of the input block; bytes of d_out up to and including `visits`
};

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
static int dalloc(hm_sp* sp, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count * sizeof(T), 8));
    if (e != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    (void)hipMemset(q, 0, std::max<size_t>(count * sizeof(T), 8));
    sp->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return 0;
}

static RulesTab* g_rules_dev = nullptr;
static int* g_polN_dev = nullptr;
static int* g_polD_dev = nullptr;
static float* g_cpuct_dev = nullptr;
static int *g_pwRoot_dev = nullptr, *g_pwNode_dev = nullptr;

static int ensure_search_tables() {
    if (g_rules_dev) return 0;
    const HostTables& h = hm_host_tables();
    if (!h.built) return hm_fail(HM_ERR_STATE, "hm_init() has not been called");
    static RulesTab rt;
    rt.att = h.dev.att; rt.zob = h.dev.zob; rt.in_hand_const = h.in_hand_const;
    // promoted-piece marks of the repetition key: splitmix64 stream (same constants as the oracle)
    uint64_t x = 0x9e3779b97f4a7c15ULL;
    for (int s = 0; s < 64; ++s) {
        x += 0x9e3779b97f4a7c15ULL;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        rt.z_promoted[s] = z ^ (z >> 31);
    }
    for (int k = 0; k < 64; ++k) {
        rt.pocket_f16[k] = (uint16_t)h.plane_consts.pocket[HM_DT_F16][k];
        rt.r50_f16[k] = (uint16_t)h.plane_consts.r50[HM_DT_F16][k];
    }
    HIPCHK(hipMalloc(&g_rules_dev, sizeof(RulesTab)));
    HIPCHK(hipMemcpy(g_rules_dev, &rt, sizeof rt, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&g_polN_dev, sizeof(h.dev.pol_normal)));
    HIPCHK(hipMemcpy(g_polN_dev, h.dev.pol_normal, sizeof(h.dev.pol_normal), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&g_polD_dev, sizeof(h.dev.pol_drop)));
    HIPCHK(hipMemcpy(g_polD_dev, h.dev.pol_drop, sizeof(h.dev.pol_drop), hipMemcpyHostToDevice));
    return 0;
}

extern "C" {

int hm_sp_create(int n_games, int max_nodes, const hm_search_config* cfg, hm_sp** out) { return hm_sp_create_ex(n_games, max_nodes, 0, cfg, out); }

int hm_sp_create_ex(int n_games, int max_nodes, int max_game_plies, const hm_search_config* cfg, hm_sp** out) {
    if (!out || n_games <= 0 || max_nodes <= 0 || max_game_plies < 0) return hm_fail(HM_ERR_INVALID, "bad hm_sp_create arguments");
    if (int rc = ensure_search_tables()) return rc;
    hm_search_config c;
    if (cfg) c = *cfg; else hm_search_config_default(&c);
    hm_sp* sp = new hm_sp();
    sp->nGames = n_games;
    sp->maxEdges = 256;
    Params& p = sp->prm;
    p.nGames = n_games;
    p.nodeCap = 3 * (max_nodes + 2 * BATCH) + 64;
    p.histGame = std::max(HIST_GAME_MIN, max_game_plies + 8);        // one key per push on a board, at most one push per macro-ply
    p.histCap = p.histGame + MAX_TRAJ + 8;
    {
        // the node mirror shares the CU's 160 KB of LDS with k_collect's static LDS (tables, wave scratch, request slots)
        hipFuncAttributes fa;
        size_t staticLds = 48 * 1024;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_collect<true>)) == hipSuccess) staticLds = fa.sharedSizeBytes;
        else (void)hipGetLastError();
        const size_t room = staticLds < 160 * 1024 ? 160 * 1024 - staticLds : 0;
        // (s_dirty in k_collect holds LDS_DIRTY_BITS node bits: pools beyond that walk their nodes in place)
        p.ldsNodes = ((size_t)p.nodeCap * sizeof(Node) <= room && p.nodeCap <= LDS_DIRTY_BITS && !std::getenv("HM_SEARCH_NO_LDS_NODES")) ? 1 : 0;
        sp->ldsMirrorMax = room;
    }
    int tt = 64;
    while (tt < 4 * p.nodeCap) tt <<= 1;
    p.ttCap = tt;
    p.arenaCap = (u32)std::min<size_t>(((size_t)(max_nodes + 2 * BATCH) * 6144 + (1 << 16)) / 8, (size_t)1 << 28);
    p.cpuctInit = c.cpuct_init; p.cpuctBase = c.cpuct_base; p.fpuReduction = c.fpu_reduction; p.drawContempt = c.draw_contempt;
    p.wdlWeight = c.wdl_value_weight; p.mlDiscount = c.moves_left_discount;
    p.enableTranspositions = c.enable_transpositions; p.enableDynamicFpu = c.enable_dynamic_fpu; p.enableWdl = c.enable_wdl_eval;
    p.qVetoDelta = 0.4f; p.qValueWeight = 1.0f;
    // the attribute is per-function process state: always the largest mirror the CU admits, so that a second, smaller engine does
    // not lower the limit under an engine that is still alive
    if (p.ldsNodes && hipFuncSetAttribute(reinterpret_cast<const void*>(k_collect<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp->ldsMirrorMax) != hipSuccess) {
        (void)hipGetLastError();
        p.ldsNodes = 0;                                            // walk the pool in place
    }
    Pools& pl = sp->pl;
    const size_t G_ = (size_t)n_games;
    int rc = 0;
    rc |= dalloc(sp, &pl.clk, 1);
    rc |= dalloc(sp, &pl.games, G_);
    rc |= dalloc(sp, &pl.nodes, G_ * p.nodeCap);
    rc |= dalloc(sp, &pl.arena, G_ * p.arenaCap);
    rc |= dalloc(sp, &pl.ttKeys, G_ * p.ttCap);
    rc |= dalloc(sp, &pl.ttVals, G_ * p.ttCap);
    rc |= dalloc(sp, &pl.ctx, G_ * 2 * BATCH);
    rc |= dalloc(sp, &pl.traj, G_ * 2 * BATCH * MAX_TRAJ);
    rc |= dalloc(sp, &pl.hist, G_ * 2 * (size_t)p.histCap);
    rc |= dalloc(sp, &pl.noise, G_ * 2 * NOISE_CAP);
    rc |= dalloc(sp, &pl.leafMoves, G_ * 2 * BATCH * 2 * HM_MAX_MOVES);
    rc |= dalloc(sp, &pl.leafCounts, G_ * 2 * BATCH * 2);
    rc |= dalloc(sp, &pl.sortedMoves, G_ * 2 * BATCH * 2 * HM_MAX_MOVES);
    rc |= dalloc(sp, &pl.sortedPriors, G_ * 2 * BATCH * 2 * HM_MAX_MOVES);
    if (rc) { hm_sp_destroy(sp); return rc; }
    // cpuct(N) and the PW schedule from the reference's own float expressions (search_params.h:307-317)
    // A node's visit count is bounded by the simulations of the searches its tree has lived through (tree reuse carries visits over
    // from one `go` to the next), so the tables are sized from the node budget: a UCI engine with max_nodes = 100 000 reaches root
    // visits far beyond 32 768, where a clamped table would freeze cpuct and stop the root from widening (search_params.h:307-317
    // evaluate both at the true N).
    const int TAB = p.tabLen = (int)std::min<long long>(std::max<long long>(MIN_VISITS_TAB, 8LL * max_nodes + 4096), 1LL << 24);
    std::vector<float> cp(TAB);
    std::vector<int> pr(TAB), pn(TAB);
    for (int v = 0; v < TAB; ++v) {
        cp[v] = std::log(((float)v + c.cpuct_base + 1.0f) / c.cpuct_base) + c.cpuct_init;
        auto allowed = [&](float coef) { return v <= 0 ? 1 : (int)std::ceil(coef * std::pow((float)v, c.pw_exponent)); };
        pr[v] = allowed(c.root_pw_coefficient);
        pn[v] = allowed(c.pw_coefficient);
    }
    float* dcp; int *dpr, *dpn;
    rc |= dalloc(sp, &dcp, TAB); rc |= dalloc(sp, &dpr, TAB); rc |= dalloc(sp, &dpn, TAB);
    if (rc) { hm_sp_destroy(sp); return rc; }
    (void)hipMemcpy(dcp, cp.data(), sizeof(float) * TAB, hipMemcpyHostToDevice);
    (void)hipMemcpy(dpr, pr.data(), sizeof(int) * TAB, hipMemcpyHostToDevice);
    (void)hipMemcpy(dpn, pn.data(), sizeof(int) * TAB, hipMemcpyHostToDevice);
    pl.cpuctTab = dcp; pl.pwRoot = dpr; pl.pwNode = dpn;
    // the alternate schedule has its own tables from the start (kernel arguments captured in a HIP graph keep these pointers);
    // hm_sp_set_pw_profiles rewrites their contents
    rc |= dalloc(sp, &sp->d_pwRootAlt, TAB); rc |= dalloc(sp, &sp->d_pwNodeAlt, TAB);
    if (rc) { hm_sp_destroy(sp); return rc; }
    (void)hipMemcpy(sp->d_pwRootAlt, pr.data(), sizeof(int) * TAB, hipMemcpyHostToDevice);
    (void)hipMemcpy(sp->d_pwNodeAlt, pn.data(), sizeof(int) * TAB, hipMemcpyHostToDevice);
    pl.pwRootAlt = sp->d_pwRootAlt; pl.pwNodeAlt = sp->d_pwNodeAlt;
    sp->pwExponent = c.pw_exponent;
    pl.rules = g_rules_dev; pl.polNormal = g_polN_dev; pl.polDrop = g_polD_dev;
    RootOut& ro = sp->ro;
    ro.maxEdges = sp->maxEdges;
    {
        // device output block: counts | rootQ | info | moveA | moveB | visits || q | prior || boards | flags
        const size_t E_ = (size_t)ro.maxEdges;
        size_t off = 0;
        auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 15) & ~(size_t)15; return o; };
        const size_t oCounts = carve(4 * G_), oRootQ = carve(4 * G_), oInfo = carve(4 * G_ * HM_SP_INFO_INTS);
        const size_t oMoveA = carve(4 * G_ * E_), oMoveB = carve(4 * G_ * E_), oVisits = carve(4 * G_ * E_);
        sp->outRootBytes = off;
        const size_t oQ = carve(4 * G_ * E_), oPrior = carve(4 * G_ * E_), oBoards = carve(sizeof(hm_board) * G_), oFlags = carve(4 * G_);
        rc |= dalloc(sp, &sp->d_out, off);
        if (rc) { hm_sp_destroy(sp); return rc; }
        unsigned char* b = sp->d_out;
        ro.counts = reinterpret_cast<int*>(b + oCounts); ro.rootQ = reinterpret_cast<float*>(b + oRootQ); ro.info = reinterpret_cast<int*>(b + oInfo);
        ro.moveA = reinterpret_cast<u32*>(b + oMoveA); ro.moveB = reinterpret_cast<u32*>(b + oMoveB); ro.visits = reinterpret_cast<int*>(b + oVisits);
        ro.q = reinterpret_cast<float*>(b + oQ); ro.prior = reinterpret_cast<float*>(b + oPrior);
        sp->d_boards = reinterpret_cast<hm_board*>(b + oBoards); sp->d_flags = reinterpret_cast<int*>(b + oFlags);
        // device input block: target | seed | mask | moveA | moveB | apply error flag
        size_t in = 0;
        auto carveIn = [&](size_t bytes) { const size_t o = in; in += (bytes + 15) & ~(size_t)15; return o; };
        const size_t iSeed = carveIn(8 * G_), iTarget = carveIn(4 * G_), iMask = carveIn(G_), iMoveA = carveIn(4 * G_), iMoveB = carveIn(4 * G_), iErr = carveIn(4);
        sp->inBytes = in;
        rc |= dalloc(sp, &sp->d_in, in);
        if (rc) { hm_sp_destroy(sp); return rc; }
        sp->d_seed = reinterpret_cast<u64*>(sp->d_in + iSeed); sp->d_target = reinterpret_cast<int*>(sp->d_in + iTarget);
        sp->d_moveA = reinterpret_cast<u32*>(sp->d_in + iMoveA); sp->d_moveB = reinterpret_cast<u32*>(sp->d_in + iMoveB);
        sp->d_mask = sp->d_in + iMask; sp->d_applyErr = reinterpret_cast<int*>(sp->d_in + iErr);
        sp->stageBytes = std::max(off, in);
        if (hipHostMalloc(reinterpret_cast<void**>(&sp->h_stage), sp->stageBytes, hipHostMallocDefault) != hipSuccess) { hm_sp_destroy(sp); return hm_fail(HM_ERR_NO_DEVICE, "hipHostMalloc failed"); }
    }
    rc |= dalloc(sp, &sp->d_rootHash, 2 * G_); rc |= dalloc(sp, &sp->d_active, 2) /* [0] active games */;
    {
        sp->queueBytes = hmq::QueueLayout((unsigned)G_, sizeof(hmq::SrvQueue)).bytes;   // queue | done | diagnostics | their snapshot | heartbeats
        rc |= dalloc(sp, &sp->d_queue, sp->queueBytes);
        // the node pool goes into LDS behind the search role's fixed LDS when both fit a CU's 160 KB
        const size_t room = 160 * 1024 - search_lds_bytes(0);
        sp->searchLdsNodes = ((size_t)p.nodeCap * sizeof(Node) <= room && p.histCap <= SEARCH_HIST_LDS && !std::getenv("HM_SEARCH_NO_LDS_NODES")) ? 1 : 0;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) sp->numCUs = prop.multiProcessorCount;
        else (void)hipGetLastError();
    }
    rc |= dalloc(sp, &sp->d_term, G_);
    rc |= dalloc(sp, &sp->raw.moves, G_ * 2 * HM_MAX_MOVES); rc |= dalloc(sp, &sp->raw.probs, G_ * 2 * HM_MAX_MOVES);
    rc |= dalloc(sp, &sp->raw.caps, G_ * 2 * HM_MAX_MOVES); rc |= dalloc(sp, &sp->raw.counts, G_ * 2); rc |= dalloc(sp, &sp->raw.onTurn, G_ * 2);
    if (rc) { hm_sp_destroy(sp); return rc; }
    sp->h_rootHash.resize(2 * G_);
    if (int rc2 = hm_sp_leg_times(sp, nullptr, nullptr, 1)) { hm_sp_destroy(sp); return rc2; }
    *out = sp;
    return 0;
}

// Progressive-widening schedule per game slot (TournamentConfig::searchConfigFor, tournament.h:34-41: each network searches
// with its own coefficient, used for root and interior nodes alike).  profiles[g] != 0 selects the alternate schedule built
// from (alt_pw_coefficient, alt_root_pw_coefficient); it stays until the next call or hm_sp_set_games.
int hm_sp_set_pw_profiles(hm_sp* sp, float alt_pw_coefficient, float alt_root_pw_coefficient, const uint8_t* profiles) {
    if (!sp || !profiles) return hm_fail(HM_ERR_INVALID, "null argument");
    if (!(alt_pw_coefficient > 0.0f) || !(alt_root_pw_coefficient > 0.0f) || !std::isfinite(alt_pw_coefficient) || !std::isfinite(alt_root_pw_coefficient))
        return hm_fail(HM_ERR_INVALID, "PW coefficients must be positive and finite");
    if (alt_pw_coefficient != sp->altPw || alt_root_pw_coefficient != sp->altRootPw) {
        const int TAB = sp->prm.tabLen;
        std::vector<int> pr(TAB), pn(TAB);
        for (int v = 0; v < TAB; ++v) {
            auto allowed = [&](float coef) { return v <= 0 ? 1 : (int)std::ceil(coef * std::pow((float)v, sp->pwExponent)); };
            pr[v] = allowed(alt_root_pw_coefficient);
            pn[v] = allowed(alt_pw_coefficient);
        }
        HIPCHK(hipMemcpy(sp->d_pwRootAlt, pr.data(), sizeof(int) * TAB, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(sp->d_pwNodeAlt, pn.data(), sizeof(int) * TAB, hipMemcpyHostToDevice));
        sp->altPw = alt_pw_coefficient; sp->altRootPw = alt_root_pw_coefficient;
    }
    HIPCHK(hipMemcpy(sp->d_mask, profiles, sp->nGames, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_set_pw_sel, dim3((sp->nGames + 63) / 64), dim3(64), 0, 0, sp->pl, sp->nGames, sp->d_mask);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
}

// Leaves per iteration of each game slot's searches (the reference's search thread collects engine->getBatchSize() leaves,
// searchthread.cc:258-273, 663; tournaments give each network its own, tournament.h:19-20): 1 .. 8 per slot, NULL = 8 everywhere.
// Stays until the next call (hm_sp_set_games leaves it).  Storage — context slots, plane rows, the leaf ring — is sized for 8.
int hm_sp_set_batch_sizes(hm_sp* sp, const uint8_t* batch) {
    if (!sp) return hm_fail(HM_ERR_INVALID, "null argument");
    if (batch) {
        for (int g = 0; g < sp->nGames; ++g)
            if (batch[g] < 1 || batch[g] > BATCH) return hm_fail(HM_ERR_INVALID, "batch sizes must be between 1 and 8");
        HIPCHK(hipMemcpy(sp->d_mask, batch, sp->nGames, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_set_batch, dim3((sp->nGames + 63) / 64), dim3(64), 0, 0, sp->pl, sp->nGames, batch ? sp->d_mask : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
}

// Leg clock (LegClock above): total device time in ms and launch counts of k_collect / forward / k_process since the last reset.
// The forward takes part when it is launched with the clock's interval pointer (hm_sp_leg_clock_net + hm_net_forward_groups_timed).
int hm_sp_leg_times(hm_sp* sp, double* ms3, uint64_t* counts3, int reset) {
    if (!sp) return hm_fail(HM_ERR_INVALID, "null argument");
    LegClock c;
    HIPCHK(hipMemcpy(&c, sp->pl.clk, sizeof c, hipMemcpyDeviceToHost));
    if (c.cEnd) { c.sumC += c.cEnd - c.cStart; c.cntC++; }          // intervals no later kernel has folded yet
    if (c.nEnd) { c.sumN += c.nEnd - c.nStart; c.cntN++; }
    if (c.pEnd) { c.sumP += c.pEnd - c.pStart; c.cntP++; }
    if (ms3) { ms3[0] = (double)c.sumC * 1e-5; ms3[1] = (double)c.sumN * 1e-5; ms3[2] = (double)c.sumP * 1e-5; }   // 100 MHz ticks
    if (counts3) { counts3[0] = c.cntC; counts3[1] = c.cntN; counts3[2] = c.cntP; }
    if (reset) {
        LegClock z;
        std::memset(&z, 0, sizeof z);
        z.cStart = z.nStart = z.pStart = ~0ULL;
        HIPCHK(hipMemcpy(sp->pl.clk, &z, sizeof z, hipMemcpyHostToDevice));
    }
    return 0;
}
// persistent searches: total time (ms) the games waited for the evaluation of their pending batches since the last reset
int hm_sp_wait_time(hm_sp* sp, double* ms) {
    if (!sp || !ms) return hm_fail(HM_ERR_INVALID, "null argument");
    LegClock c;
    HIPCHK(hipMemcpy(&c, sp->pl.clk, sizeof c, hipMemcpyDeviceToHost));
    *ms = (double)c.sumW * 1e-5;
    return 0;
}
// device pointer to the forward's (start, end) pair of the leg clock
uint64_t* hm_sp_leg_clock_net(hm_sp* sp) { return sp ? reinterpret_cast<uint64_t*>(&sp->pl.clk->nStart) : nullptr; }

int hm_sp_destroy(hm_sp* sp) {
    if (!sp) return 0;
    if (sp->h_stage) (void)hipHostFree(sp->h_stage);
    if (sp->h_qinit) (void)hipHostFree(sp->h_qinit);
    if (sp->h_rootSig) (void)hipHostFree(sp->h_rootSig);
    if (sp->h_noise) (void)hipHostFree(sp->h_noise);
    if (sp->evFork) (void)hipEventDestroy(sp->evFork);
    if (sp->evJoin) (void)hipEventDestroy(sp->evJoin);
    if (sp->evT0) (void)hipEventDestroy(sp->evT0);
    if (sp->evT1) (void)hipEventDestroy(sp->evT1);
    if (sp->sTree) (void)hipStreamDestroy(sp->sTree);
    for (void* p : sp->allocs) (void)hipFree(p);
    delete sp;
    return 0;
}

int hm_sp_set_games(hm_sp* sp, const hm_board* boards, const uint8_t* mask) {
    if (!sp || !boards) return hm_fail(HM_ERR_INVALID, "null argument");
    HIPCHK(hipMemcpy(sp->d_boards, boards, sizeof(hm_board) * sp->nGames, hipMemcpyHostToDevice));
    if (mask) HIPCHK(hipMemcpy(sp->d_mask, mask, sp->nGames, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_set_games, dim3(sp->nGames), dim3(64), 0, 0, sp->pl, sp->prm, sp->d_boards, mask ? sp->d_mask : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
}

// Dirichlet gamma draws exactly as node.h:286-315 (std::gamma_distribution<float> on std::mt19937_64
// seeded rootNoiseSeed ^ positionHash ^ salt); the device mixes them into the root priors.
int hm_sp_begin_search(hm_sp* sp, const int* target_nodes, const uint64_t* noise_seeds, float alpha, float eps, const uint8_t* mask) {
    if (!sp || !target_nodes) return hm_fail(HM_ERR_INVALID, "null argument");
    const int G_ = sp->nGames;
    // seeds | targets | mask are adjacent in the input block: one staged upload
    unsigned char* hs = sp->h_stage;
    uint64_t* seeds = reinterpret_cast<uint64_t*>(hs + (reinterpret_cast<unsigned char*>(sp->d_seed) - sp->d_in));
    if (noise_seeds) std::memcpy(seeds, noise_seeds, 8 * (size_t)G_); else std::memset(seeds, 0, 8 * (size_t)G_);
    std::memcpy(hs + (reinterpret_cast<unsigned char*>(sp->d_target) - sp->d_in), target_nodes, sizeof(int) * (size_t)G_);
    if (mask) std::memcpy(hs + (sp->d_mask - sp->d_in), mask, (size_t)G_);
    const size_t upTo = (size_t)(sp->d_mask - sp->d_in) + (size_t)G_;
    HIPCHK(hipMemcpy(sp->d_in, hs, upTo, hipMemcpyHostToDevice));
    const bool noisy = alpha > 0.0f && eps > 0.0f;
    if (noisy && !sp->h_rootSig) {
        if (hipHostMalloc(reinterpret_cast<void**>(&sp->h_rootSig), 16 * (size_t)G_, hipHostMallocDefault) != hipSuccess
            || hipHostMalloc(reinterpret_cast<void**>(&sp->h_noise), (size_t)G_ * 2 * NOISE_CAP * sizeof(float), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return hm_fail(HM_ERR_NO_DEVICE, "hipHostMalloc failed");
        }
        std::memset(sp->h_noise, 0, (size_t)G_ * 2 * NOISE_CAP * sizeof(float));
    }
    std::vector<uint64_t> seedCopy;
    if (noisy) {
        seedCopy.assign(seeds, seeds + G_);                        // (the staging block may be reused)
        for (int g = 0; g < G_; ++g) { sp->h_rootSig[2 * g] = 0; sp->h_rootSig[2 * g + 1] = ~0ULL; }
    }
    hipLaunchKernelGGL(k_begin, dim3(G_), dim3(64), 0, 0, sp->pl, sp->prm, sp->d_target, sp->d_seed, alpha, eps, mask ? sp->d_mask : nullptr, sp->d_rootHash, noisy ? sp->h_rootSig : nullptr);
    HIPCHK(hipGetLastError());
    if (noisy) {
        // The Dirichlet draws (node.h:286-315: std::gamma_distribution on std::mt19937_64 seeded from the root hash) are made here while
        // k_begin is still scanning the roots for immediate mates: each game publishes its hash and action counts as soon as they are known.
        volatile uint64_t* sig = sp->h_rootSig;
        static const uint64_t salts[2] = {0x9e3779b97f4a7c15ULL, 0xbf58476d1ce4e5b9ULL};
        unsigned spins = 0;
        for (int g = 0; g < G_; ++g) {
            while (sig[2 * g + 1] == ~0ULL) {
                if ((++spins & 0x3fffu) == 0 && hipStreamQuery(nullptr) != hipErrorNotReady) {   // the kernel has ended (or failed): whatever it wrote is there now
                    if (sig[2 * g + 1] == ~0ULL) return hm_fail(HM_ERR_STATE, "k_begin left a root without its action counts");
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            const uint64_t hash = sig[2 * g], counts = sig[2 * g + 1];
            sp->h_rootHash[2 * g] = hash; sp->h_rootHash[2 * g + 1] = counts;
            for (int b = 0; b < 2; ++b) {
                float* dst = sp->h_noise + ((size_t)g * 2 + b) * NOISE_CAP;
                int n = (int)((counts >> (32 * b)) & 0xffffffffu);
                if (n <= 1) continue;                             // a single action takes no noise (stale draws of an earlier search are never read)
                if (n > NOISE_CAP) n = NOISE_CAP;
                std::mt19937_64 eng(seedCopy[g] ^ hash ^ salts[b]);
                std::gamma_distribution<float> gamma(alpha, 1.0f);
                for (int i = 0; i < n; ++i) dst[i] = gamma(eng);
            }
        }
        HIPCHK(hipMemcpyAsync(sp->pl.noise, sp->h_noise, (size_t)G_ * 2 * NOISE_CAP * sizeof(float), hipMemcpyHostToDevice, nullptr));   // behind k_begin on its stream
    }
    sp->alpha = alpha; sp->eps = eps;
    sp->lastBeginMasked = mask != nullptr;
    sp->lastBeginActive = G_;
    if (mask) { int n = 0; for (int g = 0; g < G_; ++g) n += mask[g] ? 1 : 0; sp->lastBeginActive = n; }
    return 0;
}
// The prologue of the last hm_sp_begin_search once more, from the inputs still on the device (targets, seeds, mask, noise draws):
// puts every slot back to the start of its search after a persistent search that could not run (hm_sp_search_not_concurrent).
// Only valid with tree reuse off (refused otherwise: k_begin would adopt the abandoned search's half-built tree) and before any other
// hm_sp_* call that uploads moves or masks.
int hm_sp_begin_again(hm_sp* sp) {
    if (!sp) return hm_fail(HM_ERR_INVALID, "null argument");
    if (sp->anyReuse) return hm_fail(HM_ERR_STATE, "hm_sp_begin_again: a slot keeps its tree between searches (hm_sp_set_tree_reuse); an abandoned search cannot be repeated");
    hipLaunchKernelGGL(k_begin, dim3(sp->nGames), dim3(64), 0, 0, sp->pl, sp->prm, sp->d_target, sp->d_seed, sp->alpha, sp->eps, sp->lastBeginMasked ? sp->d_mask : nullptr, sp->d_rootHash, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
}
// 1 when the last hm_sp_search failed because no workgroup of the other role came in within 3 s (error 4: the grid was not resident
// at once — something else held the device's CUs): the lockstep calls still work.  (Round 3's two-kernel form raised it whenever the
// kernels were serialised, e.g. under a counter-collecting profiler; one launch cannot be.)
int hm_sp_search_not_concurrent(const hm_sp* sp) { return sp && sp->lastQueueError == 4u ? 1 : 0; }
// 1 when the last hm_sp_search was given up because the evaluator had nothing to do for 30 ms while games were still searching
// (hm_queue.hpp: IDLE_LIMIT_TICKS): hm_sp_begin_again + another search (persistent or lockstep) repeats it with the same result.
int hm_sp_search_stalled(const hm_sp* sp) { return sp && sp->lastQueueError == 5u ? 1 : 0; }
// 1 when the single-launch search keeps this engine's node pool in LDS for a whole search (it fits behind the search role's fixed LDS), 0 when the tree is walked in place
int hm_sp_search_lds_tree(const hm_sp* sp) { return sp && sp->searchLdsNodes ? 1 : 0; }
int hm_sp_collect_counted(hm_sp* sp, void* d_planes_next, int32_t* d_rows_next, void* stream) {
    if (!sp || !d_planes_next) return hm_fail(HM_ERR_INVALID, "null argument");
    if (sp->prm.ldsNodes) hipLaunchKernelGGL(k_collect<true>, dim3(sp->nGames), dim3(COLLECT_THREADS), (size_t)sp->prm.nodeCap * sizeof(Node), static_cast<hipStream_t>(stream), sp->pl, sp->prm,
                                             static_cast<uint16_t*>(d_planes_next), d_rows_next, sp->d_active);
    else hipLaunchKernelGGL(k_collect<false>, dim3(sp->nGames), dim3(COLLECT_THREADS), 0, static_cast<hipStream_t>(stream), sp->pl, sp->prm,
                            static_cast<uint16_t*>(d_planes_next), d_rows_next, sp->d_active);
    HIPCHK(hipGetLastError());
    return 0;
}
int hm_sp_collect(hm_sp* sp, void* d_planes_next, void* stream) { return hm_sp_collect_counted(sp, d_planes_next, nullptr, stream); }

int hm_sp_process(hm_sp* sp, const void* d_value, const void* d_pi_a, const void* d_pi_b, const void* d_wdl, const void* d_moves_left,
                  int* active_games, void* stream) {
    if (!sp || !d_value || !d_pi_a || !d_pi_b || !d_wdl || !d_moves_left) return hm_fail(HM_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    NetOut o{static_cast<const uint16_t*>(d_value), static_cast<const uint16_t*>(d_pi_a), static_cast<const uint16_t*>(d_pi_b),
             static_cast<const uint16_t*>(d_wdl), static_cast<const uint16_t*>(d_moves_left)};
    hipLaunchKernelGGL(k_process, dim3(sp->nGames), dim3(64 * (BATCH + 1)), 0, st, sp->pl, sp->prm, o, sp->d_active);
    HIPCHK(hipGetLastError());
    if (active_games) {
        HIPCHK(hipMemcpyAsync(active_games, sp->d_active, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return 0;
}

// Evaluator workgroups a single-launch search of this engine runs beside its game workgroups (0: not available): a workgroup of
// either role takes a whole CU, so all of them are resident at once — which the queue protocol relies on only for speed, every spin
// being bounded — when their number does not exceed the CU count.
// Geometry of a single-launch search of this engine: games per search workgroup, search workgroups, evaluator workgroups.
// One game per workgroup (node pool in LDS) while the games leave the evaluator three quarters of the device; more game slots than
// that (e.g. 256 concurrent games) share search workgroups (search_role_mg) so that the launch stays resident at once — up to
// MG_MAX games per workgroup.  A slow evaluator (the 384-channel network) starts from 3 games per workgroup (measured, see below).
static bool rollout_plan(const hm_sp* sp, bool slowNet, int* perWgOut, int* searchWgsOut, int* consumersOut) {
    if (!sp || sp->numCUs <= 0) return false;
    const int slots = sp->lastBeginActive > 0 ? sp->lastBeginActive : sp->nGames;
    const int act = std::min(slots, sp->nGames);               // games that search this ply (the workgroups of the other slots leave at once)
    int perWg = slowNet ? 3 : 1;                               // measured at configs[3]: 548 / 564 / 570 positions/s with 1 / 2 / 3 games per workgroup
    if (const char* e = std::getenv("HM_SEARCH_GAMES_PER_WG")) perWg = std::max(1, std::min(MG_MAX, std::atoi(e)));
    else while ((act + perWg - 1) / perWg > sp->numCUs / 4 && perWg < MG_MAX) ++perWg;       // (late in a run few games are alive: back to one per workgroup)
    const int searchWgs = perWg > 1 ? (sp->nGames + perWg - 1) / perWg : sp->nGames;
    // every CU the games leave, but no more evaluator workgroups than rows can be in flight (two batches of BATCH rows per searching game, and a few to spare)
    const int resident = perWg > 1 ? std::min(searchWgs, act) : act;
    const int consumers = std::min(sp->numCUs - resident, 2 * BATCH * act + 8);
    if (consumers < 8) return false;
    if (perWgOut) *perWgOut = perWg;
    if (searchWgsOut) *searchWgsOut = searchWgs;
    if (consumersOut) *consumersOut = consumers;
    return true;
}
int hm_sp_search_consumers(const hm_sp* sp) {
    int consumers = 0;
    return rollout_plan(sp, false, nullptr, nullptr, &consumers) ? consumers : 0;
}

#ifndef HM_ROLLOUT_WAVES8_MIN_ACT_DEFAULT
#define HM_ROLLOUT_WAVES8_MIN_ACT_DEFAULT 1000000       // (off: see DESIGN.md 4b for the measurement)
#endif
// hm_rollout.hip, one entry per search role (0: node pool in LDS, 1: tree walked in place, 2: several games per search workgroup):
// the k_rollout instantiation for a trunk of `ctiles` * 32 channels (k5: some block has a 5x5 depthwise), or nullptr
const void* hm_rollout_kernel_mode0(int ctiles, int k5, int narrow8, int* waves);
const void* hm_rollout_kernel_mode1(int ctiles, int k5, int narrow8, int* waves);
const void* hm_rollout_kernel_mode2(int ctiles, int k5, int narrow8, int* waves);

// The whole node-budget search of every slot hm_sp_begin_search left in the searching state, with the native evaluator, as ONE
// kernel launch (hm_rollout.hip: k_rollout — game workgroups and evaluator workgroups joined by the device-side queue of
// hm_queue.hpp).  Replaces the host loop collect -> forward -> process of Agent::run_search's workers (agent.cc:331-352,
// searchthread.cc:661-739); per game the order of tree operations, hence every result, is the same.  Synchronous.
int hm_sp_search(hm_sp* sp, const hm_net* net, const hm_eval_io* io, double* search_kernel_ms) {
    if (!sp || !net || !io) return hm_fail(HM_ERR_INVALID, "null argument");
    if (!io->planes[0] || !io->planes[1] || !io->value || !io->pi_a || !io->pi_b || !io->wdl || !io->moves_left
        || !io->value_2 || !io->pi_a_2 || !io->pi_b_2 || !io->wdl_2 || !io->moves_left_2) return hm_fail(HM_ERR_INVALID, "hm_sp_search needs both plane buffers and both sets of heads");
    hm_net_serve_info ni;
    if (!hm_net_can_serve(net) || hm_net_serve_info_get(net, &ni)) return hm_fail(HM_ERR_INVALID, "this network cannot be the evaluator of the single-launch search");
    int perWg = 1, searchWgs = 0, consumers = 0;
    if (!rollout_plan(sp, hm_net_serve_is_slow(net) != 0, &perWg, &searchWgs, &consumers))
        return hm_fail(HM_ERR_INVALID, "too many game slots for a single-launch search on this device (use the lockstep calls)");
    const int mode = perWg > 1 ? 2 : (sp->searchLdsNodes ? 0 : 1);
    int waves = 4;                                             // waves per workgroup of this network's kernel (k_rollout: 4, or 8 for the 384-channel trunk)
    // RISEv3-small: the eight-wave kernel (faster evaluator, slower search role) while at least this many games search — then the search is
    // bound by the evaluator; the four-wave kernel (the reverse) otherwise.  Both give identical results.  HM_ROLLOUT_WAVES8_MIN_ACT overrides.
    const int actNow = std::min(sp->lastBeginActive > 0 ? sp->lastBeginActive : sp->nGames, sp->nGames);
    static const int min8 = [] { const char* e = std::getenv("HM_ROLLOUT_WAVES8_MIN_ACT"); return e ? std::atoi(e) : HM_ROLLOUT_WAVES8_MIN_ACT_DEFAULT; }();
    const int narrow8 = actNow >= min8 ? 1 : 0;
    const void* kern = mode == 0 ? hm_rollout_kernel_mode0(ni.C / 32, ni.k5, narrow8, &waves) : mode == 1 ? hm_rollout_kernel_mode1(ni.C / 32, ni.k5, narrow8, &waves) : hm_rollout_kernel_mode2(ni.C / 32, ni.k5, narrow8, &waves);
    if (!kern) return hm_fail(HM_ERR_INVALID, "no single-launch search kernel for this trunk width");
    // dynamic LDS: the larger of the two roles' layouts (they overlay each other); the evaluator's item word sits behind its own layout
    const size_t ldsSearch = search_lds_bytes(mode) + (mode == 0 ? (size_t)sp->prm.nodeCap * sizeof(Node) : 0);
    const unsigned itemOff = (unsigned)((ni.ldsBytes + 15) & ~(size_t)15);
    const size_t ldsBytes = std::max(ldsSearch, (size_t)itemOff + 16);
    if (ldsBytes > 160 * 1024) return hm_fail(HM_ERR_INVALID, "single-launch search: the roles' LDS does not fit a CU");
    {
        bool found = false;
        for (auto& kv : sp->ldsSet)
            if (kv.first == kern) {
                found = true;
                if (kv.second < ldsBytes) { HIPCHK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); kv.second = ldsBytes; }
            }
        if (!found) { HIPCHK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes)); sp->ldsSet.emplace_back(kern, ldsBytes); }
    }
    if (!sp->evT0) {
        HIPCHK(hipEventCreate(&sp->evT0));
        HIPCHK(hipEventCreate(&sp->evT1));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sp->h_qinit), 64 + sizeof(hmq::SrvQueue) - sizeof(hmq::SrvQueue::slots), hipHostMallocDefault));
    }
    // The null stream: the search prologue (k_begin, the noise upload) and whatever the caller queued before run there, so the launch
    // is ordered behind them without a device-wide synchronisation (round 3 needed two streams that owned their hardware queues).
    hipStream_t sT = nullptr;
    const hmq::QueueLayout lay((unsigned)sp->nGames, sizeof(hmq::SrvQueue));
    hmq::SrvQueue* q = reinterpret_cast<hmq::SrvQueue*>(sp->d_queue);
    unsigned* done = reinterpret_cast<unsigned*>(sp->d_queue + lay.done);
    // every polled word starts from zero; then the two head counts (producers .. games are adjacent words)
    HIPCHK(hipMemsetAsync(sp->d_queue, 0, sp->queueBytes, sT));
    sp->h_qinit[0] = (unsigned)searchWgs; sp->h_qinit[1] = 0u; sp->h_qinit[2] = (unsigned)consumers; sp->h_qinit[3] = (unsigned)sp->nGames;
#ifdef HM_SEARCH_HB
    {
        unsigned* hbp = reinterpret_cast<unsigned*>(sp->d_queue + lay.hb);
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_hb), &hbp, sizeof(hbp)));
    }
#endif
    static_assert(offsetof(hmq::SrvQueue, error) == offsetof(hmq::SrvQueue, producers) + 4 && offsetof(hmq::SrvQueue, consumers) == offsetof(hmq::SrvQueue, producers) + 8 && offsetof(hmq::SrvQueue, games) == offsetof(hmq::SrvQueue, producers) + 12, "queue header layout");
    HIPCHK(hipMemcpyAsync(&q->producers, sp->h_qinit, 16, hipMemcpyHostToDevice, sT));
    hmq::ServeArgs a;
    a.planes[0] = static_cast<const uint16_t*>(io->planes[0]); a.planes[1] = static_cast<const uint16_t*>(io->planes[1]);
    a.value[0] = static_cast<uint16_t*>(io->value); a.value[1] = static_cast<uint16_t*>(io->value_2);
    a.piA[0] = static_cast<uint16_t*>(io->pi_a); a.piA[1] = static_cast<uint16_t*>(io->pi_a_2);
    a.piB[0] = static_cast<uint16_t*>(io->pi_b); a.piB[1] = static_cast<uint16_t*>(io->pi_b_2);
    a.wdl[0] = static_cast<uint16_t*>(io->wdl); a.wdl[1] = static_cast<uint16_t*>(io->wdl_2);
    a.ml[0] = static_cast<uint16_t*>(io->moves_left); a.ml[1] = static_cast<uint16_t*>(io->moves_left_2);
    a.q = q; a.done = done;
    a.leafMoves = sp->pl.leafMoves; a.leafCounts = sp->pl.leafCounts; a.sortedMoves = sp->pl.sortedMoves; a.sortedPriors = sp->pl.sortedPriors;
    a.polNormal = sp->pl.polNormal; a.polDrop = sp->pl.polDrop; a.noise = sp->pl.noise; a.noiseOn = (sp->alpha > 0.0f && sp->eps > 0.0f) ? 1 : 0; a.noiseEps = sp->eps;
    a.clkSum = reinterpret_cast<hmq::u64q*>(&sp->pl.clk->sumN); a.clkCnt = reinterpret_cast<hmq::u64q*>(&sp->pl.clk->cntN);   // the evaluator's ticks / positions
    a.abortAfter = 0u;
    {   // test hook: HM_SEARCH_ABORT_EVERY=k gives every k-th search up HALF-WAY — an evaluator workgroup raises "stalled" once
        // HM_SEARCH_ABORT_AFTER rows (default 150) have been published; trees, transposition tables and game records stay as they were then
        static int abortCount = 0;
        const char* ae = std::getenv("HM_SEARCH_ABORT_EVERY");
        const int every = ae ? std::atoi(ae) : 0;
        if (every > 0 && ++abortCount % every == 0) { const char* an = std::getenv("HM_SEARCH_ABORT_AFTER"); a.abortAfter = (unsigned)std::max(1, an ? std::atoi(an) : 150); }
    }
    SearchIo sio;
    sio.planes[0] = static_cast<uint16_t*>(io->planes[0]); sio.planes[1] = static_cast<uint16_t*>(io->planes[1]);
    for (int b = 0; b < 2; ++b) sio.out[b] = NetOut{a.value[b], a.piA[b], a.piB[b], a.wdl[b], a.ml[b]};
    sio.q = q; sio.done = done; sio.diag = reinterpret_cast<hmq::GameDiag*>(sp->d_queue + lay.diag); sio.netSel = nullptr; sio.ldsNodes = sp->searchLdsNodes;
    RolloutNet rn{ni.d_nd, ni.d_wh, ni.d_wf, ni.copMax, ni.uHalfs, itemOff};
    int nGames = sp->nGames, perWgArg = perWg, searchWgsArg = searchWgs;
    void* args[] = {&sp->pl, &sp->prm, &sio, &rn, &a, &nGames, &perWgArg, &searchWgsArg};
    (void)hipEventRecord(sp->evT0, sT);            // HIP events on the stream the kernel is launched on: its launch duration
    const hipError_t le = hipLaunchKernel(kern, dim3((unsigned)(searchWgs + consumers)), dim3(64u * (unsigned)waves), args, ldsBytes, sT);
    (void)hipEventRecord(sp->evT1, sT);
    if (le != hipSuccess) { (void)hipGetLastError(); return hm_fail(HM_ERR_NO_DEVICE, std::string("k_rollout launch failed: ") + hipGetErrorString(le)); }
    // the queue header comes down behind the kernel into pinned memory: one synchronisation for both
    hmq::SrvQueue* const hqp = reinterpret_cast<hmq::SrvQueue*>(reinterpret_cast<unsigned char*>(sp->h_qinit) + 64);
    HIPCHK(hipMemcpyAsync(hqp, q, offsetof(hmq::SrvQueue, slots), hipMemcpyDeviceToHost, sT));
    HIPCHK(hipStreamSynchronize(sT));
#ifdef HM_SEARCH_HB
    {
        unsigned* hbp = nullptr;                                       // the lockstep kernels share the device functions
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_hb), &hbp, sizeof(hbp)));
    }
#endif
    if (search_kernel_ms) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sp->evT0, sp->evT1) == hipSuccess) *search_kernel_ms = ms; else { *search_kernel_ms = 0.0; (void)hipGetLastError(); }
    }
    if (std::getenv("HM_SEARCH_LOG")) {                            // measurement aid: one line per search
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sp->evT0, sp->evT1) != hipSuccess) (void)hipGetLastError();
        std::fprintf(stderr, "[hm_sp_search] act %d mode %d waves %d perWg %d consumers %d ms %.3f\n", actNow, mode, waves, perWg, consumers, ms);
    }
    const hmq::SrvQueue& hq = *hqp;                                // header only (the slots follow it on the device)
    sp->lastQueueError = hq.error;
    if (!hq.error) {
        // test hook: HM_SEARCH_FAKE_STALL_EVERY=k reports every k-th COMPLETED search as stalled (HM_SEARCH_ABORT_EVERY above abandons
        // one half-way); either way the caller's recovery path (hm_sp_search_stalled -> hm_sp_begin_again -> the search once more) runs
        const char* fe = std::getenv("HM_SEARCH_FAKE_STALL_EVERY");
        const int fakeEvery = fe ? std::atoi(fe) : 0;
        static int fakeCount = 0;
        if (fakeEvery > 0 && ++fakeCount % fakeEvery == 0) {
            sp->lastQueueError = 5u;
            return hm_fail(HM_ERR_STATE, "single-launch search reported as stalled (HM_SEARCH_FAKE_STALL_EVERY)");
        }
    }
    if (hq.error) {
        // The give-up record.  Per game that had work: phase (1 collecting, 2 collected, 3 waiting for the evaluator, 4 processing, 5 the wait
        // failed), XCD, iteration, age of that mark, whether it had LEFT the kernel, and for a waiting game the rows done against the
        // rows it expects on the buffer it waits for.  From the snapshot the evaluator took before raising the error when there is one.
        const size_t G_ = (size_t)sp->nGames;
        std::vector<unsigned> dn(G_ * 2), sdn(G_ * 2);
        std::vector<hmq::GameDiag> dg(G_), sdg(G_);
        unsigned snapTime = 0;
        std::string where, census;
        if (hipMemcpy(dn.data(), sp->d_queue + lay.done, G_ * 8, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(dg.data(), sp->d_queue + lay.diag, G_ * sizeof(hmq::GameDiag), hipMemcpyDeviceToHost) == hipSuccess
            && hipMemcpy(sdn.data(), sp->d_queue + lay.snapDone, G_ * 8, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(sdg.data(), sp->d_queue + lay.snapDiag, G_ * sizeof(hmq::GameDiag), hipMemcpyDeviceToHost) == hipSuccess
            && hipMemcpy(&snapTime, sp->d_queue + lay.snapTime, 4, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long sumDone = 0, sumPub = 0;
            for (unsigned v : dn) sumDone += v;
            for (const auto& d : dg) sumPub += d.published;
            census = "published " + std::to_string(sumPub) + " tickets " + std::to_string(hq.tail >= (unsigned)consumers && hq.producers == 0 ? hq.tail - (unsigned)consumers : hq.tail)
                     + " evaluated " + std::to_string(hq.served) + " sum(done) " + std::to_string(sumDone);
            const bool snapped = hq.dbgPop[0] != 0;
            const std::vector<hmq::GameDiag>& D = snapped ? sdg : dg;
            const std::vector<unsigned>& DN = snapped ? sdn : dn;
            unsigned now = snapTime;
            if (!snapped) for (const auto& d : D) now = std::max(now, std::max(d.stamp, d.left));
            where = snapped ? " [snapshot at the give-up]" : " [final state]";
            int inNow = 0, waiting = 0, shortGames = 0;
            for (size_t g = 0; g < G_; ++g) {
                const hmq::GameDiag& d = D[g];
                if (!d.phase) continue;
                const bool left = d.left != 0;
                if (!left) ++inNow;
                where += " g" + std::to_string(g) + ":x" + std::to_string((d.phase >> 4) & 15u) + "p" + std::to_string(d.phase & 15u) + "/i" + std::to_string(d.phase >> 8)
                         + "/-" + std::to_string((int)(now - d.stamp) / 100) + "ms" + (left ? "/left" : "/IN") + "/cu" + std::to_string((d.hwId >> 8) & 15u) + "se" + std::to_string((d.hwId >> 13) & 7u);
                if (!left && d.waitBuf) {
                    ++waiting;
                    const unsigned have = DN[g * 2 + (d.waitBuf - 1)];
                    if (have < d.waitExpect) ++shortGames;
                    where += "/wait b" + std::to_string(d.waitBuf - 1) + " done " + std::to_string(have) + " of " + std::to_string(d.waitExpect);
                }
            }
            where = " games in " + std::to_string(inNow) + ", waiting " + std::to_string(waiting) + " (short of rows: " + std::to_string(shortGames) + ");" + where;
        } else (void)hipGetLastError();
        const std::string msg = "single-launch search gave up waiting (queue error " + std::to_string(hq.error) + "; head " + std::to_string(hq.head) + " tail "
                       + std::to_string(hq.tail) + " beats " + std::to_string(hq.beats) + " producers left " + std::to_string(hq.producers) + " served " + std::to_string(hq.served) + "; search workgroups in/out "
                       + std::to_string(hq.treesIn) + "/" + std::to_string(hq.treesOut) + " of " + std::to_string(searchWgs) + ", evaluator workgroups in/out "
                       + std::to_string(hq.consIn) + "/" + std::to_string(hq.consOut) + " of " + std::to_string(consumers) + "; first failed wait: game "
                       + std::to_string((int)hq.dbg[0] - 1) + " buffer " + std::to_string(hq.dbg[1]) + " expected " + std::to_string(hq.dbg[2]) + " done " + std::to_string(hq.dbg[3])
                       + " iteration " + std::to_string(hq.dbg[4]) + " waited ms " + std::to_string(hq.dbg[5]) + "; evaluator that gave up: ticket "
                       + std::to_string((int)hq.dbgPop[0] - 1) + " tail " + std::to_string(hq.dbgPop[1]) + "; tickets drawn twice " + std::to_string(hq.dupTickets) + "; census " + census
                       + ";" + where + ")";
        if (!a.abortAfter) std::fprintf(stderr, "[hivemind_amd] %s\n", msg.c_str());     // a give-up is never silent (the test hook's is expected)
        return hm_fail(HM_ERR_STATE, msg);
    }
    return 0;
}

int hm_sp_root_stats(hm_sp* sp, int* counts, hm_move* move_a, hm_move* move_b, int* visits, float* q, float* prior, float* root_q, int* info, int max_edges) {
    if (!sp || !counts) return hm_fail(HM_ERR_INVALID, "null argument");
    if (max_edges != sp->maxEdges) return hm_fail(HM_ERR_INVALID, "max_edges must equal hm_sp_max_edges()");
    const size_t G_ = sp->nGames, E = (size_t)sp->maxEdges;
    hipLaunchKernelGGL(k_root_stats, dim3(sp->nGames), dim3(64), 0, 0, sp->pl, sp->prm, sp->ro);
    HIPCHK(hipGetLastError());
    // counts | rootQ | info | moveA | moveB | visits are one device block: one download into the pinned stage
    HIPCHK(hipMemcpy(sp->h_stage, sp->d_out, sp->outRootBytes, hipMemcpyDeviceToHost));
    auto at = [&](const void* dev) { return sp->h_stage + (static_cast<const unsigned char*>(dev) - sp->d_out); };
    std::memcpy(counts, at(sp->ro.counts), 4 * G_);
    if (move_a) std::memcpy(move_a, at(sp->ro.moveA), 4 * G_ * E);
    if (move_b) std::memcpy(move_b, at(sp->ro.moveB), 4 * G_ * E);
    if (visits) std::memcpy(visits, at(sp->ro.visits), 4 * G_ * E);
    if (root_q) std::memcpy(root_q, at(sp->ro.rootQ), 4 * G_);
    if (info) std::memcpy(info, at(sp->ro.info), 4 * G_ * HM_SP_INFO_INTS);
    if (q) HIPCHK(hipMemcpy(q, sp->ro.q, 4 * G_ * E, hipMemcpyDeviceToHost));
    if (prior) HIPCHK(hipMemcpy(prior, sp->ro.prior, 4 * G_ * E, hipMemcpyDeviceToHost));
    return 0;
}
int hm_sp_max_edges(const hm_sp* sp) { return sp ? sp->maxEdges : 0; }
// Principal variations of one game's finished (or stopped) search: see k_pv.  child_idx[n_lines] are root edge indices (the caller
// orders them: visit count, solver-aware best move first -- agent.cc:917-940); moves[n_lines][max_depth][2], lens / child_type /
// child_end_in_ply [n_lines].
int hm_sp_pv_lines(hm_sp* sp, int game, int n_lines, const int* child_idx, int max_depth, hm_move* moves, int* lens, int* child_type, int* child_end_in_ply) {
    if (!sp || !child_idx || !moves || !lens) return hm_fail(HM_ERR_INVALID, "null argument");
    if (game < 0 || game >= sp->nGames || n_lines < 1 || n_lines > 500 || max_depth < 1 || max_depth > 64) return hm_fail(HM_ERR_INVALID, "game / n_lines (1..500) / max_depth (1..64) out of range");
    const size_t nInts = (size_t)n_lines * (4 + 2 * (size_t)max_depth);
    if (sp->pvInts < nInts) {                               // idx | lens | type | end | moves
        int* d = nullptr;
        if (int rc = dalloc(sp, &d, nInts)) return rc;
        sp->d_pv = d; sp->pvInts = nInts;
    }
    int* d = sp->d_pv;
    HIPCHK(hipMemcpy(d, child_idx, sizeof(int) * n_lines, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pv, dim3((n_lines + 63) / 64), dim3(64), 0, 0, sp->pl, sp->prm, game, n_lines, d, max_depth,
                       reinterpret_cast<u32*>(d + 4 * (size_t)n_lines), d + n_lines, d + 2 * (size_t)n_lines, d + 3 * (size_t)n_lines);
    HIPCHK(hipGetLastError());
    std::vector<int> h(nInts);
    HIPCHK(hipMemcpy(h.data(), d, sizeof(int) * nInts, hipMemcpyDeviceToHost));
    std::memcpy(lens, h.data() + n_lines, sizeof(int) * n_lines);
    if (child_type) std::memcpy(child_type, h.data() + 2 * (size_t)n_lines, sizeof(int) * n_lines);
    if (child_end_in_ply) std::memcpy(child_end_in_ply, h.data() + 3 * (size_t)n_lines, sizeof(int) * n_lines);
    std::memcpy(moves, h.data() + 4 * (size_t)n_lines, sizeof(int) * 2 * (size_t)max_depth * n_lines);
    return 0;
}

// diagnostic (-DHM_SEARCH_TRACE builds): select the traced game slot (clears the log) / read the log
int hm_sp_trace_select(int game) {
    unsigned int z = 0;
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_traceGame), &game, sizeof game));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_traceCount), &z, sizeof z));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_traceSeq), &z, sizeof z));
    return HM_OK;
}
int hm_sp_trace(unsigned long long* out, int cap) {
    unsigned int n = 0;
    HIPCHK(hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_traceCount), sizeof n));
    if (n > 65536u) n = 65536u;
    const int m = (int)n < cap ? (int)n : cap;
    if (m > 0) HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * (size_t)m));
    return (int)n;
}
// diagnostic (all zeros unless built with -DHM_SEARCH_PROF): out[0..31] cycles, out[32..63] counts; reset != 0 clears
int hm_sp_profile(unsigned long long* out64, int reset) {
    if (out64) HIPCHK(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 128));       // 64 sums, then 64 call counts
    if (reset) { unsigned long long z[128] = {}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z)); }
    return HM_OK;
}
int hm_sp_profile_launches(unsigned int* out, int launches) {   // [launches][64] traversal cycles per k_collect launch and game slot
#ifdef HM_SEARCH_PROF
    if (!out || launches < 0 || launches > PROF_LAUNCHES) return hm_fail(HM_ERR_INVALID, "bad launch count");
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_colDur), sizeof(unsigned int) * 64 * (size_t)launches));
    return HM_OK;
#else
    (void)out; (void)launches;
    return hm_fail(HM_ERR_STATE, "library built without -DHM_SEARCH_PROF");
#endif
}
int hm_sp_active_on(hm_sp* sp, int* pinned_out, hipStream_t stream) {
    HIPCHK(hipMemcpyAsync(pinned_out, sp->d_active, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
}
// number of games still searching after the last hm_sp_process (synchronises the null stream)
int hm_sp_active(hm_sp* sp, int* active) {
    if (!sp || !active) return hm_fail(HM_ERR_INVALID, "null argument");
    HIPCHK(hipMemcpy(active, sp->d_active, sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

static int apply_launch(hm_sp* sp, const hm_move* move_a, const hm_move* move_b, const uint8_t* mask) {
    if (!sp || !move_a || !move_b) return hm_fail(HM_ERR_INVALID, "null argument");
    const size_t G_ = sp->nGames;
    // mask | moveA | moveB | error flag (cleared) are adjacent in the input block: one staged upload
    unsigned char* hs = sp->h_stage;
    const size_t from = (size_t)(sp->d_mask - sp->d_in);
    if (mask) std::memcpy(hs + from, mask, G_);
    std::memcpy(hs + (reinterpret_cast<unsigned char*>(sp->d_moveA) - sp->d_in), move_a, 4 * G_);
    std::memcpy(hs + (reinterpret_cast<unsigned char*>(sp->d_moveB) - sp->d_in), move_b, 4 * G_);
    std::memset(hs + (reinterpret_cast<unsigned char*>(sp->d_applyErr) - sp->d_in), 0, 4);
    HIPCHK(hipMemcpy(sp->d_in + from, hs + from, sp->inBytes - from, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_apply, dim3(sp->nGames), dim3(64), 0, 0, sp->pl, sp->prm, sp->d_moveA, sp->d_moveB, mask ? sp->d_mask : nullptr, sp->d_applyErr);
    HIPCHK(hipGetLastError());
    return 0;
}
int hm_sp_apply(hm_sp* sp, const hm_move* move_a, const hm_move* move_b, const uint8_t* mask) {
    if (int rc = apply_launch(sp, move_a, move_b, mask)) return rc;
    int* err = reinterpret_cast<int*>(sp->h_stage);
    HIPCHK(hipMemcpy(err, sp->d_applyErr, sizeof(int), hipMemcpyDeviceToHost));
    if (*err) return hm_fail(HM_ERR_OVERFLOW, "game history pool full (raise max_game_plies of hm_sp_create_ex)");
    return 0;
}
// hm_sp_apply without the round trip for its error flag: the flag is checked by the hm_sp_game_state call that follows (the self-play
// driver's per-ply chain: apply -> game state is one synchronisation instead of two).  Internal (hm_host.hpp), not part of the C ABI.
int hm_sp_apply_deferred(hm_sp* sp, const hm_move* move_a, const hm_move* move_b, const uint8_t* mask) {
    if (int rc = apply_launch(sp, move_a, move_b, mask)) return rc;
    sp->applyPending = true;
    return 0;
}
int hm_sp_set_side(hm_sp* sp, const uint8_t* team, const uint8_t* time_adv) {
    if (!sp || !team || !time_adv) return hm_fail(HM_ERR_INVALID, "null argument");
    const size_t G_ = sp->nGames;
    unsigned char* hs = sp->h_stage;
    std::memcpy(hs, team, G_); std::memcpy(hs + G_, time_adv, G_);
    unsigned char* d = reinterpret_cast<unsigned char*>(sp->d_moveA);          // scratch: 8 * G bytes of the input block
    HIPCHK(hipMemcpy(d, hs, 2 * G_, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_set_side, dim3((sp->nGames + 63) / 64), dim3(64), 0, 0, sp->pl, sp->nGames, d, d + G_);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
}
// Tree reuse between the searches of a game slot (Agent::try_reuse_tree / store_next_root_candidates, agent.cc:1345-1451):
// mode[g] (NULL = leave as is) 0 off (self-play and tournaments: reset_search_state before every search), 1 reuse when the node
// budget fits behind the nodes already in the pool, 2 also shrink the budget to what fits (time-limited searches); reset != 0
// forgets the previous trees (Agent::reset_search_state: ucinewgame).
int hm_sp_set_tree_reuse(hm_sp* sp, const uint8_t* mode, int reset) {
    if (!sp) return hm_fail(HM_ERR_INVALID, "null argument");
    unsigned char* d = reinterpret_cast<unsigned char*>(sp->d_moveA);          // scratch: 8 * G bytes of the input block
    if (mode) {
        std::memcpy(sp->h_stage, mode, (size_t)sp->nGames);
        HIPCHK(hipMemcpy(d, sp->h_stage, (size_t)sp->nGames, hipMemcpyHostToDevice));
        sp->anyReuse = false;
        for (int g = 0; g < sp->nGames; ++g) sp->anyReuse |= mode[g] != 0;
    }
    hipLaunchKernelGGL(k_set_reuse, dim3((sp->nGames + 63) / 64), dim3(64), 0, 0, sp->pl, sp->nGames, mode ? d : nullptr, reset);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
}
int hm_sp_stop(hm_sp* sp, const uint8_t* mask, void* stream) {
    if (!sp) return hm_fail(HM_ERR_INVALID, "null argument");
    if (mask) HIPCHK(hipMemcpyAsync(sp->d_mask, mask, sp->nGames, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    hipLaunchKernelGGL(k_stop, dim3((sp->nGames + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), sp->pl, sp->nGames, mask ? sp->d_mask : nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}
int hm_sp_game_state(hm_sp* sp, hm_board* boards, int* flags, void* d_boards_out) {
    if (!sp) return hm_fail(HM_ERR_INVALID, "null argument");
    const size_t G_ = sp->nGames;
    hipLaunchKernelGGL(k_game_state, dim3(sp->nGames), dim3(64), 0, 0, sp->pl, sp->prm, sp->d_boards, sp->d_flags);
    HIPCHK(hipGetLastError());
    if (d_boards_out) HIPCHK(hipMemcpyAsync(d_boards_out, sp->d_boards, sizeof(hm_board) * G_, hipMemcpyDeviceToDevice, nullptr));
    const unsigned char* from = reinterpret_cast<const unsigned char*>(sp->d_boards);
    const size_t bytes = (size_t)(reinterpret_cast<const unsigned char*>(sp->d_flags) - from) + 4 * G_;
    int* const applyErr = reinterpret_cast<int*>(sp->h_stage + ((bytes + 15) & ~(size_t)15));     // behind the boards | flags image in the pinned stage
    const bool checkApply = sp->applyPending;
    sp->applyPending = false;
    if (checkApply) HIPCHK(hipMemcpyAsync(applyErr, sp->d_applyErr, sizeof(int), hipMemcpyDeviceToHost, nullptr));   // hm_sp_apply_deferred's flag rides along
    if (boards || flags) {
        // boards | flags are adjacent at the end of the output block: one download
        HIPCHK(hipMemcpyAsync(sp->h_stage, from, bytes, hipMemcpyDeviceToHost, nullptr));
    }
    HIPCHK(hipStreamSynchronize(nullptr));
    if (boards) std::memcpy(boards, sp->h_stage, sizeof(hm_board) * G_);
    if (flags) std::memcpy(flags, sp->h_stage + (reinterpret_cast<const unsigned char*>(sp->d_flags) - from), 4 * G_);
    if (checkApply && *applyErr) return hm_fail(HM_ERR_OVERFLOW, "game history pool full (raise max_game_plies of hm_sp_create_ex)");
    return 0;
}
int hm_sp_raw_policy(hm_sp* sp, const void* d_pi_a, const void* d_pi_b, hm_move* moves, float* probs, uint8_t* caps, int* counts, uint8_t* on_turn) {
    return hm_sp_policy_listing(sp, d_pi_a, d_pi_b, moves, probs, caps, counts, on_turn, 0);
}
int hm_sp_policy_listing(hm_sp* sp, const void* d_pi_a, const void* d_pi_b, hm_move* moves, float* probs, uint8_t* caps, int* counts, uint8_t* on_turn, int all_moves) {
    if (!sp || !d_pi_a || !d_pi_b || !moves || !probs || !counts) return hm_fail(HM_ERR_INVALID, "null argument");
    const size_t G_ = sp->nGames;
    hipLaunchKernelGGL(k_raw_policy, dim3(sp->nGames), dim3(64), 0, 0, sp->pl, sp->prm, static_cast<const uint16_t*>(d_pi_a),
                       static_cast<const uint16_t*>(d_pi_b), sp->raw, all_moves);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(moves, sp->raw.moves, 4 * G_ * 2 * HM_MAX_MOVES, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(probs, sp->raw.probs, 4 * G_ * 2 * HM_MAX_MOVES, hipMemcpyDeviceToHost));
    if (caps) HIPCHK(hipMemcpy(caps, sp->raw.caps, G_ * 2 * HM_MAX_MOVES, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(counts, sp->raw.counts, 4 * G_ * 2, hipMemcpyDeviceToHost));
    if (on_turn) HIPCHK(hipMemcpy(on_turn, sp->raw.onTurn, G_ * 2, hipMemcpyDeviceToHost));
    return 0;
}

int hm_sp_action_terminal(hm_sp* sp, const hm_move* move_a, const hm_move* move_b, int* out) {
    if (!sp || !move_a || !move_b || !out) return hm_fail(HM_ERR_INVALID, "null argument");
    const size_t G_ = sp->nGames;
    HIPCHK(hipMemcpy(sp->d_moveA, move_a, 4 * G_, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(sp->d_moveB, move_b, 4 * G_, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_action_terminal, dim3(sp->nGames), dim3(64), 0, 0, sp->pl, sp->prm, sp->d_moveA, sp->d_moveB, sp->d_term);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, sp->d_term, 4 * G_, hipMemcpyDeviceToHost));
    return 0;
}

int hm_sp_classify(hm_sp* sp, const int* args4, int* out4) {
    if (!sp || !args4 || !out4) return hm_fail(HM_ERR_INVALID, "null argument");
    const size_t G_ = sp->nGames;
    int *d_args = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc(&d_args, 16 * G_));
    if (hipMalloc(&d_out, 16 * G_) != hipSuccess) { (void)hipFree(d_args); return hm_fail(HM_ERR_NO_DEVICE, "hipMalloc failed"); }
    hipError_t e = hipMemcpy(d_args, args4, 16 * G_, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_classify, dim3(sp->nGames), dim3(64), 0, 0, sp->pl, sp->prm, d_args, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out4, d_out, 16 * G_, hipMemcpyDeviceToHost);
    (void)hipFree(d_args); (void)hipFree(d_out);
    if (e != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, std::string("hm_sp_classify: ") + hipGetErrorString(e));
    return 0;
}

int hm_rules_probe(const hm_board* d_boards, size_t n, int* d_out, uint64_t* d_keys) {
    if (int rc = ensure_search_tables()) return rc;
    if (!n) return 0;
    hipLaunchKernelGGL(k_rules_probe, dim3((unsigned)std::min<size_t>(n, 4096)), dim3(64), 0, 0, g_rules_dev, d_boards, (int)n, d_out, d_keys);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
}

}  // extern "C"

#ifdef HM_SINGLE_TU
// diagnostic builds: the single-launch search in this translation unit (Makefile: the probes of hm_prof.hpp are read back from here)
#define HM_ROLLOUT_MODE 0
#include "hm_rollout.hip"
#endif
