// hm_rollout.hip — the node-budget search of every game as ONE kernel launch: k_rollout.
//
// Workgroups [0, searchWgs) are games (search_role: one game per workgroup for its whole search, node pool in LDS; or search_role_mg:
// one workgroup serving several games in turn), workgroups [searchWgs, gridDim.x) are the evaluator (serve_role: the fused RISEv3
// forward of hm_net_device.hpp, one position at a time, then the leaf's prior pipeline on the logits still in LDS), joined by the
// device-side queue of hm_queue.hpp.  Replaces the host loop collect -> forward -> process of Agent::run_search's workers
// (agent.cc:331-352, searchthread.cc:661-739) and Engine::enqueueInferenceHalf / synchronizeInferenceHalf (nn/engine.h:43-81).
//
// Round 3 ran the two roles as TWO kernels on two hardware queues that spun on each other: their progress depended on both queues
// being scheduled together, counter-collecting profilers (which run kernels one at a time) could not run them at all, and the
// evaluator's stream had to be held back until the games had their CUs (k_wait_trees).  One launch has none of that: the grid is
// sized to be resident at once (one workgroup per CU: both roles take a CU's LDS), workgroups are dispatched in index order, so the
// games are in before the first evaluator workgroup, and an evaluator workgroup that finds no CU merely starts later (every one of
// them ends on a poison ticket of its own).
//
// (The workgroup size is a template parameter: see k_rollout.)
// Compiled once per search role: -DHM_ROLLOUT_MODE=0 (node pool in LDS), 1 (tree walked in place), 2 (several games per search
// workgroup), each for the network variants of rollout_kernel(); diagnostic builds (-DHM_SINGLE_TU) include this file from
// hm_search.hip instead, so that the probes of hm_prof.hpp land in the one translation unit that reads them.
#include "hm_search_device.hpp"
#include "hm_net_device.hpp"
#include <cstdlib>

namespace hms {

// One game's whole node-budget search on the first four waves of a workgroup (the body of round 3's k_search).  g: game slot.
template <bool LDS_TREE>
__device__ __forceinline__ void search_role(unsigned char* smem, const Pools& pl, const Params& prm, const SearchIo& io, const int g) {
    SearchLds<LDS_TREE>& S_ = *reinterpret_cast<SearchLds<LDS_TREE>*>(smem);
    unsigned char* const s_nodes = smem + search_lds_bytes(LDS_TREE ? 0 : 1);
    RulesTab& s_rt = S_.rt;
    WaveLds& L = S_.L;
    int& s_nextLeaf = S_.nextLeaf;
    Game& s_game = S_.game;
    constexpr int TABN = ROLE_TABN;
    float (&s_cpuct)[TABN] = S_.cpuct;
    uint16_t (&s_pwRoot)[TABN] = S_.pwRoot;
    uint16_t (&s_pwNode)[TABN] = S_.pwNode;
    SearchCtl& s_ctl = S_.ctl;
    unsigned (&s_expect)[2] = S_.expect;
    PubCtx& s_pub = S_.pub;
    auto& s_hist = S_.hist;
    hmq::GameDiag* const diag = io.diag + g;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    PROF_INIT();
    G s = make_view(pl, prm, g);
    Game* const gGame = s.g;
    Node* const gNodes = s.nodes;
    for (unsigned i = threadIdx.x; i < sizeof(Game) / 4; i += COLLECT_THREADS) reinterpret_cast<u32*>(&s_game)[i] = reinterpret_cast<const u32*>(gGame)[i];
    stage_table_wide(&s_rt, pl.rules, COLLECT_THREADS);
    {
        const bool alt = gGame->pwSel != 0;
        const int* pwr = alt ? pl.pwRootAlt : pl.pwRoot;
        const int* pwn = alt ? pl.pwNodeAlt : pl.pwNode;
        for (int i = threadIdx.x; i < TABN; i += COLLECT_THREADS) { s_cpuct[i] = pl.cpuctTab[i]; s_pwRoot[i] = (uint16_t)min(pwr[i], 65535); s_pwNode[i] = (uint16_t)min(pwn[i], 65535); }
    }
    if (threadIdx.x == 0) { s_expect[0] = s_expect[1] = 0; s_ctl.action = ACT_COLLECT; s_ctl.buf = 0; s_ctl.first = 1; s_ctl.ok = 1; L.listWords = 0; }
    __syncthreads();
    const bool searching = s_game.status == ST_SEARCHING;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(hmq::G32(&io.q->treesIn), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!searching) {                                               // idle slot, or a search k_begin already settled
        if (threadIdx.x == 0) { __hip_atomic_fetch_add(hmq::G32(&io.q->treesOut), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); hmq::producer_exit(io.q); }
        return;
    }
    constexpr bool mirror = LDS_TREE;                               // (io.ldsNodes says the same: the host picks the instantiation)
    if (mirror) {
        const int words = s_game.nodeCount * 4;
        const uint4* src = reinterpret_cast<const uint4*>(gNodes);
        uint4* dst = reinterpret_cast<uint4*>(s_nodes);
        for (int i = threadIdx.x; i < words; i += COLLECT_THREADS) dst[i] = src[i];
        s.nodes = reinterpret_cast<Node*>(s_nodes); s.ldsTree = true;
        __syncthreads();
    }
    s.g = &s_game;
    s.ldsCpuct = s_cpuct; s.ldsPwRoot = s_pwRoot; s.ldsPwNode = s_pwNode; s.tabN = TABN;
    // (every wave: pointers that depend on the wave index compile to flat accesses; only the traversal wave uses them)
    s.ackSeq = &L.ackSeq; s.typeSeq = &L.typeSeq; s.createSeq = &L.createSeq; s.createFast = &L.createRes.fast; s.gq = &L.gq; s.genAckSeq = &L.gq.ackSeq;
    // The game's repetition keys (read by every draw test and hash of the classifier wave, with the search path's keys rebuilt behind
    // them per leaf: path_rebuild_history) in LDS when the game's history and the longest path fit; nothing to write back — the search
    // only appends scratch behind the game's own keys.
    if constexpr (LDS_TREE) {                                       // (unconditional: a pointer that may be either compiles to flat accesses; hm_sp_create_ex
        for (int b = 0; b < 2; ++b) {                               //  admits this instantiation only when Params::histCap fits SEARCH_HIST_LDS)
            for (int i = threadIdx.x; i < s_game.hlen[b]; i += COLLECT_THREADS) s_hist[b][i] = s.hist[b][i];
            s.hist[b] = s_hist[b];
        }
        __syncthreads();
    }
    const int rootTeam = s_game.team;
    const bool rootAdv = s_game.adv != 0;
    const int rowBase = g * BATCH;
    ExpLds& myExp = L.exp;                                         // never touched: every expansion of this role takes the evaluator's pre-sorted priors (expand_leaf with `pre`); process_batch on wave 0 is its only user
    const PreSorted pre{pl.sortedMoves + (size_t)g * 2 * BATCH * 2 * HM_MAX_MOVES, pl.sortedPriors + (size_t)g * 2 * BATCH * 2 * HM_MAX_MOVES};
    u64 tC = 0, tW = 0, tP = 0, nIt = 0;                            // thread 0: ticks spent collecting / waiting for the evaluator / processing
    unsigned xcc;                                                   // which XCD this workgroup runs on (diagnostics of a give-up)
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    auto mark = [&](unsigned phase) {                               // thread 0 only: the diagnostics record and the hang guard's beat
        __hip_atomic_store(hmq::G32(&diag->phase), phase | (xcc << 4) | ((unsigned)nIt << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hmq::G32(&diag->stamp), (unsigned)((u64)__builtin_amdgcn_s_memrealtime() / 1000ULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(hmq::G32(&io.q->beats), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto mark_wait = [&](int buf) {                                 // thread 0 only: what the game is about to wait for
        __hip_atomic_store(hmq::G32(&diag->published), s_expect[0] + s_expect[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hmq::G32(&diag->waitExpect), s_expect[buf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hmq::G32(&diag->waitBuf), (unsigned)buf + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mark(3u);
    };
    if (threadIdx.x == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        __hip_atomic_store(hmq::G32(&diag->hwId), hw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // process of the pending batch: wait for its evaluation (unless `abortIt`), backups on wave 0 beside the expansions on waves 1..3
    auto process_pending = [&](bool abortIt) -> bool {
        const int pending = s_game.pending;
        u64 t0 = 0;
        if (threadIdx.x == 0) {
            t0 = __builtin_amdgcn_s_memrealtime();
            bool ok = true;
            mark_wait(pending);
            if (!abortIt && s_game.validCount[pending] > 0) {
                ok = hmq::wait_count(io.q, &io.done[g * 2 + pending], s_expect[pending]);
                if (ok) hmq::acquire_agent();
                else if (atomicCAS(&io.q->dbg[0], 0u, (unsigned)g + 1u) == 0u) {
                    io.q->dbg[1] = (unsigned)pending; io.q->dbg[2] = s_expect[pending];
                    io.q->dbg[3] = __hip_atomic_load(hmq::G32(&io.done[g * 2 + pending]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    io.q->dbg[4] = (unsigned)nIt; io.q->dbg[5] = (unsigned)(((u64)__builtin_amdgcn_s_memrealtime() - t0) / 100000ULL);
                }
            }
            s_ctl.ok = ok ? 1 : 0;
            if (ok) { diag->waitBuf = 0u; mark(4u); } else diag->phase = 5u | (xcc << 4) | ((unsigned)nIt << 8);     // 5: the wait failed (the stamp stays at its start)
            const u64 t1 = __builtin_amdgcn_s_memrealtime();
            tW += t1 - t0; t0 = t1;
        }
        PROF_T(tpw);
        __syncthreads();
        PROF_ADD(35, tpw);
        PROF_T(tpp);
        if (!s_ctl.ok) return false;
        const int nctx = s_game.ctxCount[pending];
        __syncthreads();                                            // every wave holds the batch header before wave 0 retires it
        if (wave == 0) {
            PROF_T(tpb);
            if (abortIt) abort_batch(s, pending);                   // discard_pending_iteration (agent.cc:343-352)
            else backup_batch(s, pending, &io.out[pending], rowBase);
            PROF_ADD(37, tpb);
        } else if (!abortIt) {
            PROF_T(tpe);
            expand_share(s, s_rt, myExp, wave, pending, nctx, rootTeam, rootAdv, &io.out[pending], rowBase, &pre);
            PROF_ADD_T(43, tpe, 64);
        }
        __threadfence_block();
        __syncthreads();
        PROF_ADD(36, tpp);
        if (threadIdx.x == 0) tP += __builtin_amdgcn_s_memrealtime() - t0;
        return true;
    };
    for (;;) {
        // ---- control: worker loop (agent.cc:331-341) + run_iteration head (searchthread.cc:661-678)
        PROF_T(tct);
        if (threadIdx.x == 0) {
            const bool fin = s_game.nodesSearched >= s_game.targetNodes || s.nodes[s_game.root].type != T_UNSOLVED || s_game.overflow;
            const bool first = s_game.pending < 0;
            s_ctl.action = fin ? ACT_FINISH : ACT_COLLECT;
            s_ctl.first = first ? 1 : 0;
            s_ctl.buf = first ? 0 : 1 - s_game.pending;
            // hand-off state of one collect phase
            L.posted = 0; L.done = 0; L.servedCnt = 0; L.servedCntB = 0; L.postCount = 0; L.reqSeq = 0; L.typeSeq = 0; L.ackSeq = 0; L.createSeq = 0; L.svcStop = 0;
            L.svcValid = 0; L.reqResult = 0; L.gq.reqSeq = 0; L.gq.ackSeq = 0;
            s_nextLeaf = 0;
            for (int i = 0; i < BATCH; ++i) L.postReady[i] = 0;
            // the root's own expansion (first batch of a search from a fresh root) mixes Dirichlet noise into the priors
            const bool rootRow = first && s_game.alpha > 0.0f && s_game.eps > 0.0f && !(s.nodes[s_game.root].flags & F_EXPANDED);
            const int bufNow = first ? 0 : 1 - s_game.pending;
            s_pub.q = io.q; s_pub.expect = &s_expect[bufNow];
            s_pub.itemBase = hmq::item_pack(g, bufNow, 0, io.netSel ? io.netSel[g] : 0) | (rootRow ? hmq::IT_ROOT : 0u);
        }
        __syncthreads();
        PROF_ADD(38, tct);
        if (s_ctl.action == ACT_FINISH) {
            // finish_pending / discard_pending_iteration (agent.cc:343-352)
            bool ok = true;
            if (s_game.pending >= 0) {
                const bool solvedOrOverflow = s.nodes[s_game.root].type != T_UNSOLVED || s_game.overflow;
                ok = process_pending(solvedOrOverflow);
            }
            if (threadIdx.x == 0) {
                s_game.pending = -1;
                if (!ok) s_game.overflow |= 128;
                s_game.status = s_game.overflow ? ST_ERROR : ST_DONE;
            }
            break;
        }
        const int buf = s_ctl.buf;
        const bool first = s_ctl.first != 0;
        u64 t0 = 0, twIter = 0;                                     // thread 0: start of the collect phase; this iteration's wait for the evaluator (inside it)
        if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memrealtime(); mark(1u); }
        // ---- collect phase (collect_batch, searchthread.cc:255-442) into plane buffer `buf`
        PROF_T(tcp);
        if (wave == 0) {
            s.inflight = -1; s.reqSeq = 0; s.svcBusy = false; s.genInflight = -1; s.genReqSeq = 0;
            int tail[2];
            collect_batch(s, s_rt, L, buf, rootTeam, rootAdv, tail);
            if (lane == 0) __hip_atomic_store(&L.svcStop, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            HB(19);
            if (!first) {
                // ---- while the classifier wave finishes this batch's last leaf (its creation step is over; what is left writes that
                // leaf's own node and context only, and no path of the batch before runs through it: it was reserved after that batch was
                // collected) and the other helpers their rows: the ordered backups of the batch whose evaluation was requested one
                // collect ago (process_batch, searchthread.cc:444-639).  The generator must be idle first: a sequential backup
                // rewrites whole nodes, `more` included.
                gen_wait(s);
                const int pending = s_game.pending;
                u64 tw0 = 0;
                if (threadIdx.x == 0) {
                    tw0 = __builtin_amdgcn_s_memrealtime();
                    bool ok = true;
                    mark_wait(pending);
                    if (s_game.validCount[pending] > 0) {
                        ok = hmq::wait_count(io.q, &io.done[g * 2 + pending], s_expect[pending]);
                        if (ok) hmq::acquire_agent();
                        else if (atomicCAS(&io.q->dbg[0], 0u, (unsigned)g + 1u) == 0u) {
                            io.q->dbg[1] = (unsigned)pending; io.q->dbg[2] = s_expect[pending];
                            io.q->dbg[3] = __hip_atomic_load(hmq::G32(&io.done[g * 2 + pending]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            io.q->dbg[4] = (unsigned)nIt; io.q->dbg[5] = (unsigned)(((u64)__builtin_amdgcn_s_memrealtime() - tw0) / 100000ULL);
                        }
                    }
                    s_ctl.ok = ok ? 1 : 0;
                    if (ok) { diag->waitBuf = 0u; mark(4u); } else diag->phase = 5u | (xcc << 4) | ((unsigned)nIt << 8);
                    twIter = __builtin_amdgcn_s_memrealtime() - tw0;
                    tW += twIter;
                }
                if (lane == 0) s_ctl.nctx = s_game.ctxCount[pending];      // the expansions' loop bound: backup_batch retires the batch header
                wave_fence();
                if (s_ctl.ok) {
                    PROF_T(tpb);
                    backup_batch(s, pending, &io.out[pending], rowBase);
                    PROF_ADD(37, tpb);
                }
            }
            collect_finish(s, L, buf, tail[0], tail[1] != 0);
        } else collect_helper_role<true>(s, s_rt, L, pl, s_game, g, io.planes[buf], wave, &s_pub);
        PROF_ADD(33, tcp);
        PROF_T(tcd);
        hmq::drain_stores();                                        // the write-through stores of this batch's plane rows have left every wave
        HB(50 + (threadIdx.x >> 6));
        __syncthreads();
        HB(60);
        PROF_ADD(39, tcd);
        PROF_ADD(32, tcp);
        PROF_T(tpu);
        if (threadIdx.x == 0) {                                     // (the batch's rows went to the evaluator one by one: post_arrive)
            const u64 t1 = __builtin_amdgcn_s_memrealtime();
            tC += t1 - t0 - twIter; nIt++;                          // (the backups of the batch before run inside this window too)
            mark(2u);
        }
        PROF_ADD(34, tpu);
        if (first) {
            // with nothing in flight the first batch went to buffer 0 and its lookahead follows
            if (wave == 0) {
                if (s_game.ctxCount[0] == 0) { if (lane == 0) s_game.overflow |= 16; }                      // no progress possible
                else if (s_game.validCount[0] == 0) process_batch(s, s_rt, L.exp, 0, rootTeam, rootAdv, nullptr, 0);
                else if (lane == 0) s_game.pending = 0;
            }
            __syncthreads();
            continue;
        }
        // ---- the rest of that batch's process step: its expansions (the backups ran on wave 0 beside the end of the collect phase;
        // the barrier above also carries wave 0's acquire of the evaluator's results to the other waves)
        if (!s_ctl.ok) {
            if (threadIdx.x == 0) { s_game.overflow |= 128; s_game.pending = -1; s_game.status = ST_ERROR; }
            break;
        }
        {
            PROF_T(tpp);
            u64 tp0 = 0;
            if (threadIdx.x == 0) tp0 = __builtin_amdgcn_s_memrealtime();
            const int pending = s_game.pending;
            const int nctx = s_ctl.nctx;
            {   // every wave (wave 0 has done its backups) draws the next leaf to expand
                PROF_T(tpe);
                for (;;) {
                    int i = 0;
                    if (lane == 0) i = atomicAdd(&s_nextLeaf, 1);
                    i = ulane(i, 0);
                    if (i >= nctx) break;
                    expand_context(s, s_rt, myExp, pending, i, rootTeam, rootAdv, &io.out[pending], rowBase, &pre);
                }
                PROF_ADD_T(43, tpe, 64);
            }
            __threadfence_block();
            __syncthreads();
            PROF_ADD(36, tpp);
            if (threadIdx.x == 0) tP += __builtin_amdgcn_s_memrealtime() - tp0;
        }
        PROF_T(ttl);
        if (wave == 0) {                                            // run_iteration tail
            const int look = 1 - s_game.pending;
            if (lane == 0) s_game.pending = -1;
            wave_fence();
            if (s_game.validCount[look] == 0) process_batch(s, s_rt, L.exp, look, rootTeam, rootAdv, nullptr, 0);
            else if (lane == 0) s_game.pending = look;
        }
        __syncthreads();
        PROF_ADD(45, ttl);
    }
    __syncthreads();
    // ---- write the tree and the game record back; the last search workgroup to leave releases the evaluator
    if (threadIdx.x == 0) { s_game.listWords += L.listWords; s_game.nodesVisited += s.nv; s_game.edgesScanned += s.es; }
    __syncthreads();
    if (mirror) {
        const int words = s_game.nodeCount * 4;
        const uint4* src = reinterpret_cast<const uint4*>(s_nodes);
        uint4* dst = reinterpret_cast<uint4*>(gNodes);
        for (int i = threadIdx.x; i < words; i += COLLECT_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < sizeof(Game) / 4; i += COLLECT_THREADS) reinterpret_cast<u32*>(gGame)[i] = reinterpret_cast<const u32*>(&s_game)[i];
    if (threadIdx.x == 0) {
        LegClock* ck = pl.clk;                                      // per game-iteration sums (100 MHz ticks)
        atomicAdd(&ck->sumC, tC); atomicAdd(&ck->sumW, tW); atomicAdd(&ck->sumP, tP);
        atomicAdd(&ck->cntC, nIt); atomicAdd(&ck->cntP, nIt);
        diag->published = s_expect[0] + s_expect[1];
        diag->left = (unsigned)((u64)__builtin_amdgcn_s_memrealtime() / 1000ULL) | 1u;                  // when it left (never 0)
        __hip_atomic_fetch_add(hmq::G32(&io.q->treesOut), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hmq::producer_exit(io.q);
    }
    PROF_FLUSH();
}

// k_search for a SLOW evaluator (the deployed 384-channel network: 0.70 ms per position against 0.11 ms of tree work per batch):
// one workgroup serves `perWg` games in turn instead of one, so that 64 games hold 16 CUs instead of 64 and the evaluator gets the
// rest.  A game's turn is either a collect phase (then it waits for the evaluation of the batch before — the workgroup moves on to its
// next game instead of spinning) or, once that evaluation is there, the process step (backups beside expansions) of that batch.  The
// tree is walked in place (no LDS mirror: the games of a workgroup would have to swap it); per game the order of tree operations is the
// one of k_search, hence every result.  Game record, rows published per buffer and the phase live in LDS per served game.
// w: this search workgroup, W: search workgroups of the launch (games w, w + W, ... are this workgroup's)
__device__ __forceinline__ void search_role_mg(unsigned char* smem, const Pools& pl, const Params& prm, const SearchIo& io, const int nGames, const int perWg, const int w, const int W) {
    SearchLdsMg& S_ = *reinterpret_cast<SearchLdsMg*>(smem);
    RulesTab& s_rt = S_.rt;
    WaveLds& L = S_.L;
    Game& s_game = S_.game;
    constexpr int TABN = ROLE_TABN;
    float (&s_cpuct)[TABN] = S_.cpuct;
    uint16_t (&s_pwRoot)[TABN] = S_.pwRoot;
    uint16_t (&s_pwNode)[TABN] = S_.pwNode;
    SearchCtl& s_ctl = S_.ctl;
    PubCtx& s_pub = S_.pub;
    MgSlot (&s_slot)[MG_MAX] = S_.slot;
    int& s_alive = S_.alive; int& s_moved = S_.moved; int& s_abort = S_.abort;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    stage_table_wide(&s_rt, pl.rules, COLLECT_THREADS);
    {
        const int g0 = w < nGames ? w : 0;
        const bool alt = pl.games[g0].pwSel != 0;                   // (one profile per launch: the caller checks)
        const int* pwr = alt ? pl.pwRootAlt : pl.pwRoot;
        const int* pwn = alt ? pl.pwNodeAlt : pl.pwNode;
        for (int i = threadIdx.x; i < TABN; i += COLLECT_THREADS) { s_cpuct[i] = pl.cpuctTab[i]; s_pwRoot[i] = (uint16_t)min(pwr[i], 65535); s_pwNode[i] = (uint16_t)min(pwn[i], 65535); }
    }
    if (threadIdx.x == 0) {
        for (int j = 0; j < MG_MAX; ++j) {
            const int g = w + j * W;
            s_slot[j].game = g; s_slot[j].expect[0] = s_slot[j].expect[1] = 0; s_slot[j].since = 0;
            s_slot[j].phase = (j < perWg && g < nGames && pl.games[g].status == ST_SEARCHING) ? MG_READY : MG_DONE;
        }
        s_ctl.ok = 1; s_abort = 0; L.listWords = 0;
        __hip_atomic_fetch_add(hmq::G32(&io.q->treesIn), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    u64 tC = 0, tW = 0, tP = 0, nIt = 0;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    ExpLds& myExp = L.exp;                                         // never touched (expand_leaf with `pre`); process_batch on wave 0 is its only user
    for (;;) {
        if (threadIdx.x == 0) { s_alive = 0; s_moved = 0; }
        __syncthreads();
        for (int j = 0; j < perWg; ++j) {
            const int phase = s_slot[j].phase;                      // uniform (LDS, written between barriers)
            if (phase == MG_DONE) continue;
            const int g = s_slot[j].game;
            // ---- a waiting game: is the evaluation of its pending batch there?  (one relaxed load; the workgroup does not spin on it)
            if (phase != MG_READY) {
                if (threadIdx.x == 0) {
                    s_alive++;
                    const int pending = s_slot[j].pending;
                    const unsigned have = __hip_atomic_load(hmq::G32(&io.done[g * 2 + pending]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int ready = have >= s_slot[j].expect[pending] ? 1 : 0;
                    if (!ready) {
                        const u64 now = __builtin_amdgcn_s_memrealtime();
                        if (__hip_atomic_load(hmq::G32(&io.q->error), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { s_abort = 1; }
                        else if (now - s_slot[j].since > hmq::MEET_LIMIT_TICKS && __hip_atomic_load(hmq::G32(&io.q->consIn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                            __hip_atomic_store(hmq::G32(&io.q->error), 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1;      // no evaluator workgroup has come in (wait_count's test)
                        }
                        else if (now - s_slot[j].since > hmq::SPIN_LIMIT_TICKS) { __hip_atomic_store(hmq::G32(&io.q->error), 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1; }
                    } else {
                        tW += (u64)__builtin_amdgcn_s_memrealtime() - s_slot[j].since;
                        hmq::acquire_agent();
                    }
                    s_ctl.first = ready;                            // (reused as the "ready" broadcast)
                }
                __syncthreads();
                const int ready = s_ctl.first;
                const int stop = s_abort;
                __syncthreads();                                    // (thread 0 reuses the broadcast words for the next game)
                if (stop) break;
                if (!ready) continue;
            } else if (threadIdx.x == 0) s_alive++;
            // ---- the game's turn: its record into LDS
            __syncthreads();
            G s = make_view(pl, prm, g);
            Game* const gGame = s.g;
            for (unsigned i = threadIdx.x; i < sizeof(Game) / 4; i += COLLECT_THREADS) reinterpret_cast<u32*>(&s_game)[i] = reinterpret_cast<const u32*>(gGame)[i];
            __syncthreads();
            s.g = &s_game;
            s.ldsCpuct = s_cpuct; s.ldsPwRoot = s_pwRoot; s.ldsPwNode = s_pwNode; s.tabN = TABN;
            if (wave == 0) { s.ackSeq = &L.ackSeq; s.typeSeq = &L.typeSeq; s.createSeq = &L.createSeq; s.createFast = &L.createRes.fast; s.gq = &L.gq; s.genAckSeq = &L.gq.ackSeq; }
            const int rootTeam = s_game.team;
            const bool rootAdv = s_game.adv != 0;
            const int rowBase = g * BATCH;
            const PreSorted pre{pl.sortedMoves + (size_t)g * 2 * BATCH * 2 * HM_MAX_MOVES, pl.sortedPriors + (size_t)g * 2 * BATCH * 2 * HM_MAX_MOVES};
            hmq::GameDiag* const diag = io.diag + g;
            auto mark = [&](unsigned ph) {
                __hip_atomic_store(hmq::G32(&diag->phase), ph | (xcc << 4) | ((unsigned)nIt << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(hmq::G32(&diag->stamp), (unsigned)((u64)__builtin_amdgcn_s_memrealtime() / 1000ULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(hmq::G32(&io.q->beats), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            auto mark_wait = [&](int buf) {
                __hip_atomic_store(hmq::G32(&diag->published), s_slot[j].expect[0] + s_slot[j].expect[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(hmq::G32(&diag->waitExpect), s_slot[j].expect[buf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(hmq::G32(&diag->waitBuf), (unsigned)buf + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mark(3u);
            };
            // process step of the pending batch (its evaluation is there, or it is discarded): backups on wave 0 beside the expansions
            auto process_now = [&](bool abortIt) {
                u64 t0 = 0;
                if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memrealtime(); diag->waitBuf = 0u; mark(4u); }
                const int pending = s_game.pending;
                const int nctx = s_game.ctxCount[pending];
                __syncthreads();                                    // every wave holds the batch header before wave 0 retires it
                if (wave == 0) {
                    if (abortIt) abort_batch(s, pending);
                    else backup_batch(s, pending, &io.out[pending], rowBase);
                } else if (!abortIt) expand_share(s, s_rt, myExp, wave, pending, nctx, rootTeam, rootAdv, &io.out[pending], rowBase, &pre);
                __threadfence_block();
                __syncthreads();
                if (threadIdx.x == 0) tP += __builtin_amdgcn_s_memrealtime() - t0;
            };
            bool finished = false;
            if (phase == MG_WAIT_FIN) {                             // finish_pending (agent.cc:343-352): the last batch, then the search ends
                process_now(false);
                finished = true;
            } else {
                if (phase == MG_WAIT_PROC) {                        // process_batch of the batch collected before the last one, then run_iteration's tail
                    process_now(false);
                    if (wave == 0) {
                        const int look = 1 - s_game.pending;
                        if (lane == 0) s_game.pending = -1;
                        wave_fence();
                        if (s_game.validCount[look] == 0) process_batch(s, s_rt, L.exp, look, rootTeam, rootAdv, nullptr, 0);
                        else if (lane == 0) s_game.pending = look;
                    }
                    __syncthreads();
                }
                // ---- control (worker loop, agent.cc:331-341) and, unless the search ends, the next collect phase
                if (threadIdx.x == 0) {
                    const bool fin = s_game.nodesSearched >= s_game.targetNodes || s.nodes[s_game.root].type != T_UNSOLVED || s_game.overflow;
                    const bool first = s_game.pending < 0;
                    s_ctl.action = fin ? ACT_FINISH : ACT_COLLECT;
                    s_ctl.first = first ? 1 : 0;
                    s_ctl.buf = first ? 0 : 1 - s_game.pending;
                    L.posted = 0; L.done = 0; L.servedCnt = 0; L.servedCntB = 0; L.postCount = 0; L.reqSeq = 0; L.typeSeq = 0; L.ackSeq = 0; L.createSeq = 0; L.svcStop = 0;
                    L.svcValid = 0; L.reqResult = 0; L.gq.reqSeq = 0; L.gq.ackSeq = 0;
                    for (int i = 0; i < BATCH; ++i) L.postReady[i] = 0;
                    const bool rootRow = first && s_game.alpha > 0.0f && s_game.eps > 0.0f && !(s.nodes[s_game.root].flags & F_EXPANDED);
                    const int bufNow = first ? 0 : 1 - s_game.pending;
                    s_pub.q = io.q; s_pub.expect = &s_slot[j].expect[bufNow];
                    s_pub.itemBase = hmq::item_pack(g, bufNow, 0, io.netSel ? io.netSel[g] : 0) | (rootRow ? hmq::IT_ROOT : 0u);
                }
                __syncthreads();
                if (s_ctl.action == ACT_FINISH) {
                    if (s_game.pending >= 0) {
                        const bool solvedOrOverflow = s.nodes[s_game.root].type != T_UNSOLVED || s_game.overflow;
                        if (solvedOrOverflow || s_game.validCount[s_game.pending] == 0) { process_now(solvedOrOverflow); finished = true; }   // discard / nothing to wait for
                        else if (threadIdx.x == 0) { s_slot[j].phase = MG_WAIT_FIN; s_slot[j].pending = s_game.pending; s_slot[j].since = __builtin_amdgcn_s_memrealtime(); mark_wait(s_game.pending); }
                    } else finished = true;
                } else {
                    const int buf = s_ctl.buf;
                    const bool first = s_ctl.first != 0;
                    u64 t0 = 0;
                    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memrealtime(); mark(1u); }
                    if (wave == 0) {
                        s.inflight = -1; s.reqSeq = 0; s.svcBusy = false; s.genInflight = -1; s.genReqSeq = 0; s.nv = 0; s.es = 0;
                        collect_batch(s, s_rt, L, buf, rootTeam, rootAdv);
                        if (lane == 0) __hip_atomic_store(&L.svcStop, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (lane == 0) { s_game.nodesVisited += s.nv; s_game.edgesScanned += s.es; }
                    } else collect_helper_role<true>(s, s_rt, L, pl, s_game, g, io.planes[buf], wave, &s_pub);
                    hmq::drain_stores();
                    __syncthreads();
                    if (threadIdx.x == 0) { tC += __builtin_amdgcn_s_memrealtime() - t0; nIt++; mark(2u); s_game.listWords += L.listWords; L.listWords = 0; }
                    if (first) {
                        if (wave == 0) {
                            if (s_game.ctxCount[0] == 0) { if (lane == 0) s_game.overflow |= 16; }
                            else if (s_game.validCount[0] == 0) process_batch(s, s_rt, L.exp, 0, rootTeam, rootAdv, nullptr, 0);
                            else if (lane == 0) s_game.pending = 0;
                            if (lane == 0) s_slot[j].phase = MG_READY;     // its lookahead batch follows at its next turn (the turn may have begun as a waiting one)
                        }
                        __syncthreads();
                    } else if (threadIdx.x == 0) {
                        // (a pending batch without network rows has nothing to wait for: expect == done, ready at its next turn)
                        s_slot[j].phase = MG_WAIT_PROC; s_slot[j].pending = s_game.pending; s_slot[j].since = __builtin_amdgcn_s_memrealtime(); mark_wait(s_game.pending);
                    }
                }
            }
            __syncthreads();
            if (finished && threadIdx.x == 0) {
                s_game.pending = -1;
                s_game.status = s_game.overflow ? ST_ERROR : ST_DONE;
                s_slot[j].phase = MG_DONE;
                diag->published = s_slot[j].expect[0] + s_slot[j].expect[1];
                diag->left = (unsigned)((u64)__builtin_amdgcn_s_memrealtime() / 1000ULL) | 1u;              // when it left (never 0)
            }
            if (threadIdx.x == 0) s_moved = 1;
            __syncthreads();
            for (unsigned i = threadIdx.x; i < sizeof(Game) / 4; i += COLLECT_THREADS) reinterpret_cast<u32*>(gGame)[i] = reinterpret_cast<const u32*>(&s_game)[i];
            __syncthreads();
        }
        __syncthreads();
        if (s_abort) {                                              // the evaluator is gone: every game still searching ends in error
            if (threadIdx.x == 0)
                for (int j = 0; j < perWg; ++j)
                    if (s_slot[j].phase != MG_DONE) { Game& gm = pl.games[s_slot[j].game]; gm.overflow |= 128; gm.pending = -1; gm.status = ST_ERROR; }
            break;
        }
        if (s_alive == 0) break;
        if (!s_moved) __builtin_amdgcn_s_sleep(16);                 // every game of this workgroup is waiting for the evaluator
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        LegClock* ck = pl.clk;
        atomicAdd(&ck->sumC, tC); atomicAdd(&ck->sumW, tW); atomicAdd(&ck->sumP, tP);
        atomicAdd(&ck->cntC, nIt); atomicAdd(&ck->cntP, nIt);
        __hip_atomic_fetch_add(hmq::G32(&io.q->treesOut), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hmq::producer_exit(io.q);
    }
}

// =======================================================================================
// evaluator role: the persistent evaluator (round 3's rise_serve) — every workgroup takes one position at a time from the queue the
// search workgroups fill, runs the whole network on it (narrow_position, 8 waves), then the leaf's prior pipeline on the logits still
// in LDS (PriorEpi), signals the owning game, until the queue hands it IT_POISON.
// =======================================================================================
// the network body for the launch's workgroup size: four waves (narrow_position4) or eight (narrow_position)
template <int CTILES, bool K5, int WAVES, typename... A>
__device__ __forceinline__ void position_body(A&&... args) {
    if constexpr (WAVES == 4) hmn::narrow_position4<CTILES, K5, true, hmn::PriorEpi>(static_cast<A&&>(args)...);
    else hmn::narrow_position<CTILES, K5, true, hmn::PriorEpi>(static_cast<A&&>(args)...);
}
template <int CTILES, bool K5, int WAVES>
__device__ __forceinline__ void serve_role(unsigned char* smem, unsigned* s_item, const hmn::NetDesc* __restrict__ ndp, const hmn::h16* __restrict__ wh, const float* __restrict__ wf,
                                           const int copMax, const int uHalfs, const hmq::ServeArgs& a) {
    using namespace hmn;
    const NetDesc& nd = *ndp;
    int dbgN = 0;
    unsigned long long ticks = 0, count = 0;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(hmq::G32(&a.q->consIn), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (threadIdx.x == 0) {
            const unsigned it = hmq::pop_item(a.q);
            if (it != hmq::IT_POISON) hmq::acquire_agent();         // the game's plane row: one invalidate, then plain loads
            *s_item = it;
        }
        __syncthreads();
        const unsigned it = *s_item;
        if (it == hmq::IT_POISON) break;                            // uniform
        if (a.abortAfter && threadIdx.x == 0 && __hip_atomic_load(hmq::G32(&a.q->tail), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.abortAfter)
            __hip_atomic_store(hmq::G32(&a.q->error), 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // test hook: the search is abandoned mid-way (this row is still evaluated)
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const int g = hmq::item_game(it), buf = hmq::item_buf(it), row = hmq::item_row(it);
        const size_t r = (size_t)g * 8 + row;
        PriorEpi epi{&a, (size_t)(g * 2 + buf) * 8 + row, g, (it & hmq::IT_ROOT) != 0};
        position_body<CTILES, K5, WAVES>(nd, wh, wf, reinterpret_cast<const h16*>(a.planes[buf]) + r * HM_PLANE_VALUES, r, copMax, uHalfs, smem,
                                                    reinterpret_cast<h16*>(a.value[buf]), reinterpret_cast<h16*>(a.piA[buf]), reinterpret_cast<h16*>(a.piB[buf]),
                                                    reinterpret_cast<h16*>(a.wdl[buf]), reinterpret_cast<h16*>(a.ml[buf]), nullptr, dbgN, epi);
        hmq::drain_stores();                                        // every wave's write-through stores of the heads have left
        __syncthreads();                                            // (also: every thread has read s_item)
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(hmq::G32(&a.done[g * 2 + buf]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticks += __builtin_amdgcn_s_memrealtime() - t0; ++count;
        }
    }
    if (threadIdx.x == 0) {
        if (a.clkSum && count) { atomicAdd(a.clkSum, ticks); atomicAdd(a.clkCnt, count); }
        __hip_atomic_fetch_add(hmq::G32(&a.q->served), (unsigned)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(hmq::G32(&a.q->consOut), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ONE launch: workgroups [0, searchWgs) search, the others evaluate.  Both roles take a whole CU, so the grid is resident at once
// when it has no more workgroups than the device has CUs.  WAVES = waves per workgroup:
//   4 (256 threads; the narrow trunks): one wave per SIMD with the lane's whole register budget (256 VGPRs + 256 AGPRs) — what the
//     traversal wave of the search role wants; the evaluator role is narrow_position4 (two accumulator tiles per weight fragment);
//   8 (512 threads; the 384-channel trunk, whose searches are bound by the evaluator): the evaluator role is the faster eight-wave
//     form (narrow_position); a search workgroup keeps its first four waves and ends the others at once — with a wave-uniform test
//     the compiler can see as such and a real s_endpgm: a plain `return` on `threadIdx.x >= 256` is a DIVERGENT exit to the compiler,
//     which keeps those waves running through the whole role with an empty exec mask.  The search role then has 256 registers per
//     lane (spills to scratch: a collect phase takes 0.145 ms instead of 0.108), which does not matter where games wait for evaluations.
template <int MODE, int CTILES, bool K5, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_rollout(Pools pl, Params prm, SearchIo io, RolloutNet rn, hmq::ServeArgs a, int nGames, int perWg, int searchWgs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x < searchWgs) {
        if constexpr (64 * WAVES > COLLECT_THREADS) {
            if (__builtin_amdgcn_readfirstlane((int)threadIdx.x) >= COLLECT_THREADS) __builtin_amdgcn_endpgm();
        }
        if constexpr (MODE == 2) search_role_mg(smem, pl, prm, io, nGames, perWg, (int)blockIdx.x, searchWgs);
        else search_role<MODE == 0>(smem, pl, prm, io, (int)blockIdx.x);
    } else {
        serve_role<CTILES, K5, WAVES>(smem, reinterpret_cast<unsigned*>(smem + rn.itemOff), static_cast<const hmn::NetDesc*>(rn.nd), static_cast<const hmn::h16*>(rn.wh),
                                      static_cast<const float*>(rn.wf), rn.copMax, rn.uHalfs, a);
    }
}

}  // namespace hms

// ---- host side: instantiation table of this translation unit's mode ---------------------------------------------------------
#ifndef HM_ROLLOUT_MODE
#error "compile with -DHM_ROLLOUT_MODE=0|1|2 (or include from hm_search.hip with HM_SINGLE_TU, which defines the three entry points itself)"
#endif
namespace {
// -> the k_rollout instantiation for a trunk of `ctiles` * 32 channels and its waves per workgroup (nullptr: none)
template <int MODE>
const void* rollout_kernel(int ctiles, int k5, int narrow8, int* waves) {
    using namespace hms;
    // (4, false) is the bench's RISEv3-small (no 5x5 block: the 5x5 depthwise code and its registers are left out); every other
    // narrow trunk runs the K5 = true instantiation of its width, which also handles 3x3 blocks.  The 384-channel trunk gets
    // eight-wave workgroups (HM_ROLLOUT_WAVES_WIDE=4 selects the four-wave kernel for measurements).
    *waves = 4;
    if (ctiles == 4 && !k5) {
        // narrow8: the caller asks for the eight-wave kernel (hm_sp_search: many games alive — the search is bound by the evaluator then)
        if (narrow8) { *waves = 8; return reinterpret_cast<const void*>(k_rollout<MODE, 4, false, 8>); }
        return reinterpret_cast<const void*>(k_rollout<MODE, 4, false, 4>);
    }
#ifdef HM_ROLLOUT_FEW
    return nullptr;                                            // diagnostic builds: RISEv3-small only (one translation unit holds all three search roles)
#else
    if (ctiles == 4) return reinterpret_cast<const void*>(k_rollout<MODE, 4, true, 4>);
    if (ctiles == 2) return reinterpret_cast<const void*>(k_rollout<MODE, 2, true, 4>);
    if (ctiles == 12) {
        const char* e = std::getenv("HM_ROLLOUT_WAVES_WIDE");
        if (e && std::atoi(e) == 4) return reinterpret_cast<const void*>(k_rollout<MODE, 12, true, 4>);
        *waves = 8;
        return reinterpret_cast<const void*>(k_rollout<MODE, 12, true, 8>);
    }
    return nullptr;
#endif
}
}  // namespace
#define HM_ROLLOUT_ENTRY_(m) hm_rollout_kernel_mode##m
#define HM_ROLLOUT_ENTRY(m) HM_ROLLOUT_ENTRY_(m)
extern "C" {
#ifdef HM_SINGLE_TU
const void* hm_rollout_kernel_mode0(int ctiles, int k5, int narrow8, int* waves) { return rollout_kernel<0>(ctiles, k5, narrow8, waves); }
const void* hm_rollout_kernel_mode1(int ctiles, int k5, int narrow8, int* waves) { return rollout_kernel<1>(ctiles, k5, narrow8, waves); }
const void* hm_rollout_kernel_mode2(int ctiles, int k5, int narrow8, int* waves) { return rollout_kernel<2>(ctiles, k5, narrow8, waves); }
#else
const void* HM_ROLLOUT_ENTRY(HM_ROLLOUT_MODE)(int ctiles, int k5, int narrow8, int* waves) { return rollout_kernel<HM_ROLLOUT_MODE>(ctiles, k5, narrow8, waves); }
#endif
}
