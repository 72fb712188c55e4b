// hm_search.hip — GPU-resident joint-action MCGS for many concurrent Bughouse games.
//
// Replaces, behind the C ABI of include/hivemind_amd.h (hm_sp_*), the reference's
//   Agent::run_search node-budget path      search/agent.cc:421-558, 331-352
//   SearchThread::{run_iteration, collect_batch, process_batch, select_and_expand,
//                  canonicalize_child, backup} search/searchthread.cc:197-916
//   Node (PUCT, PW gate, solver, reservations) search/node.h, node.cc:6-119
//   JointCandidateGenerator                  environment/joint_action.h:126-359
//   TranspositionTable (per search)          search/transposition_table.h:83-103
//
// MI355X-first design (not a port of the threaded CPU tree):
//   * one workgroup owns one game: its tree lives in HBM pools private to the game (node pool, a bump arena for per-node
//     edge arrays / candidate frontiers / cached joint positions, an open-addressing TT), so there are no locks and no
//     shared_ptr: the reference's mutex / CAS protocols (virtual loss, evaluation reservations) become plain fields
//     updated in program order by the one wave that walks the tree.
//   * all games advance in lockstep: collect (select+expand, virtual loss, terminal classification, leaf planes, leaf move
//     lists) -> one batched network call for every game's leaves -> process (masked softmax, sort, frontier seed, value
//     shaping, backup, solver).  The reference's double-buffered lookahead of one SearchThread (B=8) is reproduced per
//     game, so per-game results equal the single-thread reference schedule.
//   * k_collect is a four-wave pipeline per game (one wave per SIMD, the whole register file each; the node pool mirrored
//     in LDS for the launch): the traversal wave only selects; a classifier wave runs the terminal test and writes the
//     context record; a plane-writer wave encodes the leaf planes; a generator wave refills candidate frontiers and
//     generates move lists.  Guards (svc_wait / gen_wait) keep the sequential semantics: the traversal never reads state
//     a helper still owns, helpers serve their requests in order.
//   * k_process gives every leaf of the batch its own wave (expansion) beside one wave doing the ordered backups.
//   * PUCT child selection is lane-parallel (one edge per lane, wave arg-max by DPP row reductions with lowest-index tie
//     break); repetition scans, policy gathers, the prior sort (rank sort) and the plane writer are lane-parallel too.
//   * libm-sensitive pieces are pinned: exp is a fixed IEEE sequence (hm_expf), cpuct(N) and the progressive-widening
//     schedule come from host-built tables (std::log / std::pow, the reference's own expressions), Dirichlet gamma draws
//     are made on the host with std::gamma_distribution<float> on std::mt19937_64 exactly as node.h:286-315.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static __device__ unsigned long long g_trace[65536];   // hm_prof.hpp TRACE_EV log (diagnostic builds only)
static __device__ unsigned int g_traceCount, g_traceSeq;
static __device__ int g_traceGame = -1;
#include "hm_prof.hpp"
#include "hm_rules_device.hpp"
#include "hm_host.hpp"
#include "hm_queue.hpp"
#include "hm_policy.hpp"

using namespace hmd;

// provided by hm_kernels.hip
const HostTables& hm_host_tables();
int hm_fail(int code, const std::string& msg);

namespace hms {

constexpr int BATCH = 8;                 // SearchParams::BATCH_SIZE
constexpr int MAX_TRAJ = 96;             // search path cap (root .. leaf)
constexpr int SEARCH_HIST_LDS = 1024 + MAX_TRAJ + 8;   // k_search<true>: history keys per board it keeps in LDS (Params::histCap of the default game length)
constexpr int HIST_GAME_MIN = 1024;      // game history keys per board (grown to the run's macro-ply limit, hm_sp_create_ex)
constexpr int NOISE_CAP = hmp::NOISE_CAP; // > max actions per board (304 + pass)
constexpr int MIN_VISITS_TAB = 1 << 15;  // cpuct / PW tables: at least this long; sized from the node budget (Params::tabLen)
constexpr int NLISTS = 8;                // LDS scratch move lists per wave
constexpr int COLLECT_THREADS = 256;     // k_collect: traversal wave, classifier wave, plane-writer wave, generator wave
constexpr float Q_INIT = -1.0f;
constexpr int LDS_DIRTY_BITS = 2048;      // k_collect's LDS node mirror: dirty bits, hence the largest mirrored pool
static_assert(LDS_DIRTY_BITS / 32 <= COLLECT_THREADS, "s_dirty is cleared by one pass of the block");

enum : uint8_t { T_UNSOLVED = 0, T_WIN = 1, T_LOSS = 2, T_DRAW = 3 };
enum : uint8_t { F_PENDING = 1, F_EXPANDED = 2 };
enum : int { ST_IDLE = 0, ST_SEARCHING = 1, ST_FINISHING = 2, ST_DONE = 3, ST_NOACTION = 4, ST_ERROR = 5 };

struct Edge {            // 40 B
    int child;
    float q, vsum, prior;
    int visits, vloss;
    u32 moveA, moveB;
    uint16_t idxA, idxB;
    uint8_t ctype, has_ctype, pad0, pad1;
};
struct Node {            // 64 B
    u64 hash;
    float valueSum;
    int visits, vvsum, expanded, endInPly, unsolved, cntTypes;
    u32 edges, edgeCap, gen;           // arena offsets (8-byte units), 0 = none
    uint16_t depth;
    uint8_t team, flags, type, more;   // more: the candidate generator still holds joint actions (hasNext), kept beside the node
    u32 posOff;                        // arena offset of this node's NodePos (0 = position not computed yet)
};
// Joint position of a node, cached when the node is first reached: a descent walks node ids only and loads the leaf's
// position instead of replaying the moves of its path (Board::make_moves per level in the reference, searchthread.cc:863-897).
// Nodes are identified by Board::hash_key, which covers both positions and both per-board key sequences, so a node reached
// along another path (transposition) has the same record.  repKey = repetition key of each board's position (the key a
// push on that board appends to its history), hlen / prefix = per-board history length and chained prefix hash at this node.
struct NodePos {         // 232 B
    hm_pos pos[2];
    u64 repKey[2];
    u64 prefix[2];
    int hlen[2];
};
struct HeapEnt { float prio; uint16_t iA, iB; };
struct GenHdr {          // candidate generator state in the arena
    int nA, nB;
    u32 movesA, movesB, priorsA, priorsB;       // arena offsets; move bit 31 = capture flag
    u32 heap, heapSize, heapCap;
    u32 visited, visSize, visCap;
    uint8_t aOn, bOn, adv, aCan, bCan, pad[3];
};
struct TrajEnt { int node, childIdx; u32 moveA, moveB; };
struct Ctx {
    int leaf, trajLen;
    uint8_t team, sit, terminal, reserved;
    float termValue;
    u64 leafHash;
};
struct Game {
    // persistent game state (Board + selfplay bookkeeping)
    hm_pos pos[2];
    u32 lastMove[2];
    int hlen[2];
    u64 prefix[2];
    int team, adv;
    // search state
    int status, root, nodeCount, nodesSearched, targetNodes, pending;
    unsigned int arenaTop;
    int ctxCount[2], validCount[2];
    float alpha, eps;
    u64 noiseSeed, rootHash;
    int sameBatchCollisions, reservationCollisions, evalRows, overflow, maxDepth, ttCount;
    int ttHits;                         // lookups that found their position in the table (transposition_table.h:83-103), this search
    int nodesVisited, edgesScanned;     // traversal traffic counters (roofline accounting)
    int fresh;                          // batch `pending` was collected this iteration: its planes are in NEXT, not yet evaluated
    int listWords;                      // leaf move-list words written by the helper wave (traffic accounting)
    int live;                           // slot holds a game (set by k_set_games); dead slots are skipped by every kernel
    int pwSel;                          // progressive-widening profile of the side searching now: 0 = the engine's tables, 1 = the alternate pair (hm_sp_set_pw_profiles)    // tree reuse between searches (Agent::try_reuse_tree, agent.cc:1345-1371)
    int lastRootP1;                     // 1 + root of this slot's previous search, whose tree is still in the pool (0 = none: reset_search_state)
    int reuseMode;                      // 0 = every search starts from an empty pool; 1 = reuse when the node budget still fits; 2 = and shrink the budget to what fits
    int reusedVisits;                   // visits of the recovered root, -1 = fresh root
    int batch;                          // leaves collected per iteration (Engine::getBatchSize(), searchthread.cc:258-273, 663): 1 .. BATCH, 0 = BATCH (hm_sp_set_batch_sizes)
};

struct Params {          // device-visible configuration + pool geometry
    int nGames, nodeCap, ttCap;        // ttCap power of two
    int tabLen;                        // entries of the cpuct / progressive-widening tables (visit counts beyond clamp to the last one)
    int histGame, histCap;             // per-board history keys: game part / game + search path
    int ldsNodes;                      // k_collect keeps the game's node pool in LDS (nodeCap * 64 B fits beside its other LDS)
    u32 arenaCap;                      // 8-byte units
    float cpuctInit, cpuctBase, fpuReduction, drawContempt, wdlWeight, mlDiscount;
    int enableTranspositions, enableDynamicFpu, enableWdl;
    float qVetoDelta, qValueWeight;      // SearchParams::Q_VETO_DELTA / Q_VALUE_WEIGHT (search_params.h)
};

// Device-side clock of the three legs of a lockstep iteration (constant 100 MHz counter, s_memrealtime): every workgroup of a
// kernel min-/max-es its start / end into the leg's interval, the next kernel in stream order folds the finished interval into a
// running sum.  Gives the exact average launch duration of k_collect / the forward / k_process over ALL launches (graph-replayed
// ones included), the figure a rocprofv3 kernel trace reports.
struct LegClock { u64 cStart, cEnd, nStart, nEnd, pStart, pEnd; u64 sumC, sumN, sumP; u64 cntC, cntN, cntP; u64 sumW; };
// Persistent search (k_search / rise_serve): no launches to bracket, so the same sums count per game-iteration — sumC / sumP the
// ticks a game spent in its collect / process phases (cntC = cntP = game-iterations), sumW the ticks it waited for the evaluation
// of its pending batch, sumN / cntN the evaluator's ticks and positions.

struct Pools {
    LegClock* clk;
    Game* games;
    Node* nodes;          // [nGames][nodeCap]
    u64* arena;           // [nGames][arenaCap]
    u64* ttKeys;          // [nGames][ttCap]
    int* ttVals;
    Ctx* ctx;             // [nGames][2][BATCH]
    TrajEnt* traj;        // [nGames][2][BATCH][MAX_TRAJ]
    u64* hist;            // [nGames][2][histCap]
    float* noise;         // [nGames][2][NOISE_CAP]
    u32* leafMoves;       // [nGames][2 batches][BATCH rows][2 boards][HM_MAX_MOVES]: filtered legal lists of the network leaves
    int* leafCounts;      // [nGames][2][BATCH][2]: moves kept | side to move << 16
    u32* sortedMoves;     // [nGames][2][BATCH][2][HM_MAX_MOVES]: persistent evaluator -> tree, moves (+ capture bit) in prior order
    float* sortedPriors;  // same shape: their priors
    const float* cpuctTab;   // [Params::tabLen]
    const int* pwRoot;       // [Params::tabLen]
    const int* pwNode;
    const int* pwRootAlt;    // second schedule (tournaments give each network its own PW coefficient, tournament.h:30-41)
    const int* pwNodeAlt;
    const RulesTab* rules;
    const int* polNormal;    // [2][64][64][2]
    const int* polDrop;      // [2][64][8]
};

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
static __device__ unsigned int* g_hb;                    // diagnostics of a persistent-search give-up: last heartbeat code per (game slot, wave); nullptr outside k_search
#ifdef HM_SEARCH_HB
#define HB(c) do { unsigned int* hb_ = g_hb; if ((threadIdx.x & 63) == 0 && hb_) __hip_atomic_store(hmq::G32(&hb_[blockIdx.x * 4 + (threadIdx.x >> 6)]), (unsigned)(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define HB(c) do { } while (0)
#endif
static __device__ unsigned long long g_prof[128];     // hm_prof.hpp probes: 64 sums + 64 call counts (all zero in the product build)
#ifdef HM_SEARCH_PROF
constexpr int PROF_LAUNCHES = 8192;
static __device__ unsigned int g_colDur[PROF_LAUNCHES][64];   // traversal cycles of wave 0 per (k_collect launch, game slot < 64): straggler analysis
static __device__ unsigned int g_colLaunch;                    // launch counter, bumped by k_process
#endif
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Orderings cheaper than wave_fence (whose workgroup-scope release drains every outstanding global store: ~1-2 k cycles):
// wave_sync — the lanes of THIS wave see each other's earlier stores (wavefront scope: program order, no wait);
// lds_release — this wave's LDS writes are done before the LDS flag that follows (another wave then reads LDS only).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void lds_release() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

using hmp::hm_expf; using hmp::h2f; using hmp::finite_f; using hmp::clampf; using hmp::board_priors_sorted;   // hm_policy.hpp (shared with the persistent evaluator)

__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ULL; }
__device__ __forceinline__ bool wave_all(bool p) { return __ballot(!p) == 0ULL; }
__device__ __forceinline__ int wave_sum_i(int v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return ufirst(v);
}

struct GenQ { int node; u32 genOff; int iA, iB; int reqSeq, ackSeq; };   // traversal wave -> generator wave (LDS)
struct G {               // per-wave view of one game's pools
    Game* g;
    Node* nodes;
    u64* arena;
    u64* ttKeys;
    int* ttVals;
    Ctx* ctx;
    TrajEnt* traj;
    u64* hist[2];
    float* noise[2];
    u32* leafMoves;          // this game's [2][BATCH][2][HM_MAX_MOVES]
    int* leafCounts;         // [2][BATCH][2]
    const Params* prm;
    const Pools* pl;
    // first tabN entries of the cpuct / progressive-widening tables staged in LDS by the traversal kernel (0: none)
    const float* ldsCpuct; const uint16_t* ldsPwRoot; const uint16_t* ldsPwNode; int tabN;
    // k_collect's traversal wave hands every unsolved leaf to the block's classifier wave and goes on selecting; `inflight`
    // is the leaf whose classification has not been acknowledged yet (-1: none).  Its Node must not be read before svc_wait.
    int inflight, reqSeq;
    bool svcBusy;                    // the classifier has not finished the last request (it may still be writing the context / board image)
    const int* ackSeq;               // requests completely finished
    const int* typeSeq;              // requests whose terminal test is done (leaf Node final): all the tree guards need
    // same for the generator wave: the node whose candidate generator is being refilled after a pop (its GenHdr and Node::more
    // belong to that wave until gen_wait)
    int genInflight, genReqSeq;
    const int* genAckSeq;
    GenQ* gq;
    int nv, es;                      // nodes visited / edges scanned by this launch's descents (flushed to Game once)
    // A child reached for the first time gets its position, hash and transposition lookup from the classifier wave (request with
    // `create` set) while the traversal is already selecting again.  `jPending`: that outcome is still unknown; resolve_create()
    // waits for it at every step the traversal could not take back.
    bool jPending, jTakenBack;
    bool ldsTree;                    // the node pool is the LDS mirror: what the traversal reads of a leaf the classifier wave has finished is LDS only
    const int* createSeq;            // requests whose creation step is done
    const int* createFast;           // 1: the child stayed what the traversal assumed (a fresh leaf), 0: the traversal takes the descent back
    u32* dirty;                      // k_collect with the LDS node mirror: one bit per node the launch may have modified (nullptr otherwise)
};
// every node that enters a search path (and every first-reached child before canonicalisation) is marked: those, plus the
// nodes allocated by this launch, are the only ones the launch can have modified, and the only ones written back to HBM
__device__ __forceinline__ void mark_dirty(const G& s, int id) { if (s.dirty) s.dirty[id >> 5] |= 1u << (id & 31); }
// blocks until the classifier wave has settled the outstanding leaf's terminal test (its writes to the leaf's Node are visible after)
__device__ __forceinline__ void svc_wait(G& s) {
    if (s.inflight < 0) return;
    PROF_T(tsw);
    HB(11);
    while (__hip_atomic_load(s.typeSeq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != s.reqSeq) __builtin_amdgcn_s_sleep(1);
    s.inflight = -1;
    HB(16);
    PROF_ADD(19, tsw);
}
// blocks until the classifier wave has finished the outstanding request entirely (context record, board image, counters)
// (behind = 1: all but the latest request)
__device__ __forceinline__ void svc_join(G& s, int behind = 0) {
    if (!s.svcBusy) return;
    PROF_T(tsw);
    HB(12);
    while (__hip_atomic_load(s.ackSeq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < s.reqSeq - behind) __builtin_amdgcn_s_sleep(1);
    if (behind == 0) { s.inflight = -1; s.svcBusy = false; }
    HB(17);
    PROF_ADD(22, tsw);
}
__device__ __forceinline__ void gen_wait(G& s) {
    if (s.genInflight < 0) return;
    PROF_T(tgw);
    HB(13);
    while (__hip_atomic_load(s.genAckSeq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != s.genReqSeq) __builtin_amdgcn_s_sleep(1);
    s.genInflight = -1;
    HB(18);
    PROF_ADD(21, tgw);
}
// The creation step of the outstanding request: true when the child is the fresh leaf the traversal took it for (or nothing is
// outstanding); false when the classifier handed the descent back (a transposition, a pending evaluation, an exhausted pool): the
// caller unwinds to collect_batch, which undoes what it did on the assumption and continues that descent itself.
__device__ __forceinline__ bool resolve_create(G& s) {
    if (!s.jPending) return true;
    PROF_T(trc);
    while (__hip_atomic_load(s.createSeq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != s.reqSeq) __builtin_amdgcn_s_sleep(1);
    s.jPending = false;
    s.jTakenBack = *s.createFast == 0;
    PROF_ADD(46, trc);
    return !s.jTakenBack;
}
__device__ __forceinline__ float cpuct_of(const G& s, int v) { return v < s.tabN ? s.ldsCpuct[v] : s.pl->cpuctTab[v]; }
__device__ __forceinline__ int pw_root_of(const G& s, int v) { return v < s.tabN ? (int)s.ldsPwRoot[v] : (s.g->pwSel ? s.pl->pwRootAlt : s.pl->pwRoot)[v]; }
__device__ __forceinline__ int pw_node_of(const G& s, int v) { return v < s.tabN ? (int)s.ldsPwNode[v] : (s.g->pwSel ? s.pl->pwNodeAlt : s.pl->pwNode)[v]; }
__device__ __forceinline__ Edge* edges_of(const G& s, const Node& n) { return reinterpret_cast<Edge*>(s.arena + n.edges); }
__device__ __forceinline__ GenHdr* gen_of(const G& s, const Node& n) { return reinterpret_cast<GenHdr*>(s.arena + n.gen); }

// bump allocation in 8-byte units; offset 0 is reserved as "null"
// (lane 0 bumps the counter atomically and broadcasts: several waves of a block may allocate at once)
__device__ inline u32 arena_alloc(G& s, u32 bytes) {
    const u32 units = (bytes + 7) >> 3;
    u32 top = 0;
    if ((threadIdx.x & 63) == 0) top = atomicAdd(&s.g->arenaTop, units);
    top = (u32)ulane((int)top, 0);
    if (top + units > s.prm->arenaCap) { if ((threadIdx.x & 63) == 0) atomicOr(&s.g->overflow, 1); return 0; }
    return top;
}
__device__ inline int node_alloc(G& s, int team, int depth) {
    int id = 0;
    if ((threadIdx.x & 63) == 0) id = atomicAdd(&s.g->nodeCount, 1);
    id = ulane(id, 0);
    if (id >= s.prm->nodeCap) { if ((threadIdx.x & 63) == 0) atomicOr(&s.g->overflow, 2); return -1; }
    Node n;
    n.hash = 0; n.valueSum = 0.0f; n.visits = 0; n.vvsum = 0; n.expanded = 0; n.endInPly = 0; n.unsolved = 0; n.cntTypes = 0;
    n.edges = 0; n.edgeCap = 0; n.gen = 0; n.depth = (uint16_t)depth; n.team = (uint8_t)team; n.flags = 0; n.type = T_UNSOLVED; n.more = 0; n.posOff = 0;
    s.nodes[id] = n;
    return id;
}

// ---- transposition table: insertOrGet (transposition_table.h:83-103) ----------------------
__device__ inline int tt_insert_or_get(G& s, u64 hash, int node) {
    const int cap = s.prm->ttCap;
    u32 i = (u32)(hash ^ (hash >> 32)) & (cap - 1);
    for (int probe = 0; probe < cap; ++probe) {
        const int v = s.ttVals[i];
        if (v < 0) {
            if (s.g->ttCount * 2 >= cap) return node;        // table full: behave like a rejected insert
            s.ttKeys[i] = hash; s.ttVals[i] = node; s.g->ttCount++;
            return node;
        }
        if (s.ttKeys[i] == hash) { s.g->ttHits++; return v; }
        i = (i + 1) & (cap - 1);
    }
    return node;
}

// ---- candidate generator (joint_action.h:126-359), strict total order --------------------
__device__ inline float joint_prior(const G& s, const GenHdr& h, int iA, int iB, u32* mA, u32* mB) {   // JointActionCandidate ctor :80-105
    const u32 a = reinterpret_cast<const u32*>(s.arena + h.movesA)[iA], b = reinterpret_cast<const u32*>(s.arena + h.movesB)[iB];
    const float pA = reinterpret_cast<const float*>(s.arena + h.priorsA)[iA], pB = reinterpret_cast<const float*>(s.arena + h.priorsB)[iB];
    const u32 ma = a & 0x7fffffffu, mb = b & 0x7fffffffu;
    const bool capA = a >> 31, capB = b >> 31;
    const bool sitsA = ma == 0, sitsB = mb == 0;
    bool invalid = false;
    if (sitsA && sitsB) invalid = !is_double_sit_legal(h.adv, h.aOn, h.bOn);
    else if (sitsA && h.aCan) invalid = !is_single_pass_legal(h.adv, h.aOn, h.bOn, capB);
    else if (sitsB && h.bCan) invalid = !is_single_pass_legal(h.adv, h.aOn, h.bOn, capA);
    if (mA) *mA = ma;
    if (mB) *mB = mb;
    return invalid ? -1.0f : pA * pB;
}
__device__ inline void gen_grow(G& s, u32& off, u32& cap, u32 size, u32 elemBytes) {
    const u32 ncap = cap * 2;
    const u32 noff = arena_alloc(s, ncap * elemBytes);
    if (!noff) return;
    const u32 words = (size * elemBytes + 7) >> 3;
    for (u32 i = 0; i < words; ++i) s.arena[noff + i] = s.arena[off + i];
    off = noff; cap = ncap;
}
// pushCandidate (:146-177).  The reference recurses through invalid pairs (sit-rule violations) to their successors; what a
// push leaves behind is order-free: `visited` is a set and the frontier is popped by a strict total order, so only the
// closure matters.  It is computed breadth-first with the visited list itself as the work queue — no recursion stack, so
// no depth limit (a column of quiet-move x pass pairs is as long as the board's move list).
__device__ inline bool gen_visit(G& s, GenHdr& h, int a, int b) {
    if (a >= h.nA || b >= h.nB) return true;
    const u32 key = ((u32)a << 16) | (u32)b;
    u32* vis = reinterpret_cast<u32*>(s.arena + h.visited);
    bool seen = false;
    for (u32 i = threadIdx.x & 63; i < h.visSize; i += 64) seen |= vis[i] == key;   // lane-parallel membership test
    if (wave_any(seen)) return true;
    if (h.visSize >= h.visCap) { gen_grow(s, h.visited, h.visCap, h.visSize, 4); vis = reinterpret_cast<u32*>(s.arena + h.visited); }
    if (h.visSize >= h.visCap) return false;              // arena exhausted (overflow flagged by arena_alloc)
    vis[h.visSize++] = key;                               // every lane stores the same word: each lane later re-reads only its own stores
    return true;
}
__device__ inline void gen_push(G& s, GenHdr& h, int iA, int iB) {
    u32 q = h.visSize;
    if (!gen_visit(s, h, iA, iB)) return;
    for (; q < h.visSize; ++q) {
        const u32 key = reinterpret_cast<const u32*>(s.arena + h.visited)[q];
        const int a = key >> 16, b = key & 0xffff;
        const float jp = joint_prior(s, h, a, b, nullptr, nullptr);
        if (jp >= 0.0f) {
            if (h.heapSize >= h.heapCap) gen_grow(s, h.heap, h.heapCap, h.heapSize, 8);
            if (h.heapSize >= h.heapCap) return;
            // the frontier is an unordered array: the pop below takes the arg-best of the strict total order, which
            // is exactly the element a binary heap with that comparator would pop
            reinterpret_cast<HeapEnt*>(s.arena + h.heap)[h.heapSize++] = HeapEnt{jp, (uint16_t)a, (uint16_t)b};
        } else {
            if (!gen_visit(s, h, a + 1, b) || !gen_visit(s, h, a, b + 1)) return;
        }
    }
}
// refill = false: the caller hands the two successor pushes of the popped pair to another wave (k_collect's generator wave)
__device__ inline bool gen_next(G& s, GenHdr& h, HeapEnt* out, bool refill = true) {   // getNext :312-328
    if (h.heapSize == 0) return false;
    const int lane = threadIdx.x & 63;
    HeapEnt* hp = reinterpret_cast<HeapEnt*>(s.arena + h.heap);
    const u32 n = h.heapSize;
    // lane-parallel arg-best over the frontier: (prior desc, iA asc, iB asc) is a strict total order; priors are >= +0
    u64 bestKey = 0;
    int bi = -1;
    for (u32 base = 0; base < n; base += 64) {
        const u32 i = base + lane;
        u64 key = 0;
        if (i < n) { const HeapEnt e = hp[i]; key = ((u64)float_order_bits(e.prio) << 32) | (u32)~(((u32)e.iA << 16) | (u32)e.iB); }
        const u64 top = wave_max_u64(key);
        if (top > bestKey) { bestKey = top; bi = (int)base + (int)__builtin_ctzll(__ballot(key == top)); }   // pairs are unique
    }
    if (bi < 0) { if (lane == 0) atomicOr(&s.g->overflow, 64); return false; }   // cannot happen (n > 0 and every key is non-zero)
    const float bp = float_from_order_bits((u32)(bestKey >> 32));
    const u32 bk = ~(u32)bestKey;
    const HeapEnt best{bp, (uint16_t)(bk >> 16), (uint16_t)(bk & 0xffffu)};
    h.heapSize = n - 1;
    if ((u32)bi != n - 1) hp[bi] = hp[n - 1];
    wave_fence();
    if (refill) {
        gen_push(s, h, best.iA + 1, best.iB);
        gen_push(s, h, best.iA, best.iB + 1);
    }
    *out = best;
    return true;
}

// Node::flags lives in the dword at byte 48 (depth:16, team:8, flags:8).  k_process expands leaves (sets
// F_EXPANDED) and retires reservations (clears F_PENDING) from different waves at once: word atomics.
__device__ __forceinline__ unsigned int* flags_word(Node* n) { return reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(n) + 48); }
static_assert(offsetof(Node, flags) == 51, "Node::flags moved: fix flags_word");
__device__ __forceinline__ void node_set_flag(Node* n, unsigned int f) { atomicOr(flags_word(n), f << 24); }
__device__ __forceinline__ void node_clear_flag(Node* n, unsigned int f) { atomicAnd(flags_word(n), ~(f << 24)); }

// ---- edges -------------------------------------------------------------------------------
__device__ inline Edge* edge_append(G& s, Node& n) {
    if (n.edgeCap == 0) {
        n.edges = arena_alloc(s, 4 * sizeof(Edge)); n.edgeCap = n.edges ? 4 : 0;
    } else if ((u32)n.expanded >= n.edgeCap) {
        gen_grow(s, n.edges, n.edgeCap, (u32)n.expanded, sizeof(Edge));
    }
    if ((u32)n.expanded >= n.edgeCap) return nullptr;
    return edges_of(s, n) + n.expanded;
}

// One lane-parallel pass over a node's edges (lane i holds edge i, then i+64, ...) feeds both the widening test
// (node.h:151-175) and PUCT selection (node.cc:6-119): per edge the child's solver type is gathered once, the
// wave reductions give "all children lose", "some child unvisited" and the visited prior mass, and the first 64
// edges stay in registers for the arg-max.
struct EdgeScan {
    Edge ed;                 // this lane's edge of chunk 0 (valid when lane < expanded)
    int ct;                  // its child's node type
    bool anyNonLosing, anyUnvisited;
    float visitedPolicySum;  // priors of edges with visits + virtual losses > 0, added in index order
    int visits;              // parent visits incl. virtual visits
};
__device__ inline bool scan_edges(G& s, const Node& n, EdgeScan& o) {   // false: the outstanding creation was handed back (resolve_create)
    const int lane = threadIdx.x & 63;
    const int limit = n.expanded;
    const Edge* e = edges_of(s, n);
    o.visits = n.visits + n.vvsum;
    o.anyNonLosing = false; o.anyUnvisited = false; o.visitedPolicySum = 0.0f;
    o.ct = T_UNSOLVED;
    const bool dynFpu = s.prm->enableDynamicFpu && o.visits > 0;
    for (int base = 0; base < limit; base += 64) {
        const int i = base + lane;
        bool nonLosing = false, unvisited = false;
        Edge ed;
        ed.child = -2;
        if (i < limit) ed = e[i];
        if (s.inflight >= 0 && wave_any(ed.child == s.inflight)) {              // a child whose creation / terminal test is still running
            svc_wait(s);
            if (!resolve_create(s)) return false;
        }
        if (i < limit) {
            const int ct = s.nodes[ed.child].type;
            if (base == 0) { o.ed = ed; o.ct = ct; }
            unvisited = ed.visits + ed.vloss <= 0;
            nonLosing = ct != T_WIN;
        }
        o.anyNonLosing |= wave_any(nonLosing);
        o.anyUnvisited |= wave_any(unvisited);
    }
    // The prior mass of the visited edges feeds only the first-play value of UNVISITED edges (node.cc:33-41): when every edge has been
    // touched the selection never reads it, and the serial sum — ascending index order, as the reference adds — is skipped.
    if (dynFpu && o.anyUnvisited) {
        for (int base = 0; base < limit; base += 64) {
            const int i = base + lane;
            float pr = 0.0f;
            bool counted = false;
            if (i < limit) {
                Edge ed = o.ed;
                if (base != 0) ed = e[i];
                pr = ed.prior;
                counted = ed.visits + ed.vloss > 0;
            }
            for (u64 m = __ballot(counted); m; m &= m - 1) o.visitedPolicySum += ulanef(pr, __builtin_ctzll(m));   // ascending index order
        }
    }
    return true;
}
__device__ inline bool should_expand_new_child(G& s, const Node& n, const EdgeScan& sc) {
    const bool hasNext = n.more != 0;
    const bool allLose = n.expanded > 0 && !sc.anyNonLosing;
    if (hasNext && allLose) return true;
    if (sc.anyUnvisited) return false;
    int v = n.visits + n.vvsum;
    if (v < 0) v = 0;
    if (v >= s.prm->tabLen) v = s.prm->tabLen - 1;
    const int allowed = n.depth == 0 ? pw_root_of(s, v) : pw_node_of(s, v);
    return hasNext && n.expanded < allowed;
}

// node.h:549-613
__device__ inline void update_child_node_type(G& s, Node& n, int idx, uint8_t ct) {
    if (n.type != T_UNSOLVED) return;
    if (idx < 0 || idx >= n.cntTypes) return;
    Edge* e = edges_of(s, n);
    if (e[idx].ctype != T_UNSOLVED) return;
    e[idx].ctype = ct;
    n.unsolved--;
    if (ct == T_LOSS) {
        n.type = T_WIN;
        n.endInPly = s.nodes[e[idx].child].endInPly + 1;
        return;
    }
    const bool hasNext = n.more != 0;
    if (n.unsolved == 0 && (n.flags & F_EXPANDED) && !hasNext) {
        bool allWins = true, hasDrawn = false;
        int longest = 0;
        for (int i = 0; i < n.cntTypes; ++i) {
            if (e[i].ctype != T_WIN) allWins = false;
            if (e[i].ctype == T_DRAW) hasDrawn = true;
            const int ep = s.nodes[e[i].child].endInPly;
            if (ep > longest) longest = ep;
        }
        if (allWins) { n.type = T_LOSS; n.endInPly = longest + 1; }
        else if (hasDrawn) n.type = T_DRAW;
    }
}

// Backup of an UNSOLVED leaf: no solver bookkeeping can trigger on the way up, so every path level is an
// independent read-modify-write and lane i updates level i; the value alternates sign from the leaf (level
// len-1) upwards.  Returns false (nothing done) when the path must be walked sequentially.
__device__ inline bool backup_levels(G& s, const TrajEnt* tr, int len, float v) {
    if (len > 64) return false;
    const int lane = threadIdx.x & 63;
    // A transposition edge made on another path can close a cycle, so a node may occur twice on this
    // path after all; those (rare) paths take the sequential form.
    const int mine = lane < len ? tr[lane].node : -1 - lane;
    bool dup = false;
    for (int j = 0; j < len; ++j) dup |= j != lane && ulane(mine, j) == mine;
    if (wave_any(dup)) return false;
    for (int i = lane; i < len; i += 64) {
        const TrajEnt t = tr[i];
        const float vi = ((len - 1 - i) & 1) ? -v : v;
        Node* np = &s.nodes[t.node];
        if (t.childIdx >= 0) {
            Edge* e = edges_of(s, *np) + t.childIdx;             // update_and_remove_virtual_loss node.h:104-121
            const int visits = e->visits + 1;
            e->vloss -= 1; e->visits = visits;
            if (visits == 1) { e->vsum = vi; e->q = vi; }
            else { const float vs = e->vsum + vi; e->vsum = vs; e->q = vs / (float)visits; }
            np->vvsum -= 1;
        }
        np->valueSum += vi; np->visits += 1;
    }
    wave_fence();
    return true;
}

// searchthread.cc:197-239
__device__ inline void backup(G& s, const TrajEnt* tr, int len, float v, int leafType = -1) {
    if (len <= 0) return;
    uint8_t childType = leafType >= 0 ? (uint8_t)leafType : s.nodes[tr[len - 1].node].type;
    if (childType == T_UNSOLVED && backup_levels(s, tr, len, v)) return;
    if (childType == T_WIN) v = 1.0f;
    else if (childType == T_LOSS) v = -1.0f;
    else if (childType == T_DRAW) v = s.nodes[tr[len - 1].node].team == s.nodes[tr[0].node].team ? -s.prm->drawContempt : s.prm->drawContempt;
    for (int i = len - 1; i >= 0; --i) {
        Node n = s.nodes[tr[i].node];
        const int idx = tr[i].childIdx;
        if (idx >= 0) {
            Edge* e = edges_of(s, n) + idx;                    // update_and_remove_virtual_loss node.h:104-121
            e->vloss--; n.vvsum--; e->visits++;
            if (e->visits == 1) { e->vsum = v; e->q = v; }
            else { e->vsum += v; e->q = e->vsum / (float)e->visits; }
            n.valueSum += v; n.visits++;
            if (childType != T_UNSOLVED) {
                if (n.cntTypes < n.expanded) {                 // init_child_node_types :531-541
                    Edge* all = edges_of(s, n);
                    for (int k = n.cntTypes; k < n.expanded; ++k) all[k].ctype = T_UNSOLVED;
                    n.unsolved += n.expanded - n.cntTypes;
                    n.cntTypes = n.expanded;
                }
                update_child_node_type(s, n, idx, childType);
                childType = n.type;
            } else childType = T_UNSOLVED;
        } else { n.valueSum += v; n.visits++; }                // update_terminal
        s.nodes[tr[i].node] = n;
        v = -v;
    }
}
__device__ inline void cancel_virtual_losses(G& s, const TrajEnt* tr, int len) {   // :241-247
    for (int i = 0; i < len; ++i)
        if (tr[i].childIdx >= 0) {
            Node& n = s.nodes[tr[i].node];
            edges_of(s, n)[tr[i].childIdx].vloss--;
            n.vvsum--;
        }
}

// node.cc:6-119 — lane-parallel PUCT: lane i scores edge i (+64k), wave arg-max, lowest index wins ties.
struct Sel { int child, idx; bool reserved; int pending; };
__device__ inline Sel select_child_and_apply_virtual_loss(G& s, int nodeId, const Node& n, const EdgeScan& sc, u64* unavailMask /*LDS, 8 words*/) {
    const int lane = threadIdx.x & 63;
    const int limit = n.expanded;
    if (limit == 0) return {-1, -1, false, -1};
    Edge* e = edges_of(s, n);
    const int visits = sc.visits;
    const float sqrtVisits = sqrtf((float)visits);
    int vi = visits < 0 ? 0 : (visits >= s.prm->tabLen ? s.prm->tabLen - 1 : visits);
    const float c = cpuct_of(s, vi);
    const float explorationBase = c * sqrtVisits;
    const bool hasNonLosing = sc.anyNonLosing && n.type == T_UNSOLVED;
    const float parentQ = visits > 0 ? (n.valueSum / (float)visits) : 0.0f;
    const float fpuQ = (s.prm->enableDynamicFpu && visits > 0)
        ? clampf(parentQ - s.prm->fpuReduction * sqrtf(fmaxf(0.0f, sc.visitedPolicySum)), -1.0f, 1.0f) : Q_INIT;
    for (int w = 0; w < 8; ++w) unavailMask[w] = 0;
    int pending = -1;
    PROF_T(tq2);
    while (true) {
        u64 bestKey = 0;
        for (int base = 0; base < limit; base += 64) {
            const int i = base + lane;
            float score = -INFINITY;
            bool ok = i < limit;
            if (ok) {
                Edge ed = sc.ed;
                int ct = sc.ct;
                if (base != 0) { ed = e[i]; ct = s.nodes[ed.child].type; }      // beyond the cached chunk (rare)
                if ((unavailMask[i >> 6] >> (i & 63)) & 1) ok = false;
                else if (hasNonLosing && ct == T_WIN) ok = false;
                else {
                    const int vl = ed.vloss;
                    const uint32_t ne = (uint32_t)ed.visits + (uint32_t)vl;
                    float q;
                    if (ne == 0) q = fpuQ;
                    else if (vl == 0) q = ed.q;
                    else q = (ed.vsum - (float)vl) / (float)ne;          // VIRTUAL_LOSS style
                    const float u = explorationBase * ed.prior / (1.0f + (float)ne);
                    score = q + u;
                }
            }
            // wave arg-max of (score, lowest index).  `score > best` semantic: NaN / -inf never win (key 0).
            const bool cand = ok && score > -INFINITY;             // false for NaN as well
            const u64 key = cand ? ((u64)float_order_bits(score + 0.0f) << 32) | (u32)~(u32)i : 0ULL;
            const u64 top = wave_max_u64(key);
            if (top > bestKey) bestKey = top;                      // later chunks hold higher indices: a tie keeps the earlier one
        }
        const int bestIdx = bestKey ? (int)~(u32)bestKey : -1;
        if (bestIdx < 0) return {-1, -1, false, pending};
        const int child = bestIdx < 64 ? ulane(sc.ed.child, bestIdx) : e[bestIdx].child;
        Node& cn = s.nodes[child];
        bool reserved = false;
        // stepping onto an expanded, unsolved child only adds a virtual loss (cancel_virtual_losses takes it back); anything else —
        // a reservation, the end of the descent — waits for the outstanding creation
        if (!((cn.flags & F_EXPANDED) && cn.type == T_UNSOLVED) && !resolve_create(s)) return {-2, -2, false, pending};
        if (!(cn.flags & F_EXPANDED) && cn.type == T_UNSOLVED) {
            if (cn.flags & F_PENDING) {                        // try_reserve_evaluation failed
                pending = child;
                unavailMask[bestIdx >> 6] |= 1ULL << (bestIdx & 63);
                continue;
            }
            cn.flags |= F_PENDING;
            reserved = true;
        }
        if (bestIdx < 64) e[bestIdx].vloss = ulane(sc.ed.vloss, bestIdx) + 1;      // no read-modify-write round trip
        else e[bestIdx].vloss++;
        s.nodes[nodeId].vvsum = n.vvsum + 1;
        PROF_ADD(20, tq2);
        return {child, bestIdx, reserved, -1};
    }
}

// ---- search path bookkeeping ------------------------------------------------------------------
// A descent is a walk over node ids; the joint position (two register-resident boards + history view) exists only where it
// is needed: to give a first-reached child its position and hash, and to classify / encode the leaf.
struct Path {
    JBoard jb;
    int len;                 // trajectory length
    int posNode;             // node whose position jb holds (-1: none)
};
__device__ inline void path_reset(const G& s, Path& p) {      // the game's current position (root of every search)
    load_pos(p.jb.bd[0], &s.g->pos[0]);
    load_pos(p.jb.bd[1], &s.g->pos[1]);
    p.jb.hist[0] = s.hist[0]; p.jb.hist[1] = s.hist[1];
    p.jb.hlen[0] = s.g->hlen[0]; p.jb.hlen[1] = s.g->hlen[1];
    p.jb.prefix[0] = s.g->prefix[0]; p.jb.prefix[1] = s.g->prefix[1];
    p.len = 0;
    p.posNode = -1;
}
__device__ __forceinline__ NodePos* nodepos_of(const G& s, const Node& n) { return reinterpret_cast<NodePos*>(s.arena + n.posOff); }
// jb <- cached position of `node` (which must have one)
__device__ inline void path_load(const G& s, Path& p, int node) {
    const NodePos* np = nodepos_of(s, s.nodes[node]);
    load_pos(p.jb.bd[0], &np->pos[0]);
    load_pos(p.jb.bd[1], &np->pos[1]);
    p.jb.hist[0] = s.hist[0]; p.jb.hist[1] = s.hist[1];
    p.jb.hlen[0] = np->hlen[0]; p.jb.hlen[1] = np->hlen[1];
    p.jb.prefix[0] = np->prefix[0]; p.jb.prefix[1] = np->prefix[1];
    p.posNode = node;
}
// cache jb as the position of `node`; false when the arena is exhausted (overflow flagged)
__device__ inline bool path_store(G& s, const RulesTab& rt, const Path& p, int node) {
    const u32 off = arena_alloc(s, sizeof(NodePos));
    if (!off) return false;
    NodePos* np = reinterpret_cast<NodePos*>(s.arena + off);
    store_pos(&np->pos[0], p.jb.bd[0]);
    store_pos(&np->pos[1], p.jb.bd[1]);
    np->repKey[0] = rep_key(rt, p.jb.bd[0]); np->repKey[1] = rep_key(rt, p.jb.bd[1]);
    np->prefix[0] = p.jb.prefix[0]; np->prefix[1] = p.jb.prefix[1];
    np->hlen[0] = p.jb.hlen[0]; np->hlen[1] = p.jb.hlen[1];
    s.nodes[node].posOff = off;
    return true;
}
// History keys of the search path: entry i > 0 of the trajectory appended node i's repetition key to every board that moved
// on the edge into it (Board::push_move -> record_position, board.h:95-102).  Rebuilt behind the game's keys before the
// leaf's draw / repetition tests; lane i serves path level i.
__device__ inline void path_rebuild_history(const G& s, const TrajEnt* traj, int len) {
    const int lane = threadIdx.x & 63;
    for (int base = 1; base < len; base += 64) {
        const int i = base + lane;
        bool mvA = false, mvB = false;
        u64 kA = 0, kB = 0;
        if (i < len) {
            const TrajEnt t = traj[i];
            mvA = t.moveA != 0; mvB = t.moveB != 0;
            const NodePos* np = nodepos_of(s, s.nodes[t.node]);
            if (mvA) kA = np->repKey[0];
            if (mvB) kB = np->repKey[1];
        }
        // keys of earlier chunks (paths longer than 64 levels) were appended by the previous round
        const u64 mA = __ballot(mvA), mB = __ballot(mvB);
        const u64 below = (1ULL << lane) - 1ULL;
        int offA = 0, offB = 0;
        for (int b0 = 1; b0 < base; b0 += 64) {   // count moves of the earlier chunks (rare: only for len > 65)
            const int j = b0 + lane;
            const bool a = j < base && traj[j].moveA != 0, bb = j < base && traj[j].moveB != 0;
            offA += __popcll(__ballot(a)); offB += __popcll(__ballot(bb));
        }
        if (mvA) s.hist[0][s.g->hlen[0] + offA + __popcll(mA & below)] = kA;
        if (mvB) s.hist[1][s.g->hlen[1] + offB + __popcll(mB & below)] = kB;
    }
    wave_sync();                                                 // the keys are read back by this wave only (draw / repetition tests)
}

// searchthread.cc:741-806.  Returns: 0 = not expanded, 1 = expanded, 2 = pending (selection must abort).
// deferEdge: the edge updates (replace_child, remove_virtual_loss) are left to the caller's partner — with the LDS tree every edge
// store of a collect phase comes from the traversal wave (collect_batch applies them when the descent is handed back): the classifier
// and the traversal running ahead of it never write the same edge or the same node's virtual-visit sum, and the traversal need not
// drain its edge stores before it posts a request.
__device__ inline int canonicalize_child(G& s, const RulesTab& rt, Path& p, TrajEnt* traj, int parent, int idx, int& child, bool& reserved, bool rootAdv, int rootTeam, int* pendingOut, bool deferEdge) {
    Node& c0 = s.nodes[child];
    if (!s.prm->enableTranspositions) return (c0.flags & F_EXPANDED) ? 1 : 0;
    if (c0.hash != 0) return (c0.flags & F_EXPANDED) ? 1 : 0;
    const bool childAdv = c0.team == rootTeam ? rootAdv : !rootAdv;
    const u64 h = board_hash_key(p.jb.bd[0], p.jb.bd[1], hist_of(p.jb, 0), hist_of(p.jb, 1), childAdv, rt.zob.time_adv);
    c0.hash = h;
    const int canonical = tt_insert_or_get(s, h, child);
    if (canonical == s.inflight) svc_wait(s);
    bool isAncestor = false;
    for (int i = 0; i < p.len; ++i) isAncestor |= traj[i].node == canonical;
    const bool teamMismatch = s.nodes[canonical].team != c0.team;
    if (canonical == child || isAncestor || teamMismatch) return (c0.flags & F_EXPANDED) ? 1 : 0;
    if (reserved) { c0.flags &= ~F_PENDING; reserved = false; }
    if (!deferEdge) edges_of(s, s.nodes[parent])[idx].child = canonical;       // replace_child
    child = canonical;
    Node& cn = s.nodes[canonical];
    if (cn.flags & F_EXPANDED) return 1;
    if (cn.type != T_UNSOLVED) return 0;
    if (cn.flags & F_PENDING) {
        if (!deferEdge) {
            Node& pn = s.nodes[parent];
            edges_of(s, pn)[idx].vloss--; pn.vvsum--;         // remove_virtual_loss
        }
        *pendingOut = canonical;
        return 2;
    }
    cn.flags |= F_PENDING;
    reserved = true;
    return 0;
}

// A child reached for the first time gets its position here: parent position (cached) + the edge's joint move.
// Returns false when the pools are exhausted.
__device__ __forceinline__ bool position_child(G& s, const RulesTab& rt, Path& p, int parent, int child, u32 ma, u32 mb) {
    PROF_T(tl);
    if (p.posNode != parent) path_load(s, p, parent);
    PROF_ADD(27, tl);
    PROF_T(tm);
    jb_make(rt, p.jb, ma, mb, false);                      // keys of the path are rebuilt at the leaf (path_rebuild_history)
    PROF_ADD(4, tm);
    p.posNode = child;
    PROF_T(tst);
    const bool ok = path_store(s, rt, p, child);
    PROF_ADD(28, tst);
    return ok;
}

// searchthread.cc:818-916 as a resumable walk over node ids.  The traversal never holds a position: a child reached for the
// first time (no position record yet) ends the walk with DESC_CREATE, and the classifier wave gives it its position, hash and
// transposition lookup (serve_leaf) while the traversal starts its next descent.
struct Desc {
    int cur; bool reserved; int len;         // node under examination, whether this descent holds its evaluation reservation, trajectory length
    int nv, es;                              // nodes visited / edges scanned by this descent (counted once the descent is kept)
    int parent, idx, child; u32 ma, mb; bool childReserved, widened;   // DESC_CREATE: the edge (parent, idx) -> child to be positioned
};
enum : int { DESC_LEAF = 0, DESC_ABORT = 1, DESC_CREATE = 2, DESC_CANCEL = 3 };
__device__ __forceinline__ void desc_begin(G& s, TrajEnt* traj, Desc& d) {
    d.cur = s.g->root; d.reserved = false; d.len = 1; d.nv = 0; d.es = 0;
    traj[0] = TrajEnt{d.cur, -1, 0, 0};
    mark_dirty(s, d.cur);
}
// DESC_LEAF: d.cur is the leaf, d.reserved its reservation.  DESC_ABORT: the selection failed (pending evaluation, exhausted
// pool): the caller cancels the path's virtual losses.  DESC_CANCEL: the outstanding creation was handed back while this descent
// was running ahead of it (only virtual losses have been applied: the caller cancels them and repeats the descent later).
__device__ __forceinline__ int descend(G& s, const RulesTab& rt, TrajEnt* traj, Desc& d, u64* unavailMask) {
    (void)rt;
    while (true) {
        const int cur = d.cur;
        if (cur == s.inflight) { svc_wait(s); if (!resolve_create(s)) return DESC_CANCEL; }
        if (cur == s.genInflight) gen_wait(s);
        Node n = s.nodes[cur];
        d.nv++; d.es += n.expanded;
        if (n.type != T_UNSOLVED) return resolve_create(s) ? DESC_LEAF : DESC_CANCEL;
        if (!(n.flags & F_EXPANDED)) {
            if (!resolve_create(s)) return DESC_CANCEL;
            if (!d.reserved) {
                if (n.flags & F_PENDING) return DESC_ABORT;
                s.nodes[cur].flags = n.flags | F_PENDING;
                d.reserved = true;
            }
            return DESC_LEAF;
        }
        if (d.len >= MAX_TRAJ - 1) { if (!resolve_create(s)) return DESC_CANCEL; s.g->overflow |= 4; return DESC_ABORT; }
        int next = -1, childIdx = -1;
        u32 ma = 0, mb = 0;
        bool childReserved = false, widened = false;
        PROF_T(tw);
        EdgeScan sc;
        if (!scan_edges(s, n, sc)) return DESC_CANCEL;
        PROF_ADD(29, tw);
        const bool widen = should_expand_new_child(s, n, sc);
        PROF_ADD(1, tw);
        if (widen) {
            if (!resolve_create(s)) return DESC_CANCEL;           // a popped candidate and a new edge cannot be taken back
            // expand_next_joint_child(nullptr, 0, ..., reserveForSelection = true)  node.h:199-262
            gen_wait(s);                                          // one refill in flight
            GenHdr* gh = gen_of(s, n);
            GenHdr h = *gh;                                       // header in registers: through the pointer every field access
            HeapEnt he;                                           // is an L2 round trip that later arena stores force to repeat
            PROF_T(tg);
            const bool async = s.genAckSeq != nullptr;
            const bool got = gen_next(s, h, &he, !async);
            *gh = h;
            if (got && async) {
                // the successor pushes of the popped pair (visited-set tests, sit-rule closure, frontier growth) run on the
                // block's generator wave; this node's generator and `more` flag are not read again before gen_wait
                wave_fence();
                if ((threadIdx.x & 63) == 0) { s.gq->node = cur; s.gq->genOff = n.gen; s.gq->iA = he.iA; s.gq->iB = he.iB; }
                wave_fence();
                s.genReqSeq++;
                if ((threadIdx.x & 63) == 0) __hip_atomic_store(&s.gq->reqSeq, s.genReqSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                s.genInflight = cur;
            } else s.nodes[cur].more = h.heapSize > 0;
            PROF_ADD(2, tg);
            if (got) {
                const float jp = joint_prior(s, h, he.iA, he.iB, &ma, &mb);
                const int child = node_alloc(s, n.team ^ 1, n.depth + 1);
                Node& nn = s.nodes[cur];
                Edge* slot = child >= 0 ? edge_append(s, nn) : nullptr;
                if (!slot) return DESC_ABORT;                 // pool exhausted (overflow flagged)
                s.nodes[child].flags |= F_PENDING;
                *slot = Edge{child, Q_INIT, Q_INIT, jp, 0, 1, ma, mb, he.iA, he.iB, T_UNSOLVED, 0, 0, 0};
                nn.vvsum++;
                childIdx = nn.expanded;
                nn.expanded++;
                next = child;
                childReserved = true;
                widened = true;
            }
        }
        if (!widened) {
            PROF_T(ts);
            const Sel sel = select_child_and_apply_virtual_loss(s, cur, n, sc, unavailMask);
            PROF_ADD(3, ts);
            if (sel.child == -2) return DESC_CANCEL;
            if (sel.child < 0 || sel.idx < 0) return resolve_create(s) ? DESC_ABORT : DESC_CANCEL;
            next = sel.child; childIdx = sel.idx; childReserved = sel.reserved;
            if (childIdx < 64) { ma = (u32)ulane((int)sc.ed.moveA, childIdx); mb = (u32)ulane((int)sc.ed.moveB, childIdx); }   // edge held by lane childIdx
            else { const Edge ed = edges_of(s, s.nodes[cur])[childIdx]; ma = ed.moveA; mb = ed.moveB; }
        }
        mark_dirty(s, next);                                   // also when canonicalisation replaces it (hash, reservation, position)
        if (s.nodes[next].posOff == 0) {
            // Board::make_moves + canonicalize_child for a child that has never been reached (no position, hence no hash): the
            // classifier wave's work.  (The outstanding creation is resolved here: a widening waits for it, and so does the
            // reservation of an unexpanded child.)
            d.parent = cur; d.idx = childIdx; d.child = next; d.ma = ma; d.mb = mb; d.childReserved = childReserved; d.widened = widened;
            return DESC_CREATE;
        }
        const int cr = (s.nodes[next].flags & F_EXPANDED) ? 1 : 0;   // canonicalize_child's early outs: hash already set
        traj[d.len - 1].childIdx = childIdx;
        traj[d.len] = TrajEnt{next, -1, ma, mb};
        d.len++;
        if (widened) {
            if (cr == 1) { d.cur = next; d.reserved = false; continue; }
            d.cur = next; d.reserved = childReserved;
            return resolve_create(s) ? DESC_LEAF : DESC_CANCEL;
        }
        d.reserved = childReserved;
        d.cur = next;
    }
}

// ---- leaf planes: wave-cooperative board_to_planes (fp16) for one hm_board in LDS ----------
// WT: write-through stores (hm_queue.hpp) — the rows are read by another workgroup (the persistent evaluator) inside the launch
template <bool WT = false>
__device__ inline void write_planes_f16(const RulesTab& rt, const u64* bw /*26 words, LDS*/, uint4* dst, u64* s_mask, uint32_t* s_val) {
    const int lane = threadIdx.x & 63;
    const u64 tail = bw[25];
    const int team = (int)((tail >> 16) & 0xff), adv = (int)((tail >> 24) & 0xff);
    const uint32_t ONE = 0x3C00u;
    for (int p = lane; p < HM_NB_PLANES; p += 64) {
        const int b = p >= HM_NB_PLANES_PER_BOARD ? 1 : 0;
        const int j = p - b * HM_NB_PLANES_PER_BOARD;
        const u64* pw = bw + 12 * b;
        const bool flip = b == 0 ? team == 1 : team == 0;
        const int first = b == 0 ? team : team ^ 1;
        const u64 t1 = pw[11];
        const int castling = (int)((t1 >> 16) & 0xff), ep = (int)((t1 >> 24) & 0xff);
        const int stm = (int)((t1 >> 32) & 0xff), r50 = (int)((t1 >> 40) & 0xff);
        u64 mask = ~0ULL;
        uint32_t val = ONE;
        bool orient = false;
        if (j < 12) { const int c = j < 6 ? first : first ^ 1; mask = pw[j < 6 ? j : j - 6] & pw[6 + c]; orient = true; }
        else if (j < 22) {
            const int c = j < 17 ? first : first ^ 1, k = j < 17 ? j - 12 : j - 17;
            const int byteIdx = 80 + c * 5 + k;
            const int cnt = (int)((pw[byteIdx >> 3] >> (8 * (byteIdx & 7))) & 0xff);
            val = rt.pocket_f16[cnt & 63];
        } else if (j < 24) { const int c = j == 22 ? first : first ^ 1; mask = pw[8] & pw[6 + c]; orient = true; }
        else if (j == 24) { mask = ep < 64 ? bit(ep) : 0; orient = true; }
        else if (j == 25) val = stm == first ? ONE : 0;
        else if (j == 26) {}
        else if (j < 31) { const int c = j < 29 ? first : first ^ 1; const int right = ((j - 27) & 1) ? (c == 0 ? 2 : 8) : (c == 0 ? 1 : 4); val = (castling & right) ? ONE : 0; }
        else if (j == 31) val = adv ? ONE : 0;
        else if (j < 34) {
            const uint32_t lm = (uint32_t)(bw[24] >> (32 * b));
            mask = 0;
            if (lm != 0) {
                const bool drop = (lm & (15u << 12)) == HM_MT_DROP;
                int sq = j == 32 ? (int)((lm >> 6) & 63) : (int)(lm & 63);
                if (flip) sq ^= 56;
                if (!(j == 32 && drop)) mask = bit(sq);
            }
        } else if (j == 34) val = rt.r50_f16[r50 > 50 ? 50 : r50];
        else { const int rc = (int)((tail >> (8 * b)) & 0xff); val = rc >= (j == 35 ? 2 : 3) ? ONE : 0; }
        if (orient && flip) mask = __builtin_bswap64(mask);
        s_mask[p] = mask; s_val[p] = val;
    }
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < HM_PLANE_VALUES / 8; c += 64) {
        const int sq0 = c * 8, p = sq0 >> 6;
        const uint32_t b = (uint32_t)(s_mask[p] >> (sq0 & 63)) & 0xff, v = s_val[p];
        uint4 o;
        o.x = ((b & 1) ? v : 0) | ((b & 2) ? v << 16 : 0);
        o.y = ((b & 4) ? v : 0) | ((b & 8) ? v << 16 : 0);
        o.z = ((b & 16) ? v : 0) | ((b & 32) ? v << 16 : 0);
        o.w = ((b & 64) ? v : 0) | ((b & 128) ? v << 16 : 0);
        if constexpr (WT) { hmq::u32x4q w; w.x = o.x; w.y = o.y; w.z = o.z; w.w = o.w; hmq::store16_wt(&dst[c], w); }
        else dst[c] = o;
    }
    __builtin_amdgcn_wave_barrier();
}

struct PreSorted { const u32* moves; const float* priors; };   // this game's [2][BATCH][2][HM_MAX_MOVES] arrays sorted by the persistent evaluator
struct ExpLds {          // per-wave scratch of expand_leaf
    u32 lists[2][HM_MAX_MOVES];
    float priors[2][HM_MAX_MOVES + 8];
};
struct WaveLds {
    union {                  // expand_leaf scratch aliases lists[0..4]: never live together (k_raw_policy uses lists[0..1] == exp.lists)
        u32 lists[NLISTS][HM_MAX_MOVES];
        ExpLds exp;
    };
    u64 board[BATCH][26];    // hm_board images of this batch's network leaves, handed to the plane-writer wave
    int posted, done;        // hand-off flags (k_collect: wave 0 posts images, wave 1 writes their planes)
    int postBuf;             // which of the two batches (0/1) the posted leaves belong to
    int postRow[BATCH];      // plane row of each posted image (ring slot = post number & 7)
    int postReady[BATCH];    // persistent search: arrivals for each posted image (+1 plane writer, +1 generator, +4 / +8 classifier: network leaf / dropped)
    int servedCnt;           // posts the plane-writer wave has finished (flow control of the image ring)
    int servedCntB;          // posts whose board-B move list the generator wave has finished
    int postCount;           // images posted so far in this launch (classifier wave's counter)
    int listWords;           // wave 1: move-list words written this launch (traffic accounting)
    u32 helperLists[2][HM_MAX_MOVES];   // plane-writer wave / generator wave: legal list of the leaf board being served
    u64 pmask[HM_NB_PLANES + 6];
    uint32_t pval[HM_NB_PLANES + 6];
    u64 unavail[8];
    TrajEnt traj[MAX_TRAJ];
    // traversal wave -> classifier wave (k_collect): one leaf in flight
    // (two request slots, used alternately: the traversal may post the next leaf while the classifier is still writing the
    // previous leaf's context record and board image)
    TrajEnt trajReq[2][MAX_TRAJ];    // the leaf's path (copied: the traversal reuses `traj` for its next descent)
    // create != 0: the leaf is a child reached for the first time — the classifier first gives it its position (parent's position +
    // the joint move), hash and transposition lookup (position_child, canonicalize_child), then classifies it
    struct Req { int leaf, trajLen, ctxIdx, buf, reserved, first, create, parent, idx; u32 ma, mb; } req[2];
    int reqSeq, typeSeq, ackSeq, svcStop;   // requests posted / terminal test done / finished; svcStop: no more requests in this launch
    int createSeq;                   // requests whose creation step is done (resolve_create)
    struct { int fast, cr, next, reserved; } createRes;   // its outcome: fast = the child is the fresh, reserved leaf the traversal assumed; else canonicalize_child's result (cr 3: pool exhausted)
    int reqResult;                   // outcome of the last finished request: 0 network leaf, 1 terminal, 2 dropped (not reserved)
    int svcValid;                    // network leaves of the current batch so far (= plane rows posted)
    int batchLeaf[BATCH];            // leaves of the batch being collected (same-batch collision test)
    GenQ gq;
};

__device__ inline G make_view(const Pools& pl, const Params& prm, int g) {
    G s;
    s.g = pl.games + g;
    s.nodes = pl.nodes + (size_t)g * prm.nodeCap;
    s.arena = pl.arena + (size_t)g * prm.arenaCap;
    s.ttKeys = pl.ttKeys + (size_t)g * prm.ttCap;
    s.ttVals = pl.ttVals + (size_t)g * prm.ttCap;
    s.ctx = pl.ctx + (size_t)g * 2 * BATCH;
    s.traj = pl.traj + (size_t)g * 2 * BATCH * MAX_TRAJ;
    s.hist[0] = pl.hist + ((size_t)g * 2 + 0) * prm.histCap;
    s.hist[1] = pl.hist + ((size_t)g * 2 + 1) * prm.histCap;
    s.noise[0] = pl.noise + ((size_t)g * 2 + 0) * NOISE_CAP;
    s.noise[1] = pl.noise + ((size_t)g * 2 + 1) * NOISE_CAP;
    s.leafMoves = pl.leafMoves + (size_t)g * 2 * BATCH * 2 * HM_MAX_MOVES;
    s.leafCounts = pl.leafCounts + (size_t)g * 2 * BATCH * 2;
    s.prm = &prm; s.pl = &pl;
    s.ldsCpuct = nullptr; s.ldsPwRoot = nullptr; s.ldsPwNode = nullptr; s.tabN = 0;
    s.inflight = -1; s.reqSeq = 0; s.svcBusy = false; s.ackSeq = nullptr; s.typeSeq = nullptr;
    s.genInflight = -1; s.genReqSeq = 0; s.genAckSeq = nullptr; s.gq = nullptr; s.nv = 0; s.es = 0; s.dirty = nullptr;
    s.jPending = false; s.jTakenBack = false; s.createSeq = nullptr; s.createFast = nullptr; s.ldsTree = false;
    return s;
}

// process of one context list (searchthread.cc:444-639).  outs == nullptr for terminal-only batches.
struct NetOut { const uint16_t *value, *piA, *piB, *wdl, *ml; };

__device__ inline void expand_leaf(G& s, const RulesTab& rt, ExpLds& L, const Ctx& ctx, int buf, int row, int rootTeam, bool rootAdv, const uint16_t* piA, const uint16_t* piB,
                                   const PreSorted* pre = nullptr) {
    const int lane = threadIdx.x & 63;
    P bd[2];
    {
        const NodePos* np = nodepos_of(s, s.nodes[ctx.leaf]);   // cached when the traversal first reached the leaf
        load_pos(bd[0], &np->pos[0]);
        load_pos(bd[1], &np->pos[1]);
    }
    const int team = ctx.team;
    const bool leafAdv = team == rootTeam ? rootAdv : !rootAdv;
    const bool aOn = (int)bd[0].stm == team, bOn = (int)bd[1].stm == (team ^ 1);
    // legal moves of the on-turn boards (R/B under-promotions already erased, utils.h:169-182): generated by the
    // plane-writer wave of k_collect while the traversal went on, fetched here
    int nReal[2];
    {
        const int* cnt = s.leafCounts + ((size_t)buf * BATCH + row) * 2;
        nReal[0] = cnt[0] & 0xffff; nReal[1] = cnt[1] & 0xffff;                // (bit 16: the board's side to move, for the evaluator)
        const u32* src = s.leafMoves + ((size_t)buf * BATCH + row) * 2 * HM_MAX_MOVES;
        if (!pre)
            for (int b = 0; b < 2; ++b)
                for (int i = lane; i < nReal[b]; i += 64) L.lists[b][i] = src[(size_t)b * HM_MAX_MOVES + i];
    }
    __builtin_amdgcn_wave_barrier();
    PROF_T(te2);
    // JointCandidateGenerator::initialize (joint_action.h:195-278): rank sort by (prior desc, index asc)
    // one bump allocation for the header, both sorted move / prior arrays, the frontier, the visited list and the first four
    // edge slots (seven round trips to the game's arena counter otherwise)
    int nAct[2] = {nReal[0] + 1, nReal[1] + 1};
    auto units = [](u32 bytes) { return (bytes + 7) >> 3; };
    const u32 uHdr = units(sizeof(GenHdr)), uA = units((u32)nAct[0] * 4), uB = units((u32)nAct[1] * 4);
    const u32 uHeap = units(16 * 8), uVis = units(32 * 4), uEdges = units(4 * sizeof(Edge));
    const u32 genOff = arena_alloc(s, (uHdr + 2 * uA + 2 * uB + uHeap + uVis + uEdges) * 8);
    if (!genOff) return;
    GenHdr h;
    h.nA = nAct[0]; h.nB = nAct[1];
    h.aOn = aOn; h.bOn = bOn; h.adv = leafAdv;
    h.aCan = aOn && nReal[0] > 0; h.bCan = bOn && nReal[1] > 0;
    h.pad[0] = h.pad[1] = h.pad[2] = 0;
    u32 offM[2], offP[2];
    offM[0] = genOff + uHdr; offP[0] = offM[0] + uA;
    offM[1] = offP[0] + uA;  offP[1] = offM[1] + uB;
    h.heapCap = 16; h.heapSize = 0; h.heap = offP[1] + uB;
    h.visCap = 32; h.visSize = 0; h.visited = h.heap + uHeap;
    const u32 firstEdges = h.visited + uVis;
    // root Dirichlet noise (node.h:286-315) applies to the root's own expansion only
    const Node leaf0 = s.nodes[ctx.leaf];
    const bool noisy = leaf0.depth == 0 && s.g->alpha > 0.0f && s.g->eps > 0.0f;
    if (pre) {
        // the persistent evaluator ran the prior pipeline on the logits in its LDS (hm_policy.hpp) and left the sorted arrays
        const u32* sm = pre->moves + ((size_t)buf * BATCH + row) * 2 * HM_MAX_MOVES;
        const float* spr = pre->priors + ((size_t)buf * BATCH + row) * 2 * HM_MAX_MOVES;
        for (int b = 0; b < 2; ++b) {
            u32* outM = reinterpret_cast<u32*>(s.arena + offM[b]);
            float* outP = reinterpret_cast<float*>(s.arena + offP[b]);
            for (int i = lane; i < nAct[b]; i += 64) { outM[i] = sm[(size_t)b * HM_MAX_MOVES + i]; outP[i] = spr[(size_t)b * HM_MAX_MOVES + i]; }
        }
    } else {
        for (int b = 0; b < 2; ++b) {
            const uint16_t* pol = b == 0 ? piA : piB;
            board_priors_sorted(L.lists[b], L.priors[b], nReal[b], (int)(b ? bd[1].stm : bd[0].stm), s.pl->polNormal, s.pl->polDrop,
                                [pol](int idx) { return pol[idx]; }, noisy ? s.noise[b] : nullptr, s.g->eps,
                                reinterpret_cast<u32*>(s.arena + offM[b]), reinterpret_cast<float*>(s.arena + offP[b]));
        }
    }
    PROF_ADD_T(40, te2, 64);
    PROF_T(te3);
    wave_fence();
    PROF_ADD_T(41, te3, 64);
    PROF_T(te4);
    h.movesA = offM[0]; h.movesB = offM[1]; h.priorsA = offP[0]; h.priorsB = offP[1];
    gen_push(s, h, 0, 0);
    if (h.heapSize == 0) { gen_push(s, h, 1, 0); gen_push(s, h, 0, 1); }
    Node leaf = leaf0;
    leaf.gen = genOff;
    leaf.expanded = 0;
    // first child (try_init_and_expand node.h:317-341)
    HeapEnt he;
    if (gen_next(s, h, &he)) {
        u32 ma, mb;
        const float jp = joint_prior(s, h, he.iA, he.iB, &ma, &mb);
        const int child = node_alloc(s, leaf.team ^ 1, leaf.depth + 1);
        if (leaf.edgeCap == 0) { leaf.edges = firstEdges; leaf.edgeCap = 4; }    // the slots allocated with the generator block
        Edge* slot = child >= 0 ? edge_append(s, leaf) : nullptr;
        if (slot) {
            *slot = Edge{child, Q_INIT, Q_INIT, jp, 0, 0, ma, mb, he.iA, he.iB, T_UNSOLVED, 0, 0, 0};
            leaf.expanded = 1;
            leaf.flags |= F_EXPANDED;
        }
    }
    *reinterpret_cast<GenHdr*>(s.arena + genOff) = h;
    // only the fields expansion owns: the backups of this batch update visits / value sum of the same node concurrently
    Node* np = &s.nodes[ctx.leaf];
    np->gen = leaf.gen; np->edges = leaf.edges; np->edgeCap = leaf.edgeCap; np->expanded = leaf.expanded;
    np->more = h.heapSize > 0;
    wave_fence();
    if ((leaf.flags & F_EXPANDED) && lane == 0) node_set_flag(np, F_EXPANDED);
    PROF_ADD_T(42, te4, 64);
}

__device__ inline float shape_value(const G& s, uint16_t valueH, const uint16_t* wdl, uint16_t mlH) {   // searchthread.cc:569-619
    const float bv = h2f(valueH);
    const float scalar = finite_f(bv) ? clampf(bv, -1.0f, 1.0f) : 0.0f;
    float nv = scalar;
    if (s.prm->enableWdl) {
        const float l = h2f(wdl[0]), d = h2f(wdl[1]), w = h2f(wdl[2]);
        if (finite_f(l) && finite_f(d) && finite_f(w)) {
            const float mx = fmaxf(l, fmaxf(d, w));
            const float el = hm_expf(l - mx), ed = hm_expf(d - mx), ew = hm_expf(w - mx);
            const float sum = el + ed + ew;
            if (finite_f(sum) && sum > 0.0f) {
                const float pl = el / sum, pd = ed / sum, pw = ew / sum;
                const float wv = pw - pl - s.prm->drawContempt * pd;
                const float ww = clampf(s.prm->wdlWeight, 0.0f, 1.0f);
                nv = (1.0f - ww) * scalar + ww * wv;
            }
        }
    }
    if (s.prm->mlDiscount > 0.0f) {
        const float np = clampf(h2f(mlH), 0.0f, 1.0f);
        const float disc = clampf(s.prm->mlDiscount, 0.0f, 1.0f);
        nv *= 1.0f - disc * np;
    }
    return clampf(nv, -1.0f, 1.0f);
}

// process_batch (searchthread.cc:444-639) split for the GPU: the leaf expansions of a batch are
// independent (distinct, reserved leaves) and run one per wave; value shaping + backup then run
// sequentially in context order on wave 0 (float sums and solver propagation are order dependent).
__device__ inline int ctx_row(const G& s, int buf, int i) {   // inference row of context i (non-terminal contexts in order)
    const int lane = threadIdx.x & 63;
    const bool nn = lane < i && !s.ctx[buf * BATCH + lane].terminal;      // i <= BATCH: one context per lane
    return __popcll(__ballot(nn));
}
__device__ __forceinline__ void expand_context(G& s, const RulesTab& rt, ExpLds& L, int buf, int i, int rootTeam, bool rootAdv, const NetOut* out, int rowBase, const PreSorted* pre = nullptr) {
    const Ctx& ctx = s.ctx[buf * BATCH + i];
    if (ctx.terminal) return;
    if (s.nodes[ctx.leaf].type != T_UNSOLVED) return;
    const int slot = ctx_row(s, buf, i);               // index of this leaf among the batch's network leaves
    const int row = rowBase + slot;
    if (ctx.leafHash != 0) s.nodes[ctx.leaf].hash = ctx.leafHash;
    if (!(s.nodes[ctx.leaf].flags & F_EXPANDED))
        expand_leaf(s, rt, L, ctx, buf, slot, rootTeam, rootAdv, out->piA + (size_t)row * HM_POLICY_VALUES, out->piB + (size_t)row * HM_POLICY_VALUES, pre);
}
__device__ inline void backup_batch(G& s, int buf, const NetOut* out, int rowBase) {
    const int n = s.g->ctxCount[buf];
    const int lane = threadIdx.x & 63;
    // lane c gathers context c: header, leaf type, shaped network value, and releases the leaf's reservation.
    // (A pending leaf has no children, so no backup of this batch can pass through it or change its type, and
    // nothing below reads F_PENDING: doing this up front is equivalent to doing it at each context's turn.)
    int len = 0, term = 0, type = 0;
    float val = 0.0f;
    bool mine = lane < n;
    if (mine) {
        const Ctx& c = s.ctx[buf * BATCH + lane];
        len = c.trajLen; term = c.terminal;
        Node& ln = s.nodes[c.leaf];
        type = ln.type;
        if (term) { val = c.termValue; if (c.reserved) node_clear_flag(&ln, F_PENDING); }
        else node_clear_flag(&ln, F_PENDING);
    }
    const u64 nnMask = __ballot(mine && !term);                // contexts that own an inference row, in order
    if (mine && !term && type == T_UNSOLVED) {
        const int row = rowBase + __popcll(nnMask & ((1ULL << lane) - 1ULL));
        val = shape_value(s, out->value[row], out->wdl + (size_t)row * 3, out->ml[row]);
    }
    wave_fence();
    for (int i = 0; i < n; ++i)                                // value sums and solver propagation are order dependent
        backup(s, s.traj + (size_t)(buf * BATCH + i) * MAX_TRAJ, ulane(len, i), ulanef(val, i), ulane(type, i));
    s.g->nodesSearched += n;
    s.g->ctxCount[buf] = 0;
    s.g->validCount[buf] = 0;
}
// single-wave form (used where only one wave runs: terminal-only batches inside k_collect)
__device__ inline void process_batch(G& s, const RulesTab& rt, ExpLds& L, int buf, int rootTeam, bool rootAdv, const NetOut* out, int rowBase) {
    const int n = s.g->ctxCount[buf];
    if (out) for (int i = 0; i < n; ++i) expand_context(s, rt, L, buf, i, rootTeam, rootAdv, out, rowBase);
    wave_fence();
    backup_batch(s, buf, out, rowBase);
}
__device__ inline void abort_batch(G& s, int buf) {   // searchthread.cc:641-659
    const int n = s.g->ctxCount[buf];
    int done = 0;
    for (int i = 0; i < n; ++i) {
        Ctx& ctx = s.ctx[buf * BATCH + i];
        const TrajEnt* tr = s.traj + (size_t)(buf * BATCH + i) * MAX_TRAJ;
        if (ctx.reserved) s.nodes[ctx.leaf].flags &= ~F_PENDING;
        if (ctx.terminal) { backup(s, tr, ctx.trajLen, ctx.termValue); done++; }
        else cancel_virtual_losses(s, tr, ctx.trajLen);
    }
    s.g->nodesSearched += done;
    s.g->ctxCount[buf] = 0;
    s.g->validCount[buf] = 0;
}

// collect_batch (searchthread.cc:255-442), split over two waves of the game's block.  The traversal wave (below) selects
// leaves; every unsolved leaf is handed to the classifier wave (serve_leaf), which runs the terminal test
// (classify_terminal_position :99-139), completes the context record and, for a network leaf, posts the hm_board image to the
// plane-writer wave — while the traversal is already descending again.  The hand-off keeps the sequential semantics: the
// traversal never reads the Node of a leaf whose request is outstanding (svc_wait guards in scan_edges / select_and_expand /
// canonicalize_child), one request is in flight at a time, and requests are served in order, so context slots and plane rows
// are assigned exactly as the single-threaded loop assigns them.
// Persistent search: a posted leaf image is handed to the evaluator the moment its plane row, both move lists and the classifier's
// verdict ("a network leaf") are there — by whichever of the three waves arrives last — instead of at the end of the collect phase:
// the evaluation of the first leaves of a batch then runs beside the descents that find the last ones.  Every writer has drained its
// write-through stores (s_waitcnt vmcnt(0)) before it arrives, and the arrivals are LDS atomics, so the item follows all its bytes.
struct PubCtx { hmq::SrvQueue* q; unsigned itemBase; unsigned* expect; };
__device__ __forceinline__ void post_arrive(WaveLds& L, const PubCtx* pc, int post, int inc) {
    if ((threadIdx.x & 63) != 0) return;
    const int slot = post & (BATCH - 1);
    const int now = atomicAdd(&L.postReady[slot], inc) + inc;
    if ((now & 3) != 2 || !(now & 12)) return;
    __hip_atomic_store(&L.postReady[slot], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (now & 4) {
        const unsigned item = pc->itemBase | ((unsigned)L.postRow[slot] << 21);
        __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned*)(pc->expect), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // LDS (k_search's s_expect)
        hmq::push_items(pc->q, &item, 1);
    }
}

template <bool WT = false>
__device__ __forceinline__ void serve_leaf(G& s, const RulesTab& rt, WaveLds& L, int rootTeam, bool rootAdv, int seq, const PubCtx* pc = nullptr) {
    const int lane = threadIdx.x & 63;
    const WaveLds::Req rq = L.req[seq & 1];
    TrajEnt* trajReq = L.trajReq[seq & 1];
    const int leaf = rq.leaf, len = rq.trajLen, buf = rq.buf, slot = rq.ctxIdx;
    const bool reserved = rq.reserved != 0;
    const int valid = rq.first ? 0 : L.svcValid;
    PROF_T(tsv);
    Path p;
    if (rq.create) {
        // The child's position = the parent's cached position + the edge's joint move (Board::make_moves), then its hash and the
        // transposition lookup (canonicalize_child, searchthread.cc:741-806) — on this wave, while the traversal selects again.
        // The traversal assumed the common outcome: the child stays itself, fresh and reserved.  Anything else (the edge now
        // points to a known node, a pending evaluation, an exhausted pool) goes back to it: it undoes what it did on the
        // assumption and continues this descent itself (collect_batch).
        PROF_T(tcr);
        int next = leaf, pend = -1;
        bool childReserved = reserved;
        int cr = 3;
        p.len = len - 1;                                       // the path down to the parent (canonicalize_child's ancestor test)
        p.posNode = -1;
        if (position_child(s, rt, p, rq.parent, leaf, rq.ma, rq.mb))
            cr = canonicalize_child(s, rt, p, trajReq, rq.parent, rq.idx, next, childReserved, rootAdv, rootTeam, &pend, s.ldsTree);
        const bool fast = cr == 0 && next == leaf && childReserved == reserved;
        if (lane == 0) { L.createRes.fast = fast ? 1 : 0; L.createRes.cr = cr; L.createRes.next = next; L.createRes.reserved = childReserved ? 1 : 0; }
        // what the traversal reads once it knows the outcome: the result word and — only when the descent goes back to it — the
        // edge and the nodes canonicalize_child changed (global memory unless the tree is the LDS mirror)
        if (fast && s.ldsTree) {
            lds_release();
            if (lane == 0) __hip_atomic_store(&L.createSeq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            wave_fence();
            if (lane == 0) __hip_atomic_store(&L.createSeq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        PROF_ADD_T(47, tcr, 64);
        if (!fast) {                                           // nothing is classified: the request ends here
            if (lane == 0) L.reqResult = 3;
            wave_fence();
            if (lane == 0) __hip_atomic_store(&L.typeSeq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
    } else {
        path_load(s, p, leaf);                                 // the leaf's joint position (cached at first reach)
    }
    Ctx ctx;
    ctx.leaf = leaf; ctx.trajLen = len; ctx.reserved = reserved;
    ctx.team = s.nodes[leaf].team;
    ctx.sit = ((ctx.team == rootTeam) == rootAdv) ? 1 : 0;
    ctx.terminal = 0; ctx.termValue = 0.0f; ctx.leafHash = 0;
    const int searchPly = len - 1;
    const float drawValue = ctx.team == rootTeam ? -s.prm->drawContempt : s.prm->drawContempt;
    path_rebuild_history(s, trajReq, len);                   // the repetition keys of its path
    PROF_ADD_T(15, tsv, 64);
    // The plane-writer wave starts on this leaf now, into row `valid`, while the terminal test below runs: a leaf that turns out
    // terminal (or is dropped) simply leaves `valid` where it is and the next network leaf overwrites the row (posts are served
    // in order).  Everything the planes and the move lists need is known here.
    int myPost = -1;
    if (reserved) {
        const int np_ = L.postCount;
        myPost = np_;
        HB(22);
        while (np_ - min(__hip_atomic_load(&L.servedCnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP),
                         __hip_atomic_load(&L.servedCntB, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= BATCH) __builtin_amdgcn_s_sleep(1);   // ring slot free
        hm_board* hb = reinterpret_cast<hm_board*>(L.board[np_ & (BATCH - 1)]);
        const int rcA = repetition_count(hist_of(p.jb, 0)), rcB = repetition_count(hist_of(p.jb, 1));
        store_pos(&hb->pos[0], p.jb.bd[0]);                   // every lane writes the same words
        store_pos(&hb->pos[1], p.jb.bd[1]);
        if (lane == 0) {
            // last move per board: deepest path move on that board, else the game's last move
            u32 lm[2] = {s.g->lastMove[0], s.g->lastMove[1]};
            for (int i = 1; i < len; ++i) { if (trajReq[i].moveA) lm[0] = trajReq[i].moveA; if (trajReq[i].moveB) lm[1] = trajReq[i].moveB; }
            hb->last_move[0] = lm[0]; hb->last_move[1] = lm[1];
            hb->rep_count[0] = (uint8_t)(rcA > 3 ? 3 : rcA); hb->rep_count[1] = (uint8_t)(rcB > 3 ? 3 : rcB);
            hb->team = ctx.team; hb->time_adv = ctx.sit; hb->reserved = 0;
            L.postRow[np_ & (BATCH - 1)] = valid;
            L.postBuf = buf;
            L.postCount = np_ + 1;
        }
        lds_release();                                         // the image, its row and the counters are LDS
        if (lane == 0) __hip_atomic_store(&L.posted, np_ + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    int endInPly = 0, result = 0, newValid = valid;
    HB(23);
    const int to = classify_terminal_position(rt, p.jb, ctx.team, rootTeam, rootAdv, searchPly, &endInPly, &L.lists[0][0]);
    if (to != 0) {
        ctx.terminal = 1;
        result = 1;
        Node& ln = s.nodes[leaf];
        if (to == 1) { ctx.termValue = 1.0f; ln.type = T_WIN; ln.valueSum = 1.0f * (float)(ln.visits + 1); ln.endInPly = endInPly; }
        else if (to == 2) { ctx.termValue = -1.0f; ln.type = T_LOSS; ln.valueSum = -1.0f * (float)(ln.visits + 1); ln.endInPly = endInPly; }
        else { ctx.termValue = drawValue; ln.type = T_DRAW; ln.endInPly = 1; }
    } else if (!reserved) {
        result = 2;                                            // the traversal cancels the path's virtual losses
    }
    // the leaf's Node is final: release the traversal's tree guards before the record keeping below
    if (lane == 0) L.reqResult = result;
    if (s.ldsTree) {                                           // the leaf's node is LDS: nothing of this request that another wave reads before the batch ends is in global memory
        lds_release();
        if (lane == 0) __hip_atomic_store(&L.typeSeq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        wave_fence();
        if (lane == 0) __hip_atomic_store(&L.typeSeq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    HB(24);
    if constexpr (WT) { if (myPost >= 0) post_arrive(L, pc, myPost, result == 0 ? 4 : 8); }
    HB(25);
    PROF_ADD_T(24, tsv, 64);
    if (result == 0) {
        const bool leafAdv = ctx.team == rootTeam ? rootAdv : !rootAdv;
        ctx.leafHash = board_hash_key(p.jb.bd[0], p.jb.bd[1], hist_of(p.jb, 0), hist_of(p.jb, 1), leafAdv, rt.zob.time_adv);
        newValid = valid + 1;
    }
    if (result != 2) {
        s.ctx[buf * BATCH + slot] = ctx;
        TrajEnt* dst = s.traj + (size_t)(buf * BATCH + slot) * MAX_TRAJ;
        for (int i = lane; i < len; i += 64) dst[i] = trajReq[i];
    }
    if (lane == 0) L.svcValid = newValid;
    lds_release();                                             // (the context record and the path copy are read after the collect phase's closing barrier)
    PROF_ADD_T(23, tsv, 64);
}

// the end of a collect phase: every request served, the generator idle, the batch's counts final
__device__ __forceinline__ void collect_finish(G& s, WaveLds& L, int buf, int nctx, bool posted) {
    svc_join(s);
    gen_wait(s);
    const int valid = posted ? L.svcValid : 0;
    s.g->ctxCount[buf] = nctx;
    s.g->validCount[buf] = valid;
    s.g->evalRows += valid;
}
// tail != nullptr: return once the last request is posted and its creation step has gone the assumed way — {contexts, posted} in
// tail[0..1], collect_finish is the caller's (k_search runs the backups of the batch before while the classifier finishes this one)
__device__ __forceinline__ void collect_batch(G& s, const RulesTab& rt, WaveLds& L, int buf, int rootTeam, bool rootAdv, int* tail = nullptr) {
    const int lane = threadIdx.x & 63;
    int nctx = 0, attempts = 0;
    const int bsz = s.g->batch > 0 ? s.g->batch : BATCH;         // this slot's batch size (storage is sized for BATCH)
    bool posted = false;
    Desc d;
    bool running = false;        // d is a descent in progress (being repeated after a cancel, or continued after a hand-back)
    // the descent whose creation request is outstanding (s.jPending), as it stood before the request was posted on the
    // assumption "fresh, reserved leaf": what a hand-back has to restore
    Desc dj;
    int jSlot = 0, jNctx = 0, jMaxDepth = 0;
    bool jPosted = false;
    TRACE_SEQ();
    s.jPending = false; s.jTakenBack = false;
    for (;;) {
        int rc;
        if (s.jTakenBack) {
            // ---- the classifier handed descent j back (canonicalize_child changed the edge, met a pending evaluation, or a pool is
            // exhausted).  Undo the descent that ran ahead (virtual losses only) and j's bookkeeping, then go on exactly where
            // select_and_expand stands after canonicalize_child (searchthread.cc:846-916).
            s.jTakenBack = false;
            if (running) { cancel_virtual_losses(s, L.traj, d.len); attempts--; running = false; }
            svc_join(s);                                           // the request has ended (nothing was classified)
            s.nv -= dj.nv + (dj.widened ? 0 : 1); s.es -= dj.es;
            nctx = jNctx; posted = jPosted; s.g->maxDepth = jMaxDepth;
            for (int i = lane; i < dj.len; i += 64) L.traj[i] = L.trajReq[jSlot][i];
            wave_fence();
            d = dj;
            const int cr = L.createRes.cr, next = L.createRes.next;
            const bool childReserved = L.createRes.reserved != 0;
            if (s.ldsTree && cr != 3) {                            // canonicalize_child's edge updates, left to this wave (deferEdge)
                Node& pn = s.nodes[dj.parent];
                Edge* e = edges_of(s, pn) + dj.idx;
                if (next != dj.child) e->child = next;             // replace_child
                if (cr == 2) { e->vloss--; pn.vvsum--; }           // remove_virtual_loss
                wave_fence();
            }
            if (cr >= 2) {                                         // pending evaluation behind the transposition (its virtual loss is already removed) / pool exhausted
                L.traj[d.len - 1].childIdx = -1;                   // as select_and_expand leaves it when it gives up here
                wave_fence();
                rc = DESC_ABORT;
            }
            else {
                L.traj[d.len - 1].childIdx = d.idx;
                mark_dirty(s, next);
                L.traj[d.len] = TrajEnt{next, -1, d.ma, d.mb};
                d.len++;
                d.cur = next;
                if (d.widened && cr == 0) { d.reserved = childReserved; rc = DESC_LEAF; }
                else { d.reserved = d.widened ? false : childReserved; rc = descend(s, rt, L.traj, d, L.unavail); }
            }
        } else {
            if (!running) {
                if (!(nctx < bsz && attempts < bsz * 2)) {
                    if (resolve_create(s)) break;                  // the batch stands once the last creation went the assumed way
                    continue;
                }
                attempts++;
                HB(100 + attempts);
                desc_begin(s, L.traj, d);
                running = true;
            }
            PROF_T(t0);
            rc = descend(s, rt, L.traj, d, L.unavail);
            PROF_ADD(0, t0);
        }
        if (rc == DESC_CANCEL) continue;                           // s.jTakenBack is set: the branch above takes over
        running = false;
        s.nv += d.nv; s.es += d.es;
        if (rc == DESC_ABORT) {
            TRACE_EV(1, d.len, 0);
            s.g->reservationCollisions++;
            cancel_virtual_losses(s, L.traj, d.len);
            continue;
        }
        const bool create = rc == DESC_CREATE;
        int leaf = d.cur;
        bool reserved = d.reserved;
        if (create) {
            // assume the common outcome: the child stays itself — a fresh, reserved leaf that cannot collide with this batch's leaves
            dj = d; jNctx = nctx; jPosted = posted; jMaxDepth = s.g->maxDepth;
            L.traj[d.len - 1].childIdx = d.idx;
            L.traj[d.len] = TrajEnt{d.child, -1, d.ma, d.mb};
            d.len++;
            leaf = d.child; reserved = d.childReserved;
            if (!d.widened) s.nv++;                                // the visit of the leaf itself (the walk's next step)
            if (!reserved) { if (lane == 0) atomicOr(&s.g->overflow, 256); }   // cannot happen: a child without a position is unexpanded and unsolved, hence reserved by its selection
        } else {
            bool collision = false;
            for (int i = 0; i < nctx; ++i) collision |= L.batchLeaf[i] == leaf;
            if (collision) {
                TRACE_EV(2, d.len, 0);
                s.g->sameBatchCollisions++;
                if (leaf == s.inflight) svc_wait(s);
                if (reserved) s.nodes[leaf].flags &= ~F_PENDING;
                cancel_virtual_losses(s, L.traj, d.len);
                continue;
            }
        }
        const int searchPly = d.len - 1;
        if (searchPly > s.g->maxDepth) s.g->maxDepth = searchPly;
        if (leaf == s.inflight) svc_wait(s);
        const uint8_t solved = create ? (uint8_t)T_UNSOLVED : s.nodes[leaf].type;
        bool keep = true;
        if (solved != T_UNSOLVED) {
            TRACE_EV(3, d.len, solved);
            Ctx ctx;
            ctx.leaf = leaf; ctx.trajLen = d.len; ctx.reserved = reserved;
            ctx.team = s.nodes[leaf].team;
            ctx.sit = ((ctx.team == rootTeam) == rootAdv) ? 1 : 0;
            ctx.terminal = 1; ctx.leafHash = 0;
            const float drawValue = ctx.team == rootTeam ? -s.prm->drawContempt : s.prm->drawContempt;
            ctx.termValue = solved == T_WIN ? 1.0f : solved == T_LOSS ? -1.0f : drawValue;
            s.ctx[buf * BATCH + nctx] = ctx;
            TrajEnt* dst = s.traj + (size_t)(buf * BATCH + nctx) * MAX_TRAJ;
            for (int i = lane; i < d.len; i += 64) dst[i] = L.traj[i];
            wave_fence();
        } else {
            PROF_T(tk);
            svc_wait(s);                                           // one terminal test in flight
            svc_join(s, 1);                                        // and the request slot about to be reused is free
            const int rslot = (s.reqSeq + 1) & 1;
            for (int i = lane; i < d.len; i += 64) L.trajReq[rslot][i] = L.traj[i];
            if (lane == 0) L.req[rslot] = WaveLds::Req{leaf, d.len, nctx, buf, reserved ? 1 : 0, posted ? 0 : 1, create ? 1 : 0, d.parent, d.idx, d.ma, d.mb};
            s.reqSeq++;
            if (s.ldsTree) {                                       // the classifier reads LDS only of what this wave wrote in this phase (request, path, nodes;
                lds_release();                                     // the edges are this wave's alone: canonicalize_child's deferEdge): no drain of the edge stores
                if (lane == 0) __hip_atomic_store(&L.reqSeq, s.reqSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                wave_fence();
                if (lane == 0) __hip_atomic_store(&L.reqSeq, s.reqSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            s.inflight = leaf; s.svcBusy = true;
            posted = true;
            if (create) { s.jPending = true; jSlot = rslot; }
            if (!reserved) {                                       // rare: whether the context is kept depends on the terminal test
                svc_wait(s);
                if (s.jPending && !resolve_create(s)) continue;    // (not reachable: a creation request is always reserved)
                if (L.reqResult == 2) {
                    TRACE_EV(5, d.len, 0);
                    s.g->reservationCollisions++;
                    cancel_virtual_losses(s, L.traj, d.len);
                    keep = false;
                }
            }
            PROF_ADD(6, tk);
        }
        if (keep) {
            if (lane == 0) L.batchLeaf[nctx] = leaf;
            wave_sync();                                           // read by this wave's collision test; by others after the phase's barrier
            nctx++;
        }
    }
    if (tail) { tail[0] = nctx; tail[1] = posted ? 1 : 0; return; }
    collect_finish(s, L, buf, nctx, posted);
}

// =======================================================================================
// kernels (one workgroup per game)
// =======================================================================================
// One lockstep iteration, collect side.  Only the NEXT plane tensor is written, so this is
// independent of the network launch that reads CUR and the two overlap on separate streams
// (the reference overlaps collect_batch with the in-flight TensorRT batch the same way,
// searchthread.cc:680-688).  A game with no batch in flight collects its first batch into NEXT and
// marks it `fresh`; its lookahead follows one iteration later (same order of tree operations as
// run_iteration: collect b0, collect b1, process b0).  Runs on one wave.
__device__ __forceinline__ int collect_step(G& s, const RulesTab& rt, WaveLds& L, uint16_t* planesNext, int g) {   // returns the plane rows written
    if (s.g->status != ST_SEARCHING) return 0;
    const int rootTeam = s.g->team;
    const bool rootAdv = s.g->adv != 0;
    (void)planesNext; (void)g;
    // worker loop (agent.cc:331-341) + run_iteration head (searchthread.cc:661-678).  One collect_batch call
    // site: with nothing in flight the first batch goes to buffer 0, otherwise the lookahead to the other one.
    for (;;) {
        if (s.g->nodesSearched >= s.g->targetNodes || s.nodes[s.g->root].type != T_UNSOLVED || s.g->overflow) { s.g->status = ST_FINISHING; return 0; }
        const bool first = s.g->pending < 0;
        const int buf = first ? 0 : 1 - s.g->pending;
        collect_batch(s, rt, L, buf, rootTeam, rootAdv);
        if (!first) return s.g->validCount[buf];
        if (s.g->ctxCount[0] == 0) { s.g->overflow |= 16; s.g->status = ST_FINISHING; return 0; }   // no progress possible
        if (s.g->validCount[0] == 0) { process_batch(s, rt, L.exp, 0, rootTeam, rootAdv, nullptr, 0); continue; }
        s.g->pending = 0;
        s.g->fresh = 1;
        return s.g->validCount[0];
    }
}

// One lockstep iteration, process side, for a block of BATCH waves.  Returns on every wave; only
// wave 0 executes the ordered tail.  `activeCount` receives +1 for a game that is still searching.
__device__ inline void process_step(G& s, const RulesTab& rt, ExpLds* exp, const NetOut& out, int g, int* activeCount) {
    const int wave = threadIdx.x >> 6;
    const int st = s.g->status;
    if (st != ST_SEARCHING && st != ST_FINISHING) return;      // uniform across the block
    const int rootTeam = s.g->team;
    const bool rootAdv = s.g->adv != 0;
    const int rowBase = g * BATCH;
    if (st == ST_SEARCHING && s.g->fresh) {                    // first batch of this game is still waiting for its evaluation
        __syncthreads();
        if (threadIdx.x == 0) { s.g->fresh = 0; atomicAdd(activeCount, 1); }
        __syncthreads();
        return;
    }
    const int pending = s.g->pending;
    const int nctx = pending >= 0 ? s.g->ctxCount[pending] : 0;
    const bool solvedOrOverflow = s.nodes[s.g->root].type != T_UNSOLVED || s.g->overflow;
    const bool doProcess = pending >= 0 && !(st == ST_FINISHING && solvedOrOverflow);
    __syncthreads();                                           // every wave holds the batch header before wave 0 retires it
    // Expansions (waves 1..8, one leaf each) and the ordered backups (wave 0) touch disjoint state — a pending leaf
    // has no children, so no path of this batch runs through one, and the two sides write different fields of the
    // leaf nodes — and run side by side.
    PROF_T(te);
    if (wave == 0) {
        if (pending >= 0) {
            PROF_T(tb);
            if (st == ST_FINISHING && solvedOrOverflow) abort_batch(s, pending);   // discard_pending_iteration (agent.cc:343-352)
            else backup_batch(s, pending, &out, rowBase);
            PROF_ADD(14, tb);
        }
    } else if (doProcess && wave - 1 < nctx) {
        expand_context(s, rt, exp[wave - 1], pending, wave - 1, rootTeam, rootAdv, &out, rowBase);
    }
    __threadfence_block();
    __syncthreads();
    PROF_ADD(11, te);
    if (wave == 0) {
        // run_iteration / finish_pending tail
        if (st == ST_FINISHING) {
            if (pending >= 0) s.g->pending = -1;
            s.g->status = s.g->overflow ? ST_ERROR : ST_DONE;
        } else if (pending < 0) {
            if ((threadIdx.x & 63) == 0) atomicAdd(activeCount, 1);
        } else {
            const int look = 1 - pending;
            s.g->pending = -1;
            if (s.g->validCount[look] == 0) process_batch(s, rt, exp[0], look, rootTeam, rootAdv, nullptr, 0);
            else s.g->pending = look;
            if ((threadIdx.x & 63) == 0) atomicAdd(activeCount, 1);
        }
    }
    __threadfence_block();
    __syncthreads();
    PROF_ADD(12, te);
}

// Legal move lists of network leaf `slot` (both boards, lane 0 -> A, lane 1 -> B, R/B under-promotions erased as
// utils.h:169-182) from its hm_board image: the expansion in k_process reads them instead of generating.
template <bool WT = false>
__device__ inline void leaf_move_list(const Pools& pl, const RulesTab& rt, WaveLds& L, int g, int img, int slot, int b) {
    const int lane = threadIdx.x & 63;
    const hm_board* hb = reinterpret_cast<const hm_board*>(L.board[img]);
    const int buf = L.postBuf, team = hb->team;
    const size_t base = (((size_t)g * 2 + buf) * BATCH + slot) * 2;
    P p;
    load_pos(p, &hb->pos[b]);
    const bool on = b == 0 ? (int)p.stm == team : (int)p.stm == (team ^ 1);
    int kept = 0;
    if (on) {
        u32* list = L.helperLists[b];
        const int n = gen_legal_wave(rt.att, p, list);          // wave-cooperative, reference list order
        u32* dst = pl.leafMoves + (base + b) * HM_MAX_MOVES;
        for (int c0 = 0; c0 < n; c0 += 64) {                    // order-preserving erase of the R/B under-promotions
            const int i = c0 + lane;
            const u32 m = i < n ? list[i] : 0u;
            const bool keep = i < n && !((m & (15u << 12)) == HM_MT_PROMOTION && (((m >> 16) & 63) == HM_ROOK || ((m >> 16) & 63) == HM_BISHOP));
            const u64 km = __ballot(keep);
            if (keep) {
                // bit 31 = Position::capture(m) (JointActionCandidate's sit rules need it, joint_action.h:80-105): decided here, where
                // the position is at hand, so that whoever sorts the priors needs only the list
                const u32 v = m | (is_capture(p, m) ? hmp::CAPTURE_BIT : 0u);
                u32* q = dst + kept + __popcll(km & ((1ULL << lane) - 1ULL));
                if constexpr (WT) __hip_atomic_store(hmq::G32(q), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // read by the evaluator's workgroup
                else *q = v;
            }
            kept += __popcll(km);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        const int v = kept | ((int)p.stm << 16);
        if constexpr (WT) __hip_atomic_store(hmq::G32(reinterpret_cast<const unsigned*>(&pl.leafCounts[base + b])), (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else pl.leafCounts[base + b] = v;
        atomicAdd(&L.listWords, kept);
    }
}

// The three helper waves of a game's collect phase (k_collect and the persistent k_search): classifier (wave 1), plane writer
// (wave 2), generator (wave 3).  Each returns once the traversal wave has raised svcStop and every request has been served.
template <bool WT = false>
__device__ __forceinline__ void collect_helper_role(G& s, const RulesTab& s_rt, WaveLds& L, const Pools& pl, const Game& s_game, int g, uint16_t* planesNext, int wave, const PubCtx* pc = nullptr) {
    if (wave == 1) {
        // classifier: serves the traversal's leaf requests in order; ends once the traversal has stopped and every request is served
        const int rootTeam = s_game.team;
        const bool rootAdv = s_game.adv != 0;
        int seen = 0;
        for (;;) {
            int rs = __hip_atomic_load(&L.reqSeq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (rs == seen) {
                if (!__hip_atomic_load(&L.svcStop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) { __builtin_amdgcn_s_sleep(1); continue; }
                rs = __hip_atomic_load(&L.reqSeq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (rs == seen) break;
            }
            seen++;                                                // requests are served one by one, in order
            HB(21);
            serve_leaf<WT>(s, s_rt, L, rootTeam, rootAdv, seen, pc);
            HB(26);
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(&L.ackSeq, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // behind serve_leaf's lds_release / fences
        }
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(&L.done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        HB(29);
    } else if (wave == 3) {
        // generator wave: the two successor pushes of every pair the traversal pops (joint_action.h:312-328) and, between
        // them, the board-B legal move list of every posted leaf image (the plane-writer wave does the planes and board A)
        int seen = 0, servedB = 0;
        for (;;) {
            const int rs = __hip_atomic_load(&L.gq.reqSeq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (rs != seen) {
                const int node = L.gq.node, iA = L.gq.iA, iB = L.gq.iB;
                GenHdr* gh = reinterpret_cast<GenHdr*>(s.arena + L.gq.genOff);
                GenHdr h = *gh;
                HB(41);
                gen_push(s, h, iA + 1, iB);
                gen_push(s, h, iA, iB + 1);
                *gh = h;
                if ((threadIdx.x & 63) == 0) s.nodes[node].more = h.heapSize > 0;
                wave_fence();
                seen = rs;
                if ((threadIdx.x & 63) == 0) __hip_atomic_store(&L.gq.ackSeq, rs, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                HB(40);
                continue;
            }
            const int posted = __hip_atomic_load(&L.posted, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (servedB < posted) {
                const int img = servedB & (BATCH - 1);
                HB(42);
                leaf_move_list<WT>(pl, s_rt, L, g, img, L.postRow[img], 1);
                HB(43);
                if constexpr (WT) { hmq::drain_stores(); HB(44); post_arrive(L, pc, servedB, 1); }
                HB(45);
                servedB++;
                if ((threadIdx.x & 63) == 0) __hip_atomic_store(&L.servedCntB, servedB, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                continue;
            }
            // nothing to do: leave once the traversal has stopped (no more refills) and the classifier is done (no more images)
            if (__hip_atomic_load(&L.svcStop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) && __hip_atomic_load(&L.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
                && __hip_atomic_load(&L.gq.reqSeq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == seen
                && __hip_atomic_load(&L.posted, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == servedB) break;
            __builtin_amdgcn_s_sleep(1);
        }
        HB(49);
    } else {
        uint16_t* dst = planesNext + (size_t)g * BATCH * HM_PLANE_VALUES;
        int served = 0;
        for (;;) {                                                 // ends once the classifier has set `done` and every post is served
            int posted = __hip_atomic_load(&L.posted, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (served >= posted) {
                if (!__hip_atomic_load(&L.done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) { __builtin_amdgcn_s_sleep(4); continue; }
                posted = __hip_atomic_load(&L.posted, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (served >= posted) break;
            }
            const int img = served & (BATCH - 1), row = L.postRow[img];
            HB(31);
            write_planes_f16<WT>(s_rt, L.board[img], reinterpret_cast<uint4*>(dst + (size_t)row * HM_PLANE_VALUES), L.pmask, L.pval);
            HB(32);
            leaf_move_list<WT>(pl, s_rt, L, g, img, row, 0);
            HB(33);
            if constexpr (WT) { hmq::drain_stores(); HB(34); post_arrive(L, pc, served, 1); }
            HB(35);
            served++;
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(&L.servedCnt, served, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        HB(39);
    }
}

// =======================================================================================
// Persistent search: ONE launch per search instead of ~50 lockstep iterations of three kernels.
// =======================================================================================
// k_search keeps a game's workgroup alive for the whole node-budget search (Agent::run_search worker loop, agent.cc:331-352 +
// SearchThread::run_iteration, searchthread.cc:661-739): the node pool and the Game record stay in LDS from the first descent to
// the last backup, and the evaluator is another persistent kernel (rise_serve, hm_net.hip) fed through a device-side queue
// (hm_queue.hpp).  Per game the sequence of tree operations is exactly the lockstep one — collect b0, collect b1, process b0,
// collect b2, process b1, ... with the finish / abort rules of finish_pending / discard_pending_iteration — so every result is
// bit-identical; what changes is who waits for whom: a game waits only for the evaluation of ITS pending batch, never for the
// slowest game of the launch, and the evaluator starts on a batch as soon as its game has written it.
// Waves: 0 traversal + ordered backups, 1 classifier, 2 plane writer, 3 generator during a collect phase; during a process
// phase waves 1..3 expand the batch's leaves (leaf i on wave 1 + i % 3) beside wave 0's backups.
struct SearchIo {
    uint16_t* planes[2];                     // [nGames * BATCH][74][64] fp16 per buffer
    NetOut out[2];                           // heads of buffer 0 / 1 (rows g * BATCH + slot)
    hmq::SrvQueue* q;
    unsigned* done;                          // [nGames][2] rows evaluated per (game, buffer), cumulative over the search
    hmq::GameDiag* diag;                     // [nGames] diagnostics of a give-up (hm_queue.hpp)
    const uint8_t* netSel;                   // per game: evaluator index of its items (nullptr: 0)
    int ldsNodes;                            // the node pool fits in LDS beside the search role's other LDS
};
struct SearchCtl { int action, buf, first, ok, nctx; };   // nctx: contexts of the batch being processed (backup_batch retires the header)
enum : int { ACT_COLLECT = 0, ACT_FINISH = 1 };

__device__ __forceinline__ void expand_share(G& s, const RulesTab& rt, ExpLds& L, int wave, int pending, int nctx, int rootTeam, bool rootAdv, const NetOut* out, int rowBase, const PreSorted* pre) {
    for (int i = wave - 1; i < nctx; i += 3) expand_context(s, rt, L, pending, i, rootTeam, rootAdv, out, rowBase, pre);
}


// ---- LDS of the single-launch search (hm_rollout.hip), carved from the kernel's dynamic LDS: the evaluator role overlays the same
// bytes with its own layout.  LDS_TREE: the node pool is the LDS mirror behind the struct — a template parameter, not a launch-time
// flag, so that `s.nodes` is known to be an LDS address in that instantiation: every node access compiles to a ds_ instruction
// instead of a flat_ one (a flat access waits on BOTH memory counters, i.e. also for the wave's outstanding global stores — the
// virtual-loss store of the level above).
constexpr int ROLE_TABN = 512;
template <bool LDS_TREE>
struct SearchLds {
    RulesTab rt;
    WaveLds L;
    int nextLeaf;                                                   // process step: next context to expand (the four waves draw from it)
    __attribute__((aligned(16))) Game game;
    float cpuct[ROLE_TABN];
    uint16_t pwRoot[ROLE_TABN], pwNode[ROLE_TABN];
    SearchCtl ctl;
    unsigned expect[2];                                             // rows published per buffer so far
    PubCtx pub;
    u64 hist[LDS_TREE ? 2 : 1][LDS_TREE ? SEARCH_HIST_LDS : 1];     // G::hist of this game
};
constexpr int MG_MAX = 8;
enum : int { MG_READY = 0, MG_WAIT_PROC = 1, MG_WAIT_FIN = 2, MG_DONE = 3 };
struct MgSlot { int game, phase, pending; unsigned expect[2]; u64 since; };   // pending: the buffer whose evaluation the game waits for
struct SearchLdsMg {
    RulesTab rt;
    WaveLds L;
    __attribute__((aligned(16))) Game game;
    float cpuct[ROLE_TABN];
    uint16_t pwRoot[ROLE_TABN], pwNode[ROLE_TABN];
    SearchCtl ctl;
    PubCtx pub;
    MgSlot slot[MG_MAX];
    int alive, moved, abort;
};
// bytes of the search role's fixed LDS by mode (0: node pool in LDS — the pool follows —, 1: tree walked in place, 2: several games per workgroup)
__host__ __device__ constexpr size_t search_lds_bytes(int mode) {
    return ((mode == 0 ? sizeof(SearchLds<true>) : mode == 1 ? sizeof(SearchLds<false>) : sizeof(SearchLdsMg)) + 15) & ~(size_t)15;
}
// what the evaluator role needs of the network (hm_net_serve_info), plus where its item word sits in LDS
struct RolloutNet { const void* nd; const void* wh; const void* wf; int copMax, uHalfs; unsigned itemOff; };

}  // namespace hms
