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
#include <hip/hip_runtime.h>
#include <atomic>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

__device__ unsigned long long g_trace[65536];   // hm_prof.hpp TRACE_EV log (diagnostic builds only)
__device__ unsigned int g_traceCount, g_traceSeq;
__device__ int g_traceGame = -1;
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
__device__ unsigned int* g_hb;                    // diagnostics of a persistent-search give-up: last heartbeat code per (game slot, wave); nullptr outside k_search
#ifdef HM_SEARCH_HB
#define HB(c) do { unsigned int* hb_ = g_hb; if ((threadIdx.x & 63) == 0 && hb_) __hip_atomic_store(hmq::G32(&hb_[blockIdx.x * 4 + (threadIdx.x >> 6)]), (unsigned)(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define HB(c) do { } while (0)
#endif
__device__ unsigned long long g_prof[128];     // hm_prof.hpp probes: 64 sums + 64 call counts (all zero in the product build)
#ifdef HM_SEARCH_PROF
constexpr int PROF_LAUNCHES = 8192;
__device__ unsigned int g_colDur[PROF_LAUNCHES][64];   // traversal cycles of wave 0 per (k_collect launch, game slot < 64): straggler analysis
__device__ unsigned int g_colLaunch;                    // launch counter, bumped by k_process
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
                if (!(nctx < BATCH && attempts < BATCH * 2)) {
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
    unsigned* progress;                      // [2][nGames] diagnostics of a give-up: phase | iteration << 8, and its time stamp (10 us units)
    const uint8_t* netSel;                   // per game: evaluator index of its items (nullptr: 0)
    int ldsNodes;                            // the node pool fits in LDS beside k_search's static LDS
};
struct SearchCtl { int action, buf, first, ok, nctx; };   // nctx: contexts of the batch being processed (backup_batch retires the header)
enum : int { ACT_COLLECT = 0, ACT_FINISH = 1 };

__device__ __forceinline__ void expand_share(G& s, const RulesTab& rt, ExpLds& L, int wave, int pending, int nctx, int rootTeam, bool rootAdv, const NetOut* out, int rowBase, const PreSorted* pre) {
    for (int i = wave - 1; i < nctx; i += 3) expand_context(s, rt, L, pending, i, rootTeam, rootAdv, out, rowBase, pre);
}

// LDS_TREE: the node pool is the LDS mirror — a template parameter, not a launch-time flag, so that `s.nodes` is known to be an LDS
// address in that instantiation: every node access compiles to a ds_ instruction instead of a flat_ one (a flat access waits on BOTH
// memory counters, i.e. also for the wave's outstanding global stores — the virtual-loss store of the level above).
template <bool LDS_TREE>
__global__ __launch_bounds__(COLLECT_THREADS) void k_search(Pools pl, Params prm, SearchIo io) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_nodes[];
    __shared__ RulesTab s_rt;
    __shared__ WaveLds L;
    __shared__ ExpLds s_exp2[3];                                    // expansion scratch of waves 2, 3 and 0 (wave 1 uses L.exp)
    __shared__ int s_nextLeaf;                                      // process step: next context to expand (the four waves draw from it)
    __shared__ __attribute__((aligned(16))) Game s_game;
    constexpr int TABN = 512;
    __shared__ float s_cpuct[TABN];
    __shared__ uint16_t s_pwRoot[TABN], s_pwNode[TABN];
    __shared__ SearchCtl s_ctl;
    __shared__ unsigned s_expect[2];                                // rows published per buffer so far
    __shared__ PubCtx s_pub;
    __shared__ u64 s_hist[LDS_TREE ? 2 : 1][LDS_TREE ? SEARCH_HIST_LDS : 1];                             // G::hist of this game (see below)
    const int g = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    PROF_INIT();
    G s = make_view(pl, prm, g);
    Game* const gGame = s.g;
    Node* const gNodes = s.nodes;
    for (unsigned i = threadIdx.x; i < sizeof(Game) / 4; i += COLLECT_THREADS) reinterpret_cast<u32*>(&s_game)[i] = reinterpret_cast<const u32*>(gGame)[i];
    stage_table_wide(&s_rt, pl.rules);
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
    ExpLds& myExp = wave == 1 ? L.exp : s_exp2[wave == 0 ? 2 : wave - 2];
    const PreSorted pre{pl.sortedMoves + (size_t)g * 2 * BATCH * 2 * HM_MAX_MOVES, pl.sortedPriors + (size_t)g * 2 * BATCH * 2 * HM_MAX_MOVES};
    u64 tC = 0, tW = 0, tP = 0, nIt = 0;                            // thread 0: ticks spent collecting / waiting for the evaluator / processing
    unsigned xcc;                                                   // which XCD this workgroup runs on (diagnostics of a give-up)
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    auto mark = [&](unsigned phase) {                               // thread 0 only
        __hip_atomic_store(hmq::G32(&io.progress[g]), phase | (xcc << 4) | ((unsigned)nIt << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(hmq::G32(&io.progress[gridDim.x + g]), (unsigned)((u64)__builtin_amdgcn_s_memrealtime() / 1000ULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // process of the pending batch: wait for its evaluation (unless `abortIt`), backups on wave 0 beside the expansions on waves 1..3
    auto process_pending = [&](bool abortIt) -> bool {
        const int pending = s_game.pending;
        u64 t0 = 0;
        if (threadIdx.x == 0) {
            t0 = __builtin_amdgcn_s_memrealtime();
            bool ok = true;
            mark(3u);
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
            if (ok) mark(4u); else io.progress[g] = 5u | (xcc << 4) | ((unsigned)nIt << 8);     // 5: the wait failed (the stamp stays at its start)
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
                    mark(3u);
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
                    if (ok) mark(4u); else io.progress[g] = 5u | (xcc << 4) | ((unsigned)nIt << 8);
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
        io.progress[2 * gridDim.x + g] = (unsigned)((u64)__builtin_amdgcn_s_memrealtime() / 1000ULL);   // when it left
        io.progress[3 * gridDim.x + g] = s_expect[0] + s_expect[1] + 1u;                              // rows it published (+ 1), over the heartbeat words of wave 0..3 of slot g / 4: diagnostics only
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
constexpr int MG_MAX = 8;
enum : int { MG_READY = 0, MG_WAIT_PROC = 1, MG_WAIT_FIN = 2, MG_DONE = 3 };
struct MgSlot { int game, phase, pending; unsigned expect[2]; u64 since; };   // pending: the buffer whose evaluation the game waits for
__global__ __launch_bounds__(COLLECT_THREADS) void k_search_mg(Pools pl, Params prm, SearchIo io, int nGames, int perWg) {
    __shared__ RulesTab s_rt;
    __shared__ WaveLds L;
    __shared__ ExpLds s_exp2[2];                                    // expansion scratch of waves 2 and 3 (wave 1 uses L.exp)
    __shared__ __attribute__((aligned(16))) Game s_game;
    constexpr int TABN = 512;
    __shared__ float s_cpuct[TABN];
    __shared__ uint16_t s_pwRoot[TABN], s_pwNode[TABN];
    __shared__ SearchCtl s_ctl;
    __shared__ PubCtx s_pub;
    __shared__ MgSlot s_slot[MG_MAX];
    __shared__ int s_alive, s_moved, s_abort;
    const int w = blockIdx.x, W = gridDim.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    stage_table_wide(&s_rt, pl.rules);
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
    ExpLds& myExp = wave <= 1 ? L.exp : s_exp2[wave - 2];
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
            auto mark = [&](unsigned ph) {
                __hip_atomic_store(hmq::G32(&io.progress[g]), ph | (xcc << 4) | ((unsigned)nIt << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(hmq::G32(&io.progress[nGames + g]), (unsigned)((u64)__builtin_amdgcn_s_memrealtime() / 1000ULL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            // process step of the pending batch (its evaluation is there, or it is discarded): backups on wave 0 beside the expansions
            auto process_now = [&](bool abortIt) {
                u64 t0 = 0;
                if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memrealtime(); mark(4u); }
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
                        else if (threadIdx.x == 0) { s_slot[j].phase = MG_WAIT_FIN; s_slot[j].pending = s_game.pending; s_slot[j].since = __builtin_amdgcn_s_memrealtime(); mark(3u); }
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
                        s_slot[j].phase = MG_WAIT_PROC; s_slot[j].pending = s_game.pending; s_slot[j].since = __builtin_amdgcn_s_memrealtime(); mark(3u);
                    }
                }
            }
            __syncthreads();
            if (finished && threadIdx.x == 0) {
                s_game.pending = -1;
                s_game.status = s_game.overflow ? ST_ERROR : ST_DONE;
                s_slot[j].phase = MG_DONE;
                io.progress[2 * nGames + g] = (unsigned)((u64)__builtin_amdgcn_s_memrealtime() / 1000ULL);   // when it left
                io.progress[3 * nGames + g] = s_slot[j].expect[0] + s_slot[j].expect[1] + 1u;               // rows it published (+ 1)
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

// Holds the evaluator's stream until every search workgroup has started (k_search is launched first, on the other stream): the
// games then own their CUs before the evaluator kernel takes ALL the others.  Evaluator workgroups that find no CU simply wait in the
// dispatcher until games end (each gets its poison item then); the reverse order could leave a game workgroup waiting for a CU that
// only evaluator workgroups waiting for that game could vacate.
__global__ void k_wait_trees(hmq::SrvQueue* q, unsigned trees) {
    if (threadIdx.x != 0) return;
    const hmq::u64q t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(hmq::G32(&q->treesIn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= trees) return;
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 255u) == 255u) {
            if (__hip_atomic_load(hmq::G32(&q->error), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            if ((hmq::u64q)__builtin_amdgcn_s_memrealtime() - t0 > hmq::MEET_LIMIT_TICKS) { __hip_atomic_store(hmq::G32(&q->error), 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        }
    }
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
    int searchLdsNodes = 0;                // the node pool fits beside k_search's static LDS
    int numCUs = 0;
    hipEvent_t evFork = nullptr, evJoin = nullptr, evT0 = nullptr, evT1 = nullptr;
    hipStream_t sTree = nullptr, sNet = nullptr;   // queues of their own for the two persistent kernels
    unsigned lastQueueError = 0;           // SrvQueue::error of the last hm_sp_search (4: the two kernels did not run together)
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
        sp->queueBytes = sizeof(hmq::SrvQueue) + ((G_ * 16 * sizeof(unsigned) + 15) & ~(size_t)15);   // + done[G][2], progress[G], stamp[G]
        rc |= dalloc(sp, &sp->d_queue, sp->queueBytes);
        hipFuncAttributes fa;
        size_t staticLds = 64 * 1024;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_search<true>)) == hipSuccess) staticLds = fa.sharedSizeBytes;
        else (void)hipGetLastError();
        const size_t room = staticLds < 160 * 1024 ? 160 * 1024 - staticLds : 0;
        sp->searchLdsNodes = ((size_t)p.nodeCap * sizeof(Node) <= room && p.histCap <= SEARCH_HIST_LDS && !std::getenv("HM_SEARCH_NO_LDS_NODES")) ? 1 : 0;
        if (sp->searchLdsNodes && hipFuncSetAttribute(reinterpret_cast<const void*>(k_search<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)room) != hipSuccess) {
            (void)hipGetLastError();
            sp->searchLdsNodes = 0;
        }
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
    if (sp->sNet) (void)hipStreamDestroy(sp->sNet);
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
// Only valid with tree reuse off and before any other hm_sp_* call that uploads moves or masks.
int hm_sp_begin_again(hm_sp* sp) {
    if (!sp) return hm_fail(HM_ERR_INVALID, "null argument");
    hipLaunchKernelGGL(k_begin, dim3(sp->nGames), dim3(64), 0, 0, sp->pl, sp->prm, sp->d_target, sp->d_seed, sp->alpha, sp->eps, sp->lastBeginMasked ? sp->d_mask : nullptr, sp->d_rootHash, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
}
// 1 when the last hm_sp_search failed because its two kernels were run one after the other (a serialising profiler, or a
// runtime that put both streams on one hardware queue): the lockstep calls still work.
int hm_sp_search_not_concurrent(const hm_sp* sp) { return sp && sp->lastQueueError == 4u ? 1 : 0; }
// 1 when the last hm_sp_search was given up because the evaluator had nothing to do for 30 ms while games were still searching
// (hm_queue.hpp: IDLE_LIMIT_TICKS): hm_sp_begin_again + another search (persistent or lockstep) repeats it with the same result.
int hm_sp_search_stalled(const hm_sp* sp) { return sp && sp->lastQueueError == 5u ? 1 : 0; }
// 1 when k_search keeps this engine's node pool in LDS for a whole search (it fits beside the kernel's static LDS), 0 when the tree is walked in place
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

// Evaluator workgroups a persistent search of this engine can run beside its game workgroups: every workgroup of either kind may
// need a whole CU (the search kernel takes the full register file, the evaluator most of the LDS), so all of them are resident —
// which the queue protocol relies on only for speed, the spins being bounded — when their number does not exceed the CU count.
int hm_sp_search_consumers(const hm_sp* sp) {
    if (!sp || sp->numCUs <= 0) return 0;
    // every CU the game workgroups leave: a workgroup of either kernel takes a whole CU (the search kernel the full register file,
    // the evaluator most of it), the games are placed first (k_wait_trees), and an evaluator workgroup that finds no CU only waits
    const int slots = sp->lastBeginActive > 0 ? sp->lastBeginActive : sp->nGames;
    const int act = std::min(slots, sp->nGames);
    // no more of them than rows can be in flight (two batches of BATCH rows per searching game, and a few to spare)
    const int n = std::min(sp->numCUs - act, 2 * BATCH * act + 8);
    return (sp->numCUs - sp->nGames) >= 8 && n >= 8 ? n : 0;
}

// The whole node-budget search of every slot hm_sp_begin_search left in the searching state, with the native evaluator: two
// persistent kernels (k_search: one workgroup per game; rise_serve: the evaluator workgroups) joined by the device-side queue of
// hm_queue.hpp.  Replaces the host loop collect -> forward -> process of Agent::run_search's workers (agent.cc:331-352,
// searchthread.cc:661-739); per game the order of tree operations, hence every result, is the same.  Synchronous.
int hm_sp_search(hm_sp* sp, const hm_net* net, const hm_eval_io* io, double* search_kernel_ms) {
    if (!sp || !net || !io) return hm_fail(HM_ERR_INVALID, "null argument");
    if (!io->planes[0] || !io->planes[1] || !io->value || !io->pi_a || !io->pi_b || !io->wdl || !io->moves_left
        || !io->value_2 || !io->pi_a_2 || !io->pi_b_2 || !io->wdl_2 || !io->moves_left_2) return hm_fail(HM_ERR_INVALID, "hm_sp_search needs both plane buffers and both sets of heads");
    if (!hm_net_can_serve(net)) return hm_fail(HM_ERR_INVALID, "this network has no persistent evaluator kernel");
    int consumers = hm_sp_search_consumers(sp);
    if (consumers <= 0) return hm_fail(HM_ERR_INVALID, "too many game slots for a persistent search on this device (use the lockstep calls)");
    // A slow evaluator (the 384-channel deployed network: 0.70 ms per position against ~0.11 ms of tree work per batch) is the limit of
    // the whole search: one search workgroup then serves several games in turn (k_search_mg) and the CUs it frees go to the evaluator.
    // HM_SEARCH_GAMES_PER_WG=k overrides (1 = one workgroup per game).
    int perWg = hm_net_serve_is_slow(net) ? 3 : 1;             // measured at configs[3]: 518 / 546 / 550 / 529 positions/s with 1 / 2 / 3 / 4 games per workgroup
    if (const char* e = std::getenv("HM_SEARCH_GAMES_PER_WG")) perWg = std::max(1, std::min(MG_MAX, std::atoi(e)));
    const int searchWgs = perWg > 1 ? (sp->nGames + perWg - 1) / perWg : sp->nGames;
    if (perWg > 1) {
        const int act = std::min(sp->lastBeginActive > 0 ? sp->lastBeginActive : sp->nGames, sp->nGames);
        consumers = std::max(8, std::min(sp->numCUs - searchWgs, 2 * BATCH * act + 8));
    }
    // The two kernels must RUN TOGETHER, so they need two hardware queues: ordinary HIP streams are multiplexed onto a small pool of
    // queues and two of them may share one (the second kernel would then wait for the first to end — which waits for the second).
    // Streams created with a CU mask own their queue; the mask enables every CU.
    if (!sp->sTree) {
        const char* plain = std::getenv("HM_SEARCH_PLAIN_STREAMS");
        if (plain && plain[0] != '0') {
            // experiment (the stall): ordinary streams of different priority instead of CU-masked ones — streams of
            // different priorities never share a hardware queue
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            if (hipStreamCreateWithPriority(&sp->sTree, hipStreamDefault, hi) != hipSuccess || hipStreamCreateWithPriority(&sp->sNet, hipStreamDefault, lo) != hipSuccess) {
                (void)hipGetLastError();
                return hm_fail(HM_ERR_NO_DEVICE, "hipStreamCreateWithPriority failed");
            }
        } else {
        const int words = (sp->numCUs + 31) / 32;
        std::vector<uint32_t> all((size_t)words, 0xffffffffu);
        if (sp->numCUs % 32) all[(size_t)words - 1] = (1u << (sp->numCUs % 32)) - 1u;
        if (hipExtStreamCreateWithCUMask(&sp->sTree, (uint32_t)words, all.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&sp->sNet, (uint32_t)words, all.data()) != hipSuccess) {
            (void)hipGetLastError();
            return hm_fail(HM_ERR_NO_DEVICE, "hipExtStreamCreateWithCUMask failed (persistent search needs two hardware queues)");
        }
        }
    }
    hipStream_t sT = sp->sTree, sN = sp->sNet;
    HIPCHK(hipDeviceSynchronize());                    // planes / heads / pools may have been touched on other streams
    if (!sp->evFork) {
        HIPCHK(hipEventCreateWithFlags(&sp->evFork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sp->evJoin, hipEventDisableTiming));
        HIPCHK(hipEventCreate(&sp->evT0));
        HIPCHK(hipEventCreate(&sp->evT1));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&sp->h_qinit), 64, hipHostMallocDefault));
    }
    hmq::SrvQueue* q = reinterpret_cast<hmq::SrvQueue*>(sp->d_queue);
    unsigned* done = reinterpret_cast<unsigned*>(sp->d_queue + sizeof(hmq::SrvQueue));
    // every polled word starts from zero; then the two head counts (producers .. consumers are adjacent words)
    HIPCHK(hipMemsetAsync(sp->d_queue, 0, sp->queueBytes, sT));
    sp->h_qinit[0] = (unsigned)searchWgs; sp->h_qinit[1] = 0u; sp->h_qinit[2] = (unsigned)consumers; sp->h_qinit[3] = (unsigned)sp->nGames;
#ifdef HM_SEARCH_HB
    {
        unsigned* hbp = done + (size_t)sp->nGames * 5;
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_hb), &hbp, sizeof(hbp)));
    }
#endif
    static_assert(offsetof(hmq::SrvQueue, error) == offsetof(hmq::SrvQueue, producers) + 4 && offsetof(hmq::SrvQueue, consumers) == offsetof(hmq::SrvQueue, producers) + 8 && offsetof(hmq::SrvQueue, games) == offsetof(hmq::SrvQueue, producers) + 12, "queue header layout");
    HIPCHK(hipMemcpyAsync(&q->producers, sp->h_qinit, 16, hipMemcpyHostToDevice, sT));
    HIPCHK(hipEventRecord(sp->evFork, sT));
    HIPCHK(hipStreamWaitEvent(sN, sp->evFork, 0));
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
    // the games first: they take their CUs and start collecting; the evaluator's stream waits for all of them to be in (k_wait_trees)
    SearchIo sio;
    sio.planes[0] = static_cast<uint16_t*>(io->planes[0]); sio.planes[1] = static_cast<uint16_t*>(io->planes[1]);
    for (int b = 0; b < 2; ++b) sio.out[b] = NetOut{a.value[b], a.piA[b], a.piB[b], a.wdl[b], a.ml[b]};
    sio.q = q; sio.done = done; sio.progress = done + (size_t)sp->nGames * 2; sio.netSel = nullptr; sio.ldsNodes = sp->searchLdsNodes;
    (void)hipEventRecord(sp->evT0, sT);            // HIP events on the stream the kernel is launched on: its launch duration
    if (perWg > 1) hipLaunchKernelGGL(k_search_mg, dim3(searchWgs), dim3(COLLECT_THREADS), 0, sT, sp->pl, sp->prm, sio, sp->nGames, perWg);
    else if (sp->searchLdsNodes) hipLaunchKernelGGL(k_search<true>, dim3(sp->nGames), dim3(COLLECT_THREADS), (size_t)sp->prm.nodeCap * sizeof(Node), sT, sp->pl, sp->prm, sio);
    else hipLaunchKernelGGL(k_search<false>, dim3(sp->nGames), dim3(COLLECT_THREADS), 0, sT, sp->pl, sp->prm, sio);
    const hipError_t le = hipGetLastError();
    (void)hipEventRecord(sp->evT1, sT);
    if (le != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, std::string("k_search launch failed: ") + hipGetErrorString(le));
    hipLaunchKernelGGL(k_wait_trees, dim3(1), dim3(64), 0, sN, q, (unsigned)searchWgs);
    int rcServe = hipGetLastError() == hipSuccess ? 0 : hm_fail(HM_ERR_NO_DEVICE, "k_wait_trees launch failed");
    if (!rcServe) rcServe = hm_net_serve(net, a, consumers, sN);
    if (rcServe) {
        // the game workgroups are waiting for an evaluator that never comes: release them
        sp->h_qinit[4] = 1u;
        (void)hipMemcpyAsync(&q->error, &sp->h_qinit[4], 4, hipMemcpyHostToDevice, sN);
        (void)hipStreamSynchronize(sT); (void)hipStreamSynchronize(sN);
        return rcServe;
    }
    HIPCHK(hipStreamSynchronize(sT));
    HIPCHK(hipStreamSynchronize(sN));
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
    hmq::SrvQueue hq;                                              // header only (the slots follow it)
    HIPCHK(hipMemcpy(&hq, q, offsetof(hmq::SrvQueue, slots), hipMemcpyDeviceToHost));
    sp->lastQueueError = hq.error;
    if (!hq.error) {
        // test hook: HM_SEARCH_FAKE_STALL_EVERY=k reports every k-th completed persistent search as stalled, so that the caller's
        // recovery path (hm_sp_search_stalled -> hm_sp_begin_again -> the search once more) runs on demand
        const char* fe = std::getenv("HM_SEARCH_FAKE_STALL_EVERY");
        const int fakeEvery = fe ? std::atoi(fe) : 0;
        static int fakeCount = 0;
        if (fakeEvery > 0 && ++fakeCount % fakeEvery == 0) {
            sp->lastQueueError = 5u;
            return hm_fail(HM_ERR_STATE, "persistent search reported as stalled (HM_SEARCH_FAKE_STALL_EVERY)");
        }
    }
    if (hq.error) {
        // where every game that had work stood when the search was given up (phase: 1 collecting, 2 collected, 3 waiting for the evaluator, 4 processing, 5 the wait failed; since when; when it left)
        const size_t G_ = (size_t)sp->nGames;
        std::string census;
        {   // rows the games published (their own count) against the tickets the queue handed out, rows evaluated against the sum of the done counters
            std::vector<unsigned> dn(G_ * 2), pub(G_);
            unsigned long long sumDone = 0, sumPub = 0;
            if (hipMemcpy(dn.data(), done, dn.size() * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(pub.data(), done + G_ * 5, pub.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                for (unsigned v : dn) sumDone += v;
                for (unsigned v : pub) if (v) sumPub += v - 1u;
                census = "published " + std::to_string(sumPub) + " tickets " + std::to_string(hq.tail - (unsigned)consumers) + " evaluated " + std::to_string(hq.served) + " sum(done) " + std::to_string(sumDone);
            } else (void)hipGetLastError();
        }
        std::vector<unsigned> all(G_ * 14);                           // live progress[3G] + heartbeats[4G], then the snapshot of both
        std::string where;
        if (hipMemcpy(all.data(), done + G_ * 2, all.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            const bool snapped = hq.dbgPop[0] != 0;                    // an evaluator gave up: the snapshot it took before raising the error
            const unsigned* prog = all.data() + (snapped ? 7 * G_ : 0);
            const unsigned* hb = prog + 3 * G_;
            unsigned now = 0;
            if (snapped) now = prog[2 * G_];
            else for (size_t g = 0; g < G_; ++g) now = std::max(now, prog[2 * G_ + g]);
            where = snapped ? " [snapshot at the give-up]" : " [final state]";
            for (size_t g = 0; g < G_; ++g)
                if (prog[g]) where += " g" + std::to_string(g) + ":x" + std::to_string((prog[g] >> 4) & 15u) + "p" + std::to_string(prog[g] & 15u) + "/i" + std::to_string(prog[g] >> 8) + "/-" + std::to_string((int)(now - prog[G_ + g]) / 100) + "ms"
                                      + " w=" + std::to_string(hb[g * 4]) + "," + std::to_string(hb[g * 4 + 1]) + "," + std::to_string(hb[g * 4 + 2]) + "," + std::to_string(hb[g * 4 + 3]);
        } else (void)hipGetLastError();
        return hm_fail(HM_ERR_STATE, "persistent search gave up waiting (queue error " + std::to_string(hq.error) + "; head " + std::to_string(hq.head) + " tail "
                       + std::to_string(hq.tail) + " producers left " + std::to_string(hq.producers) + " served " + std::to_string(hq.served) + "; search workgroups in/out "
                       + std::to_string(hq.treesIn) + "/" + std::to_string(hq.treesOut) + " of " + std::to_string(sp->nGames) + ", evaluator workgroups in/out "
                       + std::to_string(hq.consIn) + "/" + std::to_string(hq.consOut) + " of " + std::to_string(consumers) + "; first failed wait: game "
                       + std::to_string((int)hq.dbg[0] - 1) + " buffer " + std::to_string(hq.dbg[1]) + " expected " + std::to_string(hq.dbg[2]) + " done " + std::to_string(hq.dbg[3])
                       + " iteration " + std::to_string(hq.dbg[4]) + " waited ms " + std::to_string(hq.dbg[5]) + "; evaluator that gave up: ticket "
                       + std::to_string((int)hq.dbgPop[0] - 1) + " tail " + std::to_string(hq.dbgPop[1]) + "; tickets drawn twice " + std::to_string(hq.dupTickets) + "; census " + census
                       + "; games:" + where + ")");
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

int hm_sp_apply(hm_sp* sp, const hm_move* move_a, const hm_move* move_b, const uint8_t* mask) {
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
    int* err = reinterpret_cast<int*>(hs);
    HIPCHK(hipMemcpy(err, sp->d_applyErr, sizeof(int), hipMemcpyDeviceToHost));
    if (*err) return hm_fail(HM_ERR_OVERFLOW, "game history pool full (raise max_game_plies of hm_sp_create_ex)");
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
    if (boards || flags) {
        // boards | flags are adjacent at the end of the output block: one download
        const unsigned char* from = reinterpret_cast<const unsigned char*>(sp->d_boards);
        const size_t bytes = (size_t)(reinterpret_cast<const unsigned char*>(sp->d_flags) - from) + 4 * G_;
        HIPCHK(hipMemcpy(sp->h_stage, from, bytes, hipMemcpyDeviceToHost));
        if (boards) std::memcpy(boards, sp->h_stage, sizeof(hm_board) * G_);
        if (flags) std::memcpy(flags, sp->h_stage + (reinterpret_cast<const unsigned char*>(sp->d_flags) - from), 4 * G_);
    }
    HIPCHK(hipStreamSynchronize(nullptr));
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
