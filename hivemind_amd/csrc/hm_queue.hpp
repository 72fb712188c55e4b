// hm_queue.hpp — device-side evaluation queue between the search workgroups and the evaluator workgroups of the single-launch
// search (hm_rollout.hip: k_rollout — roles by blockIdx: the first workgroups are games, alive for a whole search, the others
// evaluate one position at a time), replacing the host-driven lockstep of Engine::enqueueInferenceHalf /
// synchronizeInferenceHalf (nn/engine.h:43-81) inside SearchThread::run_iteration (searchthread.cc:661-739) for the native evaluator.
//
// Why: in the lockstep form every iteration of every game waits for the slowest game's collect and for one forward launch over all
// games; with the queue a game's chain collect(k+1) -> [eval(k) done?] -> process(k) advances at its own pace and the evaluator
// starts on a batch the moment its game has written it.  The order of tree operations per game — hence every result — is unchanged.
//
// Protocol (MI355X: 8 XCDs with private L2s, per-CU L1 never refreshed by other CUs' stores; cdna_hip_programming.md Guideline 16):
//   producer of bulk data (plane rows, or the network heads): WRITE-THROUGH stores (sc1: store16_wt / store2_wt), so no release
//   fence; EVERY storing wave drains (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane signals — an 8-byte {ticket tag, payload}
//   granule stored with a relaxed agent-scope atomic (queue slot), or an agent-scope atomic add (completion counter);
//   consumer: ONE lane polls the ONE word relaxed (s_sleep between polls), then ONE agent-scope acquire fence + drain, workgroup
//   barrier, then plain loads by every wave.
//   Every polled word is zeroed by a hipMemsetAsync ahead of the launches; tags / counters count within one search.
//   Every spin is bounded: on give-up the spinner sets SrvQueue::error and leaves, and every other spinner sees that word.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace hmq {

typedef unsigned long long u64q;
constexpr int QCAP = 8192;                   // ring slots (power of two) >> rows in flight (16 per game) + idle consumers
constexpr u64q SPIN_LIMIT_TICKS = 2000000000ULL;   // 20 s of the 100 MHz s_memrealtime counter: a hang guard, not a schedule
constexpr u64q MEET_LIMIT_TICKS = 300000000ULL;    // 3 s: by then a workgroup of the other role must have started (error 4: the roles are not resident together)
// Hang guard.  Every search workgroup bumps SrvQueue::beats at each phase change (collecting / collected / waiting / processing: four
// times per game-iteration, i.e. at least every few milliseconds even in the deepest tree).  An evaluator workgroup that has seen
// NEITHER the queue tail NOR the beat counter move for this long while search workgroups are still in declares the search stalled
// (error 5) after taking a snapshot of every game's diagnostics record; the caller repeats that search (hm_sp_search_stalled /
// hm_sp_begin_again), which changes no result.  (Round 3 watched the tail alone: a game deep in a forced line publishes no row for
// tens of milliseconds — every batch is one terminal leaf plus fifteen same-batch collisions — while it is making progress all the
// time; those alarms were false)
constexpr u64q IDLE_LIMIT_TICKS = 5000000ULL;      // 50 ms

// item payload (low 32 bits of a slot): game slot, plane buffer, row, evaluator-specific flags
constexpr unsigned IT_POISON = 0x80000000u;  // no more work: the consumer leaves
constexpr unsigned IT_ROOT = 0x02000000u;    // the leaf is the root of its search: Dirichlet noise is mixed into its priors
__host__ __device__ inline unsigned item_pack(int game, int buf, int row, int net) { return (unsigned)game | ((unsigned)buf << 20) | ((unsigned)row << 21) | ((unsigned)net << 24); }
__host__ __device__ inline int item_game(unsigned it) { return (int)(it & 0xfffffu); }
__host__ __device__ inline int item_buf(unsigned it) { return (int)((it >> 20) & 1u); }
__host__ __device__ inline int item_row(unsigned it) { return (int)((it >> 21) & 7u); }
__host__ __device__ inline int item_net(unsigned it) { return (int)((it >> 24) & 1u); }

// Per-game diagnostics record (global memory behind the queue; written by thread 0 of the game's workgroup with write-through
// stores, read by the host after a give-up and, for the snapshot, by the evaluator workgroup that gives up).
struct GameDiag {
    unsigned phase;                          // 1 collecting, 2 collected, 3 waiting for the evaluator, 4 processing, 5 the wait failed | XCC id << 4 | iteration << 8
    unsigned stamp;                          // when `phase` was written (10 us units of s_memrealtime)
    unsigned left;                           // when the game left the kernel (same unit), 0 = still in
    unsigned published;                      // rows it has published so far (both buffers)
    unsigned waitBuf;                        // 1 + the buffer whose evaluation it is waiting for (0: not waiting)
    unsigned waitExpect;                     // rows of that buffer that must be done
    unsigned hwId;                           // HW_REG_HW_ID of the traversal wave (SE / CU / SIMD it runs on)
    unsigned pad;
};
// [SrvQueue | done[2G] | diag[G] | snapDone[2G] | snapDiag[G] | snapTime (16 B) | heartbeats[4G] (diagnostic builds)], all 16-byte aligned
struct QueueLayout {
    size_t done, diag, snapDone, snapDiag, snapTime, hb, bytes;
    __host__ __device__ explicit QueueLayout(unsigned G, size_t queueStructBytes) {
        auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
        done = queueStructBytes;
        diag = up(done + (size_t)G * 8);
        snapDone = diag + (size_t)G * sizeof(GameDiag);
        snapDiag = up(snapDone + (size_t)G * 8);
        snapTime = snapDiag + (size_t)G * sizeof(GameDiag);
        hb = snapTime + 16;
        bytes = up(hb + (size_t)G * 16);
    }
};

struct SrvQueue {                            // one per search engine; zeroed (whole struct) before every search
    unsigned head;                           // next ticket a consumer takes
    unsigned tail;                           // next ticket a producer fills
    unsigned producers;                      // search workgroups still running (set by the host after the memset)
    unsigned error;                          // != 0: some spin gave up (code of the first), everybody leaves
    unsigned consumers;                      // evaluator workgroups of this launch (poison count)
    unsigned games;                          // game slots: behind the slots lie done[2G], GameDiag[G], and a snapshot of both + its time (QueueLayout)
    unsigned served;                         // items evaluated (statistics)
    unsigned treesIn, treesOut, consIn, consOut;   // census of both kernels (diagnostics of a give-up)
    unsigned dbg[6];                         // first failed wait of a search workgroup: game, buffer, rows expected, rows done, iteration, ms since the kernel's first workgroup started
    unsigned dupTickets;                     // pushes that found their slot already carrying their own ticket's tag (diagnostics: two producers drew one ticket)
    unsigned dbgPop[2];                      // the evaluator workgroup that gave up (error 2 / 5): its ticket + 1, the tail it saw then
    unsigned beats;                          // phase changes of all search workgroups (the hang guard's second progress signal)
    unsigned pad_[3];
    u64q slots[QCAP];                        // {ticket + 1, payload}
};
static_assert(sizeof(SrvQueue) % 16 == 0, "zeroed as one block of 16-byte multiples");

#define HMQ_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#ifndef HMQ_IDLE_SLEEP
#define HMQ_IDLE_SLEEP 8              // s_sleep argument (x 64 clocks) between two polls of an idle evaluator workgroup
#endif
// Every word two workgroups share is accessed as GLOBAL memory (address space 1: global_* instructions), never through a generic
// pointer (flat_*: the queue pointer often comes out of LDS or a by-value struct, where the compiler cannot infer the address space).
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) u64q gu64;
__device__ __forceinline__ gu32* G32(const unsigned* p) { return (gu32*)(p); }
__device__ __forceinline__ gu64* G64(const u64q* p) { return (gu64*)(p); }

__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ONE lane, after every storing wave has drained and the workgroup has met at a barrier
__device__ __forceinline__ void release_agent() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// ONE lane, after its relaxed poll has matched; the workgroup barrier that follows holds the other waves until the invalidate is done
__device__ __forceinline__ void acquire_agent() {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// Write-through (sc1) stores for bytes another workgroup will read inside the launch: they leave the XCD's L2 at once, so the
// hand-off needs no release fence — which writes back a whole L2 and, issued once per position by a hundred workgroups, was
// measured to stretch the evaluator's 0.075 ms per position to 0.5 ms.  16 bytes per lane; the asm store is invisible to the
// compiler's wait counting: the storing wave runs drain_stores() before it signals.
typedef unsigned int u32x4q __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_wt(void* p, u32x4q v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store2_wt(uint16_t* p, uint16_t v) { __hip_atomic_store((__attribute__((address_space(1))) uint16_t*)(p), v, HMQ_RLX); }
__device__ __forceinline__ bool spin_expired(u64q t0) { return (u64q)__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS; }

// producer, ONE lane (release done): n items with consecutive tickets
__device__ __forceinline__ void push_items(SrvQueue* q, const unsigned* items, int n) {
    const unsigned t = __hip_atomic_fetch_add(G32(&q->tail), (unsigned)n, HMQ_RLX);
    for (int i = 0; i < n; ++i) {
        const u64q old = __hip_atomic_exchange(G64(&q->slots[(t + i) & (QCAP - 1)]), ((u64q)(t + i + 1) << 32) | items[i], HMQ_RLX);
        if ((unsigned)(old >> 32) == t + i + 1) __hip_atomic_fetch_add(G32(&q->dupTickets), 1u, HMQ_RLX);
    }
}
// the last producer to leave releases every consumer (ONE lane)
__device__ __forceinline__ void producer_exit(SrvQueue* q) {
    if (__hip_atomic_fetch_sub(G32(&q->producers), 1u, HMQ_RLX) != 1u) return;
    const unsigned n = __hip_atomic_load(G32(&q->consumers), HMQ_RLX);
    const unsigned t = __hip_atomic_fetch_add(G32(&q->tail), n, HMQ_RLX);
    for (unsigned i = 0; i < n; ++i)
        (void)__hip_atomic_exchange(G64(&q->slots[(t + i) & (QCAP - 1)]), ((u64q)(t + i + 1) << 32) | IT_POISON, HMQ_RLX);
}
// consumer, ONE lane: next item (blocks; IT_POISON on shutdown, error or give-up).  No acquire yet.
__device__ __forceinline__ unsigned pop_item(SrvQueue* q) {
    const unsigned t = __hip_atomic_fetch_add(G32(&q->head), 1u, HMQ_RLX);
    u64q* slot = &q->slots[t & (QCAP - 1)];
    const u64q t0 = __builtin_amdgcn_s_memrealtime();
    u64q tMoved = t0;                        // when the tail or the beat counter was last seen to move
    unsigned tailSeen = ~0u, beatSeen = ~0u;
    for (unsigned spins = 0;; ++spins) {
        const u64q v = __hip_atomic_load(G64(slot), HMQ_RLX);
        if ((unsigned)(v >> 32) == t + 1) return (unsigned)v;
        __builtin_amdgcn_s_sleep(HMQ_IDLE_SLEEP);
        if ((spins & 255u) == 255u) {
            if (__hip_atomic_load(G32(&q->error), HMQ_RLX)) return IT_POISON;
            const u64q waited = (u64q)__builtin_amdgcn_s_memrealtime() - t0;
            // no search workgroup has come in while this one has been waiting: the roles are not resident together
            if (waited > MEET_LIMIT_TICKS && __hip_atomic_load(G32(&q->treesIn), HMQ_RLX) == 0u) { __hip_atomic_store(G32(&q->error), 4u, HMQ_RLX); return IT_POISON; }
            const unsigned tailNow = __hip_atomic_load(G32(&q->tail), HMQ_RLX), beatNow = __hip_atomic_load(G32(&q->beats), HMQ_RLX);
            if (tailNow != tailSeen || beatNow != beatSeen) { tailSeen = tailNow; beatSeen = beatNow; tMoved = t0 + waited; }
            const bool stalled = t0 + waited - tMoved > IDLE_LIMIT_TICKS && __hip_atomic_load(G32(&q->producers), HMQ_RLX) != 0u;
            if (stalled || waited > SPIN_LIMIT_TICKS) {
                if (atomicCAS(&q->dbgPop[0], 0u, t + 1u) == 0u) {
                    q->dbgPop[1] = tailNow;
                    // where every game stands NOW, before the error word lets anybody go: done counters and diagnostics records
                    const unsigned G = __hip_atomic_load(G32(&q->games), HMQ_RLX);
                    const QueueLayout lay(G, sizeof(SrvQueue));
                    unsigned char* base = reinterpret_cast<unsigned char*>(q);
                    unsigned* liveDone = reinterpret_cast<unsigned*>(base + lay.done);
                    unsigned* snapDone = reinterpret_cast<unsigned*>(base + lay.snapDone);
                    for (unsigned i = 0; i < 2 * G; ++i) snapDone[i] = __hip_atomic_load(G32(&liveDone[i]), HMQ_RLX);
                    unsigned* liveDiag = reinterpret_cast<unsigned*>(base + lay.diag);
                    unsigned* snapDiag = reinterpret_cast<unsigned*>(base + lay.snapDiag);
                    for (unsigned i = 0; i < G * (unsigned)(sizeof(GameDiag) / 4); ++i) snapDiag[i] = __hip_atomic_load(G32(&liveDiag[i]), HMQ_RLX);
                    *reinterpret_cast<unsigned*>(base + lay.snapTime) = (unsigned)((u64q)__builtin_amdgcn_s_memrealtime() / 1000ULL);
                }
                __hip_atomic_store(G32(&q->error), stalled ? 5u : 2u, HMQ_RLX);
                return IT_POISON;
            }
        }
    }
}
// waiter, ONE lane: until *counter >= want (false: error / give-up).  No acquire yet.
__device__ __forceinline__ bool wait_count(SrvQueue* q, unsigned* counter, unsigned want) {
    const u64q t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(G32(counter), HMQ_RLX) >= want) return true;
        __builtin_amdgcn_s_sleep(4);
        if ((spins & 255u) == 255u) {
            if (__hip_atomic_load(G32(&q->error), HMQ_RLX)) return false;
            const u64q waited = (u64q)__builtin_amdgcn_s_memrealtime() - t0;
            // no evaluator workgroup has come in in all that time: the roles are not resident together
            if (waited > MEET_LIMIT_TICKS && __hip_atomic_load(G32(&q->consIn), HMQ_RLX) == 0u) { __hip_atomic_store(G32(&q->error), 4u, HMQ_RLX); return false; }
            if (waited > SPIN_LIMIT_TICKS) { __hip_atomic_store(G32(&q->error), 3u, HMQ_RLX); return false; }
        }
    }
}

// what the persistent evaluator kernel needs besides the network: both plane buffers, both sets of heads, the queue, the
// per-(game, buffer) completion counters and optional accumulators of the per-position time
struct ServeArgs {
    const uint16_t* planes[2];
    uint16_t* value[2];
    uint16_t* piA[2];
    uint16_t* piB[2];
    uint16_t* wdl[2];
    uint16_t* ml[2];
    SrvQueue* q;
    unsigned* done;
    // the prior pipeline of every evaluated leaf runs on the evaluator (hm_policy.hpp): legal move lists in (tree -> evaluator),
    // moves and priors in prior order out; the policy logits themselves never leave the workgroup's LDS
    const uint32_t* leafMoves;               // [games][2][8][2][HM_MAX_MOVES], bit 31 = capture
    const int* leafCounts;                   // [games][2][8][2]: moves | side to move << 16
    uint32_t* sortedMoves;                   // [games][2][8][2][HM_MAX_MOVES]
    float* sortedPriors;
    const int* polNormal;                    // policy index tables (common/globals.cc:50-103)
    const int* polDrop;
    const float* noise;                      // [games][2][NOISE_CAP] gamma draws of the roots
    int noiseOn;
    float noiseEps;
    u64q* clkSum;                            // += ticks (100 MHz) spent evaluating, or nullptr
    u64q* clkCnt;                            // += positions evaluated
    unsigned abortAfter;                     // test hook (0 = off): once this many rows have been published an evaluator workgroup raises "stalled" — the search is given up half-way
};

}  // namespace hmq

struct hm_net;
// hm_net.hip: 1 when `net` can be the evaluator of the single-launch search (the narrow trunks: C = 64, 128, 384)
int hm_net_can_serve(const hm_net* net);
// hm_net.hip: 1 when one position takes this network's evaluator much longer than a game needs to collect a batch (the 384-channel trunk)
int hm_net_serve_is_slow(const hm_net* net);
// hm_net.hip: what the evaluator role needs of a network handle
struct hm_net_serve_info { const void* d_nd; const void* d_wh; const void* d_wf; int C, k5, copMax, uHalfs; size_t ldsBytes; };
int hm_net_serve_info_get(const hm_net* net, hm_net_serve_info* out);
