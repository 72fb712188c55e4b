// hm_queue.hpp — device-side evaluation queue between the persistent search kernel (k_search, hm_search.hip: one workgroup per
// game, alive for a whole search) and the persistent evaluator kernel (rise_serve, hm_net.hip: one workgroup per position at a
// time), replacing the host-driven lockstep of Engine::enqueueInferenceHalf / synchronizeInferenceHalf (nn/engine.h:43-81) inside
// SearchThread::run_iteration (searchthread.cc:661-739) for the native evaluator.
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
constexpr u64q MEET_LIMIT_TICKS = 300000000ULL;    // 3 s: by then the other kernel of the pair must have started (error 4: not concurrent)
// When NO row has been published for this long (the tail has not moved) while search workgroups are still running, a waiting
// evaluator workgroup declares the search stalled (error 5): a game's collect phase lasts ~0.1 ms, so even one live game moves
// the tail several times per millisecond (a workgroup far back in the line may itself wait much longer for its turn).  Observed on
// MI355X with the deployed network about once in 800 searches: the search workgroups stop making progress (in whatever phase they
// are in, all published rows evaluated, every counter consistent) until the evaluator kernel has left the device; the caller then
// repeats that search (hm_sp_search_stalled / hm_sp_begin_again), which costs one search and changes no result.
constexpr u64q IDLE_LIMIT_TICKS = 3000000ULL;      // 30 ms

// item payload (low 32 bits of a slot): game slot, plane buffer, row, evaluator-specific flags
constexpr unsigned IT_POISON = 0x80000000u;  // no more work: the consumer leaves
constexpr unsigned IT_ROOT = 0x02000000u;    // the leaf is the root of its search: Dirichlet noise is mixed into its priors
__host__ __device__ inline unsigned item_pack(int game, int buf, int row, int net) { return (unsigned)game | ((unsigned)buf << 20) | ((unsigned)row << 21) | ((unsigned)net << 24); }
__host__ __device__ inline int item_game(unsigned it) { return (int)(it & 0xfffffu); }
__host__ __device__ inline int item_buf(unsigned it) { return (int)((it >> 20) & 1u); }
__host__ __device__ inline int item_row(unsigned it) { return (int)((it >> 21) & 7u); }
__host__ __device__ inline int item_net(unsigned it) { return (int)((it >> 24) & 1u); }

struct SrvQueue {                            // one per search engine; zeroed (whole struct) before every search
    unsigned head;                           // next ticket a consumer takes
    unsigned tail;                           // next ticket a producer fills
    unsigned producers;                      // search workgroups still running (set by the host after the memset)
    unsigned error;                          // != 0: some spin gave up (code of the first), everybody leaves
    unsigned consumers;                      // evaluator workgroups of this launch (poison count)
    unsigned games;                          // game slots: behind the slots lie done[2G], progress[3G], heartbeats[4G] and a snapshot[7G] of the last two
    unsigned served;                         // items evaluated (statistics)
    unsigned treesIn, treesOut, consIn, consOut;   // census of both kernels (diagnostics of a give-up)
    unsigned dbg[6];                         // first failed wait of a search workgroup: game, buffer, rows expected, rows done, iteration, ms since the kernel's first workgroup started
    unsigned dupTickets;                     // pushes that found their slot already carrying their own ticket's tag (diagnostics: two producers drew one ticket)
    unsigned dbgPop[2];                      // the evaluator workgroup that gave up (error 2): its ticket, the tail and head it saw then, ms waited
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
    u64q tMoved = t0;                        // when the tail was last seen to move
    unsigned tailSeen = ~0u;
    for (unsigned spins = 0;; ++spins) {
        const u64q v = __hip_atomic_load(G64(slot), HMQ_RLX);
        if ((unsigned)(v >> 32) == t + 1) return (unsigned)v;
        __builtin_amdgcn_s_sleep(HMQ_IDLE_SLEEP);
        if ((spins & 255u) == 255u) {
            if (__hip_atomic_load(G32(&q->error), HMQ_RLX)) return IT_POISON;
            const u64q waited = (u64q)__builtin_amdgcn_s_memrealtime() - t0;
            // no search workgroup has started while this one has been waiting: the two kernels are being run one after the other
            if (waited > MEET_LIMIT_TICKS && __hip_atomic_load(G32(&q->treesIn), HMQ_RLX) == 0u) { __hip_atomic_store(G32(&q->error), 4u, HMQ_RLX); return IT_POISON; }
            const unsigned tailNow = __hip_atomic_load(G32(&q->tail), HMQ_RLX);
            if (tailNow != tailSeen) { tailSeen = tailNow; tMoved = t0 + waited; }
            const bool stalled = t0 + waited - tMoved > IDLE_LIMIT_TICKS && __hip_atomic_load(G32(&q->producers), HMQ_RLX) != 0u;
            if (stalled || waited > SPIN_LIMIT_TICKS) {
                if (atomicCAS(&q->dbgPop[0], 0u, t + 1u) == 0u) {
                    q->dbgPop[1] = __hip_atomic_load(G32(&q->tail), HMQ_RLX);
                    // where every game stands NOW, before the error word lets anybody go
                    const unsigned G = __hip_atomic_load(G32(&q->games), HMQ_RLX);
                    unsigned* live = reinterpret_cast<unsigned*>(q + 1) + 2 * G;
                    unsigned* snap = live + 7 * G;
                    for (unsigned i = 0; i < 7 * G; ++i) snap[i] = __hip_atomic_fetch_add(G32(&live[i]), 0u, HMQ_RLX);
                    snap[2 * G] = (unsigned)((u64q)__builtin_amdgcn_s_memrealtime() / 1000ULL);   // (slot 0's exit stamp is not needed: the time of the snapshot)
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
            // no evaluator workgroup has started in all that time: the two kernels are being run one after the other
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
};

}  // namespace hmq

struct hm_net;
// hm_net.hip: 1 when `net` has a persistent evaluator kernel (the 8-wave narrow trunks: C = 64, 128, 384)
int hm_net_can_serve(const hm_net* net);
// hm_net.hip: launches `grid` persistent evaluator workgroups on `stream`; they leave when the queue hands them IT_POISON
int hm_net_serve(const hm_net* net, const hmq::ServeArgs& args, int grid, hipStream_t stream);
// hm_net.hip: 1 when one position takes this network's evaluator much longer than a game needs to collect a batch (the 384-channel trunk)
int hm_net_serve_is_slow(const hm_net* net);
