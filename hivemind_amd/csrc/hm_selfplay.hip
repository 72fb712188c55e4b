// hm_selfplay.hip — host-side self-play driver over the GPU search engine (host code only; built
// with hipcc together with the kernels so the library stays one .so).
//
// Replaces run_selfplay (tools/selfplay.cc:558-748, config tools/selfplay.h:10-31) for G concurrent
// game slots per GPU: per-game RNG decisions (raw-policy opening length and temperatures, node
// jitter, visit-temperature action sampling, resignation), HVM4 record assembly
// (TrainingSample selfplay.cc:40-53, ChunkWriter :69-158) and the lockstep search loop.  The
// network stays behind the evaluator seam (nn/engine.h:66-81): the caller registers device buffers
// and an `hm_eval_fn` callback that runs its net on planes[which] and fills the head buffers.
//
// Documented deviations from the reference loop:
//   * every game draws from its own std::mt19937_64 seeded mix_seed(runId, gameIndex) (the reference
//     threads one engine through its sequential game loop), so results do not depend on how games
//     are spread over slots, ranks or GPUs;  games are striped gameIndex % world == rank.
//   * PGN text is not produced by self-play (not consumed by training).
//
// The same lockstep core drives run_tournament (tools/tournament.cc:328-465, hm_tournament_* at the end of this file): paired games
// between two networks, each search evaluated by the network of the team to move (two forwards per iteration over disjoint row
// sets), summary.json / games.pgn and the result statistics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <sys/stat.h>
#include <map>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../../include/hivemind_amd.h"
#include "hm_queue.hpp"

int hm_fail(int code, const std::string& msg);
extern "C" int hm_sp_active(hm_sp* sp, int* active);
extern "C" int hm_sp_active_on(hm_sp* sp, int* pinned_out, hipStream_t stream);
extern "C" int hm_sp_apply_deferred(hm_sp* sp, const hm_move* move_a, const hm_move* move_b, const uint8_t* mask);   // hm_search.hip (internal): hm_sp_apply whose error flag the next hm_sp_game_state checks

namespace {

uint64_t mix_seed(uint64_t seed, uint64_t value) {   // selfplay.cc:160-165
    value += 0x9e3779b97f4a7c15ULL;
    value = (value ^ (value >> 30)) * 0xbf58476d1ce4e5b9ULL;
    value = (value ^ (value >> 27)) * 0x94d049bb133111ebULL;
    return seed ^ (value ^ (value >> 31));
}

struct Sparse { uint16_t index; float probability; };
struct Sample {   // TrainingSample selfplay.cc:40-53
    uint64_t gameId = 0;
    uint32_t nodes = 0;
    uint16_t macroPly = 0, movesLeft = 0;
    uint8_t team = 0, hasTimeAdvantage = 0;
    int8_t outcome = 0;
    uint8_t wdl = 1;
    float rootQ = 0.0f;
    std::array<uint8_t, HM_PLANE_VALUES> planes{};
    std::vector<Sparse> policyA, policyB;
};

template <typename T>
void put(std::vector<uint8_t>& out, const T& v) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
    out.insert(out.end(), p, p + sizeof(T));
}
void serialize(std::vector<uint8_t>& out, const Sample& s) {   // ChunkWriter::flush body selfplay.cc:126-142
    put(out, s.gameId); put(out, s.nodes); put(out, s.macroPly); put(out, s.movesLeft);
    put(out, s.team); put(out, s.hasTimeAdvantage); put(out, s.outcome); put(out, s.wdl); put(out, s.rootQ);
    out.insert(out.end(), s.planes.begin(), s.planes.end());
    for (const auto* pol : {&s.policyA, &s.policyB}) {
        put(out, static_cast<uint16_t>(pol->size()));
        for (const Sparse& e : *pol) { put(out, e.index); put(out, e.probability); }
    }
}

void apply_temperature(std::vector<float>& p, double temperature) {   // selfplay.cc:167-185
    const double exponent = 1.0 / temperature;
    double total = 0.0;
    for (float& v : p) { v = static_cast<float>(std::pow(std::max(0.0f, v), exponent)); total += v; }
    if (!std::isfinite(total) || total <= 0.0) { std::fill(p.begin(), p.end(), 1.0f / static_cast<float>(p.size())); return; }
    for (float& v : p) v = static_cast<float>(v / total);
}

struct Slot {
    bool active = false;
    uint64_t gameIndex = 0;
    std::mt19937_64 rng;
    int team = 0;
    bool adv = false;
    size_t macroPly = 0, initLength = 0, rawPlies = 0;
    bool rawActive = false, canResign = false;
    std::array<size_t, 2> resignPlies{{0, 0}};
    std::vector<Sample> samples;
    int winner = -1;
    int termination = 0;   // 0 limit, 1 checkmate, 2 draw, 3 resignation, 4 no legal action
};

}  // namespace

struct hm_selfplay {
    hm_selfplay_config cfg;
    hm_search_config scfg;
    hm_eval_io io;
    hm_eval_fn fn;
    void* user;
    hm_sp* sp = nullptr;
    int G = 0;
    std::vector<Slot> slots;
    uint64_t nextGame = 0, gamesDone = 0, runId = 0;
    std::vector<uint8_t> records;             // serialized samples of finished games not yet handed to a chunk sink
    std::vector<size_t> sampleEnd;            // byte offset just past each pending sample
    uint64_t recordCount = 0;                 // == sampleEnd.size()
    hm_chunk_fn sink = nullptr;               // ChunkWriter::flush replacement (selfplay.cc:105-151)
    void* sinkUser = nullptr;
    std::string outDir;                       // built-in sink: <outDir>/training_data/chunk_<runId>_<idx>.hvm
    uint64_t chunkIndex = 0;
    int sinkError = 0;
    hm_selfplay_result res{};
    hm_board* d_boards = nullptr;
    uint8_t* d_u8 = nullptr;
    uint8_t* h_u8 = nullptr;                // pinned: the u8 planes of the searched positions, for the records
    std::vector<uint64_t> termCounts = std::vector<uint64_t>(5, 0);
    std::vector<hipEvent_t> evs, sync;                 // ring of per-iteration leg events
    hipStream_t sT = nullptr, sN = nullptr;      // tree / network streams (native evaluator mode)
    int* hActive = nullptr;
    // K lockstep iterations captured once as a HIP graph (collect || forward -> process, K times): one launch replays them,
    // so the dependent launches and cross-stream waits of the inner loop cost graph edges instead of host round trips
    hipGraphExec_t stepGraph = nullptr;
    hipEvent_t gFork = nullptr, gJoin = nullptr;
    hipEvent_t gCollected[2] = {nullptr, nullptr};
    int graphState = 0;                            // 0 not tried, 1 ready, -1 unavailable (eager loop)
    bool persistentOff = false;                    // hm_sp_search found its kernels serialised once: lockstep from then on
    // time-managed searches (tournaments with a movetime limit): every searching slot has its own controller of the reference's
    // polling loop (agent.cc:715-806: deadline, early stopping, time extension); 0 = node-limited searches
    int moveTimeMs = 0;
    std::vector<hm_time_manager*> timeCtl;
    int32_t* d_rows[2] = {nullptr, nullptr};   // per game slot: plane rows written into planes[k] (ragged evaluator batch)
    // second network (tournaments: contender = io.net, baseline = net2): every slot's search is evaluated by the network of
    // the team to move, so each iteration launches both forwards over disjoint row sets
    const hm_net* net2 = nullptr;
    uint8_t* d_netSel = nullptr;               // per slot: 0 = io.net, 1 = net2 (fixed during a search)
    int32_t* d_rowsNet[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [net][plane buffer]: d_rows masked by d_netSel
};

__global__ void k_split_rows(const int32_t* rows, const uint8_t* sel, int32_t* rows0, int32_t* rows1, int n) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) { const int r = rows[g]; const bool second = sel[g] != 0; rows0[g] = second ? 0 : r; rows1[g] = second ? r : 0; }
}
// collect(next) with the per-game row counts the evaluator launches need
static int collect_rows(hm_selfplay* s, int next) {
    if (int rc = hm_sp_collect_counted(s->sp, s->io.planes[next], s->d_rows[next], s->sT)) return rc;
    if (s->net2) {
        hipLaunchKernelGGL(k_split_rows, dim3((s->G + 63) / 64), dim3(64), 0, s->sT, s->d_rows[next], s->d_netSel, s->d_rowsNet[0][next], s->d_rowsNet[1][next], s->G);
        if (hipGetLastError() != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "k_split_rows launch failed");
    }
    return 0;
}
// forward(cur) on the network stream: one launch, or one per network over its slots' rows
static int forward_rows(hm_selfplay* s, int which, void** h, bool allRows) {
    uint64_t* clk = hm_sp_leg_clock_net(s->sp);      // both launches of a two-network iteration fall into the one interval
    if (!s->net2)
        return hm_net_forward_groups_timed(s->io.net, s->io.planes[which], s->G * 8, allRows ? nullptr : s->d_rows[which], 8, h[0], h[1], h[2], h[3], h[4], s->sN, clk);
    if (int rc = hm_net_forward_groups_timed(s->io.net, s->io.planes[which], s->G * 8, s->d_rowsNet[0][which], 8, h[0], h[1], h[2], h[3], h[4], s->sN, clk)) return rc;
    return hm_net_forward_groups_timed(s->net2, s->io.planes[which], s->G * 8, s->d_rowsNet[1][which], 8, h[0], h[1], h[2], h[3], h[4], s->sN, clk);
}


static bool start_game(hm_selfplay* s, Slot& sl, hm_board& out) {
    const hm_selfplay_config& c = s->cfg;
    // next game index owned by this rank (striped)
    while (s->nextGame < c.games && (int)(s->nextGame % (uint64_t)c.world) != c.rank) s->nextGame++;
    if (s->nextGame >= c.games) { sl.active = false; return false; }
    sl = Slot();
    sl.active = true;
    sl.gameIndex = s->nextGame++;
    sl.rng.seed(mix_seed(s->runId, sl.gameIndex));
    const int startingTeam = static_cast<int>(sl.rng() & 1ULL);   // selfplay.cc:590
    sl.team = startingTeam == 0 ? HM_WHITE : HM_BLACK;
    sl.adv = false;
    // sample_initialization_length selfplay.cc:187-200
    sl.initLength = 0;
    if (c.raw_policy_mean_macro_plies > 0.0 && c.raw_policy_max_macro_plies != 0) {
        std::exponential_distribution<double> d(1.0 / c.raw_policy_mean_macro_plies);
        size_t len = static_cast<size_t>(std::llround(d(sl.rng)));
        if (len > c.raw_policy_max_macro_plies) {
            std::uniform_int_distribution<size_t> clipped(0, c.raw_policy_max_macro_plies);
            len = clipped(sl.rng);
        }
        sl.initLength = len;
    }
    sl.rawActive = sl.initLength > 0;
    sl.canResign = c.resign_threshold < 0.0f
        && (c.resign_disable_fraction <= 0.0 || std::uniform_real_distribution<double>(0.0, 1.0)(sl.rng) >= c.resign_disable_fraction);
    hm_board_startpos(&out);
    out.team = (uint8_t)sl.team;
    out.time_adv = 0;
    return true;
}

// ChunkWriter::flush (selfplay.cc:105-151): the first `count` pending samples leave as one chunk
static void flush_chunk(hm_selfplay* s, size_t count) {
    if (count == 0 || count > s->sampleEnd.size()) return;
    const size_t nbytes = s->sampleEnd[count - 1];
    int rc = 0;
    if (s->sink) rc = s->sink(s->sinkUser, s->records.data(), nbytes, count, s->chunkIndex);
    else {
        char name[96];
        std::snprintf(name, sizeof name, "/training_data/chunk_%llu_%06llu.hvm", (unsigned long long)s->runId, (unsigned long long)s->chunkIndex);
        rc = hm_hvm4_write_chunk((s->outDir + name).c_str(), s->records.data(), nbytes, count);
    }
    if (rc && !s->sinkError) s->sinkError = rc;
    s->chunkIndex++;
    s->res.chunks_flushed += 1;
    s->records.erase(s->records.begin(), s->records.begin() + (std::ptrdiff_t)nbytes);
    s->sampleEnd.erase(s->sampleEnd.begin(), s->sampleEnd.begin() + (std::ptrdiff_t)count);
    for (size_t& e : s->sampleEnd) e -= nbytes;
    s->recordCount = s->sampleEnd.size();
}
static bool has_sink(const hm_selfplay* s) { return s->sink != nullptr || !s->outDir.empty(); }

static void finish_game(hm_selfplay* s, Slot& sl) {   // selfplay.cc:726-734
    const size_t perChunk = std::max<size_t>(1, (size_t)s->cfg.chunk_samples);
    for (size_t i = 0; i < sl.samples.size(); ++i) {
        Sample& sm = sl.samples[i];
        sm.outcome = sl.winner < 0 ? 0 : (sm.team == sl.winner ? 1 : -1);
        sm.wdl = static_cast<uint8_t>(sm.outcome + 1);
        sm.movesLeft = static_cast<uint16_t>(std::min<size_t>(sl.samples.size() - i, 65535));
        serialize(s->records, sm);
        s->sampleEnd.push_back(s->records.size());
        s->res.total_nodes += sm.nodes;
        if (has_sink(s) && s->sampleEnd.size() >= perChunk) flush_chunk(s, perChunk);   // ChunkWriter::append :78-85
    }
    s->recordCount = s->sampleEnd.size();
    s->res.samples += sl.samples.size();
    s->res.games += 1;
    s->res.raw_plies += sl.rawPlies;
    s->termCounts[sl.termination]++;
    s->gamesDone++;
    sl.samples.clear();
    sl.active = false;
}

static int eval_rows_sync(hm_selfplay* s, int which, int rows) {   // evaluator on the null stream (raw-policy plies, callback mode)
    if (s->io.net)
        return hm_net_forward(s->io.net, s->io.planes[which], rows, s->io.value, s->io.pi_a, s->io.pi_b, s->io.wdl, s->io.moves_left, nullptr);
    if (int rc = s->fn(s->user, which, rows)) return hm_fail(HM_ERR_STATE, "evaluator callback failed (" + std::to_string(rc) + ")");
    return 0;
}

constexpr int GRAPH_ITERS = 8;                   // even: the plane / head double buffers are back where they started

// one native lockstep iteration enqueued on (sT, sN): collect(next) on the tree stream beside forward(cur) on the network
// stream, then process(cur) on the tree stream.  `fork`/`join` are ordering-only events.
static int enqueue_iteration(hm_selfplay* s, int which, int parity, hipEvent_t fork, hipEvent_t join, bool allRows,
                             hipEvent_t collectedPrev = nullptr, hipEvent_t collectedThis = nullptr) {
    void* hv[2][5] = {{s->io.value, s->io.pi_a, s->io.pi_b, s->io.wdl, s->io.moves_left},
                      {s->io.value_2, s->io.pi_a_2, s->io.pi_b_2, s->io.wdl_2, s->io.moves_left_2}};
    void** h = hv[parity];
    if (collectedPrev) {
        // early forward: forward(i) only needs the planes collect(i-1) wrote, so it may start beside process(i-1) (used with the
        // tree and network streams pinned to disjoint CU sets; process(i-2), the last reader of this head set, precedes collect(i-1))
        (void)hipStreamWaitEvent(s->sN, collectedPrev, 0);
    } else {
        (void)hipEventRecord(fork, s->sT);         // forward(i) runs behind process(i-1), beside collect(i)
        (void)hipStreamWaitEvent(s->sN, fork, 0);
    }
    if (int rc = collect_rows(s, 1 - which)) return rc;
    if (collectedThis) (void)hipEventRecord(collectedThis, s->sT);
    if (int rc = forward_rows(s, which, h, allRows)) return rc;
    (void)hipEventRecord(join, s->sN);
    (void)hipStreamWaitEvent(s->sT, join, 0);
    return hm_sp_process(s->sp, h[0], h[1], h[2], h[3], h[4], nullptr, s->sT);
}

static void build_step_graph(hm_selfplay* s, bool allRows) {
    s->graphState = -1;
    if (std::getenv("HM_SELFPLAY_NO_GRAPH") || s->sN == s->sT) return;
    if (hipEventCreateWithFlags(&s->gFork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->gJoin, hipEventDisableTiming) != hipSuccess) return;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(s->sT, hipStreamCaptureModeThreadLocal) != hipSuccess) return;
    int rc = 0;
    const bool early = std::getenv("HM_SELFPLAY_EARLY_FORWARD") != nullptr;
    if (early && (!s->gCollected[0] && (hipEventCreateWithFlags(&s->gCollected[0], hipEventDisableTiming) != hipSuccess
                                        || hipEventCreateWithFlags(&s->gCollected[1], hipEventDisableTiming) != hipSuccess))) { (void)hipStreamEndCapture(s->sT, &graph); return; }
    for (int k = 0; k < GRAPH_ITERS && !rc; ++k)
        rc = early ? enqueue_iteration(s, k & 1, k & 1, s->gFork, s->gJoin, allRows, k > 0 ? s->gCollected[(k - 1) & 1] : nullptr, s->gCollected[k & 1])
                   : enqueue_iteration(s, k & 1, k & 1, s->gFork, s->gJoin, allRows);
    const hipError_t e = hipStreamEndCapture(s->sT, &graph);
    if (rc || e != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); (void)hipGetLastError(); return; }
    if (hipGraphInstantiate(&s->stepGraph, graph, nullptr, nullptr, 0) == hipSuccess) s->graphState = 1;
    (void)hipGraphDestroy(graph);
}

static int run_search_lockstep(hm_selfplay* s, int minTarget) {
    constexpr int RING = 256;
    int which = 0, active = 1, iters = 0;
    const bool native = s->io.net != nullptr;
    if (s->evs.empty()) {
        s->evs.resize((size_t)RING * 6);
        for (auto& e : s->evs) if (hipEventCreate(&e) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipEventCreate failed");
        s->sync.resize((size_t)RING * 2);          // ordering-only events (no timestamps): forward done, process done
        for (auto& e : s->sync) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipEventCreate failed");
    }
    if (native && !s->sT) {
        // measured alternative: HM_SELFPLAY_CU_SPLIT=<n> pins the tree stream to n CUs and the network stream to the rest
        const int split = std::getenv("HM_SELFPLAY_CU_SPLIT") ? std::atoi(std::getenv("HM_SELFPLAY_CU_SPLIT")) : 0;
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (split > 0 && hipGetDeviceProperties(&prop, dev) == hipSuccess && split < prop.multiProcessorCount) {
            const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
            std::vector<uint32_t> mT(words, 0), mN(words, 0);
            const int stride = ncu / split;
            for (int i = 0; i < ncu; ++i) {
                const bool tree = (i % stride) == 0 && (i / stride) < split;
                (tree ? mT : mN)[i >> 5] |= 1u << (i & 31);
            }
            if (hipExtStreamCreateWithCUMask(&s->sT, words, mT.data()) != hipSuccess || hipExtStreamCreateWithCUMask(&s->sN, words, mN.data()) != hipSuccess)
                return hm_fail(HM_ERR_NO_DEVICE, "hipExtStreamCreateWithCUMask failed");
        } else {
        if (hipStreamCreateWithFlags(&s->sT, hipStreamNonBlocking) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipStreamCreate failed");
        if (std::getenv("HM_SELFPLAY_SEQUENTIAL")) s->sN = s->sT;          // diagnostic: no collect/net overlap
        else if (hipStreamCreateWithFlags(&s->sN, hipStreamNonBlocking) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipStreamCreate failed");
        }
        if (hipHostMalloc(reinterpret_cast<void**>(&s->hActive), sizeof(int), hipHostMallocDefault) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipHostMalloc failed");
    }
    static const bool allRows = std::getenv("HM_SELFPLAY_ALL_ROWS") != nullptr;   // diagnostic: evaluate the unused rows too
    if (native) {
        (void)hipMemsetAsync(s->d_rows[0], 0, sizeof(int32_t) * s->G, nullptr);      // nothing collected yet for this search
        (void)hipMemsetAsync(s->d_rows[1], 0, sizeof(int32_t) * s->G, nullptr);
        if (s->net2) for (int k = 0; k < 4; ++k) (void)hipMemsetAsync(s->d_rowsNet[k >> 1][k & 1], 0, sizeof(int32_t) * s->G, nullptr);
        (void)hipStreamSynchronize(nullptr);      // the prologue ran on the null stream
    }
    // No game can finish before it has collected minTarget nodes, i.e. floor(minTarget / 8) batches: the
    // host does not poll (and so does not synchronise) before that many iterations have been enqueued.
    const int noPollBefore = minTarget / 8 - 1;
    // Leg timings.  Native evaluator: the device-side leg clock (hm_sp_leg_times) -- every launch, graph-replayed ones included,
    // stamps its own start and end, so the totals are exact and cost the host nothing; it is reset and read once per run
    // (leg_clock_begin / leg_clock_end).  Callback evaluator: HIP events on the null stream around each leg.
    int harvested = 0, nSamples = 0;
    double sum[3] = {0.0, 0.0, 0.0};
    auto is_timed = [&](int) { return !native; };
    auto harvest = [&](int upto) {                 // leg timings of the sampled iterations in [harvested, upto)
        for (int it = harvested; it < upto; ++it) {
            if (!is_timed(it)) continue;
            hipEvent_t* e = &s->evs[(size_t)(it % RING) * 6];
            float ms = 0.0f;
            bool ok = true;
            double v[3];
            for (int k = 0; k < 3; ++k) { ok = ok && hipEventElapsedTime(&ms, e[2 * k], e[2 * k + 1]) == hipSuccess; v[k] = ms; }
            if (ok) { for (int k = 0; k < 3; ++k) sum[k] += v[k]; ++nSamples; }
        }
        harvested = upto;
    };
    // Native evaluator, node-limited search, one network: the whole search is ONE kernel launch (hm_sp_search: game workgroups and
    // evaluator workgroups of k_rollout) — every game advances at its own pace instead of in lockstep iterations.
    // HM_SELFPLAY_LOCKSTEP=1 keeps the host-driven loop below (the two produce identical records; tests compare them).
    if (native && !s->net2 && s->moveTimeMs <= 0 && !s->persistentOff && !std::getenv("HM_SELFPLAY_LOCKSTEP")
        && hm_sp_search_consumers(s->sp) > 0 && hm_net_can_serve(s->io.net)) {
        double kms = 0.0;
        const int rc = hm_sp_search(s->sp, s->io.net, &s->io, &kms);
        if (!rc) {
            s->res.persistent_searches += 1;
            s->res.search_kernel_ms += kms;
            return 0;
        }
        if (hm_sp_search_stalled(s->sp)) {
            // hang guard (hm_queue.hpp: neither the queue tail nor any game's phase moved for IDLE_LIMIT_TICKS beside searching games;
            // hm_sp_search has printed the give-up record): this one search is repeated from its start on the host-driven loop
            // below — same records — and the next search is a single-launch one again
            s->res.persistent_stalls += 1;
        } else {
            if (!hm_sp_search_not_concurrent(s->sp)) return rc;
            // the two roles of the launch were not resident together (something else holds the device's CUs): this search and all
            // later ones of this driver take the host-driven loop
            s->persistentOff = true;
        }
        if (int rc2 = hm_sp_begin_again(s->sp)) return rc2;
    }
    if (native && s->graphState == 0) build_step_graph(s, allRows);
    // Time-managed search: every 5 ms (agent.cc:562) the root statistics of all slots go through their controllers; slots told to
    // stop end at their next collect (finishing the batch in flight), the search ends when no slot is left.
    const bool timed = native && s->moveTimeMs > 0;
    const auto tStart = std::chrono::steady_clock::now();
    double lastCtl = 0.0;
    std::vector<int> tcCounts, tcVisits, tcInfo, tcType, tcEnd;
    std::vector<float> tcQ;
    std::vector<uint8_t> tcStop, tcStopped;
    if (timed) {
        const int E = hm_sp_max_edges(s->sp);
        tcCounts.resize(s->G); tcVisits.resize((size_t)s->G * E); tcQ.resize((size_t)s->G * E); tcInfo.resize((size_t)s->G * HM_SP_INFO_INTS);
        tcType.resize(E); tcEnd.resize(E); tcStop.assign(s->G, 0); tcStopped.assign(s->G, 0);
        for (auto*& m : s->timeCtl) { if (m) hm_time_manager_destroy(m); m = nullptr; }
        s->timeCtl.assign(s->G, nullptr);
        for (int g = 0; g < s->G; ++g) s->timeCtl[g] = hm_time_manager_create(s->moveTimeMs);
    }
    auto time_control = [&]() -> int {             // the tree stream is idle (called right after a poll)
        const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tStart).count();
        if (now - lastCtl < 5.0) return 0;
        lastCtl = now;
        const int E = hm_sp_max_edges(s->sp);
        if (int rc = hm_sp_root_stats(s->sp, tcCounts.data(), nullptr, nullptr, tcVisits.data(), tcQ.data(), nullptr, nullptr, tcInfo.data(), E)) return rc;
        bool anyStop = false;
        std::fill(tcStop.begin(), tcStop.end(), 0);
        for (int g = 0; g < s->G; ++g) {
            const int* inf = tcInfo.data() + (size_t)g * HM_SP_INFO_INTS;
            if (inf[0] != 1 || tcStopped[g]) continue;                        // not searching (ST_SEARCHING = 1) or already told to stop
            const int n = tcCounts[g];
            std::fill(tcType.begin(), tcType.begin() + n, 0); std::fill(tcEnd.begin(), tcEnd.begin() + n, 0);
            if (inf[12] >= 0 && inf[12] < n) { tcType[inf[12]] = inf[14]; tcEnd[inf[12]] = inf[15]; }
            const int stop = hm_time_manager_poll(s->timeCtl[g], now, inf[1], n, tcVisits.data() + (size_t)g * E, tcQ.data() + (size_t)g * E, inf[6] > 0 ? inf[6] : 0,
                                                  tcType.data(), tcEnd.data(), nullptr, nullptr, 0);
            if (stop < 0) return stop;
            if (stop) { tcStop[g] = 1; tcStopped[g] = 1; anyStop = true; }
        }
        if (anyStop) return hm_sp_stop(s->sp, tcStop.data(), s->sT);
        return 0;
    };
    while (active > 0) {
        // Once captured (during the first search), the graph is replayed from the first iteration of every search, GRAPH_ITERS
        // iterations per launch, polling the active-game count once per launch.
        if (native && s->graphState == 1 && which == 0) {
            if (hipGraphLaunch(s->stepGraph, s->sT) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipGraphLaunch failed");
            iters += GRAPH_ITERS;
            s->res.eval_batches += GRAPH_ITERS;
            if (iters >= noPollBefore || timed) {
                if (int rc = hm_sp_active_on(s->sp, s->hActive, s->sT)) return rc;
                active = *s->hActive;
                if (timed && active > 0) if (int rc = time_control()) return rc;
            }
            if (iters > 100000) return hm_fail(HM_ERR_STATE, "search did not terminate");
            continue;
        }
        hipEvent_t* e = &s->evs[(size_t)(iters % RING) * 6];
        hipEvent_t* y = &s->sync[(size_t)(iters % RING) * 2];
        hipEvent_t* yPrev = &s->sync[(size_t)((iters + RING - 1) % RING) * 2];
        const bool poll = iters >= noPollBefore || (iters - harvested) >= RING - 2;
        const bool evTimed = is_timed(iters);          // leg events around this iteration (callback evaluator only); `timed` above = movetime search
        if (native) {
            // tree stream: collect(next) -- net stream: forward(cur) -- tree stream: process(cur)
            void* hv[2][5] = {{s->io.value, s->io.pi_a, s->io.pi_b, s->io.wdl, s->io.moves_left},
                              {s->io.value_2, s->io.pi_a_2, s->io.pi_b_2, s->io.wdl_2, s->io.moves_left_2}};
            void** h = hv[iters & 1];
            if (evTimed) (void)hipEventRecord(e[0], s->sT);
            if (int rc = collect_rows(s, 1 - which)) return rc;
            if (evTimed) (void)hipEventRecord(e[1], s->sT);
            // planes[which] were completed by the previous iteration's collect.  The forward could start
            // right after it, but measured on MI355X it then shares CUs with k_process (8-wave blocks,
            // 79 KB LDS), which stretches from 0.12 to 0.33 ms and lengthens the critical path; so the
            // forward is ordered behind process(i-1) and overlaps only collect(i).
            if (iters > 0) (void)hipStreamWaitEvent(s->sN, yPrev[1], 0);
            if (evTimed) (void)hipEventRecord(e[2], s->sN);
            if (int rc = forward_rows(s, which, h, allRows)) return rc;
            if (evTimed) (void)hipEventRecord(e[3], s->sN);
            (void)hipEventRecord(y[0], s->sN);
            (void)hipStreamWaitEvent(s->sT, y[0], 0);
            if (evTimed) (void)hipEventRecord(e[4], s->sT);
            if (int rc = hm_sp_process(s->sp, h[0], h[1], h[2], h[3], h[4], nullptr, s->sT)) return rc;
            if (evTimed) (void)hipEventRecord(e[5], s->sT);
            (void)hipEventRecord(y[1], s->sT);
            if (poll || timed) {
                if (int rc = hm_sp_active_on(s->sp, s->hActive, s->sT)) return rc;     // async copy + stream sync
                active = *s->hActive;
                if (timed && active > 0) if (int rc = time_control()) return rc;
            }
        } else {
            (void)hipEventRecord(e[0], nullptr);
            if (int rc = hm_sp_collect(s->sp, s->io.planes[1 - which], nullptr)) return rc;
            (void)hipEventRecord(e[1], nullptr);
            (void)hipEventRecord(e[2], nullptr);
            if (int rc = s->fn(s->user, which, s->G * 8)) return hm_fail(HM_ERR_STATE, "evaluator callback failed (" + std::to_string(rc) + ")");
            (void)hipEventRecord(e[3], nullptr);
            (void)hipEventRecord(e[4], nullptr);
            if (int rc = hm_sp_process(s->sp, s->io.value, s->io.pi_a, s->io.pi_b, s->io.wdl, s->io.moves_left, nullptr, nullptr)) return rc;
            (void)hipEventRecord(e[5], nullptr);
            if (poll) if (int rc = hm_sp_active(s->sp, &active)) return rc;     // syncs the stream
        }
        which = 1 - which;
        s->res.eval_batches += 1;
        ++iters;
        if (poll && !(native && s->graphState == 1)) harvest(iters);
        if (iters > 100000) return hm_fail(HM_ERR_STATE, "search did not terminate");
    }
    if (!native) harvest(iters);                   // native: the leg clock is read once per run (leg_clock_end)
    if (nSamples > 0) {
        const double w = (double)iters / nSamples;
        s->res.collect_ms += w * sum[0]; s->res.eval_ms += w * sum[1]; s->res.process_ms += w * sum[2];
    }
    s->res.search_iterations += iters;
    return 0;
}

extern "C" {

void hm_selfplay_config_default(hm_selfplay_config* c) {   // tools/selfplay.h:10-31
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->games = 1; c->nodes = 800; c->max_macro_plies = 400; c->chunk_samples = 16384;
    c->raw_policy_mean_macro_plies = 8.0; c->raw_policy_max_macro_plies = 30; c->raw_policy_high_temperature_probability = 0.05;
    c->mcts_temperature = 1.0; c->mcts_temperature_decay = 0.93; c->mcts_temperature_plies = 20;
    c->resign_threshold = -0.90f; c->resign_consecutive_plies = 3; c->resign_disable_fraction = 0.10;
    c->node_random_factor = 0.05; c->dirichlet_alpha = 0.3f; c->dirichlet_epsilon = 0.25f;
    c->seed = 0; c->rank = 0; c->world = 1; c->concurrent_games = 64;
}

int hm_selfplay_create(const hm_selfplay_config* cfg, const hm_search_config* scfg, const hm_eval_io* io, hm_eval_fn fn, void* user, hm_selfplay** out) {
    if (!cfg || !io || (!fn && !io->net) || !out) return hm_fail(HM_ERR_INVALID, "null argument");
    if (io->net && (!io->value_2 || !io->pi_a_2 || !io->pi_b_2 || !io->wdl_2 || !io->moves_left_2))
        return hm_fail(HM_ERR_INVALID, "native evaluator needs the second set of head buffers");
    const hm_selfplay_config& c = *cfg;
    if (c.games == 0 || c.nodes == 0 || c.max_macro_plies == 0) return hm_fail(HM_ERR_INVALID, "games, nodes, and max-macro-plies must be positive");
    if (c.raw_policy_mean_macro_plies < 0.0 || c.raw_policy_high_temperature_probability < 0.0 || c.raw_policy_high_temperature_probability > 1.0
        || c.mcts_temperature <= 0.0 || c.mcts_temperature_decay <= 0.0 || c.mcts_temperature_decay > 1.0
        || c.node_random_factor < 0.0 || c.node_random_factor >= 1.0 || c.world < 1 || c.rank < 0 || c.rank >= c.world || c.concurrent_games < 1)
        return hm_fail(HM_ERR_INVALID, "Invalid self-play exploration configuration");
    hm_selfplay* s = new hm_selfplay();
    s->cfg = c;
    if (scfg) s->scfg = *scfg; else hm_search_config_default(&s->scfg);
    s->io = *io; s->fn = fn; s->user = user;
    s->G = c.concurrent_games;
    s->runId = c.seed != 0 ? c.seed : static_cast<uint64_t>(std::chrono::system_clock::now().time_since_epoch().count());
    const int maxNodes = (int)std::llround((double)c.nodes * (1.0 + c.node_random_factor)) + 1;
    if (c.max_macro_plies > (1u << 20)) { delete s; return hm_fail(HM_ERR_INVALID, "max_macro_plies too large"); }
    if (int rc = hm_sp_create_ex(s->G, maxNodes, (int)c.max_macro_plies, &s->scfg, &s->sp)) { delete s; return rc; }
    if (hipMalloc(&s->d_boards, sizeof(hm_board) * s->G) != hipSuccess || hipMalloc(&s->d_u8, (size_t)HM_PLANE_VALUES * s->G) != hipSuccess
        || hipMalloc(&s->d_rows[0], sizeof(int32_t) * s->G) != hipSuccess || hipMalloc(&s->d_rows[1], sizeof(int32_t) * s->G) != hipSuccess) {
        hm_sp_destroy(s->sp); delete s; return hm_fail(HM_ERR_NO_DEVICE, "hipMalloc failed");
    }
    s->slots.resize(s->G);
    *out = s;
    return 0;
}

int hm_selfplay_destroy(hm_selfplay* s) {
    if (!s) return 0;
    if (s->sp) hm_sp_destroy(s->sp);
    if (s->d_boards) (void)hipFree(s->d_boards);
    if (s->d_u8) (void)hipFree(s->d_u8);
    if (s->d_rows[0]) (void)hipFree(s->d_rows[0]);
    if (s->d_rows[1]) (void)hipFree(s->d_rows[1]);
    if (s->d_netSel) (void)hipFree(s->d_netSel);
    for (int k = 0; k < 4; ++k) if (s->d_rowsNet[k >> 1][k & 1]) (void)hipFree(s->d_rowsNet[k >> 1][k & 1]);
    if (s->sT) (void)hipStreamDestroy(s->sT);
    if (s->sN && s->sN != s->sT) (void)hipStreamDestroy(s->sN);
    if (s->hActive) (void)hipHostFree(s->hActive);
    if (s->h_u8) (void)hipHostFree(s->h_u8);
    if (s->stepGraph) (void)hipGraphExecDestroy(s->stepGraph);
    if (s->gFork) (void)hipEventDestroy(s->gFork);
    if (s->gJoin) (void)hipEventDestroy(s->gJoin);
    for (hipEvent_t e : s->gCollected) if (e) (void)hipEventDestroy(e);
    for (auto& e : s->evs) if (e) (void)hipEventDestroy(e);
    for (auto& e : s->sync) if (e) (void)hipEventDestroy(e);
    for (hm_time_manager* m : s->timeCtl) if (m) hm_time_manager_destroy(m);
    delete s;
    return 0;
}

static int selfplay_run_impl(hm_selfplay* s);

int hm_selfplay_set_chunk_sink(hm_selfplay* s, hm_chunk_fn fn, void* user) {
    if (!s) return hm_fail(HM_ERR_INVALID, "null argument");
    s->sink = fn; s->sinkUser = user;
    return 0;
}
int hm_selfplay_set_output_directory(hm_selfplay* s, const char* dir) {
    if (!s) return hm_fail(HM_ERR_INVALID, "null argument");
    s->outDir = dir ? dir : "";
    if (!s->outDir.empty()) {                                  // ChunkWriter ctor: create_directories(directory_) :74
        const std::string td = s->outDir + "/training_data";
        std::string cur;
        for (size_t i = 0; i <= td.size(); ++i)
            if (i == td.size() || td[i] == '/') { if (!cur.empty() && ::mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) return hm_fail(HM_ERR_INVALID, "Unable to create " + cur); if (i < td.size()) cur += '/'; }
            else cur += td[i];
    }
    return 0;
}

// leg clock of a run with the native evaluator: cleared at the start, totals added to the result at the end (all streams idle)
static void leg_clock_begin(hm_selfplay* s) { if (s->io.net) (void)hm_sp_leg_times(s->sp, nullptr, nullptr, 1); }
static void leg_clock_end(hm_selfplay* s) {
    if (!s->io.net) return;
    (void)hipDeviceSynchronize();
    double ms[3] = {0.0, 0.0, 0.0}, wait = 0.0;
    uint64_t cnt[3] = {0, 0, 0};
    (void)hm_sp_wait_time(s->sp, &wait);
    if (hm_sp_leg_times(s->sp, ms, cnt, 1)) return;
    s->res.collect_ms += ms[0]; s->res.eval_ms += ms[1]; s->res.process_ms += ms[2];
    if (s->res.persistent_searches) { s->res.search_iterations += cnt[0]; s->res.eval_batches += cnt[0]; s->res.wait_ms += wait; }   // game-iterations
}

// run_selfplay (selfplay.cc:558-748).  Whatever the outcome, the samples of every finished game are handed to the chunk
// sink before returning (ChunkWriter::finish :87-91), so an error late in a run does not discard the games before it.
int hm_selfplay_run(hm_selfplay* s, hm_selfplay_result* out) {
    if (!s) return hm_fail(HM_ERR_INVALID, "null argument");
    const auto t0 = std::chrono::steady_clock::now();
    leg_clock_begin(s);
    const int rc = selfplay_run_impl(s);
    const std::string msg = rc ? std::string(hm_last_error()) : std::string();
    leg_clock_end(s);
    if (has_sink(s) && !s->sampleEnd.empty()) flush_chunk(s, s->sampleEnd.size());
    s->res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    s->res.record_bytes = s->records.size();
    for (int i = 0; i < 5; ++i) s->res.terminations[i] = s->termCounts[i];
    if (out) *out = s->res;
    if (rc) return hm_fail(rc, msg);
    if (s->sinkError) return hm_fail(HM_ERR_STATE, "chunk sink failed (" + std::to_string(s->sinkError) + ")");
    return 0;
}

static int selfplay_run_impl(hm_selfplay* s) {
    const hm_selfplay_config& c = s->cfg;
    const int G = s->G;
    const int E = hm_sp_max_edges(s->sp);
    std::vector<hm_board> boards(G);
    std::vector<int> flags(G), counts(G), term(G);
    std::vector<uint8_t> mask(G);
    const size_t u8bytes = (size_t)G * HM_PLANE_VALUES;
    if (!s->h_u8 && hipHostMalloc(reinterpret_cast<void**>(&s->h_u8), u8bytes, hipHostMallocDefault) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipHostMalloc failed");
    std::vector<hm_move> mA((size_t)G * E), mB((size_t)G * E), actA(G), actB(G);
    std::vector<int> visits((size_t)G * E), target(G), info((size_t)G * HM_SP_INFO_INTS);
    std::vector<float> rootQ(G);
    std::vector<uint64_t> seeds(G);
    std::vector<hm_move> rawMoves((size_t)G * 2 * HM_MAX_MOVES);
    std::vector<float> rawProbs((size_t)G * 2 * HM_MAX_MOVES);
    std::vector<uint8_t> rawCaps((size_t)G * 2 * HM_MAX_MOVES), rawOn((size_t)G * 2);
    std::vector<int> rawCounts((size_t)G * 2);

    // initial games
    {
        std::vector<hm_board> init(G);
        for (int g = 0; g < G; ++g) { hm_board_startpos(&init[g]); mask[g] = start_game(s, s->slots[g], init[g]) ? 1 : 0; }
        if (int rc = hm_sp_set_games(s->sp, init.data(), mask.data())) return rc;
    }
    while (true) {
        // ---- 1. terminal checks at the top of the macro-ply loop (selfplay.cc:607-616, 719-724)
        bool restarted = true;
        while (restarted) {
            restarted = false;
            if (int rc = hm_sp_game_state(s->sp, boards.data(), flags.data(), s->d_boards)) return rc;
            std::vector<hm_board> init(G);
            std::fill(mask.begin(), mask.end(), 0);
            for (int g = 0; g < G; ++g) {
                Slot& sl = s->slots[g];
                if (!sl.active) continue;
                bool done = false;
                if (flags[g] & 1) { sl.winner = sl.team == HM_WHITE ? 1 : 0; sl.termination = 1; done = true; }
                else if (flags[g] & 2) { sl.termination = 2; done = true; }
                else if (sl.macroPly >= c.max_macro_plies) { sl.termination = 0; done = true; }
                if (done) {
                    finish_game(s, sl);
                    if (start_game(s, sl, init[g])) { mask[g] = 1; restarted = true; }
                }
            }
            if (restarted) if (int rc = hm_sp_set_games(s->sp, init.data(), mask.data())) return rc;
        }
        bool any = false;
        for (int g = 0; g < G; ++g) any |= s->slots[g].active;
        if (!any) break;

        // ---- 2. raw-policy opening plies (selfplay.cc:618-643)
        std::vector<uint8_t> movedRaw(G, 0);
        bool anyRaw = false;
        for (int g = 0; g < G; ++g) { Slot& sl = s->slots[g]; anyRaw |= sl.active && sl.rawActive && sl.macroPly < sl.initLength; }
        if (anyRaw) {
            const auto tr0 = std::chrono::steady_clock::now();
            if (int rc = hm_encode_planes(s->d_boards, G, HM_DT_F16, s->io.planes[0], nullptr)) return rc;
            if (int rc = eval_rows_sync(s, 0, G)) return rc;
            s->res.eval_batches += 1;
            if (int rc = hm_sp_raw_policy(s->sp, s->io.pi_a, s->io.pi_b, rawMoves.data(), rawProbs.data(), rawCaps.data(), rawCounts.data(), rawOn.data())) return rc;
            std::fill(actA.begin(), actA.end(), 0); std::fill(actB.begin(), actB.end(), 0);
            for (int g = 0; g < G; ++g) {
                Slot& sl = s->slots[g];
                if (!(sl.active && sl.rawActive && sl.macroPly < sl.initLength)) continue;
                // sample_raw_policy_temperature selfplay.cc:202-216
                std::uniform_real_distribution<double> unit(0.0, 1.0);
                double temperature = 1.0;
                if (!(unit(sl.rng) >= c.raw_policy_high_temperature_probability)) {
                    const double choice = unit(sl.rng);
                    temperature = choice < 0.75 ? 2.0 : (choice < 0.95 ? 5.0 : 10.0);
                }
                std::vector<float> p[2];
                const hm_move* mv[2];
                const uint8_t* cp[2];
                int n[2];
                for (int b = 0; b < 2; ++b) {
                    const size_t base = ((size_t)g * 2 + b) * HM_MAX_MOVES;
                    n[b] = rawCounts[(size_t)g * 2 + b];
                    mv[b] = rawMoves.data() + base; cp[b] = rawCaps.data() + base;
                    p[b].assign(rawProbs.begin() + base, rawProbs.begin() + base + n[b]);
                    if (n[b] > 1) apply_temperature(p[b], temperature);   // prepare_raw_policy :277-300 (single-entry lists are {1.0})
                }
                const bool aOn = rawOn[(size_t)g * 2] != 0, bOn = rawOn[(size_t)g * 2 + 1] != 0;
                const bool aCan = aOn && n[0] > 1, bCan = bOn && n[1] > 1;
                auto valid = [&](int iA, int iB) {   // JointActionCandidate ctor joint_action.h:80-105
                    const bool sitsA = mv[0][iA] == 0, sitsB = mv[1][iB] == 0;
                    if (sitsA && sitsB) return sl.adv && (aOn != bOn);
                    if (sitsA && aCan) return sl.adv || !(aOn && bOn) || cp[1][iB] != 0;
                    if (sitsB && bCan) return sl.adv || !(aOn && bOn) || cp[0][iA] != 0;
                    return true;
                };
                std::discrete_distribution<size_t> sampleA(p[0].begin(), p[0].end());
                std::discrete_distribution<size_t> sampleB(p[1].begin(), p[1].end());
                size_t iA = sampleA(sl.rng), iB = sampleB(sl.rng);
                if (!valid((int)iA, (int)iB)) {
                    std::vector<std::pair<size_t, size_t>> legal;
                    std::vector<double> w;
                    for (size_t a = 0; a < (size_t)n[0]; ++a)
                        for (size_t b2 = 0; b2 < (size_t)n[1]; ++b2)
                            if (valid((int)a, (int)b2)) { legal.emplace_back(a, b2); w.push_back((double)p[0][a] * (double)p[1][b2]); }
                    if (std::accumulate(w.begin(), w.end(), 0.0) <= 0.0) return hm_fail(HM_ERR_STATE, "Raw policy produced no legal joint action");
                    std::discrete_distribution<size_t> ls(w.begin(), w.end());
                    const auto pr = legal[ls(sl.rng)];
                    iA = pr.first; iB = pr.second;
                }
                actA[g] = mv[0][iA]; actB[g] = mv[1][iB];
                movedRaw[g] = 1;
            }
            if (int rc = hm_sp_action_terminal(s->sp, actA.data(), actB.data(), term.data())) return rc;
            for (int g = 0; g < G; ++g) {
                if (!movedRaw[g]) continue;
                Slot& sl = s->slots[g];
                if (term[g]) { sl.rawActive = false; movedRaw[g] = 0; }   // falls through to a searched ply now
                else { sl.rawPlies++; sl.macroPly++; sl.team ^= 1; sl.adv = !sl.adv; }
            }
            if (int rc = hm_sp_apply(s->sp, actA.data(), actB.data(), movedRaw.data())) return rc;
            s->res.raw_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
        }

        // ---- 3. searched plies (selfplay.cc:645-716)
        bool anySearch = false;
        for (int g = 0; g < G; ++g) { mask[g] = s->slots[g].active && !movedRaw[g]; anySearch |= mask[g] != 0; }
        if (!anySearch) continue;
        if (int rc = hm_encode_planes(s->d_boards, G, HM_DT_U8, s->d_u8, nullptr)) return rc;
        // read after the search (the root_stats download is behind it on the same stream): the copy rides beside the search prologue
        if (hipMemcpyAsync(s->h_u8, s->d_u8, u8bytes, hipMemcpyDeviceToHost, nullptr) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipMemcpy failed");
        for (int g = 0; g < G; ++g) {
            target[g] = 1; seeds[g] = 0;
            if (!mask[g]) continue;
            Slot& sl = s->slots[g];
            std::uniform_real_distribution<double> jitter(-c.node_random_factor, c.node_random_factor);   // randomized_node_budget :218-224
            target[g] = (int)std::max<size_t>(1, static_cast<size_t>(std::llround(static_cast<double>(c.nodes) * (1.0 + jitter(sl.rng)))));
            seeds[g] = mix_seed(s->runId, sl.gameIndex * c.max_macro_plies + sl.macroPly);
        }
        const auto tp0 = std::chrono::steady_clock::now();
        if (int rc = hm_sp_begin_search(s->sp, target.data(), seeds.data(), c.dirichlet_alpha, c.dirichlet_epsilon, mask.data())) return rc;
        int minTarget = 1 << 30;
        for (int g = 0; g < G; ++g) if (mask[g] && target[g] < minTarget) minTarget = target[g];
        const auto ts0 = std::chrono::steady_clock::now();
        s->res.prologue_seconds += std::chrono::duration<double>(ts0 - tp0).count();
        if (int rc = run_search_lockstep(s, minTarget)) return rc;
        s->res.search_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts0).count();
        if (int rc = hm_sp_root_stats(s->sp, counts.data(), mA.data(), mB.data(), visits.data(), nullptr, nullptr, rootQ.data(), info.data(), E)) return rc;
        std::fill(actA.begin(), actA.end(), 0); std::fill(actB.begin(), actB.end(), 0);
        std::vector<uint8_t> applyMask(G, 0);
        std::vector<std::pair<uint16_t, uint64_t>> margin;
        margin.reserve(512);
        for (int g = 0; g < G; ++g) {
            if (!mask[g]) continue;
            Slot& sl = s->slots[g];
            if (info[(size_t)g * HM_SP_INFO_INTS + 8]) return hm_fail(HM_ERR_OVERFLOW, "search pool overflow in game slot " + std::to_string(g) + " (flags " + std::to_string(info[(size_t)g * HM_SP_INFO_INTS + 8]) + ")");
            s->res.searched_positions += 1;
            s->res.eval_rows += (uint64_t)info[(size_t)g * HM_SP_INFO_INTS + 2];
            s->res.nodes_visited += (uint64_t)info[(size_t)g * HM_SP_INFO_INTS + 10];
            s->res.edges_scanned += (uint64_t)info[(size_t)g * HM_SP_INFO_INTS + 11];
            s->res.leaf_move_words += (uint64_t)info[(size_t)g * HM_SP_INFO_INTS + 13];
            s->res.tt_hits += (uint64_t)info[(size_t)g * HM_SP_INFO_INTS + 18];
            s->res.tt_inserts += (uint64_t)info[(size_t)g * HM_SP_INFO_INTS + 19];
            const int n = counts[g];
            if (n == 0) { sl.winner = sl.team == HM_WHITE ? 1 : 0; sl.termination = 4; finish_game(s, sl); continue; }
            const hm_move* ea = mA.data() + (size_t)g * E;
            const hm_move* eb = mB.data() + (size_t)g * E;
            const int* ev = visits.data() + (size_t)g * E;
            Sample sm;
            sm.gameId = sl.gameIndex;
            sm.macroPly = static_cast<uint16_t>(std::min<size_t>(sl.macroPly, 65535));
            sm.team = sl.team == HM_WHITE ? 0 : 1;
            sm.hasTimeAdvantage = sl.adv ? 1 : 0;
            std::memcpy(sm.planes.data(), s->h_u8 + (size_t)g * HM_PLANE_VALUES, HM_PLANE_VALUES);
            uint64_t actual = 0;
            for (int i = 0; i < n; ++i) actual += (uint64_t)std::max(0, ev[i]);
            sm.nodes = static_cast<uint32_t>(std::min<uint64_t>(actual, 0xffffffffu));
            sm.rootQ = rootQ[g];
            for (int b = 0; b < 2; ++b) {   // marginal_policy :400-427 (an ordered map from policy index to visits there: here the pairs sorted by index, equal indices merged)
                margin.clear();
                uint64_t total = 0;
                for (int i = 0; i < n; ++i) {
                    if (ev[i] <= 0) continue;
                    const hm_move mv = b == 0 ? ea[i] : eb[i];
                    const int idx = hm_policy_index(mv, boards[g].pos[b].stm);
                    if (idx < 0) return hm_fail(HM_ERR_STATE, "Move is absent from policy map");
                    margin.emplace_back(static_cast<uint16_t>(idx), (uint64_t)ev[i]);
                    total += (uint64_t)ev[i];
                }
                if (total == 0) return hm_fail(HM_ERR_STATE, "Search returned no visited root edges");
                std::sort(margin.begin(), margin.end(), [](const std::pair<uint16_t, uint64_t>& x, const std::pair<uint16_t, uint64_t>& y) { return x.first < y.first; });
                auto& pol = b == 0 ? sm.policyA : sm.policyB;
                for (size_t i = 0; i < margin.size();) {
                    uint64_t v = 0;
                    size_t j = i;
                    for (; j < margin.size() && margin[j].first == margin[i].first; ++j) v += margin[j].second;
                    pol.push_back({margin[i].first, static_cast<float>(v) / static_cast<float>(total)});
                    i = j;
                }
            }
            sl.samples.push_back(std::move(sm));
            const size_t teamIdx = sl.team == HM_WHITE ? 0 : 1;
            if (sl.canResign) {   // :681-692
                if (rootQ[g] <= c.resign_threshold) {
                    if (++sl.resignPlies[teamIdx] >= c.resign_consecutive_plies) {
                        sl.winner = sl.team == HM_WHITE ? 1 : 0; sl.termination = 3; finish_game(s, sl); continue;
                    }
                } else sl.resignPlies[teamIdx] = 0;
            }
            // select_action :429-462 with mcts_temperature :226-232
            double temperature = (c.mcts_temperature_plies > 0 && sl.macroPly >= c.mcts_temperature_plies) ? 0.0
                : c.mcts_temperature * std::pow(c.mcts_temperature_decay, static_cast<double>(sl.macroPly / 2));
            int pick = 0;
            int maxVisits = ev[0], argmax = 0;
            for (int i = 1; i < n; ++i) if (ev[i] > maxVisits) { maxVisits = ev[i]; argmax = i; }
            if (temperature <= 1e-6) pick = argmax;
            else {
                std::vector<double> w(n);
                for (int i = 0; i < n; ++i)
                    w[i] = (ev[i] > 0 && maxVisits > 0) ? std::exp((std::log((double)ev[i]) - std::log((double)maxVisits)) / temperature) : 0.0;
                if (std::accumulate(w.begin(), w.end(), 0.0) <= 0.0) pick = 0;
                else { std::discrete_distribution<size_t> d(w.begin(), w.end()); pick = (int)d(sl.rng); }
            }
            actA[g] = ea[pick]; actB[g] = eb[pick];
            applyMask[g] = 1;
            sl.macroPly++; sl.team ^= 1; sl.adv = !sl.adv;
        }
        if (int rc = hm_sp_apply_deferred(s->sp, actA.data(), actB.data(), applyMask.data())) return rc;
        // slots whose game just ended (resignation / no action) restart at the top of the loop
        {
            std::vector<hm_board> init(G);
            std::fill(mask.begin(), mask.end(), 0);
            bool r = false;
            for (int g = 0; g < G; ++g) {
                Slot& sl = s->slots[g];
                if (!sl.active && !applyMask[g] && !movedRaw[g]) { if (start_game(s, sl, init[g])) { mask[g] = 1; r = true; } }
            }
            if (r) if (int rc = hm_sp_set_games(s->sp, init.data(), mask.data())) return rc;
        }
    }
    return 0;
}

uint64_t hm_selfplay_records(hm_selfplay* s, const uint8_t** data, uint64_t* count) {
    if (!s) return 0;
    if (data) *data = s->records.data();
    if (count) *count = s->recordCount;
    return s->records.size();
}

// ChunkWriter::flush header (selfplay.cc:118-124) + the serialized samples
int hm_hvm4_write_chunk(const char* path, const uint8_t* records, uint64_t nbytes, uint64_t count) {
    if (!path || (!records && nbytes)) return hm_fail(HM_ERR_INVALID, "null argument");
    const std::string tmp = std::string(path) + ".tmp";
    std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
    if (!f) return hm_fail(HM_ERR_INVALID, "Unable to create " + tmp);
    const char magic[4] = {'H', 'V', 'M', '4'};
    const uint32_t version = 4;
    const uint16_t ch = HM_NB_PLANES, pol = HM_POLICY_VALUES;
    f.write(magic, 4);
    f.write(reinterpret_cast<const char*>(&version), 4);
    f.write(reinterpret_cast<const char*>(&ch), 2);
    f.write(reinterpret_cast<const char*>(&pol), 2);
    f.write(reinterpret_cast<const char*>(&count), 8);
    f.write(reinterpret_cast<const char*>(records), (std::streamsize)nbytes);
    f.close();
    if (!f) return hm_fail(HM_ERR_INVALID, "Failed to finalize " + tmp);
    if (std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return hm_fail(HM_ERR_INVALID, std::string("Unable to publish ") + path); }
    return 0;
}

// =======================================================================================
// paired network tournament (tools/tournament.cc:328-465) on the lockstep rollout core
// =======================================================================================
}  // extern "C"

namespace {

struct TGame {                      // one finished (or running) tournament game
    int winner = -1;                // -1 none, HM_WHITE / HM_BLACK
    int termination = 3;            // 0 checkmate, 1 no legal action, 2 draw, 3 macro-ply limit
    std::vector<std::string> actions;
    bool done = false;
};
struct TSlot {
    bool active = false;
    size_t gameIndex = 0, macroPly = 0;
    int contenderTeam = 0, team = 0;
    bool adv = false;
};
const char* const kTermination[4] = {"checkmate", "no legal action", "draw", "macro-ply limit"};

uint64_t tournament_seed(uint64_t seed, uint64_t value) {   // tournament.cc:22-27
    value += 0x9e3779b97f4a7c15ULL;
    value = (value ^ (value >> 30)) * 0xbf58476d1ce4e5b9ULL;
    value = (value ^ (value >> 27)) * 0x94d049bb133111ebULL;
    return seed ^ (value ^ (value >> 31));
}
// UCI::move (Fairy-Stockfish/src/stubs.cpp:21-59) through Board::uci_move (board.h:340-350) for this variant
std::string uci_move(hm_move m) {
    if (m == 0) return "pass";
    auto sq = [](int q) { std::string r; r += char('a' + (q & 7)); r += char('1' + (q >> 3)); return r; };
    const int from = (int)((m >> 6) & 63);
    int to = (int)(m & 63);
    const uint32_t mt = m & (15u << 12);
    if (mt == HM_MT_CASTLING) to = (to > from ? 6 : 2) + (from & 56);      // king -> rook square is printed king -> g / c file
    std::string r;
    if (mt == HM_MT_DROP) { r += " PNBRQ"[(m >> 16) & 7]; r += '@'; }
    else r += sq(from);
    r += sq(to);
    if (mt == HM_MT_PROMOTION) r += " pnbrq"[(m >> 16) & 7];
    return r;
}
void fill_statistics(hm_tournament_result& r, const std::vector<double>& pairScores) {   // tournament.cc:247-326
    const uint64_t games = r.contender_wins + r.baseline_wins + r.draws;
    r.pairs = pairScores.size();
    r.contender_score = games == 0 ? 0.0 : ((double)r.contender_wins + 0.5 * (double)r.draws) / (double)games;
    const double score = r.contender_score;
    r.has_elo = !(games == 0 || score <= 0.0 || score >= 1.0);
    r.contender_elo = r.has_elo ? 400.0 * std::log10(score / (1.0 - score)) : 0.0;
    r.paired_method = pairScores.size() >= 2;
    r.has_score_ci = games != 0;
    r.score_ci[0] = r.score_ci[1] = 0.0;
    if (r.has_score_ci) {
        constexpr double z = 1.959963984540054;
        if (pairScores.size() >= 2) {
            const double count = (double)pairScores.size();
            const double mean = std::accumulate(pairScores.begin(), pairScores.end(), 0.0) / count;
            const double squaredError = std::accumulate(pairScores.begin(), pairScores.end(), 0.0,
                [mean](double total, double sc) { const double d = sc - mean; return total + d * d; });
            const double sampleVariance = squaredError / (count - 1.0);
            const double margin = z * std::sqrt(sampleVariance / count);
            r.score_ci[0] = std::max(0.0, mean - margin); r.score_ci[1] = std::min(1.0, mean + margin);
        } else {
            const double count = (double)games;
            const double denominator = 1.0 + z * z / count;
            const double center = (score + z * z / (2.0 * count)) / denominator;
            const double margin = z * std::sqrt(score * (1.0 - score) / count + z * z / (4.0 * count * count)) / denominator;
            r.score_ci[0] = std::max(0.0, center - margin); r.score_ci[1] = std::min(1.0, center + margin);
        }
    }
    r.has_elo_ci = r.has_score_ci && !(r.score_ci[0] <= 0.0 || r.score_ci[1] >= 1.0);
    auto score_to_elo = [](double sc) { return 400.0 * std::log10(sc / (1.0 - sc)); };
    r.elo_ci[0] = r.has_elo_ci ? score_to_elo(r.score_ci[0]) : 0.0;
    r.elo_ci[1] = r.has_elo_ci ? score_to_elo(r.score_ci[1]) : 0.0;
}

}  // namespace

struct hm_tournament {
    hm_tournament_config cfg;
    hm_selfplay* core = nullptr;    // lockstep engine: game slots, search pools, streams, captured iteration graph
    std::vector<TSlot> slots;
    std::vector<TGame> games;
    std::vector<uint8_t> acting;    // per slot: 1 = the contender's network evaluates the search in flight
    std::vector<double> pairScores;
    size_t nextGame = 0;
    int poolNodes = 0;              // node target handed to every search: the node budget, or the pool size of a time-limited search
    hm_tournament_result res{};
};

static bool tournament_start(hm_tournament* t, TSlot& sl, hm_board& out) {
    if (t->nextGame >= t->cfg.games) { sl.active = false; return false; }
    sl = TSlot();
    sl.active = true;
    sl.gameIndex = t->nextGame++;
    const size_t pairIndex = sl.gameIndex / 2;
    sl.contenderTeam = sl.gameIndex % 2 == 0 ? HM_WHITE : HM_BLACK;          // tournament.cc:371-376
    sl.team = pairIndex % 2 == 0 ? HM_WHITE : HM_BLACK;
    sl.adv = false;
    hm_board_startpos(&out);
    out.team = (uint8_t)sl.team;
    out.time_adv = 0;
    return true;
}

static int tournament_run_impl(hm_tournament* t) {
    hm_selfplay* s = t->core;
    const hm_tournament_config& c = t->cfg;
    const int G = s->G;
    const int E = hm_sp_max_edges(s->sp);
    std::vector<hm_board> boards(G), init(G);
    std::vector<int> flags(G), counts(G), target(G), info((size_t)G * HM_SP_INFO_INTS), visits((size_t)G * E);
    std::vector<uint8_t> mask(G), profile(G), applyMask(G), batchSz(G, 8);
    std::vector<hm_move> mA((size_t)G * E), mB((size_t)G * E), actA(G), actB(G);
    std::vector<float> rootQ(G);
    std::vector<uint64_t> seeds(G);
    t->games.assign(c.games, TGame());
    t->slots.assign(G, TSlot());
    t->acting.assign(G, 0);
    t->nextGame = 0;
    auto finish = [&](TSlot& sl, int winner, int termination) {
        TGame& g = t->games[sl.gameIndex];
        g.winner = winner; g.termination = termination; g.done = true;
        sl.active = false;
    };
    for (int g = 0; g < G; ++g) { hm_board_startpos(&init[g]); mask[g] = tournament_start(t, t->slots[g], init[g]) ? 1 : 0; }
    if (int rc = hm_sp_set_games(s->sp, init.data(), mask.data())) return rc;
    while (true) {
        // top of the macro-ply loop (tournament.cc:383-392) and its exit (:367, :417-422): checkmate, draw, macro-ply limit
        bool restarted = true;
        while (restarted) {
            restarted = false;
            if (int rc = hm_sp_game_state(s->sp, boards.data(), flags.data(), s->d_boards)) return rc;
            std::fill(mask.begin(), mask.end(), 0);
            for (int g = 0; g < G; ++g) {
                TSlot& sl = t->slots[g];
                if (!sl.active) continue;
                bool done = true;
                if (flags[g] & 1) finish(sl, sl.team ^ 1, 0);
                else if (flags[g] & 2) finish(sl, -1, 2);
                else if (sl.macroPly >= c.max_macro_plies) finish(sl, -1, 3);
                else done = false;
                if (done && tournament_start(t, sl, init[g])) { mask[g] = 1; restarted = true; }
            }
            if (restarted) if (int rc = hm_sp_set_games(s->sp, init.data(), mask.data())) return rc;
        }
        bool any = false;
        for (int g = 0; g < G; ++g) { mask[g] = t->slots[g].active; any |= mask[g] != 0; }
        if (!any) break;
        // one search per live game by the network of the team to move (:394-409)
        for (int g = 0; g < G; ++g) {
            const TSlot& sl = t->slots[g];
            const bool contenderActing = sl.active && sl.team == sl.contenderTeam;
            t->acting[g] = contenderActing ? 1 : 0;
            profile[g] = contenderActing ? 0 : 1;                              // schedule 0 = contender's PW coefficient, 1 = baseline's
            target[g] = t->poolNodes;                                          // the node budget, or the pool size of a time-limited search
            seeds[g] = sl.active ? tournament_seed(c.seed, (sl.gameIndex / 2) * c.max_macro_plies + sl.macroPly) : 0;
        }
        if (int rc = hm_sp_set_pw_profiles(s->sp, c.baseline_pw_coefficient, c.baseline_pw_coefficient, profile.data())) return rc;
        if (c.contender_batch_size != 8 || c.baseline_batch_size != 8) {      // each network searches with its own batch size (tournament.h:19-20, searchthread.cc:663)
            for (int g = 0; g < G; ++g) batchSz[g] = (uint8_t)(profile[g] ? c.baseline_batch_size : c.contender_batch_size);
            if (int rc = hm_sp_set_batch_sizes(s->sp, batchSz.data())) return rc;
        }
        if (s->d_netSel && hipMemcpy(s->d_netSel, profile.data(), (size_t)G, hipMemcpyHostToDevice) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipMemcpy failed");
        if (int rc = hm_sp_begin_search(s->sp, target.data(), seeds.data(), c.dirichlet_alpha, c.dirichlet_epsilon, mask.data())) return rc;
        if (int rc = run_search_lockstep(s, c.move_time_ms > 0 ? 0 : (int)c.nodes)) return rc;
        if (int rc = hm_sp_root_stats(s->sp, counts.data(), mA.data(), mB.data(), visits.data(), nullptr, nullptr, rootQ.data(), info.data(), E)) return rc;
        std::fill(actA.begin(), actA.end(), 0); std::fill(actB.begin(), actB.end(), 0); std::fill(applyMask.begin(), applyMask.end(), 0);
        for (int g = 0; g < G; ++g) {
            if (!mask[g]) continue;
            TSlot& sl = t->slots[g];
            if (info[(size_t)g * HM_SP_INFO_INTS + 8]) return hm_fail(HM_ERR_OVERFLOW, "search pool overflow in game slot " + std::to_string(g) + " (flags " + std::to_string(info[(size_t)g * HM_SP_INFO_INTS + 8]) + ")");
            t->res.searched_positions += 1;
            const int n = counts[g];
            if (n == 0) { finish(sl, sl.team ^ 1, 1); continue; }                     // no legal action (:403-407)
            const int* ev = visits.data() + (size_t)g * E;
            int best = 0;
            for (int i = 1; i < n; ++i) if (ev[best] < ev[i]) best = i;                // most_visited_action: first maximum (:29-39)
            for (int i = 0; i < n; ++i) t->res.total_nodes += (uint64_t)std::max(0, ev[i]);
            actA[g] = mA[(size_t)g * E + best]; actB[g] = mB[(size_t)g * E + best];
            t->games[sl.gameIndex].actions.push_back("(" + uci_move(actA[g]) + "," + uci_move(actB[g]) + ")");   // action_uci :41-49
            applyMask[g] = 1;
            sl.macroPly++; sl.team ^= 1; sl.adv = !sl.adv;
        }
        if (int rc = hm_sp_apply(s->sp, actA.data(), actB.data(), applyMask.data())) return rc;
        // slots whose game ended without a move restart at the top of the loop
        {
            std::fill(mask.begin(), mask.end(), 0);
            bool r = false;
            for (int g = 0; g < G; ++g) {
                TSlot& sl = t->slots[g];
                if (!sl.active && !applyMask[g]) { if (tournament_start(t, sl, init[g])) { mask[g] = 1; r = true; } }
            }
            if (r) if (int rc = hm_sp_set_games(s->sp, init.data(), mask.data())) return rc;
        }
    }
    return 0;
}

// result bookkeeping in game order (tournament.cc:424-452)
static void tournament_account(hm_tournament* t) {
    hm_tournament_result& r = t->res;
    t->pairScores.clear();
    double currentPairPoints = 0.0;
    for (size_t i = 0; i < t->games.size(); ++i) {
        const TGame& g = t->games[i];
        if (!g.done) break;
        const size_t pairIndex = i / 2;
        const int contenderTeam = i % 2 == 0 ? HM_WHITE : HM_BLACK;
        const int startTeam = pairIndex % 2 == 0 ? HM_WHITE : HM_BLACK;
        const bool contenderHasTimeAdvantage = contenderTeam != startTeam;
        int outcome = 0;
        if (g.winner < 0) r.draws++;
        else if (g.winner == contenderTeam) { r.contender_wins++; outcome = 1; }
        else { r.baseline_wins++; outcome = -1; }
        auto record = [&](hm_tournament_breakdown& b) { if (outcome > 0) b.wins++; else if (outcome < 0) b.losses++; else b.draws++; };
        record(contenderTeam == HM_WHITE ? r.as_white : r.as_black);
        record(contenderHasTimeAdvantage ? r.up_time : r.down_time);
        currentPairPoints += outcome > 0 ? 1.0 : outcome == 0 ? 0.5 : 0.0;
        if (i % 2 == 1) { t->pairScores.push_back(currentPairPoints / 2.0); currentPairPoints = 0.0; }
        if (g.termination == 0) r.checkmates++;
        else if (g.termination == 1) r.no_legal_actions++;
        else if (g.termination == 2) r.drawn_terminations++;
        else r.macro_ply_limits++;
    }
    fill_statistics(r, t->pairScores);
}

static std::string tournament_summary_text(const hm_tournament* t, const std::string& contenderName, const std::string& baselineName) {   // write_summary :117-187
    const hm_tournament_config& c = t->cfg;
    const hm_tournament_result& r = t->res;
    std::ostringstream o;
    o << std::fixed << std::setprecision(6) << "{\n"
      << "  \"contender\": \"" << contenderName << "\",\n" << "  \"baseline\": \"" << baselineName << "\",\n"
      << "  \"games\": " << (r.contender_wins + r.baseline_wins + r.draws) << ",\n" << "  \"nodes_per_move\": " << c.nodes << ",\n"
      << "  \"move_time_ms\": " << c.move_time_ms << ",\n" << "  \"contender_batch_size\": " << c.contender_batch_size << ",\n"
      << "  \"baseline_batch_size\": " << c.baseline_batch_size << ",\n" << "  \"seed\": " << c.seed << ",\n"
      << "  \"contender_pw_coefficient\": " << c.contender_pw_coefficient << ",\n"
      << "  \"baseline_pw_coefficient\": " << c.baseline_pw_coefficient << ",\n"
      << "  \"contender_wins\": " << r.contender_wins << ",\n" << "  \"baseline_wins\": " << r.baseline_wins << ",\n"
      << "  \"draws\": " << r.draws << ",\n" << "  \"contender_score\": " << r.contender_score << ",\n" << "  \"contender_elo\": ";
    if (r.has_elo) o << r.contender_elo; else o << "null";
    o << ",\n  \"confidence_method\": \"" << (r.paired_method ? "paired-opening normal approximation" : "game-level Wilson approximation") << "\",\n"
      << "  \"score_confidence_95\": ";
    if (r.has_score_ci) o << '[' << r.score_ci[0] << ", " << r.score_ci[1] << ']'; else o << "null";
    o << ",\n  \"elo_confidence_95\": ";
    if (r.has_elo_ci) o << '[' << r.elo_ci[0] << ", " << r.elo_ci[1] << ']'; else o << "null";
    o << ",\n  \"contender_breakdown\": {\n";
    auto bd = [&](const char* name, const hm_tournament_breakdown& b, bool comma) {
        o << "    \"" << name << "\": {\"wins\": " << b.wins << ", \"losses\": " << b.losses << ", \"draws\": " << b.draws << "}" << (comma ? "," : "") << '\n';
    };
    bd("white", r.as_white, true); bd("black", r.as_black, true); bd("up_time", r.up_time, true); bd("down_time", r.down_time, false);
    o << "  },\n" << "  \"terminations\": {\n" << "    \"checkmate\": " << r.checkmates << ",\n"
      << "    \"no_legal_action\": " << r.no_legal_actions << ",\n" << "    \"draw\": " << r.drawn_terminations << ",\n"
      << "    \"macro_ply_limit\": " << r.macro_ply_limits << "\n" << "  }\n" << "}\n";
    return o.str();
}
static std::string tournament_pgn_text(const hm_tournament* t, const std::string& contenderName, const std::string& baselineName) {   // append_game_pgn :89-115
    std::ostringstream o;
    for (size_t i = 0; i < t->games.size(); ++i) {
        const TGame& g = t->games[i];
        if (!g.done) break;
        const int contenderTeam = i % 2 == 0 ? HM_WHITE : HM_BLACK;
        const std::string result = g.winner == HM_WHITE ? "1-0" : g.winner == HM_BLACK ? "0-1" : "1/2-1/2";
        o << "[Event \"Hivemind Network Tournament\"]\n" << "[Site \"Hivemind Engine\"]\n" << "[Round \"" << (i + 1) << "\"]\n"
          << "[Variant \"bughouse\"]\n" << "[WhiteTeam \"" << (contenderTeam == HM_WHITE ? contenderName : baselineName) << "\"]\n"
          << "[BlackTeam \"" << (contenderTeam == HM_BLACK ? contenderName : baselineName) << "\"]\n"
          << "[Result \"" << result << "\"]\n" << "[Termination \"" << kTermination[g.termination] << "\"]\n\n";
        for (size_t k = 0; k < g.actions.size(); ++k) o << (k + 1) << ". " << g.actions[k] << ' ';
        o << result << "\n\n";
    }
    return o.str();
}

extern "C" {

void hm_tournament_config_default(hm_tournament_config* c) {   // tools/tournament.h:15-27
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->max_search_nodes = 0;
    c->games = 20; c->nodes = 400; c->move_time_ms = 0; c->contender_batch_size = 8; c->baseline_batch_size = 8;
    c->max_macro_plies = 400; c->dirichlet_alpha = 0.3f; c->dirichlet_epsilon = 0.10f;
    c->contender_pw_coefficient = 2.0f; c->baseline_pw_coefficient = 2.0f; c->seed = 1; c->concurrent_games = 64;
}

int hm_tournament_create(const hm_tournament_config* cfg, const hm_search_config* scfg, const hm_eval_io* io, const hm_net* baseline_net,
                         hm_eval_fn fn, void* user, hm_tournament** out) {
    if (!cfg || !io || !out) return hm_fail(HM_ERR_INVALID, "null argument");
    const hm_tournament_config& c = *cfg;
    // run_tournament's argument checks, same texts (tournament.cc:334-358)
    if (c.games == 0 || c.games % 2 != 0) return hm_fail(HM_ERR_INVALID, "Tournament games must be a positive even number");
    if ((c.nodes == 0) == (c.move_time_ms <= 0) || c.max_macro_plies == 0) return hm_fail(HM_ERR_INVALID, "Tournament requires exactly one positive nodes or movetime limit");
    if (c.contender_batch_size <= 0 || c.baseline_batch_size <= 0) return hm_fail(HM_ERR_INVALID, "Tournament batch sizes must be positive");
    if (c.dirichlet_alpha < 0.0f || c.dirichlet_epsilon < 0.0f || c.dirichlet_epsilon > 1.0f) return hm_fail(HM_ERR_INVALID, "Invalid tournament Dirichlet configuration");
    if (!std::isfinite(c.contender_pw_coefficient) || !std::isfinite(c.baseline_pw_coefficient) || c.contender_pw_coefficient <= 0.0f || c.baseline_pw_coefficient <= 0.0f)
        return hm_fail(HM_ERR_INVALID, "Tournament PW coefficients must be positive and finite");
    // what this engine does not build
    if (c.move_time_ms > 0 && !io->net) return hm_fail(HM_ERR_INVALID, "a movetime tournament is built for the native evaluator only (two networks)");
    if (c.move_time_ms > 0 && c.max_search_nodes < 0) return hm_fail(HM_ERR_INVALID, "max_search_nodes must not be negative");
    if (c.contender_batch_size > 8 || c.baseline_batch_size > 8) return hm_fail(HM_ERR_INVALID, "batch sizes above 8 are not built");
    if (c.concurrent_games < 1) return hm_fail(HM_ERR_INVALID, "concurrent_games must be positive");
    if ((io->net != nullptr) != (baseline_net != nullptr)) return hm_fail(HM_ERR_INVALID, "give both networks, or a callback that serves both");
    hm_selfplay_config sc;
    hm_selfplay_config_default(&sc);
    // a time-limited search has no node budget: its pool holds max_search_nodes (default 4096) and the search ends there at the latest
    const uint64_t poolNodes = c.move_time_ms > 0 ? (uint64_t)(c.max_search_nodes > 0 ? c.max_search_nodes : 4096) : c.nodes;
    sc.games = c.games; sc.nodes = poolNodes; sc.max_macro_plies = c.max_macro_plies; sc.node_random_factor = 0.0;
    sc.seed = c.seed ? c.seed : 1; sc.concurrent_games = (int)std::min<uint64_t>((uint64_t)c.concurrent_games, c.games);
    hm_search_config search;
    if (scfg) search = *scfg; else hm_search_config_default(&search);
    search.pw_coefficient = c.contender_pw_coefficient;                  // TournamentConfig::searchConfigFor (tournament.h:34-41):
    search.root_pw_coefficient = c.contender_pw_coefficient;             // one coefficient for root and interior nodes
    hm_tournament* t = new hm_tournament();
    t->cfg = c;
    t->cfg.concurrent_games = sc.concurrent_games;
    if (int rc = hm_selfplay_create(&sc, &search, io, fn, user, &t->core)) { delete t; return rc; }
    hm_selfplay* s = t->core;
    s->moveTimeMs = c.move_time_ms > 0 ? c.move_time_ms : 0;
    t->poolNodes = (int)poolNodes;
    if (baseline_net) {
        s->net2 = baseline_net;
        bool ok = hipMalloc(&s->d_netSel, (size_t)s->G) == hipSuccess;
        for (int k = 0; k < 4 && ok; ++k) ok = hipMalloc(&s->d_rowsNet[k >> 1][k & 1], sizeof(int32_t) * s->G) == hipSuccess;
        if (!ok) { hm_selfplay_destroy(s); delete t; return hm_fail(HM_ERR_NO_DEVICE, "hipMalloc failed"); }
        (void)hipMemset(s->d_netSel, 0, (size_t)s->G);
    }
    *out = t;
    return 0;
}

int hm_tournament_run(hm_tournament* t, hm_tournament_result* out) {
    if (!t) return hm_fail(HM_ERR_INVALID, "null argument");
    const auto t0 = std::chrono::steady_clock::now();
    t->res = hm_tournament_result{};
    t->core->res = hm_selfplay_result{};
    leg_clock_begin(t->core);
    const int rc = tournament_run_impl(t);
    const std::string msg = rc ? std::string(hm_last_error()) : std::string();
    leg_clock_end(t->core);
    tournament_account(t);
    t->res.search_iterations = t->core->res.search_iterations;
    t->res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (out) *out = t->res;
    if (rc) return hm_fail(rc, msg);
    return 0;
}

int hm_tournament_acting(const hm_tournament* t, uint8_t* acting) {
    if (!t || !acting) return hm_fail(HM_ERR_INVALID, "null argument");
    std::memcpy(acting, t->acting.data(), t->acting.size());
    return 0;
}
uint64_t hm_tournament_pair_scores(const hm_tournament* t, const double** scores) {
    if (!t) return 0;
    if (scores) *scores = t->pairScores.data();
    return t->pairScores.size();
}
static int64_t copy_text(const std::string& text, char* out, int64_t cap) {
    if ((int64_t)text.size() + 1 > cap || !out) return -(int64_t)text.size() - 1;
    std::memcpy(out, text.c_str(), text.size() + 1);
    return (int64_t)text.size();
}
int64_t hm_tournament_summary(const hm_tournament* t, const char* contender_name, const char* baseline_name, char* out, int64_t cap) {
    if (!t || !contender_name || !baseline_name) return 0;
    return copy_text(tournament_summary_text(t, contender_name, baseline_name), out, cap);
}
int64_t hm_tournament_pgn(const hm_tournament* t, const char* contender_name, const char* baseline_name, char* out, int64_t cap) {
    if (!t || !contender_name || !baseline_name) return 0;
    return copy_text(tournament_pgn_text(t, contender_name, baseline_name), out, cap);
}
int hm_tournament_write_reports(const hm_tournament* t, const char* dir, const char* contender_name, const char* baseline_name) {
    if (!t || !dir || !contender_name || !baseline_name) return hm_fail(HM_ERR_INVALID, "null argument");
    std::error_code ec;
    std::filesystem::create_directories(dir, ec);
    if (ec) return hm_fail(HM_ERR_INVALID, std::string("Unable to create ") + dir);
    const std::string base(dir);
    {
        std::ofstream f(base + "/games.pgn", std::ios::trunc);
        if (!f) return hm_fail(HM_ERR_INVALID, "Unable to append tournament PGN: " + base + "/games.pgn");
        f << tournament_pgn_text(t, contender_name, baseline_name);
    }
    const std::string path = base + "/summary.json", tmp = path + ".tmp";
    {
        std::ofstream f(tmp, std::ios::trunc);
        if (!f) return hm_fail(HM_ERR_INVALID, "Unable to write tournament summary: " + path);
        f << tournament_summary_text(t, contender_name, baseline_name);
        f.close();
        if (!f) return hm_fail(HM_ERR_INVALID, "Failed to finalize tournament summary: " + path);
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) return hm_fail(HM_ERR_INVALID, "Unable to publish " + path);
    return 0;
}
int hm_tournament_destroy(hm_tournament* t) {
    if (!t) return 0;
    if (t->core) hm_selfplay_destroy(t->core);
    delete t;
    return 0;
}
int hm_tournament_statistics(uint64_t contender_wins, uint64_t baseline_wins, uint64_t draws, const double* pair_scores, uint64_t pairs,
                             hm_tournament_result* out) {
    if (!out || (pairs && !pair_scores)) return hm_fail(HM_ERR_INVALID, "null argument");
    *out = hm_tournament_result{};
    out->contender_wins = contender_wins; out->baseline_wins = baseline_wins; out->draws = draws;
    fill_statistics(*out, std::vector<double>(pair_scores, pair_scores + pairs));
    return 0;
}
int hm_move_uci(hm_move move, char* out, int cap) {
    if (!out || cap <= 0) return hm_fail(HM_ERR_INVALID, "null argument");
    const std::string text = uci_move(move);
    std::snprintf(out, (size_t)cap, "%s", text.c_str());
    return (int)text.size();
}

}  // extern "C"
