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
//   * PGN text is not produced (not consumed by training).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <fstream>
#include <sys/stat.h>
#include <map>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../../include/hivemind_amd.h"

int hm_fail(int code, const std::string& msg);
extern "C" int hm_sp_active(hm_sp* sp, int* active);
extern "C" int hm_sp_active_on(hm_sp* sp, int* pinned_out, hipStream_t stream);

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
    std::vector<uint64_t> termCounts = std::vector<uint64_t>(5, 0);
    std::vector<hipEvent_t> evs, sync;                 // ring of per-iteration leg events
    hipStream_t sT = nullptr, sN = nullptr;      // tree / network streams (native evaluator mode)
    int* hActive = nullptr;
    // K lockstep iterations captured once as a HIP graph (collect || forward -> process, K times): one launch replays them,
    // so the dependent launches and cross-stream waits of the inner loop cost graph edges instead of host round trips
    hipGraphExec_t stepGraph = nullptr;
    hipEvent_t gFork = nullptr, gJoin = nullptr;
    int graphState = 0;                            // 0 not tried, 1 ready, -1 unavailable (eager loop)
    int32_t* d_rows[2] = {nullptr, nullptr};   // per game slot: plane rows written into planes[k] (ragged evaluator batch)
};

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
static int enqueue_iteration(hm_selfplay* s, int which, int parity, hipEvent_t fork, hipEvent_t join, bool allRows) {
    void* hv[2][5] = {{s->io.value, s->io.pi_a, s->io.pi_b, s->io.wdl, s->io.moves_left},
                      {s->io.value_2, s->io.pi_a_2, s->io.pi_b_2, s->io.wdl_2, s->io.moves_left_2}};
    void** h = hv[parity];
    (void)hipEventRecord(fork, s->sT);             // forward(i) runs behind process(i-1), beside collect(i)
    (void)hipStreamWaitEvent(s->sN, fork, 0);
    if (int rc = hm_sp_collect_counted(s->sp, s->io.planes[1 - which], s->d_rows[1 - which], s->sT)) return rc;
    if (int rc = hm_net_forward_groups(s->io.net, s->io.planes[which], s->G * 8, allRows ? nullptr : s->d_rows[which], 8,
                                       h[0], h[1], h[2], h[3], h[4], s->sN)) return rc;
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
    for (int k = 0; k < GRAPH_ITERS && !rc; ++k) rc = enqueue_iteration(s, k & 1, k & 1, s->gFork, s->gJoin, allRows);
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
        (void)hipDeviceSynchronize();              // the prologue ran on the null stream
    }
    // No game can finish before it has collected minTarget nodes, i.e. floor(minTarget / 8) batches: the
    // host does not poll (and so does not synchronise) before that many iterations have been enqueued.
    const int noPollBefore = minTarget / 8 - 1;
    // Leg timings (HIP events on the launch streams) are sampled: event records are not free.  The samples of one search
    // are averaged and stand for all of its iterations.
    int harvested = 0, nSamples = 0;
    double sum[3] = {0.0, 0.0, 0.0};
    auto is_timed = [&](int it) { return !native || (s->graphState == 1 ? (it == 3 || it == 6) : (it & 7) == 0); };
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
    if (native && s->graphState == 0) build_step_graph(s, allRows);
    while (active > 0) {
        // Leg timings are sampled on eager iterations (the first GRAPH_ITERS of every search); after that the loop replays the
        // captured graph, GRAPH_ITERS iterations per launch, polling the active-game count once per launch.
        if (native && s->graphState == 1 && iters >= GRAPH_ITERS && which == 0) {
            if (hipGraphLaunch(s->stepGraph, s->sT) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipGraphLaunch failed");
            iters += GRAPH_ITERS;
            s->res.eval_batches += GRAPH_ITERS;
            if (iters >= noPollBefore) {
                if (int rc = hm_sp_active_on(s->sp, s->hActive, s->sT)) return rc;
                active = *s->hActive;
            }
            if (iters > 100000) return hm_fail(HM_ERR_STATE, "search did not terminate");
            continue;
        }
        hipEvent_t* e = &s->evs[(size_t)(iters % RING) * 6];
        hipEvent_t* y = &s->sync[(size_t)(iters % RING) * 2];
        hipEvent_t* yPrev = &s->sync[(size_t)((iters + RING - 1) % RING) * 2];
        const bool poll = iters >= noPollBefore || (iters - harvested) >= RING - 2;
        const bool timed = is_timed(iters);
        if (native) {
            // tree stream: collect(next) -- net stream: forward(cur) -- tree stream: process(cur)
            void* hv[2][5] = {{s->io.value, s->io.pi_a, s->io.pi_b, s->io.wdl, s->io.moves_left},
                              {s->io.value_2, s->io.pi_a_2, s->io.pi_b_2, s->io.wdl_2, s->io.moves_left_2}};
            void** h = hv[iters & 1];
            if (timed) (void)hipEventRecord(e[0], s->sT);
            if (int rc = hm_sp_collect_counted(s->sp, s->io.planes[1 - which], s->d_rows[1 - which], s->sT)) return rc;
            if (timed) (void)hipEventRecord(e[1], s->sT);
            // planes[which] were completed by the previous iteration's collect.  The forward could start
            // right after it, but measured on MI355X it then shares CUs with k_process (8-wave blocks,
            // 79 KB LDS), which stretches from 0.12 to 0.33 ms and lengthens the critical path; so the
            // forward is ordered behind process(i-1) and overlaps only collect(i).
            if (iters > 0) (void)hipStreamWaitEvent(s->sN, yPrev[1], 0);
            if (timed) (void)hipEventRecord(e[2], s->sN);
            if (int rc = hm_net_forward_groups(s->io.net, s->io.planes[which], s->G * 8, allRows ? nullptr : s->d_rows[which], 8,
                                               h[0], h[1], h[2], h[3], h[4], s->sN)) return rc;
            if (timed) (void)hipEventRecord(e[3], s->sN);
            (void)hipEventRecord(y[0], s->sN);
            (void)hipStreamWaitEvent(s->sT, y[0], 0);
            if (timed) (void)hipEventRecord(e[4], s->sT);
            if (int rc = hm_sp_process(s->sp, h[0], h[1], h[2], h[3], h[4], nullptr, s->sT)) return rc;
            if (timed) (void)hipEventRecord(e[5], s->sT);
            (void)hipEventRecord(y[1], s->sT);
            if (poll) {
                if (int rc = hm_sp_active_on(s->sp, s->hActive, s->sT)) return rc;     // async copy + stream sync
                active = *s->hActive;
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
    harvest(std::min(iters, s->graphState == 1 && native ? GRAPH_ITERS : iters));      // the stream was synchronised by the last poll
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
    if (s->sT) (void)hipStreamDestroy(s->sT);
    if (s->sN && s->sN != s->sT) (void)hipStreamDestroy(s->sN);
    if (s->hActive) (void)hipHostFree(s->hActive);
    if (s->stepGraph) (void)hipGraphExecDestroy(s->stepGraph);
    if (s->gFork) (void)hipEventDestroy(s->gFork);
    if (s->gJoin) (void)hipEventDestroy(s->gJoin);
    for (auto& e : s->evs) if (e) (void)hipEventDestroy(e);
    for (auto& e : s->sync) if (e) (void)hipEventDestroy(e);
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

// run_selfplay (selfplay.cc:558-748).  Whatever the outcome, the samples of every finished game are handed to the chunk
// sink before returning (ChunkWriter::finish :87-91), so an error late in a run does not discard the games before it.
int hm_selfplay_run(hm_selfplay* s, hm_selfplay_result* out) {
    if (!s) return hm_fail(HM_ERR_INVALID, "null argument");
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = selfplay_run_impl(s);
    const std::string msg = rc ? std::string(hm_last_error()) : std::string();
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
    std::vector<uint8_t> mask(G), u8planes((size_t)G * HM_PLANE_VALUES);
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
        if (hipMemcpy(u8planes.data(), s->d_u8, u8planes.size(), hipMemcpyDeviceToHost) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipMemcpy failed");
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
        for (int g = 0; g < G; ++g) {
            if (!mask[g]) continue;
            Slot& sl = s->slots[g];
            if (info[(size_t)g * HM_SP_INFO_INTS + 8]) return hm_fail(HM_ERR_OVERFLOW, "search pool overflow in game slot " + std::to_string(g) + " (flags " + std::to_string(info[(size_t)g * HM_SP_INFO_INTS + 8]) + ")");
            s->res.searched_positions += 1;
            s->res.eval_rows += (uint64_t)info[(size_t)g * HM_SP_INFO_INTS + 2];
            s->res.nodes_visited += (uint64_t)info[(size_t)g * HM_SP_INFO_INTS + 10];
            s->res.edges_scanned += (uint64_t)info[(size_t)g * HM_SP_INFO_INTS + 11];
            s->res.leaf_move_words += (uint64_t)info[(size_t)g * HM_SP_INFO_INTS + 13];
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
            std::memcpy(sm.planes.data(), u8planes.data() + (size_t)g * HM_PLANE_VALUES, HM_PLANE_VALUES);
            uint64_t actual = 0;
            for (int i = 0; i < n; ++i) actual += (uint64_t)std::max(0, ev[i]);
            sm.nodes = static_cast<uint32_t>(std::min<uint64_t>(actual, 0xffffffffu));
            sm.rootQ = rootQ[g];
            for (int b = 0; b < 2; ++b) {   // marginal_policy :400-427
                std::map<uint16_t, uint64_t> byMove;
                uint64_t total = 0;
                for (int i = 0; i < n; ++i) {
                    if (ev[i] <= 0) continue;
                    const hm_move mv = b == 0 ? ea[i] : eb[i];
                    const int idx = hm_policy_index(mv, boards[g].pos[b].stm);
                    if (idx < 0) return hm_fail(HM_ERR_STATE, "Move is absent from policy map");
                    byMove[static_cast<uint16_t>(idx)] += (uint64_t)ev[i];
                    total += (uint64_t)ev[i];
                }
                if (total == 0) return hm_fail(HM_ERR_STATE, "Search returned no visited root edges");
                auto& pol = b == 0 ? sm.policyA : sm.policyB;
                for (const auto& kv : byMove) pol.push_back({kv.first, static_cast<float>(kv.second) / static_cast<float>(total)});
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
        if (int rc = hm_sp_apply(s->sp, actA.data(), actB.data(), applyMask.data())) return rc;
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

}  // extern "C"
