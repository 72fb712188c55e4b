// hm_uci.hip — UCI front end over the GPU search engine (host C++ only; compiled by hipcc into the same library).
//
// Replaces, behind the C ABI of include/hivemind_amd.h (hm_uci_*), the reference's
//   UCI::loop / position / go / setoption / send_uci_response / stop / new_game      interface/uci.cc:82-141, 143-231, 239-296,
//                                                                                    298-317, 396-429
//   Agent::run_search's budget handling for UCI (`go nodes N`, `go movetime T`),
//   its final `info ...` line, extract_best_move and format_uci_score                search/agent.cc:48-78, 421-558, 898-1002,
//                                                                                    1031-1049
// One game slot of the lockstep engine (hm_sp_*) plays the role of the Agent: `position` replays the move list on the device
// (history, repetition keys), `go` runs the search and prints the solver-aware best move (get_best_move_idx_with_q_weight,
// node.h:656-754 — computed on the device, hm_sp_root_stats info[12]).
// Built: uci, isready, ucinewgame, position startpos|fen ... [moves <1|2><uci> ...], go nodes N | movetime T (| neither = 1 s),
// stop, setoption name {Team, Mode, MultiPV, Ponder, DrawContemptPermille, PWCoefficientPermille, RootPWCoefficientPermille,
// PWExponentPermille, Transpositions} (Hash is accepted and reported as the reference does), policy (uci.cc:306-393), quit.
// The final `info` lines carry MultiPV principal variations (agent.cc:917-965, 1218-1290; walked on the device by hm_sp_pv_lines)
// and `bestmove` carries the ponder move (agent.cc:1054-1113) when Ponder is on.
// `go ponder` runs on a worker thread until `ponderhit` (then its budget applies, clock restarted) or `stop`; every other `go` is
// synchronous.  The search tree is carried from one `go` to the next (Agent::try_reuse_tree / store_next_root_candidates,
// agent.cc:1345-1451; on the device: k_begin's find_reusable_root) until `ucinewgame` or a setoption that rebuilds the engine.  `go movetime` follows the reference's polling loop (early exit on a
// solved root / forced mate, early stopping on an insurmountable visit lead, time extension).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hivemind_amd.h"
#include "hm_host.hpp"

const HostTables& hm_host_tables();
int hm_fail(int code, const std::string& msg);
extern "C" int hm_insurmountable_visit_lead(float best_visits, float projected_second_visits, float factor);

namespace {

const char* const kStartFen = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1";

std::string uci_text(hm_move m) {
    char buf[16];
    hm_move_uci(m, buf, sizeof buf);
    return buf;
}

// Position::set (Fairy-Stockfish position.cpp:232-470) for the bughouse subset: placement with '~' promoted markers and the
// pocket either in [brackets] or after an eighth slash, side, castling, en passant (kept only when a pawn can take,
// :395-409), halfmove clock, fullmove number.
bool parse_fen(const HostTables& h, const std::string& fen, hm_pos* out) {
    hm_pos p;
    std::memset(&p, 0, sizeof p);
    p.ep = 64;
    std::istringstream ss(fen);
    std::string placement, stm, cast = "-", ep = "-";
    if (!(ss >> placement >> stm)) return false;
    static const std::string pcs = " PNBRQK";
    int s = 56, slashes = 0;
    size_t i = 0;
    for (; i < placement.size(); ++i) {
        const char ch = placement[i];
        if (ch == '[') break;
        if (ch >= '0' && ch <= '9') s += ch - '0';
        else if (ch == '/') { if (++slashes == 8) { ++i; break; } s -= 16; }
        else {
            const size_t idx = pcs.find((char)std::toupper((unsigned char)ch));
            if (idx == std::string::npos || idx == 0) continue;
            if (s < 0 || s > 63) return false;
            const uint64_t b = 1ULL << s;
            p.by_type[idx - 1] |= b;
            p.by_color[std::islower((unsigned char)ch) ? 1 : 0] |= b;
            if (i + 1 < placement.size() && placement[i + 1] == '~') { p.promoted |= b; ++i; }
            ++s;
        }
    }
    for (; i < placement.size(); ++i) {
        const char ch = placement[i];
        const size_t idx = pcs.find((char)std::toupper((unsigned char)ch));
        if (ch == '[' || ch == ']' || idx == std::string::npos || idx == 0 || idx == 6) continue;
        uint8_t& n = p.hand[std::islower((unsigned char)ch) ? 1 : 0][idx - 1];
        if (n < 31) ++n;
    }
    if (__builtin_popcountll(p.by_type[5] & p.by_color[0]) != 1 || __builtin_popcountll(p.by_type[5] & p.by_color[1]) != 1) return false;
    p.stm = stm == "b" ? 1 : 0;
    ss >> std::ws;
    int r50 = 0, full = 1;
    if (ss.peek() != EOF && !std::isdigit(ss.peek())) ss >> cast >> ep;
    ss >> r50 >> full;
    for (const char ch : cast) {
        const int c = std::islower((unsigned char)ch) ? 1 : 0;
        const char up = (char)std::toupper((unsigned char)ch);
        const int base = c ? 56 : 0;
        const uint64_t rooks = p.by_type[3] & p.by_color[c], king = p.by_type[5] & p.by_color[c];
        if (!(king & (1ULL << (base + 4)))) continue;                              // standard chess: king on the e-file
        if (up == 'K' && (rooks & (1ULL << (base + 7)))) p.castling |= c ? HM_BLACK_OO : HM_WHITE_OO;
        else if (up == 'Q' && (rooks & (1ULL << base))) p.castling |= c ? HM_BLACK_OOO : HM_WHITE_OOO;
    }
    if (ep.size() == 2 && ep[0] >= 'a' && ep[0] <= 'h' && ep[1] >= '1' && ep[1] <= '8') {
        const int e = (ep[1] - '1') * 8 + (ep[0] - 'a'), us = p.stm, push = us == 0 ? 8 : -8, f = e & 7;
        const uint64_t occ = p.by_color[0] | p.by_color[1];
        const uint64_t ourPawns = p.by_type[0] & p.by_color[us], theirPawns = p.by_type[0] & p.by_color[us ^ 1];
        uint64_t from = 0;                                                          // squares one of our pawns would capture from
        if (e - push >= 0 && e - push < 64) {
            if (f > 0) from |= 1ULL << (e - push - 1);
            if (f < 7) from |= 1ULL << (e - push + 1);
        }
        const bool ok = (from & ourPawns) && e - push >= 0 && e - push < 64 && (theirPawns & (1ULL << (e - push)))
                        && e + push >= 0 && e + push < 64 && !(occ & ((1ULL << e) | (1ULL << (e + push))));
        p.ep = ok ? (uint8_t)e : 64;
    }
    p.rule50 = (uint8_t)std::min(std::max(r50, 0), 255);
    p.game_ply = (uint16_t)(std::max(2 * (full - 1), 0) + (p.stm ? 1 : 0));
    p.key = host_compute_key(h, p);
    *out = p;
    return true;
}

// Position::fen(false, true) (Fairy-Stockfish position.cpp:637-768) for the bughouse subset: placement with '~' after promoted pieces,
// the pocket in brackets (white then black, queen down to pawn), side, castling, en passant, halfmove clock, fullmove number.
std::string fen_text(const hm_pos& p) {
    static const char pc[] = "PNBRQK";
    std::string s;
    for (int r = 7; r >= 0; --r) {
        int empty = 0;
        for (int f = 0; f < 8; ++f) {
            const uint64_t b = 1ULL << (r * 8 + f);
            int t = -1;
            for (int k = 0; k < 6; ++k) if (p.by_type[k] & b) t = k;
            if (t < 0) { ++empty; continue; }
            if (empty) { s += (char)('0' + empty); empty = 0; }
            s += (p.by_color[1] & b) ? (char)std::tolower((unsigned char)pc[t]) : pc[t];
            if (p.promoted & b) s += '~';
        }
        if (empty) s += (char)('0' + empty);
        if (r > 0) s += '/';
    }
    s += '[';
    for (int c = 0; c < 2; ++c)
        for (int pt = 4; pt >= 0; --pt) s += std::string(p.hand[c][pt], c ? (char)std::tolower((unsigned char)pc[pt]) : pc[pt]);
    s += ']';
    s += p.stm == 0 ? " w " : " b ";
    if (p.castling & HM_WHITE_OO) s += 'K';
    if (p.castling & HM_WHITE_OOO) s += 'Q';
    if (p.castling & HM_BLACK_OO) s += 'k';
    if (p.castling & HM_BLACK_OOO) s += 'q';
    if (!(p.castling & (HM_WHITE_OO | HM_WHITE_OOO | HM_BLACK_OO | HM_BLACK_OOO))) s += '-';
    if (p.ep < 64) { s += ' '; s += (char)('a' + (p.ep & 7)); s += (char)('1' + (p.ep >> 3)); s += ' '; }
    else s += " - ";
    s += std::to_string((int)p.rule50) + " " + std::to_string(1 + ((int)p.game_ply - (p.stm == 1 ? 1 : 0)) / 2);
    return s;
}

std::string trim(const std::string& s) {
    const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

}  // namespace

// The time-managed part of Agent::run_search's polling loop (agent.cc:561-713) as a pure host object: fed the root's edge
// statistics every poll, it answers stop / continue and keeps the effective move time (SearchInfo, searchinfo.h:55-63,
// 129-180).  Constants: search_params.h:204-246.
struct hm_time_manager {
    int moveTimeMs = 0;
    double effectiveMs = 0.0;
    int extensions = 0;
    float nps = 0.0f;
    float lastCheckEval = 0.0f;
    bool evalInitialized = false;
    int lastBestChildIdx = -1;
    std::string log;                  // the `info string ...` lines the reference prints at these decisions
};
static bool tm_try_extend(hm_time_manager& t, double elapsedMs, float factor = 1.5f, int maxExtensions = 2) {   // SearchInfo::try_extend_time
    if (t.extensions >= maxExtensions) return false;
    const double remaining = t.effectiveMs - elapsedMs;
    if (remaining <= 0) return false;
    t.effectiveMs += static_cast<int>(remaining * (factor - 1.0f));
    t.extensions++;
    return true;
}
// one poll; returns true when the search should stop now
static bool tm_poll(hm_time_manager& t, double elapsedMs, int nodes, int n, const int* visits, const float* q, int rootType, const int* childType, const int* childEndInPly) {
    const double elapsedSec = elapsedMs / 1000.0;                                    // SearchInfo::update_nps
    if (elapsedSec > 0.1) {
        const float cur = static_cast<float>(nodes / elapsedSec);
        t.nps = t.nps == 0.0f ? cur : 0.9f * t.nps + 0.1f * cur;
    }
    if (n <= 0) return false;
    int firstMax = 0, secondMax = 0, firstIdx = 0, secondIdx = -1;                   // agent.cc:603-616
    for (int i = 0; i < n; ++i) {
        if (visits[i] > firstMax) { secondMax = firstMax; secondIdx = firstIdx; firstMax = visits[i]; firstIdx = i; }
        else if (visits[i] > secondMax) { secondMax = visits[i]; secondIdx = i; }
    }
    const float bestQ = q[firstIdx], secondQ = secondIdx >= 0 ? q[secondIdx] : -1.0f;
    if (!t.evalInitialized) { t.lastCheckEval = bestQ; t.evalInitialized = true; }
    // should_exit_early_winning (agent.cc:92-129)
    if (rootType != 0) { t.log += std::string("info string Early exit: root position is proven ") + (rootType == 1 ? "WIN" : rootType == 2 ? "LOSS" : "DRAW") + "\n"; return true; }
    if (childType && childType[firstIdx] == 2) { t.log += "info string Early exit: forced mate in " + std::to_string((childEndInPly[firstIdx] + 1) / 2) + " found\n"; return true; }
    if (t.nps > 0) {                                                                 // early stopping (agent.cc:634-652)
        const double remaining = t.effectiveMs - elapsedMs;
        const float projected = static_cast<float>(secondMax) + static_cast<float>(remaining * t.nps / 1000.0);
        if (hm_insurmountable_visit_lead(static_cast<float>(firstMax), projected, 2.0f) && bestQ >= secondQ) {
            t.log += "info string Early stopping: saved " + std::to_string(static_cast<int>(std::max(0.0, (double)t.moveTimeMs - elapsedMs))) + "ms\n";
            return true;
        }
    }
    {                                                                                // time extension (agent.cc:655-678)
        const float evalDrop = t.lastCheckEval - bestQ;
        if (evalDrop > 0.05f && tm_try_extend(t, elapsedMs))
            t.log += "info string Extending search time (eval dropped by " + std::to_string(static_cast<int>(evalDrop * 100)) + " cp)\n";
        t.lastCheckEval = bestQ;
        if (t.lastBestChildIdx >= 0 && firstIdx != t.lastBestChildIdx && elapsedMs > t.moveTimeMs * 0.4f && tm_try_extend(t, elapsedMs))
            t.log += "info string Extending search time (best move changed to " + std::to_string(firstIdx) + ")\n";
        t.lastBestChildIdx = firstIdx;
    }
    return elapsedMs >= t.effectiveMs;
}

struct hm_uci {
    const hm_net* net = nullptr;
    hm_eval_fn fn = nullptr;
    void* user = nullptr;
    hm_eval_io io{};
    hm_search_config scfg{};
    hm_sp* sp = nullptr;
    int maxNodes = 0;
    int team = HM_WHITE;
    bool sit = false;                 // Mode sit = teamHasTimeAdvantage (uci.cc:289-295)
    int multiPV = 1;
    bool ponder = true;
    hm_board board{};                 // host copy of the current game position
    hm_pos* d_pos = nullptr;          // scratch for legal-move queries
    hm_move* d_moves = nullptr;
    uint32_t* d_counts = nullptr;
    hm_board* d_board = nullptr;      // `policy`: the position handed to the plane encoder
    hipStream_t sN = nullptr;
    std::string out;                  // pending output text (guarded by mu: a ponder search appends from its thread)
    std::mutex mu;
    std::thread worker;               // `go ponder` runs here
    std::atomic<bool> pondering{false}, stopReq{false}, busy{false};
};

static int uci_rebuild_engine(hm_uci* u) {
    if (u->sp) { hm_sp_destroy(u->sp); u->sp = nullptr; }
    return hm_sp_create_ex(1, u->maxNodes, 2048, &u->scfg, &u->sp);
}
static int uci_sync_board(hm_uci* u) {
    int flags = 0;
    return hm_sp_game_state(u->sp, &u->board, &flags, nullptr);
}
static int uci_set_position(hm_uci* u, const hm_board& b) {
    const uint8_t one = 1;
    if (int rc = hm_sp_set_games(u->sp, &b, &one)) return rc;
    return uci_sync_board(u);
}
// UCI::to_move (Fairy-Stockfish stubs.cpp:61-75): the legal move of board `b` whose text equals `text`
static int uci_find_move(hm_uci* u, int b, std::string text, hm_move* out) {
    if (text.size() == 5) {
        if (text[4] == '=') text.pop_back();
        else if (text[1] != '@') text[4] = (char)std::tolower((unsigned char)text[4]);
    }
    if (hipMemcpy(u->d_pos, &u->board.pos[b], sizeof(hm_pos), hipMemcpyHostToDevice) != hipSuccess) return hm_fail(HM_ERR_NO_DEVICE, "hipMemcpy failed");
    if (int rc = hm_legal_moves(u->d_pos, 1, u->d_moves, u->d_counts, nullptr)) return rc;
    uint32_t n = 0;
    std::vector<hm_move> mv(HM_MAX_MOVES);
    if (hipMemcpy(&n, u->d_counts, 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(mv.data(), u->d_moves, 4 * HM_MAX_MOVES, hipMemcpyDeviceToHost) != hipSuccess)
        return hm_fail(HM_ERR_NO_DEVICE, "hipMemcpy failed");
    *out = 0;
    for (uint32_t i = 0; i < n && i < HM_MAX_MOVES; ++i)
        if (uci_text(mv[i]) == text) { *out = mv[i]; break; }
    return 0;
}

static void uci_position(hm_uci* u, std::istringstream& is) {   // uci.cc:82-141
    std::string token;
    is >> token;
    hm_board b;
    std::memset(&b, 0, sizeof b);
    const HostTables& h = hm_host_tables();
    std::string fen;
    if (token == "startpos") { fen = std::string(kStartFen) + "|" + kStartFen; is >> token; }
    else if (token == "fen") { while (is >> token && token != "moves") fen += token + " "; }
    else return;
    const size_t bar = fen.find('|');
    if (bar == std::string::npos || !parse_fen(h, trim(fen.substr(0, bar)), &b.pos[0]) || !parse_fen(h, trim(fen.substr(bar + 1)), &b.pos[1])) {
        u->out += "info string position failed: invalid FEN\n";
        return;
    }
    b.rep_count[0] = b.rep_count[1] = 1;
    b.team = (uint8_t)u->team; b.time_adv = u->sit ? 1 : 0;
    if (uci_set_position(u, b)) { u->out += std::string("info string position failed: ") + hm_last_error() + "\n"; return; }
    if (token != "moves") return;
    int moveCount = 0;
    while (is >> token) {
        if (token.empty() || token[0] < '1' || token[0] > '2') {
            u->out += "info string Error: Invalid board indicator in move '" + token + "' at move " + std::to_string(moveCount + 1) + "\n";
            break;
        }
        const int bd = token[0] - '1';
        hm_move m = 0;
        if (uci_find_move(u, bd, token.substr(1), &m) || m == 0) {
            u->out += "info string Error: Invalid move '" + token.substr(1) + "' on board " + std::to_string(bd + 1) + " at move " + std::to_string(moveCount + 1) + "\n";
            break;
        }
        const hm_move a = bd == 0 ? m : 0, c = bd == 1 ? m : 0;
        const uint8_t one = 1;
        if (hm_sp_apply(u->sp, &a, &c, &one) || uci_sync_board(u)) { u->out += std::string("info string position failed: ") + hm_last_error() + "\n"; break; }
        ++moveCount;
    }
}

static std::string uci_score(int childType, int childEndInPly, float q) {   // format_uci_score, agent.cc:48-78 (child of the root)
    if (childType == 1) return "score mate -" + std::to_string(std::max(1, (childEndInPly + 1) / 2));   // child wins = we are mated
    if (childType == 2) return "score mate " + std::to_string(std::max(1, (childEndInPly + 1) / 2));    // child loses = we mate
    return "score cp " + std::to_string(static_cast<int>(180.0f * std::tan(1.56f * q)));
}

struct GoParams { int moveTime = 0; size_t nodes = 0; bool ponder = false; };

// Agent::run_search (agent.cc:421-558, 561-713, 898-1002) for one `go`: the lockstep loop of the single game slot, the budget rules
// and the final info / bestmove lines, appended to `out`.  A ponder search (SearchOptions::isPonder, agent.h:31) ignores its budget
// until `ponderhit` clears hm_uci::pondering; the clock restarts there (SearchInfo, searchinfo.h:39) and the budget applies from then
// on.  `stop` (hm_uci::stopReq) ends either kind at the next iteration.  bestmove is never printed while still pondering.
static void uci_search(hm_uci* u, GoParams gp, std::string& out) {
    int moveTime = gp.moveTime;
    const size_t nodes = gp.nodes;
    const uint8_t team = (uint8_t)u->team, adv = u->sit ? 1 : 0, one = 1;
    if (hm_sp_set_side(u->sp, &team, &adv)) { out += "bestmove (none)\n"; return; }
    // a ponder search runs on until told otherwise: its device-side node target is the pool size, the node budget is applied by the host
    const int target = (nodes > 0 && !gp.ponder) ? (int)std::min<size_t>(nodes, (size_t)u->maxNodes) : u->maxNodes;
    const uint64_t seed = 0;
    auto t0 = std::chrono::steady_clock::now();
    auto elapsed_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    // tree reuse (ENABLE_TREE_REUSE, search_params.h:194): a node-limited search keeps its budget, a time-limited one may shrink it
    // to what the pool still holds behind the retained tree
    const uint8_t reuseMode = (nodes > 0 && !gp.ponder) ? 1 : 2;
    if (hm_sp_set_tree_reuse(u->sp, &reuseMode, 0)) { out += std::string("info string search failed: ") + hm_last_error() + "\nbestmove (none)\n"; return; }
    if (hm_sp_begin_search(u->sp, &target, &seed, 0.0f, 0.0f, &one)) { out += std::string("info string search failed: ") + hm_last_error() + "\nbestmove (none)\n"; return; }
    {   // "info string Tree reuse: N visits recovered" (agent.cc:519-522)
        int count = 0;
        std::vector<int> binfo(HM_SP_INFO_INTS);
        if (!hm_sp_root_stats(u->sp, &count, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, binfo.data(), hm_sp_max_edges(u->sp)) && binfo[16] >= 0)
            out += "info string Tree reuse: " + std::to_string(binfo[16]) + " visits recovered\n";
    }
    int active = 1, which = 0, iters = 0;
    bool stopped = false, wasPondering = gp.ponder;
    void* heads[5] = {u->io.value, u->io.pi_a, u->io.pi_b, u->io.wdl, u->io.moves_left};
    hm_time_manager tm;
    tm.moveTimeMs = moveTime; tm.effectiveMs = moveTime;
    double lastPoll = 0.0;
    int lastReportedDepth = 0;
    const int E0 = hm_sp_max_edges(u->sp);
    std::vector<int> pv(E0), pinfo(HM_SP_INFO_INTS), ptype(E0), pend(E0);
    std::vector<float> pq(E0);
    while (active > 0 && iters < 200000) {
        int rc = hm_sp_collect(u->sp, u->io.planes[1 - which], nullptr);
        if (!rc) rc = u->net ? hm_net_forward(u->net, u->io.planes[which], 8, heads[0], heads[1], heads[2], heads[3], heads[4], nullptr)
                             : u->fn(u->user, which, 8);
        if (!rc) rc = hm_sp_process(u->sp, heads[0], heads[1], heads[2], heads[3], heads[4], &active, nullptr);
        if (rc) { out += std::string("info string search failed: ") + hm_last_error() + "\nbestmove (none)\n"; return; }
        which = 1 - which;
        ++iters;
        if (!stopped && u->stopReq.load(std::memory_order_acquire)) { (void)hm_sp_stop(u->sp, nullptr, nullptr); stopped = true; }
        const bool pondering = u->pondering.load(std::memory_order_acquire);
        if (wasPondering && !pondering) {                   // ponderhit: the turn starts now
            wasPondering = false;
            t0 = std::chrono::steady_clock::now();
            lastPoll = 0.0;
        }
        if (!stopped && !pondering && gp.ponder && nodes > 0 && active > 0) {
            // node budget of a search that started as a ponder search (agent.cc:573: nodes < target once isPondering_ is cleared)
            const double now = elapsed_ms();
            if (now - lastPoll >= 1.0 || lastPoll == 0.0) {
                lastPoll = now;
                int count = 0;
                if (!hm_sp_root_stats(u->sp, &count, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, pinfo.data(), E0) && (size_t)pinfo[1] >= nodes) {
                    (void)hm_sp_stop(u->sp, nullptr, nullptr);
                    stopped = true;
                }
            }
        }
        if (!stopped && !pondering && nodes == 0 && active > 0) {
            // the reference polls every 5 ms (agent.cc:562, 577-580): early exit on a solved root / forced mate, early stopping
            // on an insurmountable visit lead, time extension on a falling evaluation or a late change of the best move
            const double now = elapsed_ms();
            bool stop = now >= tm.effectiveMs;
            if (!stop && now - lastPoll >= 5.0) {
                lastPoll = now;
                int count = 0;
                float rq = 0.0f;
                if (!hm_sp_root_stats(u->sp, &count, nullptr, nullptr, pv.data(), pq.data(), nullptr, &rq, pinfo.data(), E0)) {
                    // solver types of the children are not part of the root statistics: the forced-mate exit uses the best
                    // action's child (info[14..15]), which is the only child the rule looks at
                    for (int i = 0; i < count; ++i) { ptype[i] = 0; pend[i] = 0; }
                    if (pinfo[12] >= 0 && pinfo[12] < count) { ptype[pinfo[12]] = pinfo[14]; pend[pinfo[12]] = pinfo[15]; }
                    stop = tm_poll(tm, now, pinfo[1], count, pv.data(), pq.data(), pinfo[6] > 0 ? pinfo[6] : 0, ptype.data(), pend.data());
                    out += tm.log; tm.log.clear();
                    // "Report each completed depth once" (agent.cc:680-711): the solver-aware best line so far
                    if (!stop && pinfo[9] > lastReportedDepth && count > 0 && pinfo[12] >= 0 && pinfo[12] < count) {
                        lastReportedDepth = pinfo[9];
                        const int ci = pinfo[12];
                        int len = 0, ct = 0, ce = 0;
                        hm_move line[40];
                        if (!hm_sp_pv_lines(u->sp, 0, 1, &ci, 20, line, &len, &ct, &ce)) {
                            std::string s = "info depth " + std::to_string(pinfo[9]) + " " + uci_score(ct, ce, pq[ci]) + " nodes " + std::to_string(pinfo[1]) + " nps "
                                            + std::to_string(now > 0 ? (int)(pinfo[1] * 1000.0 / now) : 0) + " hashfull 0 tbhits 0 time " + std::to_string((int)now);
                            for (int d = 0; d < len; ++d) s += std::string(d == 0 ? " pv " : " ") + "(" + uci_text(line[2 * d]) + "," + uci_text(line[2 * d + 1]) + ")";
                            out += s + "\n";
                        }
                    }
                }
            }
            if (stop) { (void)hm_sp_stop(u->sp, nullptr, nullptr); stopped = true; }
        }
    }
    // the pool is exhausted but the GUI has not answered the ponder search yet: hold the result (UCI: no bestmove while pondering)
    while (u->pondering.load(std::memory_order_acquire) && !u->stopReq.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    const int E = hm_sp_max_edges(u->sp);
    std::vector<hm_move> mA(E), mB(E);
    std::vector<int> visits(E), info(HM_SP_INFO_INTS);
    std::vector<float> q(E);
    int count = 0;
    float rootQ = 0.0f;
    if (hm_sp_root_stats(u->sp, &count, mA.data(), mB.data(), visits.data(), q.data(), nullptr, &rootQ, info.data(), E)) { out += "bestmove (none)\n"; return; }
    const double ms = elapsed_ms();
    const int nodesDone = info[1], depth = info[9], best = info[12];
    if (count <= 0 || best < 0 || best >= count) { out += "bestmove (none)\n"; return; }
    const std::string bestText = "(" + uci_text(mA[best]) + "," + uci_text(mB[best]) + ")";
    const int nps = ms > 0 ? (int)(nodesDone * 1000.0 / ms) : 0;
    // final info lines (agent.cc:917-965): root children by visit count (the reference's std::sort and comparator), the solver-aware
    // best move first, MultiPV lines at most, each with its principal variation (20 joint actions at most, walked on the device)
    std::vector<size_t> order((size_t)count);
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return visits[a] > visits[b]; });
    {
        auto it = std::find(order.begin(), order.end(), (size_t)best);
        if (it != order.end() && it != order.begin()) { order.erase(it); order.insert(order.begin(), (size_t)best); }
    }
    constexpr int PV_DEPTH = 20;
    const int numPVs = std::min(u->multiPV, count);
    std::vector<int> childIdx((size_t)numPVs), lens((size_t)numPVs), ctype((size_t)numPVs), cend((size_t)numPVs);
    std::vector<hm_move> pvMoves((size_t)numPVs * PV_DEPTH * 2);
    for (int k = 0; k < numPVs; ++k) childIdx[k] = (int)order[k];
    if (hm_sp_pv_lines(u->sp, 0, numPVs, childIdx.data(), PV_DEPTH, pvMoves.data(), lens.data(), ctype.data(), cend.data())) {
        out += std::string("info string search failed: ") + hm_last_error() + "\nbestmove (none)\n";
        return;
    }
    auto joint_text = [&](int k, int d) { const hm_move* m = &pvMoves[((size_t)k * PV_DEPTH + d) * 2]; return "(" + uci_text(m[0]) + "," + uci_text(m[1]) + ")"; };
    for (int k = 0; k < numPVs; ++k) {
        std::string s = "info depth " + std::to_string(depth);
        if (u->multiPV > 1) s += " multipv " + std::to_string(k + 1);
        s += " " + uci_score(ctype[k], cend[k], q[childIdx[k]]) + " nodes " + std::to_string(nodesDone) + " nps " + std::to_string(nps)
             + " hashfull 0 tbhits 0 time " + std::to_string((int)ms);
        for (int d = 0; d < lens[k]; ++d) s += (d == 0 ? " pv " : " ") + joint_text(k, d);
        out += s + "\n";
    }
    char line[256];
    std::snprintf(line, sizeof line, "info string rejected selection attempts %d (same batch %d, pending evaluation %d)\n", info[3] + info[4], info[3], info[4]);
    out += line;
    // extract_ponder_move (agent.cc:1054-1113): the best child's own best reply = the second joint action of line 1
    if (u->ponder && numPVs > 0 && lens[0] >= 2) out += "bestmove " + bestText + " ponder " + joint_text(0, 1) + "\n";
    else out += "bestmove " + bestText + "\n";
}

// UCI::stop (uci.cc:30-50): end a running search and wait for its output
static void uci_stop(hm_uci* u) {
    if (!u->worker.joinable()) return;
    u->stopReq.store(true, std::memory_order_release);
    u->worker.join();
    u->stopReq.store(false, std::memory_order_release);
    u->pondering.store(false, std::memory_order_release);
}

static void uci_go(hm_uci* u, std::istringstream& is) {   // uci.cc:143-231
    std::string token;
    GoParams gp;
    while (is >> token) {
        if (token == "ponder") gp.ponder = true;
        else if (token == "movetime") is >> gp.moveTime;
        else if (token == "nodes") is >> gp.nodes;
    }
    uci_stop(u);
    if (gp.nodes == 0 && gp.moveTime <= 0) gp.moveTime = 1000;                      // "Default to 1 second if nothing specified"
    // Every search runs on the worker thread (mainSearchThread of the reference, uci.cc:192-205): the command returns at once, so
    // `stop`, `isready`, `ponderhit` and `quit` reach a running search; its text (info lines, bestmove) is fetched with later
    // commands (an empty line polls, hm_uci_busy tells whether it is still running).  A ponder search ignores its budget until
    // `ponderhit`; an ordinary one applies it from the start.
    u->pondering.store(gp.ponder, std::memory_order_release);
    int dev = 0;
    (void)hipGetDevice(&dev);
    u->busy.store(true, std::memory_order_release);
    u->worker = std::thread([u, gp, dev]() {
        (void)hipSetDevice(dev);
        std::string text;
        uci_search(u, gp, text);
        { std::lock_guard<std::mutex> lock(u->mu); u->out += text; }
        u->busy.store(false, std::memory_order_release);
    });
}

static float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, ex = (h >> 10) & 31u, man = h & 1023u;
    uint32_t bits;
    if (ex == 0) {
        if (man == 0) bits = sign;
        else { int e = -1; uint32_t m = man; do { ++e; m <<= 1; } while (!(m & 1024u)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 1023u) << 13); }
    } else if (ex == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((ex + 112u) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// UCI::policy (uci.cc:306-393): one forward of the current position as the team sees it; value, WDL (win draw loss), predicted plies,
// and per board the legal moves + pass with their normalised policy, most probable first.
static void uci_policy(hm_uci* u) {
    auto fail = [&](const char* what) { u->out += std::string("info string policy failed: ") + what + ": " + hm_last_error() + "\n"; };
    const uint8_t team = (uint8_t)u->team, adv = u->sit ? 1 : 0;
    if (hm_sp_set_side(u->sp, &team, &adv)) return fail("set_side");
    hm_board hb;
    int flags = 0;
    if (!u->d_board && hipMalloc(reinterpret_cast<void**>(&u->d_board), sizeof(hm_board)) != hipSuccess) { u->out += "info string policy failed: hipMalloc\n"; return; }
    if (hm_sp_game_state(u->sp, &hb, &flags, u->d_board)) return fail("game_state");
    // the evaluator sees batches of 8 rows: row 0 is the position, the rest of the buffer is cleared
    if (hipMemset(u->io.planes[0], 0, (size_t)8 * HM_NB_PLANES * 64 * sizeof(uint16_t)) != hipSuccess) { u->out += "info string policy failed: hipMemset\n"; return; }
    if (hm_encode_planes(u->d_board, 1, HM_DT_F16, u->io.planes[0], nullptr)) return fail("encode_planes");
    if (hipDeviceSynchronize() != hipSuccess) { u->out += "info string policy failed: device\n"; return; }
    const int rc = u->net ? hm_net_forward(u->net, u->io.planes[0], 8, u->io.value, u->io.pi_a, u->io.pi_b, u->io.wdl, u->io.moves_left, nullptr)
                          : u->fn(u->user, 0, 8);
    if (rc) { u->out += "Inference failed\n"; return; }
    if (hipDeviceSynchronize() != hipSuccess) { u->out += "info string policy failed: device\n"; return; }
    uint16_t hv = 0, hml = 0, hwdl[3] = {0, 0, 0};
    if (hipMemcpy(&hv, u->io.value, 2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&hml, u->io.moves_left, 2, hipMemcpyDeviceToHost) != hipSuccess
        || hipMemcpy(hwdl, u->io.wdl, 6, hipMemcpyDeviceToHost) != hipSuccess) { u->out += "info string policy failed: hipMemcpy\n"; return; }
    std::vector<hm_move> moves(2 * HM_MAX_MOVES);
    std::vector<float> probs(2 * HM_MAX_MOVES);
    int counts[2] = {0, 0};
    uint8_t onTurn[2] = {0, 0};
    if (hm_sp_policy_listing(u->sp, u->io.pi_a, u->io.pi_b, moves.data(), probs.data(), nullptr, counts, onTurn, 1)) return fail("policy_listing");
    std::ostringstream os;                                  // the reference prints through std::cout's default float format
    const float wdl[3] = {half_to_float(hwdl[0]), half_to_float(hwdl[1]), half_to_float(hwdl[2])};
    os << "Value: " << half_to_float(hv) << "\n";
    const float maxWdl = std::max({wdl[0], wdl[1], wdl[2]});
    const float lossExp = std::exp(wdl[0] - maxWdl), drawExp = std::exp(wdl[1] - maxWdl), winExp = std::exp(wdl[2] - maxWdl);
    const float wdlTotal = lossExp + drawExp + winExp;
    os << "WDL: " << winExp / wdlTotal << " " << drawExp / wdlTotal << " " << lossExp / wdlTotal << "\n";
    os << "Predicted plies to end: " << half_to_float(hml) * 100.0f << "\n\n";
    for (int b = 0; b < 2; ++b) {
        os << "Board " << (b == 0 ? 'A' : 'B') << " (" << fen_text(hb.pos[b]) << "):\n";
        if (onTurn[b]) {
            const hm_move* mv = &moves[(size_t)b * HM_MAX_MOVES];
            const float* pr = &probs[(size_t)b * HM_MAX_MOVES];
            std::vector<size_t> idx((size_t)counts[b]);
            for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
            std::sort(idx.begin(), idx.end(), [&](size_t i1, size_t i2) { return pr[i1] > pr[i2]; });     // argsort, utils.h:26-34
            for (const size_t i : idx) os << "  " << uci_text(mv[i]) << ": " << pr[i] << "\n";
        } else os << "  (not our turn)\n";
        if (b == 0) os << "\n";
    }
    u->out += os.str();
}

static void uci_setoption(hm_uci* u, std::istringstream& is) {   // uci.cc:239-296
    std::string token, name, value;
    is >> token;
    if (token != "name") return;
    is >> name >> token;
    if (token != "value") return;
    is >> value;
    auto permille = [&](int lo, int hi) { return std::min(std::max(std::atoi(value.c_str()), lo), hi); };
    bool rebuild = false;
    if (name == "Hash") u->out += "info string Hash table set to " + value + " MB\n";            // per-search table sized from the node budget here
    else if (name == "MultiPV") { const int v = std::atoi(value.c_str()); if (v >= 1 && v <= 500) { u->multiPV = v; u->out += "info string MultiPV set to " + std::to_string(v) + "\n"; } }
    else if (name == "Ponder") { if (value == "true" || value == "false") { u->ponder = value == "true"; u->out += "info string Ponder set to " + value + "\n"; } }
    else if (name == "DrawContemptPermille") { const int v = permille(0, 1000); u->scfg.draw_contempt = (float)v / 1000.0f; rebuild = true; u->out += "info string DrawContemptPermille set to " + std::to_string(v) + "\n"; }
    else if (name == "PWCoefficientPermille") { const int v = permille(1, 10000); u->scfg.pw_coefficient = (float)v / 1000.0f; rebuild = true; u->out += "info string PWCoefficientPermille set to " + std::to_string(v) + "\n"; }
    else if (name == "RootPWCoefficientPermille") { const int v = permille(1, 10000); u->scfg.root_pw_coefficient = (float)v / 1000.0f; rebuild = true; u->out += "info string RootPWCoefficientPermille set to " + std::to_string(v) + "\n"; }
    else if (name == "PWExponentPermille") { const int v = permille(1, 1000); u->scfg.pw_exponent = (float)v / 1000.0f; rebuild = true; u->out += "info string PWExponentPermille set to " + std::to_string(v) + "\n"; }
    else if (name == "Transpositions") { if (value == "true" || value == "false") { u->scfg.enable_transpositions = value == "true"; rebuild = true; u->out += "info string Transpositions set to " + value + "\n"; } }
    else if (name == "Team") { if (value == "white") u->team = HM_WHITE; else if (value == "black") u->team = HM_BLACK; }
    else if (name == "Mode") { if (value == "sit") u->sit = true; else if (value == "go") u->sit = false; }
    if (rebuild) {
        // the search configuration lives in the engine handle: rebuild it and put the game back (its history restarts, as after
        // a `position` command, which the GUI sends before every `go` anyway)
        const hm_board keep = u->board;
        if (uci_rebuild_engine(u) || uci_set_position(u, keep)) u->out += std::string("info string setoption failed: ") + hm_last_error() + "\n";
    }
}

extern "C" {

// SearchParams::has_insurmountable_visit_lead (search_params.h:322-326)
int hm_insurmountable_visit_lead(float best_visits, float projected_second_visits, float factor) { return projected_second_visits * factor < best_visits ? 1 : 0; }

// the movetime controller alone, for tests: feed polls, read decisions (host-only)
hm_time_manager* hm_time_manager_create(int move_time_ms) { hm_time_manager* t = new hm_time_manager(); t->moveTimeMs = move_time_ms; t->effectiveMs = move_time_ms; return t; }
int hm_time_manager_poll(hm_time_manager* t, double elapsed_ms, int nodes, int n, const int* visits, const float* q, int root_type, const int* child_type,
                         const int* child_end_in_ply, double* effective_ms, char* log, int cap) {
    if (!t || (n > 0 && (!visits || !q))) return hm_fail(HM_ERR_INVALID, "null argument");
    const bool stop = tm_poll(*t, elapsed_ms, nodes, n, visits, q, root_type, child_type, child_end_in_ply);
    if (effective_ms) *effective_ms = t->effectiveMs;
    if (log && cap > 0) { std::snprintf(log, (size_t)cap, "%s", t->log.c_str()); }
    t->log.clear();
    return stop ? 1 : 0;
}
void hm_time_manager_destroy(hm_time_manager* t) { delete t; }

int hm_uci_create(const hm_net* net, const hm_eval_io* io, hm_eval_fn fn, void* user, int max_nodes, hm_uci** out) {
    if (!io || !out || (!net && !fn) || max_nodes <= 0) return hm_fail(HM_ERR_INVALID, "null argument");
    if (!io->planes[0] || !io->planes[1] || !io->value || !io->pi_a || !io->pi_b || !io->wdl || !io->moves_left) return hm_fail(HM_ERR_INVALID, "evaluator buffers missing");
    hm_uci* u = new hm_uci();
    u->net = net; u->fn = fn; u->user = user; u->io = *io; u->maxNodes = max_nodes;
    hm_search_config_default(&u->scfg);
    if (int rc = uci_rebuild_engine(u)) { delete u; return rc; }
    if (hipMalloc(&u->d_pos, sizeof(hm_pos)) != hipSuccess || hipMalloc(&u->d_moves, 4 * HM_MAX_MOVES) != hipSuccess || hipMalloc(&u->d_counts, 4) != hipSuccess) {
        hm_uci_destroy(u);
        return hm_fail(HM_ERR_NO_DEVICE, "hipMalloc failed");
    }
    hm_board b;
    hm_board_startpos(&b);
    if (int rc = uci_set_position(u, b)) { hm_uci_destroy(u); return rc; }
    *out = u;
    return 0;
}

// One UCI command line in, the engine's output text out (0-terminated, possibly several lines).  Returns the text length,
// -(needed size) when `cap` is too small (the command has been executed; call again with an empty line to fetch the text),
// and HM_UCI_QUIT for `quit`.
int64_t hm_uci_command(hm_uci* u, const char* line, char* out, int64_t cap) {
    if (!u || !line) return hm_fail(HM_ERR_INVALID, "null argument");
    std::istringstream is(line);
    std::string token;
    is >> std::skipws >> token;
    bool quit = false;
    auto say = [&](const char* s) { std::lock_guard<std::mutex> lock(u->mu); u->out += s; };
    if (token == "uci") {   // send_uci_response, uci.cc:298-317
        say("id name hivemind\nid author aminwoo\n\n"
                  "option name Hash type spin default 16 min 1 max 33554432\n"
                  "option name MultiPV type spin default 1 min 1 max 500\n"
                  "option name Ponder type check default true\n"
                  "option name DrawContemptPermille type spin default 0 min 0 max 1000\n"
                  "option name PWCoefficientPermille type spin default 1000 min 1 max 10000\n"
                  "option name RootPWCoefficientPermille type spin default 4000 min 1 max 10000\n"
                  "option name PWExponentPermille type spin default 300 min 1 max 1000\n"
                  "option name Transpositions type check default true\n"
                  "option name Team type combo default white var white var black\n"
                  "option name Mode type combo default go var sit var go\n"
                  "info string HIP engines 1 search workers 1 (one wavefront pipeline per game)\n"
                  "uciok\n");
    } else if (token == "isready") say("readyok\n");
    else if (token == "go") uci_go(u, is);
    else if (token == "ponderhit") u->pondering.store(false, std::memory_order_release);      // Agent::ponderhit (agent.cc:1304-1310)
    else if (token == "stop") uci_stop(u);
    else if (token == "setoption") { uci_stop(u); uci_setoption(u, is); }
    else if (token == "position") { uci_stop(u); uci_position(u, is); }
    else if (token == "ucinewgame") {          // UCI::new_game (uci.cc:77-86): reset_search_state drops the retained tree
        uci_stop(u);
        hm_board b;
        hm_board_startpos(&b);
        (void)uci_set_position(u, b);
        (void)hm_sp_set_tree_reuse(u->sp, nullptr, 1);
    }
    else if (token == "policy") { uci_stop(u); uci_policy(u); }
    else if (token == "quit") { uci_stop(u); quit = true; }
    std::lock_guard<std::mutex> lock(u->mu);
    if ((int64_t)u->out.size() + 1 > cap || !out) return quit ? HM_UCI_QUIT : -(int64_t)u->out.size() - 1;
    std::memcpy(out, u->out.c_str(), u->out.size() + 1);
    const int64_t n = (int64_t)u->out.size();
    u->out.clear();
    return quit ? HM_UCI_QUIT : n;
}

// 1 while a search (`go ...`) is running on its worker thread (its text arrives with later commands; an empty line polls).
int hm_uci_busy(hm_uci* u) { return u && u->busy.load(std::memory_order_acquire) ? 1 : 0; }

// Board::fen(board) (environment/board.h:172-174 -> Position::fen(false, true)) of a compact board; host-only.  Returns the text
// length, or -(needed size) when cap is too small.
int hm_board_fen(const hm_board* b, int board, char* out, int cap) {
    if (!b || board < 0 || board > 1) return hm_fail(HM_ERR_INVALID, "null board / board index not 0 or 1");
    const std::string s = fen_text(b->pos[board]);
    if (!out || (int)s.size() + 1 > cap) return -(int)s.size() - 1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

// the current game position (after the last `position` command) for tests and GUIs that mirror the board
int hm_uci_board(hm_uci* u, hm_board* out) {
    if (!u || !out) return hm_fail(HM_ERR_INVALID, "null argument");
    *out = u->board;
    out->team = (uint8_t)u->team; out->time_adv = u->sit ? 1 : 0;
    return 0;
}

int hm_uci_destroy(hm_uci* u) {
    if (!u) return 0;
    uci_stop(u);
    if (u->sp) hm_sp_destroy(u->sp);
    if (u->d_pos) (void)hipFree(u->d_pos);
    if (u->d_moves) (void)hipFree(u->d_moves);
    if (u->d_counts) (void)hipFree(u->d_counts);
    if (u->d_board) (void)hipFree(u->d_board);
    delete u;
    return 0;
}

}  // extern "C"
