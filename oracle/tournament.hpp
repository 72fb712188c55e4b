// oracle/tournament.hpp — CPU restatement of the reference's paired network tournament.
//
// TEST INFRASTRUCTURE ONLY (see bughouse.hpp).  Follows tools/tournament.cc line by line:
//   tournament_seed :22-27, most_visited_action :29-39, action_uci :41-49, pgn_result :51-59,
//   record_outcome :61-69, append_game_pgn :89-115, write_summary :117-187,
//   TournamentResult statistics :247-326, run_tournament :328-465; TournamentConfig tools/tournament.h:15-42;
//   UCI::move Fairy-Stockfish/src/stubs.cpp:21-59 for the bughouse variant (no gating, no chess960).
// on top of oracle::Search (one search thread = the reference's only deterministic configuration).
//
// PARITY STATUS: tools/tournament.cc cannot be built in this image (search/agent.h -> nn/engine.h -> TensorRT).  The
// statistics are pinned by the reference's own known answers (engine/tests/test_tournament.cc:7-72, replayed in
// tests/test_oracle_tournament.py); move strings are compared with the reference build's Board::uci_move on the
// reference playouts; the game loop itself has no reference-held fixture ("parity unpinned"), it is pinned through its
// parts (search known answers, Board rules from the reference build).
#pragma once
#include <cstdio>
#include <iomanip>
#include <numeric>
#include <optional>
#include <sstream>
#include <string>

#include "search.hpp"

namespace hmo {

struct TournamentConfig {   // tools/tournament.h:15-42
    size_t games = 20, nodes = 400;
    int moveTimeMs = 0;
    int contenderBatchSize = 8, baselineBatchSize = 8;
    size_t maxMacroPlies = 400;
    float dirichletAlpha = 0.3f, dirichletEpsilon = 0.10f;
    float contenderPwCoefficient = 2.0f, baselinePwCoefficient = 2.0f;
    uint64_t seed = 1;
    float pwCoefficientFor(bool isContender) const { return isContender ? contenderPwCoefficient : baselinePwCoefficient; }
};

struct TournamentBreakdown {
    size_t wins = 0, losses = 0, draws = 0;
    size_t games() const { return wins + losses + draws; }
};

struct TournamentResult {   // tools/tournament.h:50-73, tournament.cc:247-326
    size_t contenderWins = 0, baselineWins = 0, draws = 0;
    TournamentBreakdown asWhite, asBlack, upTime, downTime;
    size_t checkmates = 0, noLegalActions = 0, drawnTerminations = 0, macroPlyLimits = 0;
    std::vector<double> pairScores;

    size_t games() const { return contenderWins + baselineWins + draws; }
    double contenderScore() const {
        if (games() == 0) return 0.0;
        return (static_cast<double>(contenderWins) + 0.5 * static_cast<double>(draws)) / static_cast<double>(games());
    }
    std::optional<double> contenderElo() const {
        const double score = contenderScore();
        if (games() == 0 || score <= 0.0 || score >= 1.0) return std::nullopt;
        return 400.0 * std::log10(score / (1.0 - score));
    }
    std::optional<std::pair<double, double>> scoreConfidence95() const {
        if (games() == 0) return std::nullopt;
        constexpr double z = 1.959963984540054;
        if (pairScores.size() >= 2) {
            const double count = static_cast<double>(pairScores.size());
            const double mean = std::accumulate(pairScores.begin(), pairScores.end(), 0.0) / count;
            const double squaredError = std::accumulate(pairScores.begin(), pairScores.end(), 0.0,
                [mean](double total, double score) { const double d = score - mean; return total + d * d; });
            const double sampleVariance = squaredError / (count - 1.0);
            const double margin = z * std::sqrt(sampleVariance / count);
            return std::pair{std::max(0.0, mean - margin), std::min(1.0, mean + margin)};
        }
        const double count = static_cast<double>(games());
        const double score = contenderScore();
        const double denominator = 1.0 + z * z / count;
        const double center = (score + z * z / (2.0 * count)) / denominator;
        const double margin = z * std::sqrt(score * (1.0 - score) / count + z * z / (4.0 * count * count)) / denominator;
        return std::pair{std::max(0.0, center - margin), std::min(1.0, center + margin)};
    }
    std::optional<std::pair<double, double>> eloConfidence95() const {
        const auto si = scoreConfidence95();
        if (!si || si->first <= 0.0 || si->second >= 1.0) return std::nullopt;
        auto score_to_elo = [](double score) { return 400.0 * std::log10(score / (1.0 - score)); };
        return std::pair{score_to_elo(si->first), score_to_elo(si->second)};
    }
    std::string confidenceMethod() const {
        return pairScores.size() >= 2 ? "paired-opening normal approximation" : "game-level Wilson approximation";
    }
};

inline uint64_t tournament_seed(uint64_t seed, uint64_t value) {   // :22-27
    value += 0x9e3779b97f4a7c15ULL;
    value = (value ^ (value >> 30)) * 0xbf58476d1ce4e5b9ULL;
    value = (value ^ (value >> 27)) * 0x94d049bb133111ebULL;
    return seed ^ (value ^ (value >> 31));
}

// UCI::move (stubs.cpp:21-59) for this variant: castling is encoded king -> rook square and printed king -> g/c file
// (a two-square king move is never a pseudo-legal plain king move, so the 960 fallback of :37-38 cannot trigger).
inline std::string move_uci(Move m) {
    if (m == MOVE_NONE) return "(none)";
    auto sq = [](int s) { std::string r; r += char('a' + (s & 7)); r += char('1' + (s >> 3)); return r; };
    int from = (int)((m >> 6) & 63), to = (int)(m & 63);
    const uint32_t mt = m & (15u << 12);
    if (mt == HM_MT_CASTLING) to = (to > from ? 6 : 2) + (from & 56);
    std::string s;
    if (mt == HM_MT_DROP) { s += " PNBRQ"[(m >> 16) & 63]; s += '@'; }
    else s += sq(from);
    s += sq(to);
    if (mt == HM_MT_PROMOTION) s += " pnbrq"[(m >> 16) & 63];
    return s;
}
inline std::string board_uci_move(Move m) { return m == MOVE_NONE ? "pass" : move_uci(m); }   // Board::uci_move board.h:340-350

struct TournamentGame {
    int winner = -1;                  // -1 none, WHITE / BLACK (tournament.cc:51-59)
    std::string termination = "macro-ply limit";
    std::vector<std::string> actions;
    uint64_t searched = 0, nodes = 0;
};

class Tournament {
public:
    TournamentConfig cfg;
    Search search;                    // tie_mode / exp_mode set by the caller; the evaluator is switched per macro-ply
    Evaluator contender, baseline;
    TournamentResult result;
    std::vector<TournamentGame> games;

    void validate() const {   // :334-358
        if (cfg.games == 0 || cfg.games % 2 != 0) throw std::invalid_argument("Tournament games must be a positive even number");
        if ((cfg.nodes == 0) == (cfg.moveTimeMs <= 0) || cfg.maxMacroPlies == 0)
            throw std::invalid_argument("Tournament requires exactly one positive nodes or movetime limit");
        if (cfg.contenderBatchSize <= 0 || cfg.baselineBatchSize <= 0) throw std::invalid_argument("Tournament batch sizes must be positive");
        if (cfg.dirichletAlpha < 0.0f || cfg.dirichletEpsilon < 0.0f || cfg.dirichletEpsilon > 1.0f)
            throw std::invalid_argument("Invalid tournament Dirichlet configuration");
        if (!std::isfinite(cfg.contenderPwCoefficient) || !std::isfinite(cfg.baselinePwCoefficient)
            || cfg.contenderPwCoefficient <= 0.0f || cfg.baselinePwCoefficient <= 0.0f)
            throw std::invalid_argument("Tournament PW coefficients must be positive and finite");
    }

    TournamentGame play_game(size_t gameIndex) {   // body of the game loop :367-415
        TournamentGame g;
        Board board;
        const size_t pairIndex = gameIndex / 2;
        const int contenderTeam = gameIndex % 2 == 0 ? WHITE : BLACK;
        int team = pairIndex % 2 == 0 ? WHITE : BLACK;
        bool adv = false;
        for (size_t macroPly = 0; macroPly < cfg.maxMacroPlies; ++macroPly) {
            if (board.is_checkmate(team, adv)) { g.winner = team ^ 1; g.termination = "checkmate"; break; }
            if (board.is_draw()) { g.termination = "draw"; break; }
            const bool contenderActing = team == contenderTeam;
            search.evaluator = contenderActing ? contender : baseline;
            search.cfg.batchSize = contenderActing ? cfg.contenderBatchSize : cfg.baselineBatchSize;   // the acting network's Engine batch size (tournament.h:19-20; searchthread.cc:663 ensureBufferSize(engine->getBatchSize()))
            search.cfg.pwCoefficient = cfg.pwCoefficientFor(contenderActing);      // searchConfigFor tournament.h:34-41
            search.cfg.rootPwCoefficient = cfg.pwCoefficientFor(contenderActing);
            search.cfg.rootDirichletAlpha = cfg.dirichletAlpha;
            search.cfg.rootDirichletEpsilon = cfg.dirichletEpsilon;
            search.cfg.rootNoiseSeed = tournament_seed(cfg.seed, pairIndex * cfg.maxMacroPlies + macroPly);
            const bool ok = search.run(board, team, adv, (int)cfg.nodes);
            const std::vector<RootEdge> edges = ok ? search.root_edge_stats() : std::vector<RootEdge>();
            if (edges.empty()) { g.winner = team ^ 1; g.termination = "no legal action"; break; }
            g.searched++;
            for (const RootEdge& e : edges) g.nodes += (uint64_t)std::max(0, e.visits);
            const RootEdge& action = *std::max_element(edges.begin(), edges.end(),   // most_visited_action :29-39
                [](const RootEdge& l, const RootEdge& r) { return l.visits < r.visits; });
            g.actions.push_back("(" + board_uci_move(action.moveA) + "," + board_uci_move(action.moveB) + ")");   // action_uci :41-49
            board.make_moves(action.moveA, action.moveB);
            team ^= 1; adv = !adv;
        }
        // after the loop (macro-ply limit reached): :417-422.  `team` / `adv` are the side to move now.
        // (when the loop broke on its own test, winner / termination are already final.)
        return finish(g, board, team, adv);
    }

    void run() {
        validate();
        result = TournamentResult();
        games.clear();
        double currentPairPoints = 0.0;
        for (size_t gameIndex = 0; gameIndex < cfg.games; ++gameIndex) {
            games.push_back(play_game(gameIndex));
            account(gameIndex, games.back(), currentPairPoints);
        }
    }

    // bookkeeping of one finished game (:424-452)
    void account(size_t gameIndex, const TournamentGame& g, double& currentPairPoints) {
        const size_t pairIndex = gameIndex / 2;
        const int contenderTeam = gameIndex % 2 == 0 ? WHITE : BLACK;
        const int startTeam = pairIndex % 2 == 0 ? WHITE : BLACK;
        const bool contenderHasTimeAdvantage = contenderTeam != startTeam;
        int contenderOutcome = 0;
        if (g.winner < 0) result.draws++;
        else if (g.winner == contenderTeam) { result.contenderWins++; contenderOutcome = 1; }
        else { result.baselineWins++; contenderOutcome = -1; }
        auto record = [&](TournamentBreakdown& b) { if (contenderOutcome > 0) b.wins++; else if (contenderOutcome < 0) b.losses++; else b.draws++; };
        record(contenderTeam == WHITE ? result.asWhite : result.asBlack);
        record(contenderHasTimeAdvantage ? result.upTime : result.downTime);
        currentPairPoints += contenderOutcome > 0 ? 1.0 : contenderOutcome == 0 ? 0.5 : 0.0;
        if (gameIndex % 2 == 1) { result.pairScores.push_back(currentPairPoints / 2.0); currentPairPoints = 0.0; }
        if (g.termination == "checkmate") result.checkmates++;
        else if (g.termination == "no legal action") result.noLegalActions++;
        else if (g.termination == "draw") result.drawnTerminations++;
        else result.macroPlyLimits++;
    }

    static std::string pgn_result(int winner) { return winner == WHITE ? "1-0" : winner == BLACK ? "0-1" : "1/2-1/2"; }   // :51-59

    // games.pgn as append_game_pgn (:89-115) leaves it after the last game
    std::string pgn(const std::string& contenderName, const std::string& baselineName) const {
        std::ostringstream o;
        for (size_t i = 0; i < games.size(); ++i) {
            const TournamentGame& g = games[i];
            const int contenderTeam = i % 2 == 0 ? WHITE : BLACK;
            const std::string res = pgn_result(g.winner);
            o << "[Event \"Hivemind Network Tournament\"]\n" << "[Site \"Hivemind Engine\"]\n" << "[Round \"" << (i + 1) << "\"]\n"
              << "[Variant \"bughouse\"]\n" << "[WhiteTeam \"" << (contenderTeam == WHITE ? contenderName : baselineName) << "\"]\n"
              << "[BlackTeam \"" << (contenderTeam == BLACK ? contenderName : baselineName) << "\"]\n"
              << "[Result \"" << res << "\"]\n" << "[Termination \"" << g.termination << "\"]\n\n";
            for (size_t k = 0; k < g.actions.size(); ++k) o << (k + 1) << ". " << g.actions[k] << ' ';
            o << res << "\n\n";
        }
        return o.str();
    }

    // summary.json as write_summary (:117-187) leaves it after the last game
    std::string summary(const std::string& contenderName, const std::string& baselineName) const {
        std::ostringstream s;
        s << std::fixed << std::setprecision(6) << "{\n"
          << "  \"contender\": \"" << contenderName << "\",\n" << "  \"baseline\": \"" << baselineName << "\",\n"
          << "  \"games\": " << result.games() << ",\n" << "  \"nodes_per_move\": " << cfg.nodes << ",\n"
          << "  \"move_time_ms\": " << cfg.moveTimeMs << ",\n" << "  \"contender_batch_size\": " << cfg.contenderBatchSize << ",\n"
          << "  \"baseline_batch_size\": " << cfg.baselineBatchSize << ",\n" << "  \"seed\": " << cfg.seed << ",\n"
          << "  \"contender_pw_coefficient\": " << cfg.contenderPwCoefficient << ",\n"
          << "  \"baseline_pw_coefficient\": " << cfg.baselinePwCoefficient << ",\n"
          << "  \"contender_wins\": " << result.contenderWins << ",\n" << "  \"baseline_wins\": " << result.baselineWins << ",\n"
          << "  \"draws\": " << result.draws << ",\n" << "  \"contender_score\": " << result.contenderScore() << ",\n"
          << "  \"contender_elo\": ";
        if (const auto elo = result.contenderElo()) s << *elo; else s << "null";
        s << ",\n  \"confidence_method\": \"" << result.confidenceMethod() << "\",\n" << "  \"score_confidence_95\": ";
        if (const auto iv = result.scoreConfidence95()) s << '[' << iv->first << ", " << iv->second << ']'; else s << "null";
        s << ",\n  \"elo_confidence_95\": ";
        if (const auto iv = result.eloConfidence95()) s << '[' << iv->first << ", " << iv->second << ']'; else s << "null";
        s << ",\n  \"contender_breakdown\": {\n";
        auto bd = [&](const char* name, const TournamentBreakdown& b, bool comma) {
            s << "    \"" << name << "\": {\"wins\": " << b.wins << ", \"losses\": " << b.losses << ", \"draws\": " << b.draws << "}" << (comma ? "," : "") << '\n';
        };
        bd("white", result.asWhite, true); bd("black", result.asBlack, true); bd("up_time", result.upTime, true); bd("down_time", result.downTime, false);
        s << "  },\n" << "  \"terminations\": {\n" << "    \"checkmate\": " << result.checkmates << ",\n"
          << "    \"no_legal_action\": " << result.noLegalActions << ",\n" << "    \"draw\": " << result.drawnTerminations << ",\n"
          << "    \"macro_ply_limit\": " << result.macroPlyLimits << "\n" << "  }\n" << "}\n";
        return s.str();
    }

private:
    static TournamentGame finish(TournamentGame g, Board& board, int team, bool adv) {   // :417-422
        if (g.winner < 0 && g.termination == "macro-ply limit") {
            if (board.is_checkmate(team, adv)) { g.winner = team ^ 1; g.termination = "checkmate"; }
            else if (board.is_draw()) g.termination = "draw";
        }
        return g;
    }
};

}  // namespace hmo
