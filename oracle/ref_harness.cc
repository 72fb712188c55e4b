// oracle/ref_harness.cc — thin C wrapper over the REFERENCE's own sources.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with the
// reference's unmodified sources where they lie under /root/reference (see
// oracle/Makefile) into oracle/_ref/libhmref.so.  No reference source is copied into
// the repo, and no stand-in header is used: only the parts of the reference that build
// with this image's toolchain as-is are included (Fairy-Stockfish, environment/board.cc,
// environment/zobrist.cc, common/globals.cc and the header-only joint_action.h /
// search_params.h).  environment/planes.cc, common/utils.h, search/* and tools/* need
// <cuda_fp16.h>/<nv/target>/TensorRT, which this image lacks -> unbuildable here.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "environment/board.h"
#include "environment/constants.h"
#include "environment/joint_action.h"
#include "common/globals.h"
#include "search/search_params.h"
#include "Fairy-Stockfish/src/bitboard.h"
#include "Fairy-Stockfish/src/movegen.h"
#include "Fairy-Stockfish/src/piece.h"
#include "Fairy-Stockfish/src/position.h"
#include "Fairy-Stockfish/src/thread.h"
#include "Fairy-Stockfish/src/types.h"

#include "../include/hivemind_amd.h"

using namespace Stockfish;

extern "C" {

void ref_init() {
    static bool done = false;
    if (done) return;
    done = true;
    // main.cc:75-81
    pieceMap.init();
    variants.init();
    Bitboards::init();
    Position::init();
    Threads.set(1);
    init_policy_index();
}

void* ref_board_new() { ref_init(); return new Board(); }
void* ref_board_clone(void* h) { return new Board(*static_cast<Board*>(h)); }
void ref_board_free(void* h) { delete static_cast<Board*>(h); }
void ref_board_set(void* h, const char* fen) { static_cast<Board*>(h)->set(fen); }
void ref_board_set_fen(void* h, int b, const char* fen) { static_cast<Board*>(h)->set_fen(b, fen); }

int ref_legal_moves(void* h, int b, uint32_t* out) {
    int n = 0;
    for (Move m : static_cast<Board*>(h)->legal_moves(b)) out[n++] = (uint32_t)m;
    return n;
}
void ref_push(void* h, int b, uint32_t m) { static_cast<Board*>(h)->push_move(b, Move(m)); }
void ref_pop(void* h, int b) { static_cast<Board*>(h)->pop_move(b); }
int ref_make_moves(void* h, uint32_t a, uint32_t b) {
    try { static_cast<Board*>(h)->make_moves(Move(a), Move(b)); } catch (const std::logic_error&) { return -1; }
    return 0;
}
void ref_unmake_moves(void* h, uint32_t a, uint32_t b) { static_cast<Board*>(h)->unmake_moves(Move(a), Move(b)); }
int ref_is_checkmate(void* h, int side, int adv) { return static_cast<Board*>(h)->is_checkmate(Color(side), adv != 0); }
int ref_is_draw(void* h, int ply) { return static_cast<Board*>(h)->is_draw(ply); }
uint64_t ref_hash_key(void* h, int adv) { return static_cast<Board*>(h)->hash_key(adv != 0); }
uint64_t ref_board_only_key(void* h, int b) { return static_cast<Board*>(h)->board_only_key(b); }
uint64_t ref_pos_key(void* h, int b) { return static_cast<Board*>(h)->pos[b]->key(); }
int ref_repetition_count(void* h, int b) { return static_cast<Board*>(h)->repetition_count(b); }
int ref_gives_check(void* h, int b, uint32_t m) { return static_cast<Board*>(h)->gives_check(b, Move(m)); }
int ref_is_capture(void* h, int b, uint32_t m) { return static_cast<Board*>(h)->is_capture(b, Move(m)); }
int ref_in_check(void* h, int b) { return static_cast<Board*>(h)->is_in_check(b); }
int ref_fen(void* h, int b, char* buf, int cap) {
    std::string s = static_cast<Board*>(h)->fen(b);
    std::strncpy(buf, s.c_str(), cap - 1);
    buf[cap - 1] = 0;
    return (int)s.size();
}
int ref_uci(void* h, int b, uint32_t m, char* buf, int cap) {
    std::string s = static_cast<Board*>(h)->uci_move(b, Move(m));
    std::strncpy(buf, s.c_str(), cap - 1);
    buf[cap - 1] = 0;
    return (int)s.size();
}

// Compact state straight from the reference's Position / Board accessors.
void ref_compact(void* h, int team, int adv, hm_board* o) {
    Board& bd = *static_cast<Board*>(h);
    std::memset(o, 0, sizeof *o);
    static const PieceType pts[6] = {PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING};
    for (int b = 0; b < 2; ++b) {
        hm_pos& q = o->pos[b];
        for (int i = 0; i < 6; ++i) q.by_type[i] = bd.pos[b]->pieces(pts[i]);
        q.by_color[0] = bd.pieces(b, WHITE);
        q.by_color[1] = bd.pieces(b, BLACK);
        q.promoted = bd.promoted_pieces(b);
        q.key = bd.pos[b]->state()->key;
        for (int c = 0; c < 2; ++c)
            for (int i = 0; i < 5; ++i) q.hand[c][i] = (uint8_t)bd.count_in_hand(b, Color(c), pts[i]);
        q.castling = (uint8_t)((bd.can_castle(b, WHITE_OO) ? 1 : 0) | (bd.can_castle(b, WHITE_OOO) ? 2 : 0)
                             | (bd.can_castle(b, BLACK_OO) ? 4 : 0) | (bd.can_castle(b, BLACK_OOO) ? 8 : 0));
        Square ep = bd.ep_square(b);
        q.ep = (uint8_t)(ep == SQ_NONE ? 64 : int(ep));
        q.stm = (uint8_t)bd.side_to_move(b);
        int r50 = bd.rule50_count(b);
        q.rule50 = (uint8_t)(r50 > 255 ? 255 : r50);
        q.game_ply = (uint16_t)bd.game_ply(b);
        o->last_move[b] = (uint32_t)bd.last_move(b);
        int rc = bd.repetition_count(b);
        o->rep_count[b] = (uint8_t)(rc > 3 ? 3 : rc);
    }
    o->team = (uint8_t)team;
    o->time_adv = adv ? 1 : 0;
}

// tools/benchmark.cc:59-76 restated over the reference Board (benchmark.cc itself needs CUDA headers)
static long long perft_rec(Board& board, int depth) {
    if (depth == 0) return 1;
    auto movesA = board.legal_moves(BOARD_A);
    auto movesB = board.legal_moves(BOARD_B);
    if (depth == 1) return (long long)movesA.size() * (long long)movesB.size();
    long long nodes = 0;
    for (const auto& a : movesA)
        for (const auto& b : movesB) {
            board.make_moves(a, b);
            nodes += perft_rec(board, depth - 1);
            board.unmake_moves(a, b);
        }
    return nodes;
}
long long ref_perft(void* h, int depth) { return perft_rec(*static_cast<Board*>(h), depth); }

// single-board perft over Position (engine/tests/test_move_gen.cc:1526-1571 known answers)
static long long perft1_rec(Board& board, int b, int depth) {
    auto moves = board.legal_moves(b);
    if (depth == 1) return (long long)moves.size();
    long long n = 0;
    for (auto m : moves) { board.push_move(b, m); n += perft1_rec(board, b, depth - 1); board.pop_move(b); }
    return n;
}
long long ref_perft_single(void* h, int b, int depth) { return perft1_rec(*static_cast<Board*>(h), b, depth); }
// timing helper of tools/cpu_baseline.py: `reps` calls of Board::legal_moves(b) on this position; returns the moves counted
long long ref_time_legal_moves(void* h, int b, int reps) {
    Board& board = *static_cast<Board*>(h);
    long long n = 0;
    for (int r = 0; r < reps; ++r) n += (long long)board.legal_moves(b).size();
    return n;
}

void ref_policy_tables(int* normal /*2*64*64*2*/, int* drop /*2*64*8*/) {
    ref_init();
    std::memcpy(normal, POLICY_TABLE_NORMAL, sizeof(POLICY_TABLE_NORMAL));
    std::memcpy(drop, POLICY_TABLE_DROP, sizeof(POLICY_TABLE_DROP));
}
int ref_policy_label(int idx, char* buf, int cap) {
    const std::string& s = UCI_MOVES[idx];
    std::strncpy(buf, s.c_str(), cap - 1);
    buf[cap - 1] = 0;
    return (int)s.size();
}
uint64_t ref_time_advantage_key() { return Zobrist::timeAdvantage; }

// search_params.h progressive-widening schedule + joint_action.h sit rules / generator
int ref_pw_allowed_children(int visits, int isRoot) {
    return SearchParams::get_allowed_children(visits,
        isRoot ? SearchParams::ROOT_PW_COEFFICIENT : SearchParams::PW_COEFFICIENT, SearchParams::PW_EXPONENT);
}
float ref_get_cpuct(float totalVisits) { return SearchParams::get_cpuct(totalVisits); }

}  // extern "C"
