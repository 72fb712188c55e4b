/*
 * hivemind_amd.h — C ABI of the MI355X-native Bughouse rollout engine.
 *
 * This is the drop-in boundary for the self-play hot path of aminwoo/hivemind.
 * Every entry point names the reference interface it replaces (file:line is
 * relative to the reference tree's engine/src/).  Plain pointers and sizes
 * only; device pointers are HIP device memory (e.g. torch.Tensor.data_ptr()).
 * All functions return 0 on success and a negative hm_status on failure;
 * hm_last_error() gives the message.  Nothing here falls back to a CPU path:
 * a call that needs the GPU fails with HM_ERR_NO_DEVICE when there is none.
 */
#ifndef HIVEMIND_AMD_H
#define HIVEMIND_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants (environment/constants.h:10-22) ---- */
#define HM_NB_PLANES            74      /* NB_INPUT_CHANNELS */
#define HM_NB_PLANES_PER_BOARD  37
#define HM_PLANE_VALUES         4736    /* NB_INPUT_VALUES() = 74*8*8 */
#define HM_POLICY_VALUES        4672    /* NB_POLICY_VALUES() = 73*8*8 */
#define HM_MAX_MOVES            512     /* per-board legal move cap of the batched movegen
                                           (reference MAX_MOVES=1024, types.h:226; bound: 48 P-drops + 4*62 drops + 8 king = 304) */

/* Move encoding is Fairy-Stockfish's 32-bit Move (types.h:237-263,735-799):
 * bits 0-5 to, 6-11 from, 12-15 type, 16-21 promotion / dropped piece type,
 * 22-27 in-hand piece type.  0 == MOVE_NONE == pass/sit (board.h:340-343). */
typedef uint32_t hm_move;
#define HM_MOVE_NONE       0u
#define HM_MT_NORMAL       (0u << 12)
#define HM_MT_EN_PASSANT   (1u << 12)
#define HM_MT_CASTLING     (2u << 12)   /* king-from -> rook-square */
#define HM_MT_PROMOTION    (3u << 12)
#define HM_MT_DROP         (4u << 12)

enum { HM_WHITE = 0, HM_BLACK = 1 };
enum { HM_PAWN = 1, HM_KNIGHT = 2, HM_BISHOP = 3, HM_ROOK = 4, HM_QUEEN = 5, HM_KING = 6 };
/* castling bits == Stockfish::CastlingRights (types.h:262-277) */
enum { HM_WHITE_OO = 1, HM_WHITE_OOO = 2, HM_BLACK_OO = 4, HM_BLACK_OOO = 8 };

typedef enum hm_status {
    HM_OK = 0,
    HM_ERR_INVALID = -1,      /* bad argument */
    HM_ERR_NO_DEVICE = -2,    /* no HIP device / kernel launch failed */
    HM_ERR_ILLEGAL = -3,      /* illegal joint action (board.cc:317-322 throws logic_error) */
    HM_ERR_OVERFLOW = -4,     /* a caller-owned buffer or pool is too small */
    HM_ERR_STATE = -5         /* call made in the wrong state (e.g. second enqueue before sync) */
} hm_status;

/* One board.  Replaces Stockfish::Position + StateInfo (position.h:40-85,338-358)
 * for the bughouse variant: 96 bytes, no pointers, trivially copyable (copy-make
 * instead of do/undo).  `key` is bit-identical to StateInfo::key. */
typedef struct hm_pos {
    uint64_t by_type[6];    /* P,N,B,R,Q,K bitboards, both colours; a1 = bit 0 */
    uint64_t by_color[2];   /* white, black */
    uint64_t promoted;      /* Position::promotedPieces */
    uint64_t key;           /* Zobrist key incl. side, castling, ep file, in-hand counts */
    uint8_t  hand[2][5];    /* pieceCountInHand[colour][P,N,B,R,Q] */
    uint8_t  castling;      /* CastlingRights bits */
    uint8_t  ep;            /* en-passant square or 64 = none */
    uint8_t  stm;           /* side to move */
    uint8_t  rule50;        /* halfmove clock */
    uint16_t game_ply;
} hm_pos;

/* Two boards + the history-derived scalars the plane encoder needs.
 * Replaces `Board` (environment/board.h:25-454) as the encoder's input. 208 bytes. */
typedef struct hm_board {
    hm_pos   pos[2];        /* BOARD_A, BOARD_B */
    uint32_t last_move[2];  /* Board::last_move(b) (board.h:319-324); 0 = none */
    uint8_t  rep_count[2];  /* Board::repetition_count(b) (board.h:326-332), saturated at 3 */
    uint8_t  team;          /* teamSide the planes are oriented for */
    uint8_t  time_adv;      /* hasTimeAdvantage */
    uint32_t reserved;
} hm_board;

/* plane output element types */
enum { HM_DT_F16 = 0, HM_DT_F32 = 1, HM_DT_U8 = 2 };

/* ------------------------------------------------------------------ */
/* library                                                              */
/* ------------------------------------------------------------------ */

/* Builds attack/Zobrist/policy tables on the host and uploads them to the
 * current HIP device.  Replaces the start-up sequence of main.cc:75-81
 * (Bitboards::init, Position::init, init_policy_index).  Idempotent.
 * Process model: ONE process per GPU.  The rule / policy-index tables are
 * process-wide device allocations made on the device that was current at
 * the first hm_init; the reference's "one process, a list of devices"
 * shape (main.cc:154-173) maps to one rank per device (bench.py --gpus N
 * starts them; torchrun does the same), not to one process driving several
 * devices, which this library does not support. */
int hm_init(int device);
/* Message of the last failure on this thread (never NULL). */
const char* hm_last_error(void);
/* 1 when a HIP device is usable by this library build, else 0. */
int hm_device_available(void);
/* Library ABI version. */
int hm_abi_version(void);

/* ------------------------------------------------------------------ */
/* positions                                                            */
/* ------------------------------------------------------------------ */

/* Fills `out` with the dual start position (Board::Board(), board.cc:52-69). */
int hm_board_startpos(hm_board* out);

/* Host-side helper: policy index of a move for the side to move
 * (get_fast_policy_index, common/utils.h:184-216). -1 = unrepresentable. */
int hm_policy_index(hm_move m, int stm);

/* ------------------------------------------------------------------ */
/* batched plane encoder: board_to_planes (environment/planes.h:20-22,  */
/* planes.cc:213-265), one launch for n positions.                      */
/* d_boards: device, n * sizeof(hm_board).  d_out: device,              */
/* n * 4736 elements of `dtype` (U8 = round(v*255), selfplay.cc:464-476).*/
/* stream: hipStream_t as void* (NULL = default stream).                */
/* ------------------------------------------------------------------ */
int hm_encode_planes(const hm_board* d_boards, size_t n, int dtype, void* d_out, void* stream);

/* ------------------------------------------------------------------ */
/* batched legal move generation: Board::legal_moves(board_num)         */
/* (board.cc:133-139 -> generate<LEGAL>, movegen.cpp:439-456), in the   */
/* reference's list order.  d_pos: n positions; d_moves: n*HM_MAX_MOVES;*/
/* d_counts: n.                                                         */
/* ------------------------------------------------------------------ */
int hm_legal_moves(const hm_pos* d_pos, size_t n, hm_move* d_moves, uint32_t* d_counts, void* stream);

/* Legal move COUNT only (MoveList<LEGAL>::size(), used by is_checkmate board.cc:169-208 and
 * the perft depth-1 shortcut benchmark.cc:66). d_counts: n. */
/* Test hook: the same lists from the wave-cooperative generator the search kernels use (one wavefront per position). */
int hm_legal_moves_wave(const hm_pos* d_pos, size_t n, hm_move* d_moves, uint32_t* d_counts, void* stream);
int hm_count_moves(const hm_pos* d_pos, size_t n, uint32_t* d_counts, void* stream);

/* Batched joint make: Board::make_moves (board.cc:316-341) applied to
 * d_boards[i] with (d_move_a[i], d_move_b[i]); writes d_out[i]; no legality
 * re-check (the caller passes moves produced by hm_legal_moves). */
int hm_make_moves(const hm_board* d_boards, const hm_move* d_move_a, const hm_move* d_move_b,
                  size_t n, hm_board* d_out, void* stream);

/* ------------------------------------------------------------------ */
/* joint perft: benchmark_movegen / perft (tools/benchmark.cc:59-97).   */
/* Counts with the reference's convention (depth 1 = |A|*|B|) from      */
/* `root` (host pointer).  Runs entirely on the device.                 */
/* shard/nshards split the depth-2 frontier across ranks (weak scaling  */
/* by game/stripe: no collective).  *nodes = this shard's count.        */
/* ------------------------------------------------------------------ */
int hm_perft(const hm_board* root, int depth, int shard, int nshards, uint64_t* nodes, double* seconds);

/* ================================================================== */
/* GPU-resident self-play search engine (rollout seam).                */
/*                                                                      */
/* Replaces Agent::run_search + SearchThread + Node + TranspositionTable*/
/* (search/agent.h:136-144, search/searchthread.cc:255-916) for G       */
/* concurrent games, and the evaluator seam of class Engine             */
/* (nn/engine.h:43-81) as a pair of device tensors: the engine WRITES   */
/* fp16 planes [G*8, 74, 8, 8] (collect) and READS fp16 heads (process):*/
/* value [G*8], pi_a / pi_b [G*8, 4672] raw logits plane-major, wdl     */
/* [G*8, 3] ordered loss,draw,win, moves_left [G*8]                     */
/* (searchthread.cc:474-484, 576-578, 609-617).  Row g*8+k belongs to   */
/* game g; unused rows are ignored.  One lockstep iteration =           */
/* hm_sp_collect(planes_next) || net(planes_cur) -> hm_sp_process, then */
/* the caller swaps planes_cur / planes_next (the reference's           */
/* double-buffered lookahead, searchthread.cc:661-708): collect only    */
/* writes planes_next, so it may overlap the network on another stream. */
/* ================================================================== */
typedef struct hm_sp hm_sp;

/* search/search_params.h:26-295 (RuntimeConfig + constants) */
typedef struct hm_search_config {
    float cpuct_init, cpuct_base;          /* 2.5, 19652 */
    float fpu_reduction;                   /* 1.0 */
    float draw_contempt;                   /* 0.0 */
    float wdl_value_weight;                /* 0.25 */
    float moves_left_discount;             /* 0.005 */
    float pw_coefficient, root_pw_coefficient, pw_exponent;   /* 2, 4, 0.4 */
    int   enable_transpositions;           /* 1 */
    int   enable_dynamic_fpu;              /* 1 */
    int   enable_wdl_eval;                 /* 1 */
} hm_search_config;
void hm_search_config_default(hm_search_config* cfg);

/* n_games game slots, trees sized for searches of up to max_nodes nodes. */
int hm_sp_create(int n_games, int max_nodes, const hm_search_config* cfg, hm_sp** out);
/* As hm_sp_create, with the per-board game history sized for games of up to max_game_plies macro-plies
 * (Board::positionHistory grows by one key per push, board.h:95-102); 0 = the default of 1024 keys. */
int hm_sp_create_ex(int n_games, int max_nodes, int max_game_plies, const hm_search_config* cfg, hm_sp** out);
int hm_sp_destroy(hm_sp* sp);
/* Side to act per game slot (team[g] = HM_WHITE / HM_BLACK, time_adv[g] = the team sits with the time advantage), positions and
 * history untouched: the UCI options Team / Mode (interface/uci.cc:283-296) after a `position ... moves` replay. */
int hm_sp_set_side(hm_sp* sp, const uint8_t* team, const uint8_t* time_adv);
/* Ends the searches of the masked slots (all if NULL) at their next collect: Agent::set_is_running(false) — UCI `stop`, movetime. */
int hm_sp_stop(hm_sp* sp, const uint8_t* mask, void* stream);
/* Progressive-widening schedule per game slot (TournamentConfig::searchConfigFor, tools/tournament.h:34-41): profiles[g] != 0
 * makes slot g search with the alternate coefficients (root and interior); 0 = the engine's hm_search_config. */
int hm_sp_set_pw_profiles(hm_sp* sp, float alt_pw_coefficient, float alt_root_pw_coefficient, const uint8_t* profiles);
/* Leaves collected per search iteration, per game slot: the reference's SearchThread collects engine->getBatchSize() leaves
 * (search/searchthread.cc:258-273, 663; Engine(deviceId, batchSize) nn/engine.h:43; tournaments give each network its own,
 * tools/tournament.h:19-20).  batch[g] in 1 .. 8 (NULL: 8 everywhere, the reference's SearchParams::BATCH_SIZE); stays until the
 * next call.  Larger batches are not built (context slots, plane rows and the leaf ring are sized for 8). */
int hm_sp_set_batch_sizes(hm_sp* sp, const uint8_t* batch);
/* (Re)start games from host boards[n_games] (Board::set, board.cc:27-49: history restarts);
 * team / time_adv of each hm_board give the side to act.  mask[g]==0 leaves game g alone. */
int hm_sp_set_games(hm_sp* sp, const hm_board* boards, const uint8_t* mask);
/* Agent::run_search prologue for every (masked) game: early outs, 1-ply root mate scan, root +
 * TT setup (agent.cc:421-558).  Dirichlet noise (alpha, eps, per-game rootNoiseSeed) as
 * node.h:286-315; alpha == 0 disables it. */
int hm_sp_begin_search(hm_sp* sp, const int* target_nodes, const uint64_t* noise_seeds, float alpha, float eps, const uint8_t* mask);
int hm_sp_collect(hm_sp* sp, void* d_planes_next, void* stream);
/* As hm_sp_collect; additionally d_rows_next[g] (device, n_games ints) = the number of plane rows game g
 * wrote, i.e. the batchSize its search thread hands to Engine::run (searchthread.cc:474-484). */
int hm_sp_collect_counted(hm_sp* sp, void* d_planes_next, int32_t* d_rows_next, void* stream);
/* active_games (host, optional): number of games still searching after this step (forces a sync). */
int hm_sp_process(hm_sp* sp, const void* d_value, const void* d_pi_a, const void* d_pi_b, const void* d_wdl,
                  const void* d_moves_left, int* active_games, void* stream);
/* Leg clock: exact device time (ms) and launch counts of k_collect / the forward / k_process since the last reset — every
 * workgroup min-/max-es the constant 100 MHz device clock into the leg's interval, the next kernel in stream order folds it into a
 * sum, so graph-replayed launches are covered too (ms3 / counts3 may be NULL).  The forward takes part when it is launched through
 * hm_net_forward_groups_timed with hm_sp_leg_clock_net(sp). */
int hm_sp_leg_times(hm_sp* sp, double* ms3, uint64_t* counts3, int reset);
uint64_t* hm_sp_leg_clock_net(hm_sp* sp);
/* The whole node-budget search of every slot hm_sp_begin_search left searching, with the native evaluator, as TWO PERSISTENT
 * KERNELS joined by a device-side queue (hivemind_amd/csrc/hm_queue.hpp) instead of the host loop
 * hm_sp_collect || forward -> hm_sp_process: one workgroup per game stays alive for the search (Agent's worker loop,
 * agent.cc:331-352, around SearchThread::run_iteration, searchthread.cc:661-739) with its node pool in LDS, and hands each
 * collected batch to evaluator workgroups (Engine::enqueueInferenceHalf / synchronizeInferenceHalf, nn/engine.h:43-81, one
 * position per workgroup) the moment it is written.  Per game the order of tree operations — hence every visit count — is the
 * lockstep one; a game no longer waits for the slowest game of an iteration.  io: both plane buffers and both sets of heads
 * (hm_eval_io of the native mode).  Synchronous (synchronises the device first).  ONE kernel launch (k_rollout, hm_rollout.hip):
 * the first workgroups of the grid are the games, the others the evaluator, joined by a device-side queue (hm_queue.hpp) — so it
 * also runs under profilers that serialise kernels (rocprofv3 --pmc).  Needs hm_sp_search_consumers(sp) > 0 (game slots plus at
 * least 8 evaluator workgroups within the device's CU count) and a network whose trunk has 64 / 128 / 384 channels; otherwise
 * HM_ERR_INVALID — use the lockstep calls.  A search the hang guard gives up (hm_sp_search_stalled) prints its give-up record on
 * stderr and returns HM_ERR_STATE. */
struct hm_net;
struct hm_eval_io;
int hm_sp_search(hm_sp* sp, const struct hm_net* net, const struct hm_eval_io* io,
                 double* search_kernel_ms /* optional: duration of the k_rollout launch, HIP events on its stream */);
/* 1 when the last hm_sp_search gave up because no workgroup of the other role came in within 3 s (the grid was not resident at once:
 * something else held the device's CUs); hm_sp_begin_again then restores the state hm_sp_begin_search had left (same targets,
 * seeds, mask and noise; refused with HM_ERR_STATE while a slot keeps its tree between searches) so that the lockstep calls can
 * run the search instead. */
int hm_sp_search_not_concurrent(const hm_sp* sp);
/* 1 when the last hm_sp_search was given up by its hang guard: neither the queue tail nor any game's phase had moved for 50 ms
 * while games were still searching (hm_queue.hpp: IDLE_LIMIT_TICKS).  hm_sp_begin_again puts every slot back to the start of
 * that search; running it again — single-launch or lockstep — gives the same result. */
int hm_sp_search_stalled(const hm_sp* sp);
/* 1 when the single-launch search keeps this engine's node pool in LDS for a whole search (the pool fits behind the search role's
 * fixed LDS), 0 when it walks the tree in place. */
int hm_sp_search_lds_tree(const hm_sp* sp);
int hm_sp_begin_again(hm_sp* sp);
/* Evaluator workgroups a persistent search of this engine runs (0: not available for this many game slots on this device). */
int hm_sp_search_consumers(const hm_sp* sp);
/* Persistent searches: total time (ms) the games spent waiting for the evaluation of their pending batches since the last
 * hm_sp_leg_times reset.  In that mode hm_sp_leg_times reports per game-iteration sums: [0] collect phases, [2] process phases
 * (counts = game-iterations), [1] the evaluator's time and positions. */
int hm_sp_wait_time(hm_sp* sp, double* ms);
/* Games still searching after the last hm_sp_process (synchronises). */
int hm_sp_active(hm_sp* sp, int* active);
/* Agent::root_edge_stats / root_q (agent.cc:1004-1024) for all games -> host arrays
 * [n_games][max_edges]; info[g][HM_SP_INFO_INTS] = status, nodes, eval rows, same-batch collisions,
 * reservation collisions, node count, root type, root visits, overflow flags, max depth,
 * tree nodes visited and edges scanned during selection (traffic accounting), [12] = index of the
 * joint action Agent::run_search returns (get_best_move_idx_with_q_weight, node.h:656-754 with
 * Q_VETO_DELTA 0.4 / Q_VALUE_WEIGHT 1.0, then the fallbacks of agent.cc:872-886), [13] = leaf move-list words written
 * (traffic accounting), [14] / [15] = solver type (0 unsolved, 1 win, 2 loss, 3 draw — from the child's side) and endInPly of that
 * action's child node (format_uci_score, agent.cc:48-78), [16] = visits of the root recovered by tree reuse (-1 = fresh root),
 * [17] = node budget in force (hm_sp_set_tree_reuse), [18] / [19] = transposition-table lookups of this search that found their
 * position / entries inserted (TranspositionTable::insertOrGet, transposition_table.h:83-103). */
#define HM_SP_INFO_INTS 20
int hm_sp_max_edges(const hm_sp* sp);
int hm_sp_root_stats(hm_sp* sp, int* counts, hm_move* move_a, hm_move* move_b, int* visits, float* q, float* prior,
                     float* root_q, int* info, int max_edges);
/* Tree reuse between the searches of a game slot (Agent::try_reuse_tree / store_next_root_candidates, agent.cc:1345-1451): the
 * next hm_sp_begin_search takes the previous search's selected child, or one of the replies generated below it, as its root when
 * that node is the new position (hash, side, both boards, history) -- hm_sp_root_stats info[16] = visits recovered, -1 = fresh root.
 * mode[g] (NULL = unchanged): 0 off (default; self-play and tournaments reset the agent before every search), 1 reuse when the node
 * budget fits behind the nodes the pool already holds, 2 also shrink the budget to what fits (info[17] = the budget in force).
 * reset != 0 forgets the previous trees (Agent::reset_search_state, agent.cc:403-412). */
int hm_sp_set_tree_reuse(hm_sp* sp, const uint8_t* mode, int reset);
/* Principal variations of one game's search (Agent::extract_pv_from_child, agent.cc:1218-1290): line l starts with root edge
 * child_idx[l] and follows the final-move rule through expanded nodes, max_depth joint actions at most.  The caller orders the
 * root edges (visit count, solver-aware best move first: agent.cc:917-940).  moves[n_lines][max_depth][2]; lens, child_type and
 * child_end_in_ply [n_lines] (the last two feed format_uci_score, agent.cc:48-78; optional). */
int hm_sp_pv_lines(hm_sp* sp, int game, int n_lines, const int* child_idx, int max_depth, hm_move* moves, int* lens,
                   int* child_type, int* child_end_in_ply);
/* Board::push_move of the chosen joint action, then team / time-advantage flip (selfplay.cc:694-716).
 * HM_ERR_OVERFLOW when a game's history pool is full (nothing is applied to that game). */
int hm_sp_apply(hm_sp* sp, const hm_move* move_a, const hm_move* move_b, const uint8_t* mask);
/* Current boards (+ rep counts, last moves) and flags: bit0 is_checkmate(team, adv), bit1 is_draw()
 * (selfplay.cc:608-616).  d_boards_out (device, optional) receives the same hm_board array. */
int hm_sp_game_state(hm_sp* sp, hm_board* boards, int* flags, void* d_boards_out);
/* Raw-policy opening (selfplay.cc:277-300): per game and board the action list (+pass last) and
 * masked-softmax probabilities from policy heads [n_games, 4672] (one row per game). */
int hm_sp_raw_policy(hm_sp* sp, const void* d_pi_a, const void* d_pi_b, hm_move* moves, float* probs, uint8_t* caps,
                     int* counts, uint8_t* on_turn);
/* The same listing for the UCI `policy` command (uci.cc:306-393): all_moves != 0 keeps rook / bishop promotions in the list with
 * probability 0 (get_fast_policy_index has no plane for them, utils.h:183-216) instead of dropping them as the search does. */
int hm_sp_policy_listing(hm_sp* sp, const void* d_pi_a, const void* d_pi_b, hm_move* moves, float* probs, uint8_t* caps,
                         int* counts, uint8_t* on_turn, int all_moves);
/* action_leads_to_terminal (selfplay.cc:378-390). */
int hm_sp_action_terminal(hm_sp* sp, const hm_move* move_a, const hm_move* move_b, int* out);
/* Test hook: Board::is_checkmate x4, in-check x2, classify_terminal_position x2 and hash keys for
 * history-free boards (d_out: n*8 ints, d_keys: n*4 u64: hash_key(adv=0), hash_key(adv=1), repetition keys). */
int hm_rules_probe(const hm_board* d_boards, size_t n, int* d_out, uint64_t* d_keys);
/* Deterministic stand-in network for full-size parity runs (tests / tools): a hash of each row's planes fills the five fp16
 * heads — the same function as the CPU restatement's hash evaluator (oracle/search.hpp), `salt` selects one of many. */
int hm_hash_evaluator(const void* d_planes, int rows, uint64_t salt, void* d_value, void* d_pi_a, void* d_pi_b, void* d_wdl,
                      void* d_moves_left, void* stream);
/* Test hook: classify_terminal_position (searchthread.cc:101-139), Board::is_draw(ply) and repetition_count on every
 * game's CURRENT position with its real game history.  args4[g] = {teamToPlay, rootTeam, rootAdv, searchPly} (host);
 * out4[g] = {outcome | endInPly << 8, is_draw, repetition_count(A), repetition_count(B)} (host). */
int hm_sp_classify(hm_sp* sp, const int* args4, int* out4);

/* Diagnostics of the tree kernels (no reference counterpart; all no-ops / zeros in the product build).
 * hm_sp_profile: cycle accounting of game slot 0, out64[0..31] cycles and out64[32..63] call counts per probe
 * (library built with -DHM_SEARCH_PROF; tools/profile_search.py).  hm_sp_trace_select / hm_sp_trace: per-attempt
 * event log of collect_batch for one game slot, encoded as oracle Search::ctxTrace (library built with
 * -DHM_SEARCH_TRACE; tools/dbg_ctx.py).  hm_sp_trace returns the number of events logged. */
int hm_sp_profile(unsigned long long* out64, int reset);
int hm_sp_profile_launches(unsigned int* out, int launches);   /* [launches][64] wave-0 cycles per k_collect launch and game slot */
int hm_sp_trace_select(int game);
int hm_sp_trace(unsigned long long* out, int cap);

/* ================================================================== */
/* RISEv3 forward as one kernel launch (evaluator hot op; replaces the  */
/* TensorRT FP16 plan of nn/engine.cc:290-401,577-650).  A network is a */
/* handle: hm_net_create takes the packed descriptor (hivemind_amd/     */
/* net.py FusedNet: hmn::NetDesc as int32s) and the fp16 / fp32         */
/* parameter buffers in MFMA fragment order (device, caller-owned);     */
/* hm_net_create_host takes them from host memory and owns the device   */
/* copies (what Engine::loadNetwork does with a plan file).  Handles are*/
/* independent: no process-global state.  d_planes = fp16 [n,74,8,8];   */
/* heads as in hm_sp_process.                                           */
/* ================================================================== */
typedef struct hm_net hm_net;
int hm_net_create(const int32_t* desc, size_t desc_ints, const void* d_wh, const void* d_wf, hm_net** out);
int hm_net_create_host(const int32_t* desc, size_t desc_ints, const void* h_wh, size_t wh_bytes, const void* h_wf, size_t wf_bytes, hm_net** out);
int hm_net_destroy(hm_net* net);
int hm_net_forward(const hm_net* net, const void* d_planes, int n,
                   void* d_value, void* d_pi_a, void* d_pi_b, void* d_wdl, void* d_moves_left, void* stream);
/* Ragged batch: rows come in groups of `group` (one group per game slot); only the first d_group_rows[g]
 * rows of group g are evaluated (d_group_rows from hm_sp_collect_counted), the rest are skipped. */
int hm_net_forward_groups(const hm_net* net, const void* d_planes, int n, const int32_t* d_group_rows, int group,
                          void* d_value, void* d_pi_a, void* d_pi_b, void* d_wdl, void* d_moves_left, void* stream);
/* As hm_net_forward_groups; the launch is also stamped into d_interval[0..1] = (start, end) of a leg clock
 * (hm_sp_leg_clock_net), every workgroup min-/max-ing the 100 MHz device clock into it. */
int hm_net_forward_groups_timed(const hm_net* net, const void* d_planes, int n, const int32_t* d_group_rows, int group,
                                void* d_value, void* d_pi_a, void* d_pi_b, void* d_wdl, void* d_moves_left, void* stream,
                                uint64_t* d_interval);
/* Diagnostic variant: same launch, d_stamps[256] (device u64) receives the shader clock at the phase
 * boundaries of workgroup 0 (tools/profile_net.py). */
int hm_net_profile(const hm_net* net, const void* d_planes, int n,
                   void* d_value, void* d_pi_a, void* d_pi_b, void* d_wdl, void* d_moves_left, void* stream, uint64_t* d_stamps);

/* ================================================================== */
/* Evaluator seam as an object: class Engine (nn/engine.h:43-81).       */
/*   Engine(int deviceId, int batchSize = 8)      hm_engine_create      */
/*   bool loadNetwork(onnxFile, engineFile)       hm_engine_load_network (packed network in memory) /           */
/*                                                hm_engine_load_network_file (the "plan" file FusedNet.save writes) */
/*   bool enqueueInferenceHalf(obs, worker)       hm_engine_enqueue_half */
/*   bool synchronizeInferenceHalf(out&, worker)  hm_engine_sync_half   */
/*   bool runInferenceHalf(obs, out&, worker)     hm_engine_run_half    */
/*   bool runInference(float* ..., worker)        hm_engine_run_f32     */
/*   int  getBatchSize()                          hm_engine_batch_size  */
/* HM_ENGINE_WORKERS (= SearchParams::NUM_SEARCH_THREADS, 4) execution  */
/* states, each with its own stream, device buffers and pinned host     */
/* output buffers; each worker may have ONE request in flight: a second */
/* enqueue before the sync, or a sync with nothing pending, returns     */
/* HM_ERR_STATE (the reference returns false, engine.cc:585-590,        */
/* 659-664).  obs: caller-owned batch x 4736 fp16, host (pinned or      */
/* pageable) or device memory; it must stay unchanged until the sync.   */
/* Output pointers are engine-owned pinned host memory, valid until the */
/* next enqueue on that worker.  Output semantics as hm_sp_process.     */
/* ================================================================== */
#define HM_ENGINE_WORKERS 4
typedef struct hm_engine hm_engine;
typedef struct hm_half_outputs {          /* Engine::HalfInferenceOutputs, nn/engine.h:45-51 (fp16 bit patterns) */
    const uint16_t* value; const uint16_t* policy_a; const uint16_t* policy_b; const uint16_t* wdl; const uint16_t* moves_left;
} hm_half_outputs;
int hm_engine_create(int device, int batch_size, hm_engine** out);
int hm_engine_destroy(hm_engine* e);
int hm_engine_load_network(hm_engine* e, const int32_t* desc, size_t desc_ints, const void* h_wh, size_t wh_bytes, const void* h_wf, size_t wf_bytes);
int hm_engine_load_network_file(hm_engine* e, const char* path);
int hm_engine_enqueue_half(hm_engine* e, const void* obs, size_t worker);
int hm_engine_sync_half(hm_engine* e, hm_half_outputs* out, size_t worker);
int hm_engine_run_half(hm_engine* e, const void* obs, hm_half_outputs* out, size_t worker);
int hm_engine_run_f32(hm_engine* e, const float* obs, float* value, float* pi_a, float* pi_b, float* wdl, float* moves_left, size_t worker);
int hm_engine_batch_size(const hm_engine* e);
/* The loaded network, for callers that keep their buffers on the device (hm_eval_io.net). */
const hm_net* hm_engine_net(const hm_engine* e);
/* Packed-network file ("plan"): 'HMNP' u32 version=1, u64 desc_ints, u64 wh_bytes, u64 wf_bytes, then the three blobs. */
int hm_net_save_file(const char* path, const int32_t* desc, size_t desc_ints, const void* h_wh, size_t wh_bytes, const void* h_wf, size_t wf_bytes);

/* ================================================================== */
/* self-play driver: run_selfplay (tools/selfplay.h:10-33,              */
/* tools/selfplay.cc:558-748) for `concurrent_games` slots on one GPU.  */
/* ================================================================== */
typedef struct hm_selfplay hm_selfplay;

typedef struct hm_selfplay_config {       /* SelfPlayConfig, tools/selfplay.h:10-31 */
    uint64_t games, nodes, max_macro_plies, chunk_samples;
    double   raw_policy_mean_macro_plies;
    uint64_t raw_policy_max_macro_plies;
    double   raw_policy_high_temperature_probability;
    double   mcts_temperature, mcts_temperature_decay;
    uint64_t mcts_temperature_plies;
    float    resign_threshold;
    uint64_t resign_consecutive_plies;
    double   resign_disable_fraction;
    double   node_random_factor;
    float    dirichlet_alpha, dirichlet_epsilon;
    uint64_t seed;                        /* 0 = wall clock, as the reference */
    int      rank, world;                 /* games with index % world == rank are played here */
    int      concurrent_games;            /* game slots searched in lockstep on this GPU */
} hm_selfplay_config;
void hm_selfplay_config_default(hm_selfplay_config* cfg);

/* Evaluator seam (class Engine, nn/engine.h:43-81): device buffers owned by the caller.
 * planes[k]: fp16 [concurrent_games*8, 74, 8, 8]; heads as in hm_sp_process. */
typedef struct hm_eval_io {
    void* planes[2];
    void* value; void* pi_a; void* pi_b; void* wdl; void* moves_left;
    /* Optional native evaluator: when net != NULL the driver calls hm_net_forward itself
     * (no callback) on a second stream, overlapping it with hm_sp_collect; *_2 is the second set
     * of head buffers the overlap needs. */
    const hm_net* net;
    void* value_2; void* pi_a_2; void* pi_b_2; void* wdl_2; void* moves_left_2;
} hm_eval_io;
/* Runs the network on the first `rows` rows of planes[which] and fills the head buffers
 * (Engine::enqueueInferenceHalf + synchronizeInferenceHalf).  Return 0 on success. */
typedef int (*hm_eval_fn)(void* user, int which, int rows);

typedef struct hm_selfplay_result {
    uint64_t games, samples, searched_positions, total_nodes, eval_rows, eval_batches, search_iterations, raw_plies;
    uint64_t record_bytes;
    uint64_t terminations[5];            /* macro-ply limit, checkmate, draw, resignation, no legal action */
    double   seconds;
    /* device time of the three legs of a lockstep iteration, HIP events on the launch stream */
    double   collect_ms, eval_ms, process_ms;
    uint64_t nodes_visited, edges_scanned;   /* tree nodes / edges read by selection (roofline accounting) */
    /* wall-clock split of `seconds`: lockstep search loop, search prologue (hm_sp_begin_search), raw-policy plies;
     * the remainder is host bookkeeping (terminal checks, record building, root statistics) */
    double   search_seconds, prologue_seconds, raw_seconds;
    uint64_t chunks_flushed;             /* chunks handed to the sink / written to the output directory */
    uint64_t leaf_move_words;            /* 4-byte move-list entries the traversal wrote for its network leaves (roofline accounting) */
    /* single-launch searches (hm_sp_search): launches of k_rollout, their total duration (HIP events on the launch stream) and the
     * time the games waited for evaluations; collect_ms / process_ms are then sums over game-iterations (search_iterations counts
     * game-iterations), eval_ms the evaluator workgroups' time over eval_rows positions */
    uint64_t persistent_searches;
    double   search_kernel_ms, wait_ms;
    uint64_t persistent_stalls;          /* persistent searches given up as stalled and repeated on the lockstep loop (hm_sp_search_stalled) */
    uint64_t tt_hits, tt_inserts;        /* transposition-table lookups that found their position / inserted it, over all searches */
} hm_selfplay_result;

int hm_selfplay_create(const hm_selfplay_config* cfg, const hm_search_config* search_cfg, const hm_eval_io* io,
                       hm_eval_fn fn, void* user, hm_selfplay** out);
/* ChunkWriter (selfplay.cc:69-158).  With a sink, finished samples leave the driver every `chunk_samples` samples
 * (ChunkWriter::append :78-85) and once more at the end of hm_selfplay_run — also when the run fails, so the games
 * finished before an error are not lost (ChunkWriter::finish :87-91) — and host memory stays bounded by one chunk.
 *   hm_selfplay_set_output_directory: built-in sink writing <dir>/training_data/chunk_<runId>_<index 6 digits>.hvm
 *                                     (published atomically via .tmp + rename), the reference's file naming;
 *   hm_selfplay_set_chunk_sink:       caller's sink (e.g. the per-chunk gather to rank 0); `records` is valid during the
 *                                     call only; a non-zero return fails the run with HM_ERR_STATE after the remaining
 *                                     chunks were offered.
 * Without either, every sample stays in memory until hm_selfplay_records (tests, single-shot benchmarks). */
typedef int (*hm_chunk_fn)(void* user, const uint8_t* records, uint64_t nbytes, uint64_t count, uint64_t chunk_index);
int hm_selfplay_set_chunk_sink(hm_selfplay* sp, hm_chunk_fn fn, void* user);
int hm_selfplay_set_output_directory(hm_selfplay* sp, const char* dir);
int hm_selfplay_run(hm_selfplay* sp, hm_selfplay_result* out);
/* Serialized TrainingSample records of the finished games that no sink has taken (HVM4 sample layout,
 * selfplay.cc:126-142); returns the byte count; pointers stay valid until the next run / destroy. */
uint64_t hm_selfplay_records(hm_selfplay* sp, const uint8_t** data, uint64_t* count);
int hm_selfplay_destroy(hm_selfplay* sp);
/* ChunkWriter::flush (selfplay.cc:105-151): header 'HVM4' u32 4, u16 74, u16 4672, u64 count + samples,
 * published atomically via .tmp + rename. */
int hm_hvm4_write_chunk(const char* path, const uint8_t* records, uint64_t nbytes, uint64_t count);

/* ================================================================== */
/* paired network tournament: run_tournament (tools/tournament.h:15-75, */
/* tools/tournament.cc:328-465) for `concurrent_games` slots on one GPU. */
/* Game i: pair i/2, the contender plays White in even games, the pair's  */
/* starting team alternates, every macro-ply is searched with `nodes`     */
/* nodes by the network of the team to move (its own PW coefficient for    */
/* root and interior nodes), root noise seeded by (seed, pair, macro-ply), */
/* and the most visited joint action is played.  Games are independent     */
/* (no shared RNG), so they run in lockstep slots; results, summary.json   */
/* and games.pgn are produced in game order and are byte-identical to the  */
/* sequential loop under the same evaluators.                              */
/* move_time_ms > 0 (instead of nodes): every search runs until its slot's */
/* controller of the reference's polling loop says stop (deadline, early   */
/* stopping, time extension: agent.cc:715-806) -- wall-clock dependent, so */
/* not reproducible run to run; native evaluator only.  Not built: batch   */
/* sizes other than 8 -> HM_ERR_INVALID at create.                         */
/* ================================================================== */
typedef struct hm_tournament hm_tournament;
typedef struct hm_tournament_config {     /* TournamentConfig, tools/tournament.h:15-42 */
    uint64_t games, nodes;
    int32_t  move_time_ms, contender_batch_size, baseline_batch_size;
    uint64_t max_macro_plies;
    float    dirichlet_alpha, dirichlet_epsilon;
    float    contender_pw_coefficient, baseline_pw_coefficient;
    uint64_t seed;
    int32_t  concurrent_games;            /* game slots searched in lockstep on this GPU */
    int32_t  max_search_nodes;            /* move_time_ms > 0: nodes the pool of one time-limited search holds (0 = 4096); the search ends there at the latest */
} hm_tournament_config;
void hm_tournament_config_default(hm_tournament_config* cfg);
typedef struct hm_tournament_breakdown { uint64_t wins, losses, draws; } hm_tournament_breakdown;
typedef struct hm_tournament_result {     /* TournamentResult, tools/tournament.h:50-73 + its statistics (tournament.cc:247-326) */
    uint64_t contender_wins, baseline_wins, draws;
    hm_tournament_breakdown as_white, as_black, up_time, down_time;
    uint64_t checkmates, no_legal_actions, drawn_terminations, macro_ply_limits;
    uint64_t pairs;                       /* pair scores recorded (hm_tournament_pair_scores) */
    double   contender_score;
    int32_t  has_elo, has_score_ci, has_elo_ci, paired_method;   /* paired_method: 1 = "paired-opening normal approximation" */
    double   contender_elo, score_ci[2], elo_ci[2];
    uint64_t searched_positions, total_nodes, search_iterations;
    double   seconds;
} hm_tournament_result;
/* io->net = the contender's network; baseline_net = the baseline's (both NULL: callback evaluator, which must serve both
 * networks: hm_tournament_acting tells it which slots the contender is searching). */
int hm_tournament_create(const hm_tournament_config* cfg, const hm_search_config* search_cfg, const hm_eval_io* io, const hm_net* baseline_net,
                         hm_eval_fn fn, void* user, hm_tournament** out);
int hm_tournament_run(hm_tournament* t, hm_tournament_result* out);
/* acting[concurrent_games]: 1 where the contender's network evaluates the slot's current search, 0 baseline (valid inside the callback). */
int hm_tournament_acting(const hm_tournament* t, uint8_t* acting);
uint64_t hm_tournament_pair_scores(const hm_tournament* t, const double** scores);
/* summary.json (write_summary :117-187) / games.pgn (append_game_pgn :89-115) of the finished run as text; returns the
 * length, or -(needed size) when cap is too small. */
int64_t hm_tournament_summary(const hm_tournament* t, const char* contender_name, const char* baseline_name, char* out, int64_t cap);
int64_t hm_tournament_pgn(const hm_tournament* t, const char* contender_name, const char* baseline_name, char* out, int64_t cap);
/* <dir>/summary.json (atomically via .tmp + rename) and <dir>/games.pgn, the reference's report files. */
int hm_tournament_write_reports(const hm_tournament* t, const char* dir, const char* contender_name, const char* baseline_name);
int hm_tournament_destroy(hm_tournament* t);
/* The statistics alone (host-only): score, Elo and the 95 % intervals of a W-L-D record with optional pair scores. */
int hm_tournament_statistics(uint64_t contender_wins, uint64_t baseline_wins, uint64_t draws, const double* pair_scores, uint64_t pairs,
                             hm_tournament_result* out);
/* UCI::move text of a move of this variant (Board::uci_move, environment/board.h:340-350; MOVE_NONE -> "pass"). */
int hm_move_uci(hm_move move, char* out, int cap);

/* ================================================================== */
/* UCI front end (interface/uci.cc:82-317, 396-429) over one game slot  */
/* of the GPU search engine: `position` replays the game on the device, */
/* `go nodes N` / `go movetime T` searches and prints `info ...` and     */
/* `bestmove (<moveA>,<moveB>)` with the solver-aware move rule          */
/* (agent.cc:1031-1049).  Built: uci, isready, ucinewgame, position      */
/* startpos|fen [moves <1|2><uci>...], go, stop, setoption (Team, Mode,  */
/* MultiPV, Ponder, DrawContemptPermille, PWCoefficientPermille,          */
/* RootPWCoefficientPermille, PWExponentPermille, Transpositions; Hash      */
/* accepted), policy (uci.cc:306-393), quit; `go movetime` with the         */
/* reference's early exit / early stopping / time extension rules; final    */
/* `info` lines with MultiPV principal variations (agent.cc:917-965,        */
/* 1218-1290) and `bestmove ... ponder ...` (agent.cc:1054-1113).           */
/* Every `go` searches on a worker thread (mainSearchThread, uci.cc:192-205): */
/* hm_uci_command returns at once, `stop` / `isready` / `quit` reach a running */
/* search, its text is fetched with later commands while hm_uci_busy() is 1.   */
/* `go ponder ...` ignores its budget until `ponderhit` (then it applies) or   */
/* `stop`; the tree is reused from one `go` to the next (agent.cc:1345-1451)   */
/* until `ucinewgame`.                                                       */
/* ================================================================== */
typedef struct hm_uci hm_uci;
#define HM_UCI_QUIT (-1000000)
/* net: loaded network (hm_net_create / hm_engine_net), or NULL with a callback evaluator `fn` (rows 0..7 of io->planes[which]);
 * io: caller-owned device buffers for 8 rows (planes[2], five heads); max_nodes: largest node budget of a `go`. */
int hm_uci_create(const hm_net* net, const hm_eval_io* io, hm_eval_fn fn, void* user, int max_nodes, hm_uci** out);
/* One command line in, the engine's output text out (several lines possible).  Returns the text length, -(needed size)
 * when cap is too small (the command HAS run; call again with "" to fetch the text) and HM_UCI_QUIT for `quit`.
 * `go` only starts the search: poll with "" until hm_uci_busy() is 0 to collect its `info` / `bestmove` lines. */
int64_t hm_uci_command(hm_uci* uci, const char* line, char* out, int64_t cap);
/* The current game position (after the last `position`), team / time_adv = the Team / Mode options. */
int hm_uci_board(hm_uci* uci, hm_board* out);
/* 1 while a search is running on the worker thread (its text arrives with later commands; an empty line polls). */
int hm_uci_busy(hm_uci* uci);
/* Board::fen(board) (environment/board.h:172-174 -> Position::fen(false, true)) of a compact board; host-only.  Returns the text
 * length, or -(needed size) when cap is too small. */
int hm_board_fen(const hm_board* board_pair, int board, char* out, int cap);
int hm_uci_destroy(hm_uci* uci);
/* SearchParams::has_insurmountable_visit_lead (search/search_params.h:322-326). */
int hm_insurmountable_visit_lead(float best_visits, float projected_second_visits, float factor);
/* The movetime controller of Agent::run_search's polling loop (agent.cc:561-713; SearchInfo, searchinfo.h:129-180) alone, host-only:
 * one poll = the root's edge visits / Q values, root solver type (0 unsolved, 1 win, 2 loss, 3 draw), the children's solver types and
 * endInPly; returns 1 = stop now, 0 = go on; *effective_ms = move time after extensions; log = the `info string` lines of the decision. */
typedef struct hm_time_manager hm_time_manager;
hm_time_manager* hm_time_manager_create(int move_time_ms);
int hm_time_manager_poll(hm_time_manager* tm, double elapsed_ms, int nodes, int n, const int* visits, const float* q, int root_type,
                         const int* child_type, const int* child_end_in_ply, double* effective_ms, char* log, int cap);
void hm_time_manager_destroy(hm_time_manager* tm);

#ifdef __cplusplus
}
#endif
#endif /* HIVEMIND_AMD_H */
