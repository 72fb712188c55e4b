"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when present, the reference
build (oracle/_ref/libhmref.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libhmref.so")

POS_DTYPE = np.dtype([
    ("by_type", "<u8", (6,)), ("by_color", "<u8", (2,)), ("promoted", "<u8"), ("key", "<u8"),
    ("hand", "u1", (2, 5)), ("castling", "u1"), ("ep", "u1"), ("stm", "u1"), ("rule50", "u1"),
    ("game_ply", "<u2"),
])
BOARD_DTYPE = np.dtype([
    ("pos", POS_DTYPE, (2,)), ("last_move", "<u4", (2,)), ("rep_count", "u1", (2,)),
    ("team", "u1"), ("time_adv", "u1"), ("reserved", "<u4"),
])
DT = {"f16": 0, "f32": 1, "u8": 2}
NPDT = {"f16": np.uint16, "f32": np.float32, "u8": np.uint8}

if not os.path.exists(ORACLE_PATH):
    import subprocess
    subprocess.run(["make", "-s", "liboracle.so"], cwd=os.path.join(ROOT, "oracle"), check=True)
lib = C.CDLL(ORACLE_PATH)
ref = C.CDLL(REF_PATH) if os.path.exists(REF_PATH) else None

_vp, _u32, _u64, _i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int


def _sig(l, prefix):
    s = {
        "board_new": (_vp, []), "board_clone": (_vp, [_vp]), "board_free": (None, [_vp]),
        "board_set": (None, [_vp, C.c_char_p]), "board_set_fen": (None, [_vp, _i, C.c_char_p]),
        "legal_moves": (_i, [_vp, _i, _vp]), "push": (None, [_vp, _i, _u32]), "pop": (None, [_vp, _i]),
        "make_moves": (_i, [_vp, _u32, _u32]), "unmake_moves": (None, [_vp, _u32, _u32]),
        "is_checkmate": (_i, [_vp, _i, _i]), "is_draw": (_i, [_vp, _i]), "hash_key": (_u64, [_vp, _i]),
        "pos_key": (_u64, [_vp, _i]), "repetition_count": (_i, [_vp, _i]), "in_check": (_i, [_vp, _i]),
        "gives_check": (_i, [_vp, _i, _u32]), "is_capture": (_i, [_vp, _i, _u32]),
        "compact": (None, [_vp, _i, _i, _vp]), "uci": (_i, [_vp, _i, _u32, C.c_char_p, _i]),
        "perft": (_u64 if prefix == "ora_" else C.c_longlong, [_vp, _i]),
        "perft_single": (_u64 if prefix == "ora_" else C.c_longlong, [_vp, _i, _i]),
        "policy_tables": (None, [_vp, _vp]), "policy_label": (_i, [_i, C.c_char_p, _i]),
    }
    for name, (res, args) in s.items():
        fn = getattr(l, prefix + name)
        fn.restype, fn.argtypes = res, args


_sig(lib, "ora_")
lib.ora_board_from_compact.restype, lib.ora_board_from_compact.argtypes = None, [_vp, _vp]
lib.ora_rep_key.restype, lib.ora_rep_key.argtypes = _u64, [_vp, _i]
lib.ora_perft_fast.restype, lib.ora_perft_fast.argtypes = _u64, [_vp, _i]
lib.ora_legal_moves_pos.restype, lib.ora_legal_moves_pos.argtypes = _i, [_vp, _vp]
lib.ora_planes.restype, lib.ora_planes.argtypes = None, [_vp, C.c_size_t, _i, _vp]
lib.ora_make_moves_compact.restype, lib.ora_make_moves_compact.argtypes = None, [_vp, _u32, _u32, _vp]
lib.ora_policy_index.restype, lib.ora_policy_index.argtypes = _i, [_u32, _i]
lib.ora_random_positions.restype, lib.ora_random_positions.argtypes = C.c_size_t, [_u64, C.c_size_t, _i, _vp]
lib.ora_time_planes.restype, lib.ora_time_planes.argtypes = C.c_double, [_vp, C.c_size_t, _i, _vp, _i]
if ref is not None:
    ref.ref_init.restype = None
    ref.ref_init()
    _sig(ref, "ref_")
    ref.ref_board_only_key.restype, ref.ref_board_only_key.argtypes = _u64, [_vp, _i]
    ref.ref_fen.restype, ref.ref_fen.argtypes = _i, [_vp, _i, C.c_char_p, _i]
    ref.ref_pw_allowed_children.restype, ref.ref_pw_allowed_children.argtypes = _i, [_i, _i]
    ref.ref_get_cpuct.restype, ref.ref_get_cpuct.argtypes = C.c_float, [C.c_float]


class Board:
    """Handle over either implementation: Board(impl='ora'|'ref')."""

    def __init__(self, impl="ora", fen=None):
        self.l = lib if impl == "ora" else ref
        self.p = impl + "_"
        if self.l is None:
            raise RuntimeError("reference build not available")
        self.h = self._f("board_new")()
        if fen:
            self.set(fen)

    def _f(self, name):
        return getattr(self.l, self.p + name)

    def __del__(self):
        try:
            self._f("board_free")(self.h)
        except Exception:
            pass

    def from_compact(self, board):
        board = np.ascontiguousarray(board)
        lib.ora_board_from_compact(self.h, board.ctypes.data)

    def set(self, fen): self._f("board_set")(self.h, fen.encode())
    def set_fen(self, b, fen): self._f("board_set_fen")(self.h, b, fen.encode())

    def legal_moves(self, b):
        buf = np.zeros(1024, dtype=np.uint32)
        n = self._f("legal_moves")(self.h, b, buf.ctypes.data)
        return buf[:n].copy()

    def push(self, b, m): self._f("push")(self.h, b, int(m))
    def pop(self, b): self._f("pop")(self.h, b)
    def make_moves(self, a, b): return self._f("make_moves")(self.h, int(a), int(b))
    def unmake_moves(self, a, b): self._f("unmake_moves")(self.h, int(a), int(b))
    def is_checkmate(self, side, adv=False): return bool(self._f("is_checkmate")(self.h, side, int(adv)))
    def is_draw(self, ply=0): return bool(self._f("is_draw")(self.h, ply))
    def hash_key(self, adv=False): return int(self._f("hash_key")(self.h, int(adv)))
    def in_check(self, b): return bool(self._f("in_check")(self.h, b))
    def repetition_count(self, b): return int(self._f("repetition_count")(self.h, b))
    def perft(self, d): return int(self._f("perft")(self.h, d))
    def perft_single(self, b, d): return int(self._f("perft_single")(self.h, b, d))

    def compact(self, team=0, adv=False):
        out = np.zeros(1, dtype=BOARD_DTYPE)
        self._f("compact")(self.h, team, int(adv), out.ctypes.data)
        return out

    def fen(self, b):
        buf = C.create_string_buffer(160)
        (lib.ora_fen if self.p == "ora_" else ref.ref_fen)(self.h, b, buf, 160)
        return buf.value.decode()

    def uci(self, b, m):
        buf = C.create_string_buffer(16)
        self._f("uci")(self.h, b, int(m), buf, 16)
        return buf.value.decode()

    def find_move(self, b, uci):
        for m in self.legal_moves(b):
            if self.uci(b, m) == uci:
                return int(m)
        return 0


def random_positions(seed, n, max_plies=120):
    out = np.zeros(n, dtype=BOARD_DTYPE)
    got = lib.ora_random_positions(seed, n, max_plies, out.ctypes.data)
    assert got == n
    return out


def planes(boards, dtype="f16"):
    boards = np.ascontiguousarray(boards)
    out = np.zeros((len(boards), 4736), dtype=NPDT[dtype])
    lib.ora_planes(boards.ctypes.data, len(boards), DT[dtype], out.ctypes.data)
    return out


def legal_moves_pos(pos):
    pos = np.ascontiguousarray(pos)
    buf = np.zeros(1024, dtype=np.uint32)
    n = lib.ora_legal_moves_pos(pos.ctypes.data, buf.ctypes.data)
    return buf[:n].copy()


def make_moves_compact(board, a, b):
    board = np.ascontiguousarray(board)
    out = np.zeros(1, dtype=BOARD_DTYPE)
    lib.ora_make_moves_compact(board.ctypes.data, int(a), int(b), out.ctypes.data)
    return out


def policy_tables(impl="ora"):
    normal = np.zeros((2, 64, 64, 2), dtype=np.int32)
    drop = np.zeros((2, 64, 8), dtype=np.int32)
    (lib.ora_policy_tables if impl == "ora" else ref.ref_policy_tables)(normal.ctypes.data, drop.ctypes.data)
    return normal, drop


# ---- search oracle (oracle/search.hpp) -------------------------------------------------------
lib.ora_search_new.restype, lib.ora_search_new.argtypes = _vp, [_i, _i]
lib.ora_search_free.restype, lib.ora_search_free.argtypes = None, [_vp]
lib.ora_search_set_noise.restype, lib.ora_search_set_noise.argtypes = None, [_vp, C.c_float, C.c_float, _u64]
lib.ora_search_tt_hits.restype, lib.ora_search_tt_hits.argtypes = _i, [_vp]
lib.ora_search_set_transpositions.restype, lib.ora_search_set_transpositions.argtypes = None, [_vp, _i]
lib.ora_search_set_batch_size.restype, lib.ora_search_set_batch_size.argtypes = None, [_vp, _i]
lib.ora_search_run.restype, lib.ora_search_run.argtypes = _i, [_vp, _vp, _i, _i, _i]
lib.ora_search_edges.restype, lib.ora_search_edges.argtypes = _i, [_vp, _vp, _vp, _vp, _vp, _vp, _i]
lib.ora_search_root_q.restype, lib.ora_search_root_q.argtypes = C.c_float, [_vp]
lib.ora_search_info.restype, lib.ora_search_info.argtypes = None, [_vp, _vp]
lib.ora_search_best_move.restype, lib.ora_search_best_move.argtypes = _i, [_vp]
lib.ora_search_pv_lines.restype, lib.ora_search_pv_lines.argtypes = _i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]
lib.ora_fen.restype, lib.ora_fen.argtypes = _i, [_vp, _i, C.c_char_p, _i]
lib.ora_search_set_tree_reuse.restype, lib.ora_search_set_tree_reuse.argtypes = None, [_vp, _i]
lib.ora_search_reset.restype, lib.ora_search_reset.argtypes = None, [_vp]
lib.ora_search_reused_visits.restype, lib.ora_search_reused_visits.argtypes = _i, [_vp]
lib.ora_search_retained.restype, lib.ora_search_retained.argtypes = _i, [_vp, _vp, _i]
lib.ora_search_trace.restype, lib.ora_search_trace.argtypes = _i, [_vp, _vp, _i]
lib.ora_classify.restype, lib.ora_classify.argtypes = _i, [_vp, _i, _i, _i, _i]
lib.ora_search_ctx_trace.restype, lib.ora_search_ctx_trace.argtypes = _i, [_vp, _vp, _i]
lib.ora_hash_evaluator.restype, lib.ora_hash_evaluator.argtypes = None, [_vp, _i, _vp, _vp, _vp, _vp, _vp]
lib.ora_pw_allowed_children.restype, lib.ora_pw_allowed_children.argtypes = _i, [_i, _i]
lib.ora_get_cpuct.restype, lib.ora_get_cpuct.argtypes = C.c_float, [C.c_float]
lib.ora_portable_expf.restype, lib.ora_portable_expf.argtypes = C.c_float, [C.c_float]
EVAL_CB = C.CFUNCTYPE(None, _vp, _i, _vp, _vp, _vp, _vp, _vp)
lib.ora_search_set_callback.restype, lib.ora_search_set_callback.argtypes = None, [_vp, EVAL_CB]


class Search:
    """Single-thread reference-schedule search (tie_mode/exp_mode: see oracle/search.hpp)."""

    def __init__(self, tie_mode=1, exp_mode=1):
        self.h = lib.ora_search_new(tie_mode, exp_mode)
        self._cb = None

    def __del__(self):
        try:
            lib.ora_search_free(self.h)
        except Exception:
            pass

    def set_noise(self, alpha, eps, seed): lib.ora_search_set_noise(self.h, alpha, eps, seed)
    def set_batch_size(self, b): lib.ora_search_set_batch_size(self.h, int(b))                # Engine::getBatchSize(): leaves per iteration (default 8)
    def set_tree_reuse(self, on=True): lib.ora_search_set_tree_reuse(self.h, int(on))      # ENABLE_TREE_REUSE between run() calls (the UCI front end)
    def reset(self): lib.ora_search_reset(self.h)                                           # Agent::reset_search_state
    def reused_visits(self): return int(lib.ora_search_reused_visits(self.h))               # -1 = fresh root

    def retained(self):
        """next-root candidates kept after the last run: rows (move_a, move_b, visits, type, end_in_ply, is_principal_reply); row 0 = the selected child"""
        out = np.zeros((512, 6), np.int32)
        n = lib.ora_search_retained(self.h, out.ctypes.data, 512)
        return out[:n]

    def set_evaluator(self, fn):
        """fn(planes uint16 [n,4736]) -> (value[n], piA[n,4672], piB[n,4672], wdl[n,3], ml[n]) uint16 arrays."""
        def cb(planes, n, value, pia, pib, wdl, ml):
            pl = np.ctypeslib.as_array(C.cast(planes, C.POINTER(C.c_uint16)), shape=(n, 4736))
            v, a, b, w, m = fn(pl)
            for dst, src, cnt in ((value, v, n), (pia, a, n * 4672), (pib, b, n * 4672), (wdl, w, n * 3), (ml, m, n)):
                C.memmove(dst, np.ascontiguousarray(src, dtype=np.uint16).ctypes.data, 2 * cnt)
        self._cb = EVAL_CB(cb)
        lib.ora_search_set_callback(self.h, self._cb)

    def run(self, board, team, adv, nodes):
        return bool(lib.ora_search_run(self.h, board.h, team, int(adv), nodes))

    def ctx_trace(self):
        """collect_batch event log: rows of (collect#, code, path length, outcome); codes: 1 no leaf, 2 same-batch
        collision, 3 solved leaf, 4 classified terminal, 5 leaf not reserved, 6 network leaf."""
        cap = 1 << 16
        buf = np.zeros(cap, np.uint64)
        n = min(lib.ora_search_ctx_trace(self.h, buf.ctypes.data, cap), cap)
        v = buf[:n]
        return np.stack([(v >> np.uint64(32)), (v >> np.uint64(24)) & np.uint64(0xff), (v >> np.uint64(8)) & np.uint64(0xffff), v & np.uint64(0xff)], axis=1).astype(np.int64)

    def edges(self):
        cap = 1024
        ma = np.zeros(cap, np.uint32); mb = np.zeros(cap, np.uint32); v = np.zeros(cap, np.int32)
        q = np.zeros(cap, np.float32); p = np.zeros(cap, np.float32)
        n = lib.ora_search_edges(self.h, ma.ctypes.data, mb.ctypes.data, v.ctypes.data, q.ctypes.data, p.ctypes.data, cap)
        return dict(move_a=ma[:n], move_b=mb[:n], visits=v[:n], q=q[:n], prior=p[:n])

    def info(self):
        o = np.zeros(8, np.int32)
        lib.ora_search_info(self.h, o.ctypes.data)
        return dict(nodes=int(o[0]), eval_rows=int(o[1]), eval_calls=int(o[2]), same_batch=int(o[3]),
                    reservation=int(o[4]), node_count=int(o[5]), root_type=int(o[6]), root_visits=int(o[7]))

    def tt_hits(self): return int(lib.ora_search_tt_hits(self.h))
    def root_q(self): return float(lib.ora_search_root_q(self.h))
    def best_move(self): return int(lib.ora_search_best_move(self.h))

    def pv_lines(self, multi_pv=1, max_depth=20):
        """-> list of dicts: child index / type / end-in-ply / q of each line and its joint actions [(moveA, moveB), ...]"""
        idx, typ, end, ln = (np.zeros(multi_pv, np.int32) for _ in range(4))
        q = np.zeros(multi_pv, np.float32)
        mv = np.zeros((multi_pv, max_depth, 2), np.uint32)
        n = lib.ora_search_pv_lines(self.h, multi_pv, max_depth, idx.ctypes.data, typ.ctypes.data, end.ctypes.data, ln.ctypes.data, q.ctypes.data, mv.ctypes.data)
        return [dict(child=int(idx[l]), type=int(typ[l]), end_in_ply=int(end[l]), q=float(q[l]), moves=[(int(a), int(b)) for a, b in mv[l, :ln[l]]]) for l in range(n)]


def hash_evaluator(planes):
    planes = np.ascontiguousarray(planes, dtype=np.uint16).reshape(-1, 4736)
    n = len(planes)
    v = np.zeros(n, np.uint16); a = np.zeros((n, 4672), np.uint16); b = np.zeros((n, 4672), np.uint16)
    w = np.zeros((n, 3), np.uint16); m = np.zeros(n, np.uint16)
    lib.ora_hash_evaluator(planes.ctypes.data, n, v.ctypes.data, a.ctypes.data, b.ctypes.data, w.ctypes.data, m.ctypes.data)
    return v, a, b, w, m


# ---- self-play loop oracle (oracle/selfplay.hpp) ------------------------------------------------
class SelfPlayCfg(C.Structure):
    """Layout of hm_selfplay_config (include/hivemind_amd.h) = SelfPlayConfig (tools/selfplay.h:10-31) + sharding."""
    _fields_ = [("games", C.c_uint64), ("nodes", C.c_uint64), ("max_macro_plies", C.c_uint64), ("chunk_samples", C.c_uint64),
                ("raw_policy_mean_macro_plies", C.c_double), ("raw_policy_max_macro_plies", C.c_uint64),
                ("raw_policy_high_temperature_probability", C.c_double),
                ("mcts_temperature", C.c_double), ("mcts_temperature_decay", C.c_double), ("mcts_temperature_plies", C.c_uint64),
                ("resign_threshold", C.c_float), ("resign_consecutive_plies", C.c_uint64), ("resign_disable_fraction", C.c_double),
                ("node_random_factor", C.c_double), ("dirichlet_alpha", C.c_float), ("dirichlet_epsilon", C.c_float),
                ("seed", C.c_uint64), ("rank", C.c_int), ("world", C.c_int), ("concurrent_games", C.c_int)]


def selfplay_cfg(**kw):
    """Reference defaults (tools/selfplay.h:10-31) with overrides."""
    c = SelfPlayCfg(games=1, nodes=800, max_macro_plies=400, chunk_samples=16384, raw_policy_mean_macro_plies=8.0,
                    raw_policy_max_macro_plies=30, raw_policy_high_temperature_probability=0.05, mcts_temperature=1.0,
                    mcts_temperature_decay=0.93, mcts_temperature_plies=20, resign_threshold=-0.90, resign_consecutive_plies=3,
                    resign_disable_fraction=0.10, node_random_factor=0.05, dirichlet_alpha=0.3, dirichlet_epsilon=0.25,
                    seed=0, rank=0, world=1, concurrent_games=64)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


lib.ora_selfplay_new.restype, lib.ora_selfplay_new.argtypes = _vp, [_vp, _i, _i]
lib.ora_selfplay_free.restype, lib.ora_selfplay_free.argtypes = None, [_vp]
lib.ora_selfplay_set_callback.restype, lib.ora_selfplay_set_callback.argtypes = None, [_vp, EVAL_CB]
lib.ora_selfplay_game.restype, lib.ora_selfplay_game.argtypes = C.c_int64, [_vp, _u64, _vp, _u64, _vp]
lib.ora_selfplay_last_actions.restype, lib.ora_selfplay_last_actions.argtypes = _i, [_vp, _vp, _vp, _i]


class SelfPlayOracle:
    """Sequential restatement of run_selfplay's game loop (per-game RNG streams), hash evaluator by default."""

    def __init__(self, cfg, tie_mode=1, exp_mode=1):
        self.cfg = cfg
        self.h = lib.ora_selfplay_new(C.byref(cfg), tie_mode, exp_mode)

    def __del__(self):
        try:
            lib.ora_selfplay_free(self.h)
        except Exception:
            pass

    def set_evaluator(self, fn):
        """fn(planes uint16 [n,4736]) -> (value[n], piA[n,4672], piB[n,4672], wdl[n,3], ml[n]) uint16 arrays (as Search.set_evaluator)."""
        def cb(planes, n, value, pia, pib, wdl, ml):
            pl = np.ctypeslib.as_array(C.cast(planes, C.POINTER(C.c_uint16)), shape=(n, 4736))
            v, a, b, w, m = fn(pl)
            for dst, src, cnt in ((value, v, n), (pia, a, n * 4672), (pib, b, n * 4672), (wdl, w, n * 3), (ml, m, n)):
                C.memmove(dst, np.ascontiguousarray(src, dtype=np.uint16).ctypes.data, 2 * cnt)
        self._cb = EVAL_CB(cb)
        lib.ora_selfplay_set_callback(self.h, self._cb)

    def game(self, index):
        """-> (record bytes of the game, dict(samples, raw_plies, winner, termination, nodes), actions [(moveA, moveB, raw)])"""
        cap = 1 << 22
        buf = np.zeros(cap, np.uint8)
        info = np.zeros(8, np.int64)
        n = lib.ora_selfplay_game(self.h, index, buf.ctypes.data, cap, info.ctypes.data)
        assert n >= 0, n
        a = np.zeros(2048, np.uint32); b = np.zeros(2048, np.uint32); r = np.zeros(2048, np.uint8)
        k = lib.ora_selfplay_last_actions(a.ctypes.data, b.ctypes.data, r.ctypes.data, 2048)
        return (buf[:n].tobytes(), dict(samples=int(info[0]), raw_plies=int(info[1]), winner=int(info[2]), termination=int(info[3]), nodes=int(info[4])),
                list(zip(a[:k].tolist(), b[:k].tolist(), r[:k].tolist())))


# ---- tournament oracle (oracle/tournament.hpp) -----------------------------------------------------
class TournamentCfg(C.Structure):
    """Layout of hm_tournament_config (include/hivemind_amd.h) = TournamentConfig (tools/tournament.h:15-42) + slot count."""
    _fields_ = [("games", C.c_uint64), ("nodes", C.c_uint64), ("move_time_ms", C.c_int32), ("contender_batch_size", C.c_int32),
                ("baseline_batch_size", C.c_int32), ("max_macro_plies", C.c_uint64), ("dirichlet_alpha", C.c_float),
                ("dirichlet_epsilon", C.c_float), ("contender_pw_coefficient", C.c_float), ("baseline_pw_coefficient", C.c_float),
                ("seed", C.c_uint64), ("concurrent_games", C.c_int32), ("max_search_nodes", C.c_int32)]


def tournament_cfg(**kw):
    """Reference defaults (tools/tournament.h:15-27) with overrides."""
    c = TournamentCfg(games=20, nodes=400, move_time_ms=0, contender_batch_size=8, baseline_batch_size=8, max_macro_plies=400,
                      dirichlet_alpha=0.3, dirichlet_epsilon=0.10, contender_pw_coefficient=2.0, baseline_pw_coefficient=2.0,
                      seed=1, concurrent_games=64, max_search_nodes=0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


lib.ora_tournament_new.restype, lib.ora_tournament_new.argtypes = _vp, [_vp, _i, _i, _u64, _u64]
lib.ora_tournament_free.restype, lib.ora_tournament_free.argtypes = None, [_vp]
lib.ora_tournament_run.restype, lib.ora_tournament_run.argtypes = _i, [_vp, _vp, _i]
lib.ora_tournament_summary.restype, lib.ora_tournament_summary.argtypes = C.c_int64, [_vp, C.c_char_p, C.c_char_p, _vp, C.c_int64]
lib.ora_tournament_pgn.restype, lib.ora_tournament_pgn.argtypes = C.c_int64, [_vp, C.c_char_p, C.c_char_p, _vp, C.c_int64]
lib.ora_tournament_stats.restype, lib.ora_tournament_stats.argtypes = None, [_u64, _u64, _u64, _vp, _i, _vp]
lib.ora_move_uci.restype, lib.ora_move_uci.argtypes = _i, [C.c_uint32, _vp, _i]
lib.ora_hash_evaluator_salted.restype, lib.ora_hash_evaluator_salted.argtypes = None, [_vp, _i, _u64, _vp, _vp, _vp, _vp, _vp]


def hash_evaluator_salted(planes_u16, salt):
    """The deterministic stand-in network with a salt: (value, piA, piB, wdl, ml) as uint16 fp16 bit patterns."""
    p = np.ascontiguousarray(planes_u16, dtype=np.uint16).reshape(-1, 4736)
    n = len(p)
    v = np.zeros(n, np.uint16); a = np.zeros((n, 4672), np.uint16); b = np.zeros((n, 4672), np.uint16)
    w = np.zeros((n, 3), np.uint16); m = np.zeros(n, np.uint16)
    lib.ora_hash_evaluator_salted(p.ctypes.data, n, salt, v.ctypes.data, a.ctypes.data, b.ctypes.data, w.ctypes.data, m.ctypes.data)
    return v, a, b, w, m


def move_uci(m):
    buf = C.create_string_buffer(16)
    lib.ora_move_uci(int(m), buf, 16)
    return buf.value.decode()


def tournament_stats(contender_wins, baseline_wins, draws, pairs=()):
    p = np.ascontiguousarray(pairs, dtype=np.float64)
    out = np.zeros(10, np.float64)
    lib.ora_tournament_stats(contender_wins, baseline_wins, draws, p.ctypes.data, len(p), out.ctypes.data)
    return dict(score=out[0], elo=out[2] if out[1] else None, score_ci=(out[4], out[5]) if out[3] else None,
                elo_ci=(out[7], out[8]) if out[6] else None,
                method="paired-opening normal approximation" if out[9] else "game-level Wilson approximation")


class TournamentOracle:
    """Sequential restatement of run_tournament with two salted hash evaluators."""

    def __init__(self, cfg, salt_contender=0, salt_baseline=0x5EED, tie_mode=1, exp_mode=1):
        self.h = lib.ora_tournament_new(C.byref(cfg), tie_mode, exp_mode, salt_contender, salt_baseline)

    def __del__(self):
        try:
            lib.ora_tournament_free(self.h)
        except Exception:
            pass

    def run(self):
        err = C.create_string_buffer(256)
        if lib.ora_tournament_run(self.h, err, 256) != 0:
            raise ValueError(err.value.decode())

    def _text(self, fn, a, b):
        cap = 1 << 22
        buf = C.create_string_buffer(cap)
        n = fn(self.h, a.encode(), b.encode(), buf, cap)
        assert n >= 0
        return buf.value.decode()

    def summary(self, a="contender", b="baseline"):
        return self._text(lib.ora_tournament_summary, a, b)

    def pgn(self, a="contender", b="baseline"):
        return self._text(lib.ora_tournament_pgn, a, b)
