"""run_tournament (tools/tournament.h:69-75, tools/tournament.cc:328-465) on one GPU: paired games between two
networks on the lockstep rollout engine.  The C++ driver (hm_tournament_*) owns the game loop; this module is the
ctypes mirror: configuration, result statistics, summary.json / games.pgn text."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .search import BATCH, SearchConfig, default_config
from .selfplay import EVAL_FN, EvalIO


class TournamentConfig(C.Structure):
    """TournamentConfig (tools/tournament.h:15-42) + slot count."""
    _fields_ = [("games", C.c_uint64), ("nodes", C.c_uint64), ("move_time_ms", C.c_int32), ("contender_batch_size", C.c_int32),
                ("baseline_batch_size", C.c_int32), ("max_macro_plies", C.c_uint64), ("dirichlet_alpha", C.c_float),
                ("dirichlet_epsilon", C.c_float), ("contender_pw_coefficient", C.c_float), ("baseline_pw_coefficient", C.c_float),
                ("seed", C.c_uint64), ("concurrent_games", C.c_int32), ("max_search_nodes", C.c_int32)]


class TournamentBreakdown(C.Structure):
    _fields_ = [("wins", C.c_uint64), ("losses", C.c_uint64), ("draws", C.c_uint64)]


class TournamentResult(C.Structure):
    """TournamentResult (tools/tournament.h:50-73) with its statistics evaluated (tournament.cc:247-326)."""
    _fields_ = [("contender_wins", C.c_uint64), ("baseline_wins", C.c_uint64), ("draws", C.c_uint64),
                ("as_white", TournamentBreakdown), ("as_black", TournamentBreakdown), ("up_time", TournamentBreakdown),
                ("down_time", TournamentBreakdown),
                ("checkmates", C.c_uint64), ("no_legal_actions", C.c_uint64), ("drawn_terminations", C.c_uint64),
                ("macro_ply_limits", C.c_uint64), ("pairs", C.c_uint64), ("contender_score", C.c_double),
                ("has_elo", C.c_int32), ("has_score_ci", C.c_int32), ("has_elo_ci", C.c_int32), ("paired_method", C.c_int32),
                ("contender_elo", C.c_double), ("score_ci", C.c_double * 2), ("elo_ci", C.c_double * 2),
                ("searched_positions", C.c_uint64), ("total_nodes", C.c_uint64), ("search_iterations", C.c_uint64),
                ("seconds", C.c_double)]

    @property
    def games(self):
        return self.contender_wins + self.baseline_wins + self.draws

    @property
    def confidence_method(self):
        return "paired-opening normal approximation" if self.paired_method else "game-level Wilson approximation"


_vp, _i = C.c_void_p, C.c_int
_SIGS = {
    "hm_tournament_config_default": (None, [C.POINTER(TournamentConfig)]),
    "hm_tournament_create": (_i, [C.POINTER(TournamentConfig), C.POINTER(SearchConfig), C.POINTER(EvalIO), _vp, EVAL_FN, _vp, C.POINTER(_vp)]),
    "hm_tournament_run": (_i, [_vp, C.POINTER(TournamentResult)]),
    "hm_tournament_acting": (_i, [_vp, _vp]),
    "hm_tournament_pair_scores": (C.c_uint64, [_vp, C.POINTER(C.POINTER(C.c_double))]),
    "hm_tournament_summary": (C.c_int64, [_vp, C.c_char_p, C.c_char_p, _vp, C.c_int64]),
    "hm_tournament_pgn": (C.c_int64, [_vp, C.c_char_p, C.c_char_p, _vp, C.c_int64]),
    "hm_tournament_write_reports": (_i, [_vp, C.c_char_p, C.c_char_p, C.c_char_p]),
    "hm_tournament_destroy": (_i, [_vp]),
    "hm_tournament_statistics": (_i, [C.c_uint64, C.c_uint64, C.c_uint64, _vp, C.c_uint64, C.POINTER(TournamentResult)]),
    "hm_move_uci": (_i, [C.c_uint32, _vp, _i]),
}
for _n, (_r, _a) in _SIGS.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a
_lib.EXPORTED_SYMBOLS = tuple(_lib.EXPORTED_SYMBOLS) + tuple(_SIGS)


def default_tournament_config(**kw) -> TournamentConfig:
    c = TournamentConfig()
    lib.hm_tournament_config_default(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def statistics(contender_wins, baseline_wins, draws, pair_scores=()) -> TournamentResult:
    """Score, Elo and 95 % intervals of a W-L-D record (host-only; TournamentResult::contenderScore ... eloConfidence95)."""
    r = TournamentResult()
    p = np.ascontiguousarray(pair_scores, dtype=np.float64)
    check(lib.hm_tournament_statistics(contender_wins, baseline_wins, draws, p.ctypes.data if p.size else None, p.size, C.byref(r)))
    return r


def move_uci(move: int) -> str:
    """Board::uci_move (environment/board.h:340-350): 'pass' for MOVE_NONE, else UCI::move of the variant."""
    buf = C.create_string_buffer(16)
    lib.hm_move_uci(int(move), buf, 16)
    return buf.value.decode()


class Tournament:
    """Paired tournament between `contender` and `baseline`.  Either both are FusedNet objects (native: both forwards run
    on the network stream inside the captured iteration graph), or `evaluator(planes[rows], acting) -> 5 fp16 heads` is a
    callback serving both networks (acting[slot] = 1 where the contender evaluates)."""

    def __init__(self, config: TournamentConfig, contender=None, baseline=None, evaluator=None, search_config: SearchConfig = None, device=None):
        from . import _require_init
        _require_init()
        self.cfg = config
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        G = int(min(config.concurrent_games, config.games)) if config.games else config.concurrent_games
        G = max(G, 1)
        rows = G * BATCH
        f16 = dict(dtype=torch.float16, device=self.device)
        self.planes = [torch.zeros((rows, 74, 8, 8), **f16) for _ in range(2)]
        self.heads = [(torch.zeros(rows, **f16), torch.zeros((rows, 4672), **f16), torch.zeros((rows, 4672), **f16),
                       torch.zeros((rows, 3), **f16), torch.zeros(rows, **f16)) for _ in range(2)]
        self.io = EvalIO()
        self.io.planes[0], self.io.planes[1] = self.planes[0].data_ptr(), self.planes[1].data_ptr()
        (self.io.value, self.io.pi_a, self.io.pi_b, self.io.wdl, self.io.moves_left) = [t.data_ptr() for t in self.heads[0]]
        (self.io.value_2, self.io.pi_a_2, self.io.pi_b_2, self.io.wdl_2, self.io.moves_left_2) = [t.data_ptr() for t in self.heads[1]]
        self.nets = (contender, baseline)
        self._error = None
        self.G = G
        baseline_handle = None
        if contender is not None and baseline is not None:
            self.io.net = contender.handle
            baseline_handle = baseline.handle
        elif evaluator is None:
            raise ValueError("give two FusedNet networks or an evaluator callback")

        def cb(_user, which, rows_):
            try:
                acting = np.zeros(G, np.uint8)
                check(lib.hm_tournament_acting(self.h, acting.ctypes.data))
                out = evaluator(self.planes[which][:rows_], acting)
                for dst, src in zip(self.heads[0], out):
                    dst[:rows_].copy_(src.reshape(dst[:rows_].shape))
                return 0
            except Exception as e:  # surfaced after hm_tournament_run returns
                self._error = e
                return 1
        self._cb = EVAL_FN(cb)
        self.h = _vp()
        scfg = search_config or default_config()
        check(lib.hm_tournament_create(C.byref(config), C.byref(scfg), C.byref(self.io), baseline_handle, self._cb, None, C.byref(self.h)))

    def run(self) -> TournamentResult:
        res = TournamentResult()
        rc = lib.hm_tournament_run(self.h, C.byref(res))
        if self._error is not None:
            raise self._error
        check(rc)
        return res

    def pair_scores(self):
        p = C.POINTER(C.c_double)()
        n = lib.hm_tournament_pair_scores(self.h, C.byref(p))
        return [p[i] for i in range(n)]

    def _text(self, fn, contender_name, baseline_name):
        n = fn(self.h, contender_name.encode(), baseline_name.encode(), None, 0)
        buf = C.create_string_buffer(int(-n) + 1)
        fn(self.h, contender_name.encode(), baseline_name.encode(), buf, len(buf))
        return buf.value.decode()

    def summary(self, contender_name="contender", baseline_name="baseline") -> str:
        return self._text(lib.hm_tournament_summary, contender_name, baseline_name)

    def pgn(self, contender_name="contender", baseline_name="baseline") -> str:
        return self._text(lib.hm_tournament_pgn, contender_name, baseline_name)

    def write_reports(self, directory, contender_name="contender", baseline_name="baseline"):
        check(lib.hm_tournament_write_reports(self.h, str(directory).encode(), contender_name.encode(), baseline_name.encode()))

    def close(self):
        if self.h:
            lib.hm_tournament_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
