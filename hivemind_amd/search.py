"""Host-side mirror of the reference's rollout seam over the hm_sp_* C ABI.

  SearchEngine.begin_search / step / root_stats   <->  Agent::run_search + root_edge_stats
                                                       (search/agent.h:136-144)
  evaluator(planes fp16 [G*8,74,8,8]) -> (value, pi_a, pi_b, wdl, moves_left) fp16 device tensors
                                                  <->  Engine::enqueueInferenceHalf /
                                                       synchronizeInferenceHalf (nn/engine.h:66-81)
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import BOARD_DTYPE, MAX_MOVES, PLANE_VALUES, POLICY_VALUES, HivemindError, check, lib

BATCH = 8   # SearchParams::BATCH_SIZE (search_params.h:26)
INFO_INTS = 20   # HM_SP_INFO_INTS
ST_IDLE, ST_SEARCHING, ST_FINISHING, ST_DONE, ST_NOACTION, ST_ERROR = range(6)


class SearchConfig(C.Structure):
    _fields_ = [("cpuct_init", C.c_float), ("cpuct_base", C.c_float), ("fpu_reduction", C.c_float),
                ("draw_contempt", C.c_float), ("wdl_value_weight", C.c_float), ("moves_left_discount", C.c_float),
                ("pw_coefficient", C.c_float), ("root_pw_coefficient", C.c_float), ("pw_exponent", C.c_float),
                ("enable_transpositions", C.c_int), ("enable_dynamic_fpu", C.c_int), ("enable_wdl_eval", C.c_int)]


_vp, _i = C.c_void_p, C.c_int
_SIGS = {
    "hm_search_config_default": (None, [C.POINTER(SearchConfig)]),
    "hm_sp_create": (_i, [_i, _i, C.POINTER(SearchConfig), C.POINTER(_vp)]),
    "hm_sp_create_ex": (_i, [_i, _i, _i, C.POINTER(SearchConfig), C.POINTER(_vp)]),
    "hm_sp_destroy": (_i, [_vp]),
    "hm_sp_set_games": (_i, [_vp, _vp, _vp]),
    "hm_sp_begin_search": (_i, [_vp, _vp, _vp, C.c_float, C.c_float, _vp]),
    "hm_sp_collect": (_i, [_vp, _vp, _vp]),
    "hm_sp_collect_counted": (_i, [_vp, _vp, _vp, _vp]),
    "hm_sp_process": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i), _vp]),
    "hm_sp_max_edges": (_i, [_vp]),
    "hm_sp_active": (_i, [_vp, C.POINTER(_i)]),
    "hm_sp_root_stats": (_i, [_vp] + [_vp] * 8 + [_i]),
    "hm_sp_apply": (_i, [_vp, _vp, _vp, _vp]),
    "hm_sp_game_state": (_i, [_vp, _vp, _vp, _vp]),
    "hm_sp_raw_policy": (_i, [_vp] * 8),
    "hm_sp_policy_listing": (_i, [_vp] * 8 + [_i]),
    "hm_sp_action_terminal": (_i, [_vp, _vp, _vp, _vp]),
    "hm_rules_probe": (_i, [_vp, C.c_size_t, _vp, _vp]),
    "hm_sp_classify": (_i, [_vp, _vp, _vp]),
    "hm_sp_set_pw_profiles": (_i, [_vp, C.c_float, C.c_float, _vp]),
    "hm_sp_set_batch_sizes": (_i, [_vp, _vp]),
    "hm_sp_set_side": (_i, [_vp, _vp, _vp]),
    "hm_sp_stop": (_i, [_vp, _vp, _vp]),
    "hm_sp_profile": (_i, [_vp, _i]),
    "hm_sp_profile_launches": (_i, [_vp, _i]),
    "hm_sp_leg_times": (_i, [_vp, _vp, _vp, _i]),
    "hm_sp_set_tree_reuse": (_i, [_vp, _vp, _i]),
    "hm_sp_pv_lines": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "hm_sp_leg_clock_net": (_vp, [_vp]),
    "hm_sp_search": (_i, [_vp, _vp, _vp, C.POINTER(C.c_double)]),
    "hm_sp_search_consumers": (_i, [_vp]),
    "hm_sp_search_not_concurrent": (_i, [_vp]),
    "hm_sp_search_stalled": (_i, [_vp]),
    "hm_sp_search_lds_tree": (_i, [_vp]),
    "hm_sp_begin_again": (_i, [_vp]),
    "hm_sp_wait_time": (_i, [_vp, C.POINTER(C.c_double)]),
    "hm_sp_trace_select": (_i, [_i]),
    "hm_sp_trace": (_i, [_vp, _i]),
}
for _n, (_r, _a) in _SIGS.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a
_lib.EXPORTED_SYMBOLS = tuple(_lib.EXPORTED_SYMBOLS) + tuple(_SIGS)


def default_config() -> SearchConfig:
    c = SearchConfig()
    lib.hm_search_config_default(C.byref(c))
    return c


def _p(a):
    return None if a is None else a.ctypes.data


class SearchEngine:
    """G concurrent games, each searched by one wavefront; all games advance in lockstep."""

    def __init__(self, n_games: int, max_nodes: int, config: SearchConfig = None, device=None, max_game_plies: int = 0):
        from . import _require_init
        _require_init()
        self.G = n_games
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.h = _vp()
        cfg = config or default_config()
        check(lib.hm_sp_create_ex(n_games, max_nodes, max_game_plies, C.byref(cfg), C.byref(self.h)))
        self.max_edges = lib.hm_sp_max_edges(self.h)
        rows = n_games * BATCH
        self.planes = [torch.zeros((rows, 74, 8, 8), dtype=torch.float16, device=self.device) for _ in range(2)]
        self.cur = 0

    def close(self):
        if self.h:
            lib.hm_sp_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- game state -------------------------------------------------------------------
    def set_games(self, boards: np.ndarray, mask: np.ndarray = None):
        boards = np.ascontiguousarray(boards, dtype=BOARD_DTYPE)
        assert len(boards) == self.G
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        check(lib.hm_sp_set_games(self.h, boards.ctypes.data, _p(m)))

    def game_state(self, device_out: torch.Tensor = None):
        boards = np.zeros(self.G, dtype=BOARD_DTYPE)
        flags = np.zeros(self.G, dtype=np.int32)
        check(lib.hm_sp_game_state(self.h, boards.ctypes.data, flags.ctypes.data,
                                   None if device_out is None else device_out.data_ptr()))
        return boards, flags

    def apply(self, move_a, move_b, mask=None):
        a = np.ascontiguousarray(move_a, dtype=np.uint32)
        b = np.ascontiguousarray(move_b, dtype=np.uint32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        check(lib.hm_sp_apply(self.h, a.ctypes.data, b.ctypes.data, _p(m)))

    def set_batch_sizes(self, batch=None):
        """Leaves per search iteration and slot (Engine::getBatchSize(), searchthread.cc:663): 1 .. 8 each, None = 8 everywhere."""
        b = None if batch is None else np.ascontiguousarray(np.broadcast_to(np.asarray(batch, dtype=np.uint8), (self.G,)))
        check(lib.hm_sp_set_batch_sizes(self.h, _p(b)))

    # ---- search -----------------------------------------------------------------------
    def begin_search(self, target_nodes, noise_seeds=None, alpha=0.0, eps=0.0, mask=None):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(target_nodes, dtype=np.int32), (self.G,)))
        s = None if noise_seeds is None else np.ascontiguousarray(noise_seeds, dtype=np.uint64)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        check(lib.hm_sp_begin_search(self.h, t.ctypes.data, _p(s), float(alpha), float(eps), _p(m)))
        self.cur = 0

    def collect(self, rows_next=None):
        """Collect the next batch; rows_next (int32 CUDA tensor [G], optional) receives the plane rows each game wrote."""
        st = torch.cuda.current_stream().cuda_stream
        if rows_next is None:
            check(lib.hm_sp_collect(self.h, self.planes[1 - self.cur].data_ptr(), st))
        else:
            check(lib.hm_sp_collect_counted(self.h, self.planes[1 - self.cur].data_ptr(), rows_next.data_ptr(), st))
        return self.planes[self.cur]

    def process(self, value, pi_a, pi_b, wdl, moves_left, want_active=True) -> int:
        st = torch.cuda.current_stream().cuda_stream
        for t in (value, pi_a, pi_b, wdl, moves_left):
            if t.dtype != torch.float16 or not t.is_contiguous() or t.device.type != "cuda":
                raise HivemindError("network heads must be contiguous fp16 CUDA tensors")
        act = _i(0)
        check(lib.hm_sp_process(self.h, value.data_ptr(), pi_a.data_ptr(), pi_b.data_ptr(), wdl.data_ptr(),
                                moves_left.data_ptr(), C.byref(act) if want_active else None, st))
        self.cur = 1 - self.cur
        return act.value

    def run(self, evaluator, max_iters=100000, poll_every=1):
        """Lockstep loop until every game is done.  evaluator(planes) -> 5 fp16 tensors."""
        it = 0
        while it < max_iters:
            planes = self.collect()
            heads = evaluator(planes)
            it += 1
            active = self.process(*heads, want_active=(it % poll_every == 0))
            if it % poll_every == 0 and active == 0:
                break
        return it

    def search_persistent(self, net):
        """The whole search of every slot begin_search left searching as ONE kernel launch (hm_sp_search: game workgroups and
        evaluator workgroups of k_rollout): `net` is a FusedNet.  Same results as run(net), game by game.  -> kernel ms."""
        from .selfplay import EvalIO
        f16 = dict(dtype=torch.float16, device=self.device)
        rows = self.G * BATCH
        if not hasattr(self, "_heads2"):
            self._heads2 = [(torch.zeros(rows, **f16), torch.zeros((rows, POLICY_VALUES), **f16), torch.zeros((rows, POLICY_VALUES), **f16),
                             torch.zeros((rows, 3), **f16), torch.zeros(rows, **f16)) for _ in range(2)]
        io = EvalIO()
        io.planes[0], io.planes[1] = self.planes[0].data_ptr(), self.planes[1].data_ptr()
        (io.value, io.pi_a, io.pi_b, io.wdl, io.moves_left) = [t.data_ptr() for t in self._heads2[0]]
        (io.value_2, io.pi_a_2, io.pi_b_2, io.wdl_2, io.moves_left_2) = [t.data_ptr() for t in self._heads2[1]]
        io.net = net.handle
        torch.cuda.synchronize(self.device)
        ms = C.c_double(0.0)
        rc = lib.hm_sp_search(self.h, net.handle, C.byref(io), C.byref(ms))
        if rc and (lib.hm_sp_search_stalled(self.h) or lib.hm_sp_search_not_concurrent(self.h)):
            # the hang guard gave the search up (or the launch was not resident at once): the same recovery as the self-play driver's —
            # every slot back to the start of its search, then the host-driven loop (same results)
            self.stalls = getattr(self, "stalls", 0) + 1
            check(lib.hm_sp_begin_again(self.h))
            self.cur = 0
            self.run(net)
            return ms.value
        check(rc)
        return ms.value

    def search_consumers(self) -> int:
        return int(lib.hm_sp_search_consumers(self.h))

    def search_lds_tree(self) -> bool:
        return bool(lib.hm_sp_search_lds_tree(self.h))

    def leg_times(self, reset=False):
        """Device-clock totals since the last reset: (ms[collect, forward, process], launches[3]).  Synchronise first."""
        ms, cnt = np.zeros(3, np.float64), np.zeros(3, np.uint64)
        check(lib.hm_sp_leg_times(self.h, ms.ctypes.data, cnt.ctypes.data, int(reset)))
        return ms, cnt

    def root_stats(self):
        G, E = self.G, self.max_edges
        out = dict(counts=np.zeros(G, np.int32), move_a=np.zeros((G, E), np.uint32), move_b=np.zeros((G, E), np.uint32),
                   visits=np.zeros((G, E), np.int32), q=np.zeros((G, E), np.float32), prior=np.zeros((G, E), np.float32),
                   root_q=np.zeros(G, np.float32), info=np.zeros((G, INFO_INTS), np.int32))
        check(lib.hm_sp_root_stats(self.h, *[out[k].ctypes.data for k in
                                             ("counts", "move_a", "move_b", "visits", "q", "prior", "root_q", "info")], E))
        return out

    # ---- raw-policy opening (selfplay.cc:277-390) ----------------------------------------
    def raw_policy(self, pi_a: torch.Tensor, pi_b: torch.Tensor):
        G = self.G
        moves = np.zeros((G, 2, MAX_MOVES), np.uint32)
        probs = np.zeros((G, 2, MAX_MOVES), np.float32)
        caps = np.zeros((G, 2, MAX_MOVES), np.uint8)
        counts = np.zeros((G, 2), np.int32)
        on_turn = np.zeros((G, 2), np.uint8)
        check(lib.hm_sp_raw_policy(self.h, pi_a.data_ptr(), pi_b.data_ptr(), moves.ctypes.data, probs.ctypes.data,
                                   caps.ctypes.data, counts.ctypes.data, on_turn.ctypes.data))
        return moves, probs, caps, counts, on_turn

    def classify(self, team, root_team, root_adv, ply):
        """Test hook (hm_sp_classify): classify_terminal_position / is_draw / repetition counts on the games' current
        positions with their game history.  Arguments broadcast over games.  -> int32 [G, 4]."""
        args = np.ascontiguousarray(np.stack(np.broadcast_arrays(*[np.asarray(x, np.int32) for x in (team, root_team, root_adv, ply)],
                                                                 np.zeros(self.G, np.int32))[:4], axis=1), dtype=np.int32)
        out = np.zeros((self.G, 4), np.int32)
        check(lib.hm_sp_classify(self.h, args.ctypes.data, out.ctypes.data))
        return out

    def action_terminal(self, move_a, move_b):
        a = np.ascontiguousarray(move_a, dtype=np.uint32)
        b = np.ascontiguousarray(move_b, dtype=np.uint32)
        out = np.zeros(self.G, np.int32)
        check(lib.hm_sp_action_terminal(self.h, a.ctypes.data, b.ctypes.data, out.ctypes.data))
        return out


def rules_probe(boards: torch.Tensor):
    """Test hook (see hm_rules_probe)."""
    from . import _require_init
    _require_init()
    n = boards.shape[0]
    out = torch.zeros((n, 8), dtype=torch.int32, device=boards.device)
    keys = torch.zeros((n, 4), dtype=torch.int64, device=boards.device)
    check(lib.hm_rules_probe(boards.data_ptr(), n, out.data_ptr(), keys.data_ptr()))
    return out.cpu().numpy(), keys.cpu().numpy().view(np.uint64)
