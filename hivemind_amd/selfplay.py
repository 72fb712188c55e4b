"""run_selfplay (tools/selfplay.h:33) on one GPU: the C++ driver (hm_selfplay_*) owns the game
loop and RNG decisions, PyTorch runs the RISEv3 evaluator behind the Engine seam, and finished
HVM4 records are gathered to rank 0 (the only cross-GPU exchange)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .search import BATCH, SearchConfig, default_config


class SelfPlayConfig(C.Structure):
    """SelfPlayConfig (tools/selfplay.h:10-31) + sharding / slot count."""
    _fields_ = [("games", C.c_uint64), ("nodes", C.c_uint64), ("max_macro_plies", C.c_uint64), ("chunk_samples", C.c_uint64),
                ("raw_policy_mean_macro_plies", C.c_double), ("raw_policy_max_macro_plies", C.c_uint64),
                ("raw_policy_high_temperature_probability", C.c_double),
                ("mcts_temperature", C.c_double), ("mcts_temperature_decay", C.c_double), ("mcts_temperature_plies", C.c_uint64),
                ("resign_threshold", C.c_float), ("resign_consecutive_plies", C.c_uint64), ("resign_disable_fraction", C.c_double),
                ("node_random_factor", C.c_double), ("dirichlet_alpha", C.c_float), ("dirichlet_epsilon", C.c_float),
                ("seed", C.c_uint64), ("rank", C.c_int), ("world", C.c_int), ("concurrent_games", C.c_int)]


class EvalIO(C.Structure):
    _fields_ = [("planes", C.c_void_p * 2), ("value", C.c_void_p), ("pi_a", C.c_void_p), ("pi_b", C.c_void_p),
                ("wdl", C.c_void_p), ("moves_left", C.c_void_p),
                ("net", C.c_void_p),
                ("value_2", C.c_void_p), ("pi_a_2", C.c_void_p), ("pi_b_2", C.c_void_p), ("wdl_2", C.c_void_p),
                ("moves_left_2", C.c_void_p)]


class SelfPlayResult(C.Structure):
    _fields_ = [("games", C.c_uint64), ("samples", C.c_uint64), ("searched_positions", C.c_uint64), ("total_nodes", C.c_uint64),
                ("eval_rows", C.c_uint64), ("eval_batches", C.c_uint64), ("search_iterations", C.c_uint64), ("raw_plies", C.c_uint64),
                ("record_bytes", C.c_uint64), ("terminations", C.c_uint64 * 5), ("seconds", C.c_double),
                ("collect_ms", C.c_double), ("eval_ms", C.c_double), ("process_ms", C.c_double),
                ("nodes_visited", C.c_uint64), ("edges_scanned", C.c_uint64),
                ("search_seconds", C.c_double), ("prologue_seconds", C.c_double), ("raw_seconds", C.c_double),
                ("chunks_flushed", C.c_uint64), ("leaf_move_words", C.c_uint64),
                ("persistent_searches", C.c_uint64), ("search_kernel_ms", C.c_double), ("wait_ms", C.c_double),
                ("persistent_stalls", C.c_uint64), ("tt_hits", C.c_uint64), ("tt_inserts", C.c_uint64)]


EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)
CHUNK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint64, C.c_uint64, C.c_uint64)
_vp, _i = C.c_void_p, C.c_int
_SIGS = {
    "hm_selfplay_config_default": (None, [C.POINTER(SelfPlayConfig)]),
    "hm_selfplay_create": (_i, [C.POINTER(SelfPlayConfig), C.POINTER(SearchConfig), C.POINTER(EvalIO), EVAL_FN, _vp, C.POINTER(_vp)]),
    "hm_selfplay_run": (_i, [_vp, C.POINTER(SelfPlayResult)]),
    "hm_selfplay_set_chunk_sink": (_i, [_vp, CHUNK_FN, _vp]),
    "hm_selfplay_set_output_directory": (_i, [_vp, C.c_char_p]),
    "hm_selfplay_records": (C.c_uint64, [_vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]),
    "hm_selfplay_destroy": (_i, [_vp]),
    "hm_hvm4_write_chunk": (_i, [C.c_char_p, _vp, C.c_uint64, C.c_uint64]),
}
for _n, (_r, _a) in _SIGS.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a
_lib.EXPORTED_SYMBOLS = tuple(_lib.EXPORTED_SYMBOLS) + tuple(_SIGS)


def default_selfplay_config(**kw) -> SelfPlayConfig:
    c = SelfPlayConfig()
    lib.hm_selfplay_config_default(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class SelfPlay:
    """One rank's self-play worker.  `net(planes[rows]) -> (value, pi_a, pi_b, wdl, moves_left)` fp16."""

    def __init__(self, config: SelfPlayConfig, net, search_config: SearchConfig = None, device=None,
                 output_directory: str = None, chunk_sink=None):
        """output_directory: ChunkWriter (selfplay.cc:69-158) writing <dir>/training_data/chunk_<runId>_<idx>.hvm every
        config.chunk_samples samples; chunk_sink(records uint8 array, count, chunk_index): caller's sink instead (e.g. a
        per-chunk gather).  With neither, records stay in memory until records()."""
        from . import _require_init
        _require_init()
        self.cfg = config
        self.net = net
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        G = config.concurrent_games
        rows = G * BATCH
        f16 = dict(dtype=torch.float16, device=self.device)
        self.planes = [torch.zeros((rows, 74, 8, 8), **f16) for _ in range(2)]
        self.value = torch.zeros(rows, **f16)
        self.pi_a = torch.zeros((rows, 4672), **f16)
        self.pi_b = torch.zeros((rows, 4672), **f16)
        self.wdl = torch.zeros((rows, 3), **f16)
        self.ml = torch.zeros(rows, **f16)
        self.io = EvalIO()
        self.io.planes[0], self.io.planes[1] = self.planes[0].data_ptr(), self.planes[1].data_ptr()
        self.io.value, self.io.pi_a, self.io.pi_b = self.value.data_ptr(), self.pi_a.data_ptr(), self.pi_b.data_ptr()
        self.io.wdl, self.io.moves_left = self.wdl.data_ptr(), self.ml.data_ptr()
        self.eval_rows = 0
        self._error = None
        if hasattr(net, "wh") and getattr(net, "native", True):
            # FusedNet: hand the packed network to the C++ driver (no Python in the search loop; the
            # forward runs on its own stream, overlapped with leaf collection)
            self._heads2 = (torch.zeros(rows, **f16), torch.zeros((rows, 4672), **f16), torch.zeros((rows, 4672), **f16),
                            torch.zeros((rows, 3), **f16), torch.zeros(rows, **f16))
            self.io.net = net.handle
            (self.io.value_2, self.io.pi_a_2, self.io.pi_b_2, self.io.wdl_2, self.io.moves_left_2) = [t.data_ptr() for t in self._heads2]

        fused = hasattr(self.net, "wh")            # FusedNet: writes the registered head buffers directly

        def cb(_user, which, rows_):
            try:
                if fused:
                    self.net(self.planes[which][:rows_], out=(self.value[:rows_], self.pi_a[:rows_], self.pi_b[:rows_],
                                                               self.wdl[:rows_], self.ml[:rows_]))
                    self.eval_rows += rows_
                    return 0
                with torch.no_grad():
                    v, a, b, w, m = self.net(self.planes[which][:rows_])
                self.value[:rows_].copy_(v.reshape(-1)); self.pi_a[:rows_].copy_(a); self.pi_b[:rows_].copy_(b)
                self.wdl[:rows_].copy_(w); self.ml[:rows_].copy_(m.reshape(-1))
                self.eval_rows += rows_
                return 0
            except Exception as e:  # surfaced after hm_selfplay_run returns
                self._error = e
                return 1
        self._cb = EVAL_FN(cb)
        self.h = _vp()
        scfg = search_config or default_config()
        check(lib.hm_selfplay_create(C.byref(config), C.byref(scfg), C.byref(self.io), self._cb, None, C.byref(self.h)))
        self._sink = None
        if chunk_sink is not None:
            def sink(_user, data, nbytes, count, index):
                try:
                    chunk_sink(np.ctypeslib.as_array(data, shape=(nbytes,)).copy(), int(count), int(index))
                    return 0
                except Exception as e:
                    self._error = e
                    return 1
            self._sink = CHUNK_FN(sink)
            check(lib.hm_selfplay_set_chunk_sink(self.h, self._sink, None))
        elif output_directory is not None:
            check(lib.hm_selfplay_set_output_directory(self.h, str(output_directory).encode()))

    def run(self) -> SelfPlayResult:
        res = SelfPlayResult()
        rc = lib.hm_selfplay_run(self.h, C.byref(res))
        if self._error is not None:
            raise self._error
        check(rc)
        return res

    def records(self):
        data = C.POINTER(C.c_uint8)()
        count = C.c_uint64(0)
        n = lib.hm_selfplay_records(self.h, C.byref(data), C.byref(count))
        buf = np.ctypeslib.as_array(data, shape=(n,)).copy() if n else np.zeros(0, np.uint8)
        return buf, int(count.value)

    def close(self):
        if self.h:
            lib.hm_selfplay_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_chunk(path: str, records: np.ndarray, count: int):
    records = np.ascontiguousarray(records, dtype=np.uint8)
    check(lib.hm_hvm4_write_chunk(path.encode(), records.ctypes.data, records.size, count))


def gather_records(records: np.ndarray, count: int, dist=None):
    """The one cross-GPU exchange: variable-length gather of finished HVM4 sample bytes to rank 0.  An all_gather of
    (byte count, sample count) pairs, then every other rank sends exactly its payload to rank 0 (point-to-point over
    RCCL / gloo): nothing is padded and no rank other than 0 receives anything."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return records, count
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    world, rank = dist.get_world_size(), dist.get_rank()
    meta = torch.tensor([records.size, count], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    sizes = [int(m[0]) for m in metas]
    if rank != 0:
        if records.size:
            dist.send(torch.from_numpy(np.ascontiguousarray(records)).to(dev), dst=0)
        return np.zeros(0, np.uint8), 0
    parts = [np.ascontiguousarray(records)]
    for r in range(1, world):
        if sizes[r]:
            buf = torch.empty(sizes[r], dtype=torch.uint8, device=dev)
            dist.recv(buf, src=r)
            parts.append(buf.cpu().numpy())
    return np.concatenate(parts), sum(int(m[1]) for m in metas)


def read_hvm4(path: str):
    """Minimal HVM4 reader / validator written from the reference reader's constants
    (src/preprocessing/convert_selfplay_data.py:24-28, 69-146): struct strings '<4sIHHQ',
    '<QIHHBBbBf', '<Hf'; returns a list of dict samples."""
    import struct
    data = open(path, "rb").read()
    magic, version, channels, policy, count = struct.unpack_from("<4sIHHQ", data, 0)
    if magic != b"HVM4" or version != 4 or channels != 74 or policy != 4672:
        raise ValueError("not an HVM4 chunk")
    off = struct.calcsize("<4sIHHQ")
    out = []
    for _ in range(count):
        gid, nodes, mply, mleft, team, adv, outcome, wdl, rootq = struct.unpack_from("<QIHHBBbBf", data, off)
        off += struct.calcsize("<QIHHBBbBf")
        planes = np.frombuffer(data, np.uint8, 4736, off).reshape(74, 8, 8)
        off += 4736
        pols = []
        for _b in range(2):
            (n,) = struct.unpack_from("<H", data, off)
            off += 2
            ent = np.frombuffer(data, np.dtype([("index", "<u2"), ("prob", "<f4")]), n, off)
            off += 6 * n
            pols.append(ent)
        out.append(dict(game_id=gid, nodes=nodes, macro_ply=mply, moves_left=mleft, team=team, time_adv=adv,
                        outcome=outcome, wdl=wdl, root_q=rootq, planes=planes, policy_a=pols[0], policy_b=pols[1]))
    if off != len(data):
        raise ValueError("trailing bytes in chunk")
    return out
