"""hivemind_amd — MI355X-native Bughouse self-play rollout engine (host-side mirror of the
reference's rollout interfaces over the C ABI in include/hivemind_amd.h).

Reference interfaces mirrored (engine/src):
  board_to_planes      environment/planes.h:20-22
  Board.legal_moves    environment/board.h:146 / board.cc:133-139
  Board.make_moves     environment/board.cc:316-341
  benchmark_movegen    tools/benchmark.cc:78-97  (-> perft)
PyTorch is used only for device memory and streams.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import (BOARD_DTYPE, POS_DTYPE, DT_F16, DT_F32, DT_U8, MAX_MOVES, NB_PLANES,
                   PLANE_VALUES, POLICY_VALUES, HivemindError, check, lib)

from .search import SearchEngine, SearchConfig, default_config, rules_probe  # noqa: E402,F401
from .selfplay import (SelfPlay, SelfPlayConfig, default_selfplay_config, gather_records, read_hvm4,  # noqa: E402,F401
                       write_chunk)
from .tournament import (Tournament, TournamentConfig, TournamentResult, default_tournament_config,  # noqa: E402,F401
                         move_uci, statistics as tournament_statistics)
from .uci import Uci  # noqa: E402,F401

_TORCH_DT = {DT_F16: torch.float16, DT_F32: torch.float32, DT_U8: torch.uint8}
_NAME_DT = {"f16": DT_F16, "f32": DT_F32, "u8": DT_U8, torch.float16: DT_F16,
            torch.float32: DT_F32, torch.uint8: DT_U8}
_initialised = None


def init(device: int = 0) -> None:
    """Build tables and upload them to `device` (main.cc:75-81)."""
    global _initialised
    if not torch.cuda.is_available():
        raise HivemindError("no GPU visible to torch: hivemind_amd has no CPU path")
    torch.cuda.set_device(device)
    check(lib.hm_init(device))
    _initialised = device


def _require_init():
    if _initialised is None:
        init(torch.cuda.current_device() if torch.cuda.is_available() else 0)


def startpos() -> np.ndarray:
    """Dual start position (Board::Board(), board.cc:52-69) as a BOARD_DTYPE scalar array."""
    out = np.zeros(1, dtype=BOARD_DTYPE)
    check(lib.hm_board_startpos(out.ctypes.data))
    return out


def policy_index(move: int, stm: int) -> int:
    """get_fast_policy_index (common/utils.h:184-216)."""
    return int(lib.hm_policy_index(int(move) & 0xFFFFFFFF, int(stm)))


def to_device(arr: np.ndarray) -> torch.Tensor:
    """Structured numpy array -> uint8 device tensor [n, itemsize]."""
    a = np.ascontiguousarray(arr)
    return torch.from_numpy(a.view(np.uint8).reshape(a.shape[0], a.dtype.itemsize).copy()).cuda()


def _stream_ptr(stream):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def board_to_planes(boards: torch.Tensor, dtype="f16", out: torch.Tensor = None, stream=None) -> torch.Tensor:
    """Batched board_to_planes: boards = uint8 device tensor [n, 208] (hm_board); returns
    [n, 74, 8, 8] of `dtype` (u8 = the HVM4 record quantisation, selfplay.cc:464-476)."""
    _require_init()
    dt = _NAME_DT[dtype]
    if boards.device.type != "cuda" or boards.dtype != torch.uint8 or boards.shape[-1] != 208 or not boards.is_contiguous():
        raise HivemindError("boards must be a contiguous uint8 CUDA tensor of shape [n, 208]")
    n = boards.shape[0]
    if out is None:
        out = torch.empty((n, NB_PLANES, 8, 8), dtype=_TORCH_DT[dt], device=boards.device)
    elif out.dtype != _TORCH_DT[dt] or out.numel() != n * PLANE_VALUES or not out.is_contiguous():
        raise HivemindError("out has the wrong dtype/size")
    check(lib.hm_encode_planes(boards.data_ptr(), n, dt, out.data_ptr(), _stream_ptr(stream)))
    return out


def legal_moves(pos: torch.Tensor, stream=None):
    """Batched Board::legal_moves: pos = uint8 device tensor [n, 96] (hm_pos). Returns
    (moves int32 [n, 512] in the reference's list order, counts int32 [n])."""
    _require_init()
    if pos.device.type != "cuda" or pos.dtype != torch.uint8 or pos.shape[-1] != 96 or not pos.is_contiguous():
        raise HivemindError("pos must be a contiguous uint8 CUDA tensor of shape [n, 96]")
    n = pos.shape[0]
    moves = torch.empty((n, MAX_MOVES), dtype=torch.int32, device=pos.device)
    counts = torch.empty((n,), dtype=torch.int32, device=pos.device)
    check(lib.hm_legal_moves(pos.data_ptr(), n, moves.data_ptr(), counts.data_ptr(), _stream_ptr(stream)))
    return moves, counts


def legal_moves_wave(pos: torch.Tensor, stream=None):
    """As legal_moves, from the wave-cooperative generator of the search kernels (test hook)."""
    _require_init()
    if pos.device.type != "cuda" or pos.dtype != torch.uint8 or pos.shape[-1] != 96 or not pos.is_contiguous():
        raise HivemindError("pos must be a contiguous uint8 CUDA tensor of shape [n, 96]")
    n = pos.shape[0]
    moves = torch.zeros((n, MAX_MOVES), dtype=torch.int32, device=pos.device)
    counts = torch.empty((n,), dtype=torch.int32, device=pos.device)
    check(lib.hm_legal_moves_wave(pos.data_ptr(), n, moves.data_ptr(), counts.data_ptr(), _stream_ptr(stream)))
    return moves, counts


def count_moves(pos: torch.Tensor, stream=None) -> torch.Tensor:
    _require_init()
    n = pos.shape[0]
    counts = torch.empty((n,), dtype=torch.int32, device=pos.device)
    check(lib.hm_count_moves(pos.data_ptr(), n, counts.data_ptr(), _stream_ptr(stream)))
    return counts


def make_moves(boards: torch.Tensor, move_a: torch.Tensor, move_b: torch.Tensor, stream=None) -> torch.Tensor:
    """Batched Board::make_moves (no legality re-check)."""
    _require_init()
    n = boards.shape[0]
    out = torch.empty_like(boards)
    check(lib.hm_make_moves(boards.data_ptr(), move_a.contiguous().data_ptr(), move_b.contiguous().data_ptr(),
                            n, out.data_ptr(), _stream_ptr(stream)))
    return out


def perft(depth: int, root: np.ndarray = None, shard: int = 0, nshards: int = 1):
    """Joint perft (tools/benchmark.cc:59-76). Returns (nodes of this shard, seconds)."""
    _require_init()
    r = startpos() if root is None else np.ascontiguousarray(root)
    nodes, secs = C.c_uint64(0), C.c_double(0.0)
    check(lib.hm_perft(r.ctypes.data, depth, shard, nshards, C.byref(nodes), C.byref(secs)))
    return int(nodes.value), float(secs.value)


def random_positions(n: int, seed: int = 42, max_plies: int = 120, games: int = 2048) -> torch.Tensor:
    """Synthetic workload: n Bughouse positions from seeded random playouts run entirely on the
    GPU with this library's own movegen/make kernels (uniform board, uniform legal move, restart
    after max_plies or on a dead end; team / time-advantage flags random bits).  Returns a uint8
    device tensor [n, 208] (hm_board)."""
    _require_init()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    start = to_device(startpos())
    cur = start.repeat(games, 1).contiguous()
    ply = torch.zeros(games, dtype=torch.int64, device="cuda")
    out, have = [], 0
    zero = torch.zeros(games, dtype=torch.int32, device="cuda")
    while have < n:
        snap = cur.clone()
        bits = torch.randint(0, 4, (games,), device="cuda", generator=gen, dtype=torch.int64)
        snap[:, 202] = (bits & 1).to(torch.uint8)          # team
        snap[:, 203] = (bits >> 1).to(torch.uint8)         # time_adv
        out.append(snap)
        have += games
        board = torch.randint(0, 2, (games,), device="cuda", generator=gen)
        pos = cur[:, :192].reshape(games, 2, 96)
        sel = pos[torch.arange(games, device="cuda"), board].contiguous()
        mv, cnt = legal_moves(sel)
        r = torch.randint(0, 1 << 30, (games,), device="cuda", generator=gen)
        pick = mv[torch.arange(games, device="cuda"), (r % cnt.clamp(min=1).to(torch.int64))]
        pick = torch.where(cnt > 0, pick, zero)
        ma = torch.where(board == 0, pick, zero)
        mb = torch.where(board == 1, pick, zero)
        cur = make_moves(cur, ma, mb)
        ply += 1
        restart = (ply >= max_plies) | (cnt == 0)
        cur = torch.where(restart[:, None], start.expand(games, -1), cur).contiguous()
        ply = torch.where(restart, torch.zeros_like(ply), ply)
    return torch.cat(out)[:n].contiguous()
