"""UCI front end (interface/uci.cc) over the GPU search engine: ctypes mirror of hm_uci_* plus the stdin/stdout loop of
UCI::loop (uci.cc:396-429).  `python -m hivemind_amd.uci --model small` plays through any UCI GUI that speaks the reference's
bughouse dialect (moves prefixed with the board number, bestmove as a pair)."""
import ctypes as C
import sys

import torch

from . import _lib
from ._lib import check, lib
from .selfplay import EVAL_FN, EvalIO

UCI_QUIT = -1000000
_vp, _i = C.c_void_p, C.c_int
_SIGS = {
    "hm_uci_create": (_i, [_vp, C.POINTER(EvalIO), EVAL_FN, _vp, _i, C.POINTER(_vp)]),
    "hm_uci_command": (C.c_int64, [_vp, C.c_char_p, _vp, C.c_int64]),
    "hm_uci_board": (_i, [_vp, _vp]),
    "hm_uci_destroy": (_i, [_vp]),
    "hm_uci_busy": (_i, [_vp]),
    "hm_board_fen": (_i, [_vp, _i, C.c_char_p, _i]),
    "hm_insurmountable_visit_lead": (_i, [C.c_float, C.c_float, C.c_float]),
    "hm_time_manager_create": (_vp, [_i]),
    "hm_time_manager_poll": (_i, [_vp, C.c_double, _i, _i, _vp, _vp, _i, _vp, _vp, C.POINTER(C.c_double), _vp, _i]),
    "hm_time_manager_destroy": (None, [_vp]),
}
for _n, (_r, _a) in _SIGS.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a
_lib.EXPORTED_SYMBOLS = tuple(_lib.EXPORTED_SYMBOLS) + tuple(_SIGS)


def board_fen(board, b: int) -> str:
    """Board::fen(b) of a compact board record (numpy BOARD_DTYPE, one element); host-only."""
    import numpy as np
    board = np.ascontiguousarray(board)
    buf = C.create_string_buffer(160)
    n = lib.hm_board_fen(board.ctypes.data, int(b), buf, len(buf))
    if n < 0:
        raise ValueError("hm_board_fen failed")
    return buf.value.decode()


class Uci:
    """`net`: a FusedNet (native evaluator) or any callable planes[8 rows] -> five fp16 heads (callback evaluator)."""

    def __init__(self, net, max_nodes=100000, device=None):
        from . import _require_init
        _require_init()
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        f16 = dict(dtype=torch.float16, device=self.device)
        self.planes = [torch.zeros((8, 74, 8, 8), **f16) for _ in range(2)]
        self.heads = (torch.zeros(8, **f16), torch.zeros((8, 4672), **f16), torch.zeros((8, 4672), **f16), torch.zeros((8, 3), **f16), torch.zeros(8, **f16))
        self.io = EvalIO()
        self.io.planes[0], self.io.planes[1] = self.planes[0].data_ptr(), self.planes[1].data_ptr()
        (self.io.value, self.io.pi_a, self.io.pi_b, self.io.wdl, self.io.moves_left) = [t.data_ptr() for t in self.heads]
        self.net = net
        self._error = None
        native = hasattr(net, "handle") and getattr(net, "native", True)

        def cb(_user, which, rows):
            try:
                out = net(self.planes[which][:rows])
                for dst, src in zip(self.heads, out):
                    dst[:rows].copy_(src.reshape(dst[:rows].shape))
                return 0
            except Exception as e:
                self._error = e
                return 1
        self._cb = EVAL_FN(cb)
        self.h = _vp()
        check(lib.hm_uci_create(net.handle if native else None, C.byref(self.io), self._cb, None, int(max_nodes), C.byref(self.h)))
        self._buf = C.create_string_buffer(1 << 16)

    def _call(self, line: str):
        n = lib.hm_uci_command(self.h, line.encode(), self._buf, len(self._buf))
        if self._error is not None:
            err, self._error = self._error, None
            raise err
        if n < 0 and n != UCI_QUIT:                              # the text did not fit: the command has run, fetch with an empty line
            self._buf = C.create_string_buffer(int(-n) + 1)
            n = lib.hm_uci_command(self.h, b"", self._buf, len(self._buf))
        return self._buf.value.decode() if n != 0 else "", n == UCI_QUIT

    def command(self, line: str, wait: bool = True):
        """-> (output text, quit flag).  The C ABI never blocks on a search (`go` starts it on the engine's worker thread);
        with wait=True an ordinary `go` is followed here until its bestmove has been printed, which is what scripts and tests
        want.  `go ponder` always returns at once (it ends with `ponderhit` / `stop`); the stdin loop passes wait=False."""
        text, quit_ = self._call(line)
        tok = line.split()
        if wait and tok and tok[0] == "go" and "ponder" not in tok:
            import time
            while self.busy():
                time.sleep(0.0005)                               # releases the GIL: a callback evaluator runs on the worker thread
                more, _ = self._call("")
                text += more
            more, _ = self._call("")
            text += more
        return text, quit_

    def busy(self) -> bool:
        """a search is still running on the engine's worker thread (poll its text with command(""))"""
        return bool(lib.hm_uci_busy(self.h))

    def board(self):
        import numpy as np
        out = np.zeros(1, dtype=_lib.BOARD_DTYPE)
        check(lib.hm_uci_board(self.h, out.ctypes.data))
        return out

    def close(self):
        if self.h:
            lib.hm_uci_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def main(argv=None):
    import argparse
    from . import init, net as N
    ap = argparse.ArgumentParser(description="hivemind UCI engine on the MI355X search engine")
    ap.add_argument("--model", default="small", choices=["small", "full"])
    ap.add_argument("--checkpoint", default=None, help="reference trainer checkpoint (.tar) to load")
    ap.add_argument("--max-nodes", type=int, default=100000)
    a = ap.parse_args(argv)
    init(0)
    torch.manual_seed(0)
    model = N.load_checkpoint(a.checkpoint) if a.checkpoint else (N.rise_v3_small() if a.model == "small" else N.rise_v33())
    uci = Uci(N.FusedNet(model), a.max_nodes)
    import queue
    import threading

    def emit(text):
        if text:
            sys.stdout.write(text)
            sys.stdout.flush()
    # stdin is read on its own thread, line by line, into a queue: the loop below never mixes select() on the descriptor with
    # Python's buffered readline (two commands arriving in one pipe write would leave the second in the buffer, unseen by select)
    lines = queue.Queue()

    def reader():
        for ln in sys.stdin:
            lines.put(ln)
        lines.put(None)
    threading.Thread(target=reader, daemon=True).start()
    while True:
        try:
            line = lines.get(timeout=0.005 if uci.busy() else None)
        except queue.Empty:
            emit(uci.command("", wait=False)[0])                 # a running search prints as it goes / when it ends
            continue
        if line is None:
            line = "quit"                                        # end of input is `quit` (uci.cc:399-401), which stops a running search
        text, quit_ = uci.command(line.strip(), wait=False)
        emit(text)
        if quit_:
            break
    emit(uci.command("")[0])
    uci.close()


if __name__ == "__main__":
    main()
