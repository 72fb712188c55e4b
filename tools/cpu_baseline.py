#!/usr/bin/env python3
"""CPU baselines of the integer side of the path: joint perft and Board::legal_moves, timed

  * on the REFERENCE's own sources compiled where they lie (oracle/_ref/libhmref.so — only in the build container), and
  * on the CPU restatement (oracle/liboracle.so — also on the MI355X host, where the reference cannot go),

one pinned core each, median of 3.  With both present the script also prints the restatement / reference ratio that turns a
restatement time measured on the GPU host into a DERIVED reference time there.  TEST / BENCH INFRASTRUCTURE ONLY.

  python tools/cpu_baseline.py [--json out.json] [--cpu 2]
"""
import argparse, ctypes as C, json, os, statistics, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_py as O

ap = argparse.ArgumentParser()
ap.add_argument("--json", default=None)
ap.add_argument("--cpu", type=int, default=None, help="core to pin to (default: the last one this process may use)")
a = ap.parse_args()
cpu = a.cpu if a.cpu is not None else max(os.sched_getaffinity(0))
os.sched_setaffinity(0, {cpu})

LIBS = {"restatement": (O.lib, "ora_")}
if O.ref is not None:
    LIBS["reference"] = (O.ref, "ref_")
for lib, pre in LIBS.values():
    f = getattr(lib, pre + "time_legal_moves")
    f.restype, f.argtypes = C.c_longlong, [C.c_void_p, C.c_int, C.c_int]


def med3(fn):
    ts, val = [], None
    for _ in range(3):
        t0 = time.perf_counter()
        val = fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), val


class B:
    """a Board of either library"""
    def __init__(self, lib, pre):
        self.lib, self.pre = lib, pre
        self.h = C.c_void_p(getattr(lib, pre + "board_new")())

    def f(self, name):
        return getattr(self.lib, self.pre + name)

    def close(self):
        self.f("board_free")(self.h)


def playout_positions(n_games=12, max_plies=180, seed=7):
    """move sequences of seeded random playouts (the oracle plays them; both libraries replay them): [(board index, move), ...] per game"""
    rng = np.random.default_rng(seed)
    games = []
    for _ in range(n_games):
        b = O.Board()
        seq = []
        for _ in range(max_plies):
            bd = int(rng.integers(0, 2))
            mv = b.legal_moves(bd)
            if len(mv) == 0:
                bd ^= 1
                mv = b.legal_moves(bd)
                if len(mv) == 0:
                    break
            m = int(mv[int(rng.integers(0, len(mv)))])
            seq.append((bd, m))
            b.push(bd, m)
        games.append(seq)
    return games


out = {"cpu": cpu, "host": os.uname().nodename, "libs": {}}
games = playout_positions()
n_pos = sum(len(g) for g in games)
for name, (lib, pre) in LIBS.items():
    r = {}
    # joint perft from the dual start position, single thread (the reference's perft is single-threaded: tools/benchmark.cc:59-97)
    for depth in (2, 3):
        b = B(lib, pre)
        t, nodes = med3(lambda: int(b.f("perft")(b.h, depth)))
        r[f"perft{depth}"] = {"seconds": t, "nodes": nodes, "nodes_per_s": nodes / t}
        b.close()
    # perft(4) sampled: 8 of the 400 depth-1 joint moves (every 50th), each followed by perft(3)
    b = B(lib, pre)
    la, lb = (np.zeros(512, np.uint32) for _ in range(2))
    na, nb = b.f("legal_moves")(b.h, 0, la.ctypes.data), b.f("legal_moves")(b.h, 1, lb.ctypes.data)
    picks = [(int(la[k // nb]), int(lb[k % nb])) for k in range(0, na * nb, 50)]

    def sampled():
        tot = 0
        for ma, mb in picks:
            b.f("make_moves")(b.h, ma, mb)
            tot += int(b.f("perft")(b.h, 3))
            b.f("unmake_moves")(b.h, ma, mb)
        return tot
    t, nodes = med3(sampled)
    r["perft4_sampled"] = {"seconds": t, "nodes": nodes, "nodes_per_s": nodes / t, "sample": f"{len(picks)} of {na * nb} depth-1 joint moves (every 50th), perft(3) below each"}
    b.close()
    # Board::legal_moves over random-playout positions: 200 calls per position and board inside the library
    REPS = 200

    def movegen():
        tot = 0
        for seq in games:
            bb = B(lib, pre)
            for bd, m in seq:
                tot += int(bb.f("time_legal_moves")(bb.h, 0, REPS)) + int(bb.f("time_legal_moves")(bb.h, 1, REPS))
                bb.f("push")(bb.h, bd, m)
            bb.close()
        return tot
    t, moves = med3(movegen)
    calls = 2 * REPS * n_pos
    r["legal_moves"] = {"seconds": t, "calls": calls, "us_per_call": t / calls * 1e6, "moves_per_call": moves / calls,
                        "positions": n_pos, "note": "includes one push per position (1 / 400 of the calls)"}
    out["libs"][name] = r
if "reference" in out["libs"]:
    ref, ora = out["libs"]["reference"], out["libs"]["restatement"]
    assert ref["perft3"]["nodes"] == ora["perft3"]["nodes"] == 79245604 and ref["perft4_sampled"]["nodes"] == ora["perft4_sampled"]["nodes"]
    assert abs(ref["legal_moves"]["moves_per_call"] - ora["legal_moves"]["moves_per_call"]) < 1e-9
    out["restatement_over_reference"] = {k: ora[k]["seconds"] / ref[k]["seconds"] for k in ("perft3", "perft4_sampled", "legal_moves")}
print(json.dumps(out, indent=1))
if a.json:
    os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
    json.dump(out, open(a.json, "w"), indent=1)
