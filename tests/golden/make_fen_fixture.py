#!/usr/bin/env python3
"""Writes tests/golden/fen_playout.json: random playouts of the REFERENCE build (oracle/_ref, build container only) with the
reference's own Board::fen(board) of both boards after every ply.  Data only: start FEN pair, the moves played (board, move word),
the FEN strings.  Positions with pockets, promoted pieces ('~'), en-passant squares and lost castling rights all occur."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_py as O  # noqa: E402

assert O.ref is not None, "needs the reference build (make -C oracle ref)"
rng = np.random.RandomState(20261004)
games = []
for g in range(5):
    r = O.Board("ref")
    plies = []
    for ply in range(140):
        lists = [r.legal_moves(0), r.legal_moves(1)]
        bd = int(rng.randint(2))
        if len(lists[bd]) == 0:
            bd ^= 1
        if len(lists[bd]) == 0:
            break
        m = int(lists[bd][rng.randint(len(lists[bd]))])
        r.push(bd, m)
        plies.append([bd, m, r.fen(0), r.fen(1)])
    games.append(plies)
json.dump(dict(source="reference build: Board::fen after every ply of random playouts", games=games),
          open(os.path.join(HERE, "fen_playout.json"), "w"), indent=0)
print(sum(len(g) for g in games), "plies")
