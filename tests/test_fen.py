"""Board::fen (environment/board.h:172-174 -> Position::fen(false, true), Fairy-Stockfish position.cpp:637-768) — the text the UCI
`policy` command prints per board (uci.cc:348, 369).  The CPU restatement (oracle/bughouse.hpp Position::fen) and the product
(hm_board_fen, host-only) against strings the reference build itself produced over random playouts (tests/golden/fen_playout.json,
written by tests/golden/make_fen_fixture.py), and FEN -> set -> fen round trips."""
import json
import os

import numpy as np
import pytest

import oracle_py as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _product_fen(board, b):
    from hivemind_amd.uci import board_fen
    return board_fen(board.compact(0, False), b)


def test_fen_matches_reference_strings_over_playouts():
    d = json.load(open(os.path.join(G, "fen_playout.json")))
    n = 0
    for plies in d["games"]:
        b = O.Board()
        for bd, m, fen_a, fen_b in plies:
            assert int(m) in [int(x) for x in b.legal_moves(bd)]
            b.push(bd, m)
            assert b.fen(0) == fen_a and b.fen(1) == fen_b, (b.fen(0), fen_a, b.fen(1), fen_b)
            assert _product_fen(b, 0) == fen_a and _product_fen(b, 1) == fen_b
            n += 1
    assert n >= 600


def test_fen_round_trip_through_set():
    """set(fen) -> fen() is the identity on every string of the fixture (both the CPU restatement's parser and the UCI front end's
    parse_fen feed the same writer)"""
    d = json.load(open(os.path.join(G, "fen_playout.json")))
    b = O.Board()
    for plies in d["games"][:2]:
        for _, _, fen_a, fen_b in plies[::3]:
            b.set(fen_a + "|" + fen_b)
            assert b.fen(0) == fen_a and b.fen(1) == fen_b
            assert _product_fen(b, 0) == fen_a and _product_fen(b, 1) == fen_b


def test_start_position_and_argument_checks():
    import ctypes as C
    import hivemind_amd  # noqa: F401
    from hivemind_amd._lib import lib
    b = O.Board()
    start = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR[] w KQkq - 0 1"
    assert b.fen(0) == b.fen(1) == _product_fen(b, 0) == start
    board = np.ascontiguousarray(b.compact(0, False))
    buf = C.create_string_buffer(8)
    assert lib.hm_board_fen(board.ctypes.data, 0, buf, 8) == -(len(start) + 1)          # too small: the needed size
    assert lib.hm_board_fen(board.ctypes.data, 2, buf, 8) < 0 and b"board index" in lib.hm_last_error()
    assert lib.hm_board_fen(None, 0, buf, 8) < 0


@pytest.mark.skipif(O.ref is None, reason="reference build only exists in the build container")
def test_fen_matches_reference_build_live():
    rng = np.random.RandomState(11)
    for g in range(3):
        r, b = O.Board("ref"), O.Board()
        for ply in range(120):
            lists = [r.legal_moves(0), r.legal_moves(1)]
            bd = int(rng.randint(2))
            if len(lists[bd]) == 0:
                bd ^= 1
            if len(lists[bd]) == 0:
                break
            m = int(lists[bd][rng.randint(len(lists[bd]))])
            r.push(bd, m)
            b.push(bd, m)
            for k in range(2):
                assert r.fen(k) == b.fen(k) == _product_fen(b, k), (g, ply, k)
