"""CPU suite for the tournament restatement (oracle/tournament.hpp <- tools/tournament.cc) and the host-only parts of the
product's tournament ABI: the reference's own known answers for the statistics (engine/tests/test_tournament.cc:7-72),
move strings against the reference build's Board::uci_move, argument checks with the reference's texts, and the
properties of a played tournament (pairing, colour / time-advantage alternation, report files).  The GPU driver is compared
with this restatement byte for byte in tests/test_gpu_tournament.py."""
import json
import math
import os
import re

import numpy as np
import pytest

import oracle_py as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def both_stats(cw, bw, d, pairs=()):
    """the oracle's and the product's (host-only C ABI) evaluation of the same record"""
    import hivemind_amd as hm
    o = O.tournament_stats(cw, bw, d, pairs)
    r = hm.tournament_statistics(cw, bw, d, pairs)
    p = dict(score=r.contender_score, elo=r.contender_elo if r.has_elo else None,
             score_ci=(r.score_ci[0], r.score_ci[1]) if r.has_score_ci else None,
             elo_ci=(r.elo_ci[0], r.elo_ci[1]) if r.has_elo_ci else None, method=r.confidence_method)
    assert o == p, (o, p)                                        # the same doubles, bit for bit
    return o


def test_reference_known_answers_score_and_elo():
    """TournamentResultTest.ComputesScoreAndElo / EloIsUndefinedAtScoreEndpoints (test_tournament.cc:20-40)"""
    s = both_stats(6, 2, 2)
    assert s["score"] == 0.7 and abs(s["elo"] - 147.1907) < 1e-3
    s = both_stats(0, 0, 0)
    assert s["score"] == 0.0 and s["elo"] is None and s["score_ci"] is None
    assert both_stats(2, 0, 0)["elo"] is None


def test_reference_known_answers_confidence_interval():
    """TournamentResultTest.ComputesFirstTournamentConfidenceInterval (test_tournament.cc:42-64)"""
    pairs = [1.0] * 23 + [0.5] * 26 + [0.0]
    s = both_stats(72, 28, 0, pairs)
    assert s["score"] == 0.72 and abs(s["elo"] - 164.069786) < 1e-6
    assert abs(s["score_ci"][0] - 0.645078483) < 1e-9 and abs(s["score_ci"][1] - 0.794921517) < 1e-9
    assert abs(s["elo_ci"][0] - 103.792091) < 1e-6 and abs(s["elo_ci"][1] - 235.361663) < 1e-6
    assert s["method"] == "paired-opening normal approximation"
    # fewer than two pairs: the game-level Wilson interval (tournament.cc:288-300)
    w = both_stats(6, 2, 2, [0.75])
    z, n, p = 1.959963984540054, 10.0, 0.7
    den = 1 + z * z / n
    c = (p + z * z / (2 * n)) / den
    m = z * math.sqrt(p * (1 - p) / n + z * z / (4 * n * n)) / den
    assert w["method"] == "game-level Wilson approximation" and abs(w["score_ci"][0] - (c - m)) < 1e-12 and abs(w["score_ci"][1] - (c + m)) < 1e-12


def test_move_strings_on_reference_positions():
    """Board::uci_move (board.h:340-350 -> UCI::move, stubs.cpp:21-59) on legal moves of the reference playouts: the
    tournament's move text (oracle and product) equals the Board-level formatter of the oracle; drops, promotions,
    castling (king -> g/c file) and plain moves all occur."""
    import hivemind_amd as hm
    d = np.load(os.path.join(G, "ref_playout.npz"))
    boards = d["boards"].view(O.BOARD_DTYPE).reshape(-1)
    offs, moves = d["offsets"], d["moves"]
    ob = O.Board()
    kinds = set()
    for i in range(0, len(boards), 7):
        ob.from_compact(boards[i:i + 1])
        for bd in range(2):
            for m in moves[offs[2 * i + bd]:offs[2 * i + bd + 1]]:
                text = O.move_uci(m)
                assert text == hm.move_uci(m) == ob.uci(bd, m), (i, bd, int(m))
                kinds.add((int(m) >> 12) & 15)
    assert {0, 2, 3, 4} <= kinds                                  # normal, castling, promotion, drop all occurred
    assert O.move_uci(0) == hm.move_uci(0) == "pass"


@pytest.mark.skipif(O.ref is None, reason="reference build only exists in the build container")
def test_move_strings_match_reference_build():
    """the same text as the reference's own Board::uci_move, move by move over random playouts of the reference build"""
    import hivemind_amd as hm
    rng = np.random.RandomState(5)
    kinds = set()
    for g in range(8):
        r = O.Board("ref")
        for ply in range(160):
            lists = [r.legal_moves(0), r.legal_moves(1)]
            for bd in range(2):
                for m in lists[bd]:
                    want = r.uci(bd, m)
                    assert O.move_uci(m) == hm.move_uci(m) == want, (g, ply, bd, int(m), want)
                    kinds.add((int(m) >> 12) & 15)
            bd = int(rng.randint(2))
            if len(lists[bd]) == 0:
                bd ^= 1
            if len(lists[bd]) == 0:
                break
            r.push(bd, lists[bd][rng.randint(len(lists[bd]))])
    assert {0, 3, 4} <= kinds


def test_argument_checks_carry_the_reference_texts():
    """run_tournament's std::invalid_argument texts (tournament.cc:334-358), oracle and product alike"""
    import ctypes as C
    import hivemind_amd as hm
    from hivemind_amd.selfplay import EvalIO, EVAL_FN
    cases = [(dict(games=3), "Tournament games must be a positive even number"),
             (dict(games=0), "Tournament games must be a positive even number"),
             (dict(nodes=0), "Tournament requires exactly one positive nodes or movetime limit"),
             (dict(nodes=100, move_time_ms=50), "Tournament requires exactly one positive nodes or movetime limit"),
             (dict(max_macro_plies=0), "Tournament requires exactly one positive nodes or movetime limit"),
             (dict(contender_batch_size=0), "Tournament batch sizes must be positive"),
             (dict(dirichlet_epsilon=1.5), "Invalid tournament Dirichlet configuration"),
             (dict(baseline_pw_coefficient=0.0), "Tournament PW coefficients must be positive and finite"),
             (dict(contender_pw_coefficient=float("inf")), "Tournament PW coefficients must be positive and finite")]
    io = EvalIO()
    cb = EVAL_FN(lambda *_: 0)
    for kw, text in cases:
        with pytest.raises(ValueError, match=re.escape(text)):
            O.TournamentOracle(O.tournament_cfg(**kw)).run()
        h = C.c_void_p()
        rc = hm.lib.hm_tournament_create(C.byref(hm.default_tournament_config(**kw)), None, C.byref(io), None, cb, None, C.byref(h))
        assert rc != 0 and hm.lib.hm_last_error().decode() == text, (kw, hm.lib.hm_last_error())
    # what the GPU engine does not build (a movetime tournament over a callback evaluator, batch sizes above 8) is refused with its
    # own text, not silently ignored
    for kw in (dict(nodes=0, move_time_ms=100), dict(contender_batch_size=16)):
        h = C.c_void_p()
        assert hm.lib.hm_tournament_create(C.byref(hm.default_tournament_config(**kw)), None, C.byref(io), None, cb, None, C.byref(h)) != 0
        assert b"not built" in hm.lib.hm_last_error() or b"is built" in hm.lib.hm_last_error()
    d = hm.default_tournament_config()
    r = O.tournament_cfg()
    for f, _ in d._fields_:
        assert getattr(d, f) == getattr(r, f), f                  # TournamentConfig defaults (tournament.h:15-27)


def test_oracle_tournament_pairing_and_reports():
    cfg = O.tournament_cfg(games=4, nodes=40, max_macro_plies=24, seed=9, contender_pw_coefficient=1.5, baseline_pw_coefficient=2.5)
    t = O.TournamentOracle(cfg)
    t.run()
    s = json.loads(t.summary("new", "old"))
    assert s["contender"] == "new" and s["baseline"] == "old" and s["games"] == 4 and s["nodes_per_move"] == 40 and s["seed"] == 9
    assert s["contender_wins"] + s["baseline_wins"] + s["draws"] == 4
    b = s["contender_breakdown"]
    # the contender plays White in even games and Black in odd ones; pairs alternate the starting team, so the contender
    # moves second (has the time advantage, tournament.cc:376-378) in games 1 and 2 of every four
    assert sum(b["white"].values()) == 2 and sum(b["black"].values()) == 2 and sum(b["up_time"].values()) == 2 and sum(b["down_time"].values()) == 2
    assert sum(s["terminations"].values()) == 4
    assert s["confidence_method"] == "paired-opening normal approximation"
    assert abs(s["contender_pw_coefficient"] - 1.5) < 1e-6 and abs(s["baseline_pw_coefficient"] - 2.5) < 1e-6
    pgn = t.pgn("new", "old")
    rounds = re.findall(r'\[Round "(\d+)"\]', pgn)
    assert rounds == ["1", "2", "3", "4"]
    assert re.findall(r'\[WhiteTeam "(\w+)"\]', pgn) == ["new", "old", "new", "old"]
    first = pgn.split("\n\n")[1]
    assert re.match(r"1\. \((pass|[a-h][1-8][a-h][1-8][nbrq]?|[PNBRQ]@[a-h][1-8]),(pass|[a-h][1-8][a-h][1-8][nbrq]?|[PNBRQ]@[a-h][1-8])\) ", first), first
    # determinism, and independence of a game from the games before it (no shared RNG): games 3-4 of this run are
    # games 3-4 of a run that only differs in the node count of... nothing: same config replays bit for bit
    t2 = O.TournamentOracle(cfg)
    t2.run()
    assert t2.pgn("new", "old") == pgn and t2.summary("new", "old") == t.summary("new", "old")
    # different PW coefficients really reach the searches: equal coefficients play a different tournament
    t3 = O.TournamentOracle(O.tournament_cfg(games=4, nodes=40, max_macro_plies=24, seed=9))
    t3.run()
    assert t3.pgn("new", "old") != pgn
