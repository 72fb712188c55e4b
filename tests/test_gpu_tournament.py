"""GPU tournament driver (hm_tournament_*, tools/tournament.cc:328-465) against the sequential restatement
(oracle/tournament.hpp): with the same two deterministic stand-in networks, summary.json and games.pgn are identical
byte for byte — games in lockstep slots, finished out of order, reported in game order."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_py as O

pytestmark = pytest.mark.gpu

SALT_C, SALT_B = 0, 0x5EED


@pytest.fixture(scope="module")
def hm():
    import hivemind_amd as hm
    hm.init(0)
    return hm


def stand_in_networks(planes, acting):
    """two salted hash evaluators; every slot's 8 rows are served by the network of the team to move there"""
    h = planes.cpu().numpy().view(np.uint16).reshape(-1, 4736)
    outs = [O.hash_evaluator_salted(h, SALT_C), O.hash_evaluator_salted(h, SALT_B)]
    rows = np.repeat(acting.astype(bool), 8)[:len(h)]
    pick = [np.where(rows.reshape((-1,) + (1,) * (c.ndim - 1)), c, b) for c, b in zip(outs[0], outs[1])]
    return tuple(torch.from_numpy(np.ascontiguousarray(x).view(np.float16)).cuda() for x in pick)


@pytest.mark.parametrize("kw", [dict(games=6, nodes=48, max_macro_plies=30, seed=3, concurrent_games=4),
                                dict(games=4, nodes=64, max_macro_plies=20, seed=11, concurrent_games=8,
                                     contender_pw_coefficient=1.25, baseline_pw_coefficient=3.0, dirichlet_epsilon=0.25),
                                # per-side Engine batch sizes (tournament.h:19-20): each network's searches collect that many leaves per iteration
                                dict(games=4, nodes=48, max_macro_plies=20, seed=5, concurrent_games=4, contender_batch_size=4, baseline_batch_size=6)],
                         ids=["6x48", "4x64-pw", "4x48-batch4v6"])
def test_tournament_reports_match_oracle_bytes(hm, kw):
    t = hm.Tournament(hm.default_tournament_config(**kw), evaluator=stand_in_networks)
    res = t.run()
    ora = O.TournamentOracle(O.tournament_cfg(**kw), SALT_C, SALT_B)
    ora.run()
    assert t.pgn("new", "old") == ora.pgn("new", "old")
    assert t.summary("new", "old") == ora.summary("new", "old")
    s = json.loads(t.summary("new", "old"))
    assert res.games == kw["games"] == s["games"] and res.contender_wins == s["contender_wins"] and res.draws == s["draws"]
    assert res.pairs == kw["games"] // 2 and len(t.pair_scores()) == res.pairs
    assert res.searched_positions > 0 and res.total_nodes >= res.searched_positions * kw["nodes"] * 0.9
    t.close()


def test_tournament_native_networks_and_report_files(hm, tmp_path):
    """two FusedNet networks evaluated inside the captured iteration graph (one forward per network over its slots'
    rows); report files as the reference writes them; the run is deterministic"""
    from hivemind_amd import net as N
    torch.manual_seed(0)
    a = N.FusedNet(N.rise_v3_small())
    torch.manual_seed(1)
    b = N.FusedNet(N.rise_v3_small())
    cfg = dict(games=4, nodes=48, max_macro_plies=16, seed=5, concurrent_games=4)
    texts = []
    for _ in range(2):
        t = hm.Tournament(hm.default_tournament_config(**cfg), contender=a, baseline=b)
        res = t.run()
        assert res.games == 4 and res.macro_ply_limits + res.checkmates + res.drawn_terminations + res.no_legal_actions == 4
        texts.append((t.pgn("a", "b"), t.summary("a", "b")))
        t.write_reports(tmp_path / "tournament_results", "a", "b")
        t.close()
    assert texts[0] == texts[1]
    assert (tmp_path / "tournament_results" / "games.pgn").read_text() == texts[0][0]
    assert json.loads((tmp_path / "tournament_results" / "summary.json").read_text())["games"] == 4
    # the same network on both sides is a different tournament from two different networks (the baseline really plays)
    t = hm.Tournament(hm.default_tournament_config(**cfg), contender=a, baseline=a)
    t.run()
    assert t.pgn("a", "b") != texts[0][0]
    t.close()


def test_tournament_with_a_movetime_limit(hm):
    """TournamentConfig::moveTimeMs instead of nodes (tournament.h:17-18): every search runs under its slot's controller of the
    reference's polling loop (agent.cc:715-806).  Wall-clock dependent, so the checks are behavioural: searches last about the move
    time (early stopping may cut them short, two extensions of 1.5x may lengthen them), every game ends, the reports are well formed,
    far more nodes per move than a handful, and no search outgrows its pool."""
    import time
    from hivemind_amd import net as N
    torch.manual_seed(0)
    a = N.FusedNet(N.rise_v3_small())
    torch.manual_seed(1)
    b = N.FusedNet(N.rise_v3_small())
    move_ms, plies = 30, 8
    cfg = hm.default_tournament_config(games=4, nodes=0, move_time_ms=move_ms, max_macro_plies=plies, seed=9, concurrent_games=4, max_search_nodes=3000)
    t = hm.Tournament(cfg, contender=a, baseline=b)
    t0 = time.time()
    res = t.run()
    dt = time.time() - t0
    assert res.games == 4 and res.macro_ply_limits + res.checkmates + res.drawn_terminations + res.no_legal_actions == 4
    searches = res.searched_positions
    assert 4 <= searches <= 4 * plies
    # all four games search in lockstep: at most `plies` rounds, each between a fraction of the move time and 2.25x + slack
    assert dt < plies * (move_ms * 2.25 + 40) * 1e-3 + 2.0, dt
    assert res.total_nodes / searches > 20, (res.total_nodes, searches)
    assert res.total_nodes / searches <= 3000 * 1.05 + 16
    s = json.loads(t.summary("a", "b"))
    assert s["move_time_ms"] == move_ms and s["nodes_per_move"] == 0 and s["games"] == 4
    assert t.pgn("a", "b").count("[Event ") == 4
    t.close()
    # a node budget and a move time together are refused (tournament.cc:338-341)
    with pytest.raises(hm.HivemindError, match="exactly one positive nodes or movetime"):
        hm.Tournament(hm.default_tournament_config(games=2, nodes=100, move_time_ms=10), contender=a, baseline=b)


def test_movetime_is_enforced_without_the_step_graph(hm, monkeypatch):
    """The eager lockstep loop (no captured graph: HM_SELFPLAY_NO_GRAPH, or a failed capture) must poll the slots' time controllers
    too: with a pool of 60 000 nodes per search, a search that ignored the clock would run ~150 x the move time."""
    import time
    from hivemind_amd import net as N
    monkeypatch.setenv("HM_SELFPLAY_NO_GRAPH", "1")
    torch.manual_seed(0)
    a = N.FusedNet(N.rise_v3_small())
    move_ms, plies = 40, 4
    cfg = hm.default_tournament_config(games=2, nodes=0, move_time_ms=move_ms, max_macro_plies=plies, seed=5, concurrent_games=2, max_search_nodes=60000)
    t = hm.Tournament(cfg, contender=a, baseline=a)
    t0 = time.time()
    res = t.run()
    dt = time.time() - t0
    t.close()
    assert res.games == 2 and 2 <= res.searched_positions <= 2 * plies
    assert dt < plies * (move_ms * 2.25 + 60) * 1e-3 + 2.0, dt
    assert res.total_nodes / res.searched_positions < 30000
