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
                                     contender_pw_coefficient=1.25, baseline_pw_coefficient=3.0, dirichlet_epsilon=0.25)])
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
