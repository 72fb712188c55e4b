"""The TIMED path against the oracle, directly.

bench.py times hm_sp_search: the persistent search workgroups (k_search / k_search_mg) fed by the persistent evaluator
(the fused RISEv3 forward + the leaf's prior pipeline on the logits in LDS) through the device-side queue of hm_queue.hpp.
The other suites pin the LOCKSTEP kernels to the oracle under a callback evaluator and the persistent pair to lockstep; here
the persistent pair itself meets oracle/search.hpp (mode 1,1) and oracle/selfplay.hpp — whose evaluator callback is the same
GPU fused forward (hm_net_forward on the rows the oracle asks for), so both sides see bit-identical logits and everything
else (selection, widening, transpositions, priors, noise mix, backups, best move; RNG draw order, records) is the oracle's
own CPU code.  Reference semantics: searchthread.cc:255-639 (collect_batch / process_batch), agent.cc:331-352, selfplay.cc:558-748."""
import numpy as np
import pytest
import torch

import oracle_py as O
from test_gpu_search import _transposing_roots
from test_gpu_selfplay_parity import _split_by_game

pytestmark = pytest.mark.gpu


def _net(seed=0):
    from hivemind_amd import net as N
    torch.manual_seed(seed)
    return N.FusedNet(N.rise_v3_small())


def _oracle_eval(net):
    """evaluator callback of the oracle = the product's fused forward on the GPU (any row count; one workgroup per position, so a
    row's logits do not depend on what else is in the batch — tests/test_gpu_net.py)"""
    def ev(planes_u16):
        x = torch.from_numpy(np.ascontiguousarray(planes_u16).view(np.float16)).cuda().reshape(-1, 74, 8, 8)
        out = net(x)
        torch.cuda.synchronize()
        return tuple(t.cpu().numpy().view(np.uint16) for t in out)
    return ev


def _compare_with_oracle(eng, st, roots, net, nodes, noise, seeds, alpha, eps, tt=False):
    ev = _oracle_eval(net)
    searched = hits = 0
    for g in range(len(roots)):
        b = O.Board()
        b.from_compact(roots[g:g + 1])
        s = O.Search(1, 1)
        s.set_evaluator(ev)
        if noise:
            s.set_noise(alpha, eps, int(seeds[g]))
        ok = s.run(b, int(roots["team"][g]), bool(roots["time_adv"][g]), nodes)
        info = st["info"][g]
        if not ok:
            assert info[0] == 4, (g, info)                            # ST_NOACTION on both sides
            continue
        searched += 1
        e, oi, n = s.edges(), s.info(), st["counts"][g]
        assert info[8] == 0, ("overflow", g, info)
        assert n == len(e["visits"]), (g, n, len(e["visits"]))
        for k in ("move_a", "move_b", "visits"):
            assert np.array_equal(st[k][g, :n], e[k]), (g, k, st[k][g, :n], e[k])
        for k in ("prior", "q"):                                      # bit patterns
            assert np.array_equal(st[k][g, :n].view(np.uint32), e[k].view(np.uint32)), (g, k, st[k][g, :n], e[k])
        assert (info[1], info[2], info[3], info[4], info[5]) == (oi["nodes"], oi["eval_rows"], oi["same_batch"], oi["reservation"], oi["node_count"]), (g, info, oi)
        assert info[18] == s.tt_hits(), (g, info[18], s.tt_hits())
        assert st["root_q"][g] == np.float32(s.root_q())
        assert info[12] == s.best_move(), (g, info[12], s.best_move())
        hits += int(info[18])
    return searched, hits


@pytest.mark.parametrize("nodes,noise,cap", [(400, False, 421), (400, True, 421), (1600, True, 1700)],
                         ids=["400-ldstree", "400-noise-ldstree", "1600-noise-in-place"])
def test_persistent_search_matches_oracle_with_the_gpu_forward(hm, nodes, noise, cap):
    """24 random-playout roots + the 12 transposition-rich king endings: root edges / visits / priors / Q / counters /
    transposition hits of hm_sp_search (k_search<LDS_TREE> at 400 nodes, the tree walked in place at 1600) bit-equal to the oracle's."""
    net = _net()
    rnd = O.random_positions(77 + nodes, 24 * 37, 90)[::37][:24].copy()
    rnd[0] = O.Board().compact(0, False)[0]
    roots = np.concatenate([rnd, _transposing_roots()])
    G = len(roots)
    seeds = (np.arange(G, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(4711)
    alpha, eps = (0.3, 0.25) if noise else (0.0, 0.0)
    eng = hm.SearchEngine(G, cap)
    assert eng.search_consumers() > 0
    assert eng.search_lds_tree() == (cap <= 421)
    eng.set_games(roots)
    eng.begin_search(nodes, seeds, alpha, eps)
    assert eng.search_persistent(net) > 0.0
    st = eng.root_stats()
    searched, hits = _compare_with_oracle(eng, st, roots, net, nodes, noise, seeds, alpha, eps)
    assert searched >= 24 and hits >= 20, (searched, hits)
    eng.close()


def test_several_games_per_search_workgroup_match_oracle(hm, monkeypatch):
    """k_search_mg (the form the deployed network's searches take) forced onto the small network: same comparison"""
    monkeypatch.setenv("HM_SEARCH_GAMES_PER_WG", "3")
    net = _net()
    rnd = O.random_positions(515, 20 * 13, 120)[::13][:20].copy()
    roots = np.concatenate([rnd, _transposing_roots()[:6]])
    G = len(roots)
    seeds = (np.arange(G, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(31)
    eng = hm.SearchEngine(G, 421)
    eng.set_games(roots)
    eng.begin_search(400, seeds, 0.3, 0.25)
    assert eng.search_persistent(net) > 0.0
    searched, _ = _compare_with_oracle(eng, eng.root_stats(), roots, net, 400, True, seeds, 0.3, 0.25)
    assert searched >= 16, searched
    eng.close()


def test_selfplay_configs2_sampled_games_match_the_oracle_loop(hm, tmp_path):
    """BASELINE configs[2] at full size (64 games, nodes 400, RISEv3-small) through the persistent pair — the run bench.py times —:
    the HVM4 bytes of sampled games equal oracle/selfplay.hpp driven by the same GPU forward."""
    net = _net()
    kw = dict(games=64, nodes=400, seed=1, concurrent_games=64)
    sp = hm.SelfPlay(hm.default_selfplay_config(**kw), net)
    res = sp.run()
    rec, cnt = sp.records()
    sp.close()
    assert res.persistent_searches > 0
    got = _split_by_game(hm, tmp_path, rec, cnt, "gpu.hvm")
    ora = O.SelfPlayOracle(O.selfplay_cfg(**kw), 1, 1)
    ora.set_evaluator(_oracle_eval(net))
    lens = sorted((len(v), g) for g, v in got.items())
    sample = sorted({lens[0][1], lens[len(lens) // 2][1], lens[-1][1], 0, 37})       # shortest, median, longest game + two fixed ones
    for g in sample:
        want, info, _ = ora.game(g)
        assert got.get(g, b"") == want, (g, info, len(got.get(g, b"")), len(want))
