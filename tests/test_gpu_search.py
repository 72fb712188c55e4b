"""GPU search parity: the wave-per-game MCGS (hm_sp_*) against the CPU oracle search
(oracle/search.hpp, tie_mode=1 / exp_mode=1) under the shared deterministic hash evaluator:
root edge lists (moves, visit counts, priors, Q), node counts and collision counters must be
identical for every root."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


def _hash_eval_gpu(planes):
    import torch
    h = planes.cpu().numpy().view(np.uint16).reshape(-1, 4736)
    v, a, b, w, m = O.hash_evaluator(h)
    tc = lambda x: torch.from_numpy(x.view(np.float16)).cuda()
    return tc(v), tc(a), tc(b), tc(w), tc(m)


def test_rules_probe_matches_oracle(hm):
    boards = np.concatenate([O.random_positions(21, 1500, 200), O.random_positions(22, 500, 40)])
    out, keys = hm.rules_probe(hm.to_device(boards))
    b = O.Board()
    for i in range(len(boards)):
        b.from_compact(boards[i:i + 1])
        want = [b.is_checkmate(0, False), b.is_checkmate(0, True), b.is_checkmate(1, False), b.is_checkmate(1, True),
                b.in_check(0), b.in_check(1)]
        assert [bool(x) for x in out[i, :6]] == want, i
        assert int(keys[i, 0]) == b.hash_key(False) and int(keys[i, 1]) == b.hash_key(True), i
        assert int(keys[i, 2]) == int(O.lib.ora_rep_key(b.h, 0)) and int(keys[i, 3]) == int(O.lib.ora_rep_key(b.h, 1))
        # classify_terminal_position at search ply 1 for either team to play (incl. the waiting-board mate rule)
        team, adv = int(boards["team"][i]), int(boards["time_adv"][i])
        assert int(out[i, 6]) == int(O.lib.ora_classify(b.h, team, team, adv, 1)), (i, "classify own")
        assert int(out[i, 7]) == int(O.lib.ora_classify(b.h, team ^ 1, team, adv, 1)), (i, "classify other")


def _roots(n, seed):
    boards = O.random_positions(seed, n * 37, 90)[::37][:n].copy()
    return boards


@pytest.mark.parametrize("nodes,noise", [(400, False), (400, True), (100, False), (1600, True), (200, "sweep")])
def test_search_matches_oracle(hm, nodes, noise):
    G = 24 if nodes <= 400 else 8           # BASELINE configs[4]: nodes=1600, transposition-sharing MCGS + Dirichlet noise
    if noise == "sweep":                    # many more roots (late-game positions included): rare rules get exercised
        G, noise = 128, True
        roots = O.random_positions(4242, G * 11, 160)[::11][:G].copy()
    else:
        roots = _roots(G, 77 + nodes)
    roots[0] = O.Board().compact(0, False)[0]
    eng = hm.SearchEngine(G, 1700)
    eng.set_games(roots)
    seeds = (np.arange(G, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(12345)
    alpha, eps = (0.3, 0.25) if noise else (0.0, 0.0)
    eng.begin_search(nodes, seeds, alpha, eps)
    eng.run(_hash_eval_gpu)
    st = eng.root_stats()
    exact = 0
    for g in range(G):
        b = O.Board()
        b.from_compact(roots[g:g + 1])
        s = O.Search(1, 1)
        if noise:
            s.set_noise(alpha, eps, int(seeds[g]))
        ok = s.run(b, int(roots["team"][g]), bool(roots["time_adv"][g]), nodes)
        info = st["info"][g]
        if not ok:
            assert info[0] == 4, (g, info)          # ST_NOACTION: both sides say "bestmove (none)"
            exact += 1
            continue
        e = s.edges()
        n = st["counts"][g]
        oi = s.info()
        assert info[8] == 0, ("overflow", g, info)
        same = (n == len(e["visits"]) and np.array_equal(st["move_a"][g, :n], e["move_a"])
                and np.array_equal(st["move_b"][g, :n], e["move_b"]) and np.array_equal(st["visits"][g, :n], e["visits"])
                and np.array_equal(st["prior"][g, :n], e["prior"]) and np.array_equal(st["q"][g, :n], e["q"])
                and info[1] == oi["nodes"] and info[2] == oi["eval_rows"] and info[5] == oi["node_count"])
        assert same, (g, n, len(e["visits"]), info, oi, st["visits"][g, :n], e["visits"])
        assert st["root_q"][g] == np.float32(s.root_q())
        assert info[12] == s.best_move(), (g, info[12], s.best_move())      # Agent::run_search's returned action (agent.cc:859-889)
        exact += 1
    assert exact == G
    eng.close()


def test_collect_row_counts(hm):
    import torch
    """hm_sp_collect_counted: per-game batch sizes add up to the evaluator rows the search reports."""
    G = 8
    roots = _roots(G, 901)
    eng = hm.SearchEngine(G, 500)
    eng.set_games(roots)
    eng.begin_search(200)
    rows = torch.zeros(G, dtype=torch.int32, device="cuda")
    total = np.zeros(G, np.int64)
    iters = 0
    eng.leg_times(reset=True)
    for _ in range(2000):
        planes = eng.collect(rows_next=rows)
        iters += 1
        r = rows.cpu().numpy()
        assert ((0 <= r) & (r <= 8)).all()
        total += r
        if eng.process(*_hash_eval_gpu(planes)) == 0:
            break
    st = eng.root_stats()
    # leg clock: one k_collect and one k_process interval per iteration, each a plausible launch duration; no timed forward here
    ms, cnt = eng.leg_times(reset=True)
    assert cnt[0] == cnt[2] == iters and cnt[1] == 0, (cnt, iters)
    assert 0.005 * iters < ms[0] < 50.0 * iters and 0.005 * iters < ms[2] < 50.0 * iters and ms[1] == 0.0, (ms, iters)
    assert eng.leg_times()[1].sum() == 0
    # rows written after a game's last processed batch (aborted lookahead) are counted by collect but not evaluated
    assert (total >= st["info"][:, 2]).all() and (total - st["info"][:, 2] <= 8).all(), (total, st["info"][:, 2])


def test_search_with_non_finite_logits_matches_oracle(hm):
    """normalize_logits' fallbacks inside the search (utils.h:127-167; reference known answers test_move_gen.cc:643-657 pin
    the oracle): the evaluator poisons a share of the policy / value / wdl outputs with NaN and infinities; GPU expansion
    (lane-parallel masked softmax, shape_value) must still equal the oracle bit for bit."""
    import torch

    def by_content(h):
        v, a, b, w, m = O.hash_evaluator(h)
        v, a, b, w, m = v.copy(), a.copy(), b.copy(), w.copy(), m.copy()
        for r in range(len(h)):
            k = int(h[r].astype(np.uint64).sum() % 8)
            if k == 0:
                a[r, :] = 0x7E00
            elif k == 1:
                a[r, 0::3] = 0xFC00; b[r, 1::5] = 0x7E00
            elif k == 2:
                b[r, :] = 0xFC00; w[r, 1] = 0x7E00
            elif k == 3:
                v[r] = 0x7C00; a[r, 2::7] = 0x7C00
        return v, a, b, w, m
    G = 16
    roots = _roots(G, 4711)
    eng = hm.SearchEngine(G, 400)
    eng.set_games(roots)
    eng.begin_search(200)
    eng.run(lambda planes: tuple(torch.from_numpy(x.view(np.float16)).cuda() for x in by_content(planes.cpu().numpy().view(np.uint16).reshape(-1, 4736))))
    st = eng.root_stats()
    for g in range(G):
        b = O.Board()
        b.from_compact(roots[g:g + 1])
        s = O.Search(1, 1)
        s.set_evaluator(by_content)
        ok = s.run(b, int(roots["team"][g]), bool(roots["time_adv"][g]), 200)
        if not ok:
            assert st["info"][g][0] == 4
            continue
        e = s.edges()
        n = st["counts"][g]
        assert n == len(e["visits"]) and np.array_equal(st["visits"][g, :n], e["visits"]), (g, st["visits"][g, :n], e["visits"])
        assert np.array_equal(st["prior"][g, :n], e["prior"]) and np.array_equal(st["q"][g, :n], e["q"]), g
    eng.close()


@pytest.mark.parametrize("nodes", [560, 575, 600])
def test_search_at_the_lds_mirror_boundary(hm, nodes):
    """node pools just below / above what fits beside k_collect's static LDS (the mirror is switched off above): the search
    runs and matches the oracle on both sides of the boundary"""
    roots = _roots(4, 5)
    eng = hm.SearchEngine(4, nodes)
    eng.set_games(roots)
    eng.begin_search(nodes)
    eng.run(_hash_eval_gpu)
    st = eng.root_stats()
    for g in range(4):
        b = O.Board()
        b.from_compact(roots[g:g + 1])
        s = O.Search(1, 1)
        if not s.run(b, int(roots["team"][g]), bool(roots["time_adv"][g]), nodes):
            continue
        e = s.edges()
        n = st["counts"][g]
        assert st["info"][g][8] == 0 and n == len(e["visits"]) and np.array_equal(st["visits"][g, :n], e["visits"]), g


# Blocked-pawn king endings: two or three legal moves per board, so a few hundred nodes reach depth 6+ and the same joint position
# comes up along several move orders (5-36 transposition-table hits per search, up to 1 300 same-batch collisions and 140
# reservation collisions) — random middle-game roots give none at these budgets.  Exercises canonicalize_child's replace_child /
# pending-evaluation outcomes (searchthread.cc:741-806): on the GPU the descent that ran ahead is cancelled, the edge updates are
# applied by the traversal wave and the handed-back descent continues (collect_batch).
_TRANSPOSING = [("k7/p7/P7/8/8/8/8/7K[] w - - 0 1", "7k/8/8/8/8/p7/P7/K7[] b - - 0 1"),
                ("k7/p7/P7/8/8/7p/7P/7K[] w - - 0 1", "k7/p7/P7/8/8/7p/7P/7K[] b - - 0 1"),
                ("k7/p7/P7/8/8/7p/7P/7K[] w - - 0 1", "k7/p7/P7/8/8/7p/7P/7K[] w - - 0 1")]


def _transposing_roots():
    rows = []
    for fa, fb in _TRANSPOSING:
        for adv in (False, True):
            for team in (0, 1):
                b = O.Board()
                b.set_fen(0, fa)
                b.set_fen(1, fb)
                rows.append(b.compact(team, adv))
    return np.concatenate(rows)


@pytest.mark.parametrize("nodes", [400, 1600])
def test_search_with_many_transpositions_matches_oracle(hm, nodes):
    roots = _transposing_roots()
    G = len(roots)
    eng = hm.SearchEngine(G, 1700)
    eng.set_games(roots)
    eng.begin_search(nodes, None, 0.0, 0.0)
    eng.run(_hash_eval_gpu)
    st = eng.root_stats()
    hits = collisions = searched = 0
    for g in range(G):
        b = O.Board()
        b.from_compact(roots[g:g + 1])
        s = O.Search(1, 1)
        ok = s.run(b, int(roots["team"][g]), bool(roots["time_adv"][g]), nodes)
        info = st["info"][g]
        if not ok:
            assert info[0] == 4, (g, info)
            continue
        searched += 1
        e, oi, n = s.edges(), s.info(), st["counts"][g]
        assert info[8] == 0, ("overflow", g, info)
        assert n == len(e["visits"]), (g, n, len(e["visits"]))
        for k in ("move_a", "move_b", "visits", "prior", "q"):
            assert np.array_equal(st[k][g, :n], e[k]), (g, k, st[k][g, :n], e[k])
        assert (info[1], info[2], info[3], info[4], info[5]) == (oi["nodes"], oi["eval_rows"], oi["same_batch"], oi["reservation"], oi["node_count"]), (g, info, oi)
        assert info[18] == s.tt_hits(), (g, info[18], s.tt_hits())
        assert st["root_q"][g] == np.float32(s.root_q())
        assert info[12] == s.best_move()
        hits += int(info[18]); collisions += int(info[3]) + int(info[4])
    assert searched >= 8 and hits >= (40 if nodes == 400 else 150) and collisions > 500, (searched, hits, collisions)
    eng.close()


def test_persistent_search_with_many_transpositions_equals_lockstep(hm):
    """the same roots through k_search + rise_serve (creation on the classifier wave, hand-backs, deferred edge updates) and through
    the lockstep kernels, with a network as evaluator: identical root statistics and counters, transposition hits included"""
    import torch
    from hivemind_amd import net as N
    torch.manual_seed(0)
    net = N.FusedNet(N.rise_v3_small())
    roots = _transposing_roots()
    G = len(roots)
    for nodes, cap in ((400, 421), (1600, 1700)):                   # LDS tree / tree walked in place
        eng = hm.SearchEngine(G, cap)
        eng.set_games(roots)
        eng.begin_search(nodes, None, 0.0, 0.0)
        eng.run(net)
        want = eng.root_stats()
        eng.set_games(roots)
        eng.begin_search(nodes, None, 0.0, 0.0)
        assert eng.search_persistent(net) > 0.0
        got = eng.root_stats()
        for g in range(G):
            n = want["counts"][g]
            assert n == got["counts"][g], g
            for k in ("move_a", "move_b", "visits"):
                assert np.array_equal(want[k][g, :n], got[k][g, :n]), (g, k)
            for k in ("q", "prior"):
                assert np.array_equal(want[k][g, :n].view(np.uint32), got[k][g, :n].view(np.uint32)), (g, k)
            assert np.array_equal(want["info"][g, :20], got["info"][g, :20]), (g, want["info"][g], got["info"][g])
        assert int(want["info"][:, 18].sum()) > 0
        eng.close()


@pytest.mark.parametrize("batch", [1, 3, 4, 7])
def test_search_with_other_batch_sizes_matches_oracle(hm, batch):
    """Engine::getBatchSize() leaves per iteration (searchthread.cc:258-273, 663) instead of the default 8: hm_sp_set_batch_sizes against
    the oracle with the same cfg.batchSize — lockstep kernels under the hash evaluator; slots with different sizes in one engine."""
    G = 12
    roots = _roots(G, 4100 + batch)
    sizes = np.full(G, batch, np.uint8)
    sizes[1::3] = 8                                                          # mixed: every third slot keeps the default
    eng = hm.SearchEngine(G, 500)
    eng.set_games(roots)
    eng.set_batch_sizes(sizes)
    eng.begin_search(200, None, 0.0, 0.0)
    eng.run(_hash_eval_gpu)
    st = eng.root_stats()
    searched = 0
    for g in range(G):
        b = O.Board()
        b.from_compact(roots[g:g + 1])
        s = O.Search(1, 1)
        s.set_batch_size(int(sizes[g]))
        if not s.run(b, int(roots["team"][g]), bool(roots["time_adv"][g]), 200):
            assert st["info"][g][0] == 4
            continue
        searched += 1
        e, oi, n = s.edges(), s.info(), st["counts"][g]
        assert n == len(e["visits"]), (g, n, len(e["visits"]))
        for k in ("move_a", "move_b", "visits", "prior", "q"):
            assert np.array_equal(st[k][g, :n], e[k]), (g, k)
        info = st["info"][g]
        assert (info[1], info[2], info[3], info[4], info[5]) == (oi["nodes"], oi["eval_rows"], oi["same_batch"], oi["reservation"], oi["node_count"]), (g, info, oi)
        assert 200 <= oi["nodes"] < 200 + 2 * int(sizes[g])                  # the budget is overshot by less than two batches of this size
    assert searched >= 8
    eng.set_batch_sizes(None)                                                # back to 8 everywhere
    eng.close()
