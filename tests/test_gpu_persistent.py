"""Single-launch search (hm_sp_search: the game and evaluator workgroups of k_rollout joined by the device-side queue of hm_queue.hpp) against the
host-driven lockstep loop (hm_sp_collect || forward -> hm_sp_process), which the other suites pin to the oracle: per game the
order of tree operations is the same, so root edge lists / visits / priors / Q, node counts, collision counters and whole
self-play records must be IDENTICAL — whatever the relative timing of games and evaluator workgroups in a given run.
(searchthread.cc:661-739, agent.cc:331-352 are the reference loop both forms implement.)"""
import numpy as np
import pytest
import torch

import oracle_py as O

pytestmark = pytest.mark.gpu


def _net(seed=0):
    from hivemind_amd import net as N
    torch.manual_seed(seed)
    return N.FusedNet(N.rise_v3_small())


def _stats_equal(a, b, G):
    for g in range(G):
        n = a["counts"][g]
        assert n == b["counts"][g], (g, n, b["counts"][g])
        for k in ("move_a", "move_b", "visits"):
            assert np.array_equal(a[k][g, :n], b[k][g, :n]), (g, k)
        for k in ("q", "prior"):                                             # bit patterns, not tolerances
            assert np.array_equal(a[k][g, :n].view(np.uint32), b[k][g, :n].view(np.uint32)), (g, k)
        assert a["root_q"][g].tobytes() == b["root_q"][g].tobytes(), g
        # status, nodes, eval rows, both collision counters, node count, root type, root visits, overflow, max depth,
        # nodes visited / edges scanned by selection, best move, move-list words, best child's type / end
        assert np.array_equal(a["info"][g, :16], b["info"][g, :16]), (g, a["info"][g], b["info"][g])


@pytest.mark.parametrize("G,nodes,noise", [(24, 400, False), (48, 400, True), (64, 100, True), (8, 37, False)])
def test_persistent_search_equals_lockstep_search(hm, G, nodes, noise):
    net = _net()
    roots = O.random_positions(900 + G, G * 13, 140)[::13][:G].copy()
    roots[0] = O.Board().compact(0, False)[0]
    seeds = (np.arange(G, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(777)
    alpha, eps = (0.3, 0.25) if noise else (0.0, 0.0)
    eng = hm.SearchEngine(G, 421)
    assert eng.search_consumers() > 0
    eng.set_games(roots)
    eng.begin_search(nodes, seeds, alpha, eps)
    eng.run(net)
    want = eng.root_stats()
    for rep in range(2):                                                     # timing differs from run to run; results must not
        eng.set_games(roots)
        eng.begin_search(nodes, seeds, alpha, eps)
        ms = eng.search_persistent(net)
        assert ms > 0.0
        got = eng.root_stats()
        _stats_equal(want, got, G)
    assert int((want["info"][:, 0] == 3).sum()) >= G // 2                    # most roots really searched (ST_DONE)
    eng.close()


def test_persistent_search_with_masked_and_dead_slots(hm):
    """slots that do not search (mask) leave the queue protocol intact: their workgroups only sign off"""
    net = _net()
    G = 16
    roots = O.random_positions(31, G * 7, 60)[::7][:G].copy()
    mask = (np.arange(G) % 3 != 0).astype(np.uint8)
    eng = hm.SearchEngine(G, 200)
    eng.set_games(roots)
    eng.begin_search(120, None, 0.0, 0.0, mask)
    eng.run(net)
    want = eng.root_stats()
    eng.set_games(roots)
    eng.begin_search(120, None, 0.0, 0.0, mask)
    eng.search_persistent(net)
    _stats_equal(want, eng.root_stats(), G)
    # no slot searching at all: both kernels start and leave
    eng.begin_search(120, None, 0.0, 0.0, np.zeros(G, np.uint8))
    eng.search_persistent(net)
    eng.close()


def test_persistent_search_without_the_lds_node_mirror(hm, monkeypatch):
    """pools too large for LDS walk their nodes in HBM (BASELINE configs[4]: nodes = 1600); same results"""
    net = _net()
    G = 6
    roots = O.random_positions(5, G * 9, 80)[::9][:G].copy()
    eng = hm.SearchEngine(G, 1700)
    eng.set_games(roots)
    eng.begin_search(1600, None, 0.3, 0.25)
    eng.run(net)
    want = eng.root_stats()
    eng.set_games(roots)
    eng.begin_search(1600, None, 0.3, 0.25)
    eng.search_persistent(net)
    _stats_equal(want, eng.root_stats(), G)
    eng.close()


def _selfplay(hm, net, **kw):
    sp = hm.SelfPlay(hm.default_selfplay_config(**kw), net)
    res = sp.run()
    rec, cnt = sp.records()
    sp.close()
    return res, rec.tobytes(), cnt


@pytest.mark.parametrize("kw", [dict(games=16, nodes=100, seed=3, concurrent_games=16, max_macro_plies=80),
                                dict(games=12, nodes=48, seed=11, concurrent_games=6, max_macro_plies=60)])
def test_selfplay_records_persistent_equal_lockstep(hm, monkeypatch, kw):
    net = _net()
    monkeypatch.setenv("HM_SELFPLAY_LOCKSTEP", "1")
    res_l, rec_l, cnt_l = _selfplay(hm, net, **kw)
    assert res_l.persistent_searches == 0
    monkeypatch.delenv("HM_SELFPLAY_LOCKSTEP")
    res_p, rec_p, cnt_p = _selfplay(hm, net, **kw)
    assert res_p.persistent_searches > 0 and res_p.search_kernel_ms > 0
    assert cnt_p == cnt_l and rec_p == rec_l
    assert (res_p.samples, res_p.total_nodes, res_p.eval_rows, res_p.nodes_visited, res_p.edges_scanned) == \
           (res_l.samples, res_l.total_nodes, res_l.eval_rows, res_l.nodes_visited, res_l.edges_scanned)


def test_selfplay_configs2_full_size_persistent_equals_lockstep(hm, monkeypatch):
    """BASELINE configs[2] at full size (64 games, nodes 400, RISEv3-small): byte-identical records from both loops"""
    net = _net()
    kw = dict(games=64, nodes=400, seed=1, concurrent_games=64)
    res_p, rec_p, cnt_p = _selfplay(hm, net, **kw)
    monkeypatch.setenv("HM_SELFPLAY_LOCKSTEP", "1")
    res_l, rec_l, cnt_l = _selfplay(hm, net, **kw)
    assert res_p.persistent_searches > 0 and res_l.persistent_searches == 0
    assert cnt_p == cnt_l and rec_p == rec_l


def test_stalled_persistent_search_is_repeated_with_the_same_records(hm, monkeypatch):
    """Recovery path of a persistent search given up as stalled (hm_queue.hpp: no row published for 30 ms beside searching games;
    seen on MI355X about once in a few thousand searches with the deployed network): hm_sp_search_stalled -> hm_sp_begin_again ->
    the same search on the lockstep loop.  The test hook reports every third completed persistent search as stalled; the records
    must not change, and the following searches are persistent ones again."""
    net = _net()
    kw = dict(games=12, nodes=64, seed=5, concurrent_games=12, max_macro_plies=60)
    res_a, rec_a, cnt_a = _selfplay(hm, net, **kw)
    assert res_a.persistent_searches > 3 and res_a.persistent_stalls == 0
    monkeypatch.setenv("HM_SEARCH_FAKE_STALL_EVERY", "3")
    res_b, rec_b, cnt_b = _selfplay(hm, net, **kw)
    monkeypatch.delenv("HM_SEARCH_FAKE_STALL_EVERY")
    assert res_b.persistent_stalls > 0 and res_b.persistent_searches > 0
    assert res_b.persistent_stalls + res_b.persistent_searches == res_a.persistent_searches
    assert cnt_b == cnt_a and rec_b == rec_a
    assert (res_b.samples, res_b.total_nodes) == (res_a.samples, res_a.total_nodes)


@pytest.mark.parametrize("per_wg", [1, 3])
def test_search_abandoned_half_way_is_repeated_with_the_same_records(hm, monkeypatch, per_wg):
    """Recovery from a search the hang guard gives up MID-WAY: the hook makes an evaluator workgroup raise "stalled" once
    150 rows of every third search have been published — games are left in any phase, trees / transposition tables / game records
    half-built, rows published to a queue nobody serves any more.  hm_sp_begin_again + the lockstep loop must produce the records of
    an undisturbed run, for one game per search workgroup (search_role, node pool in LDS) and for several (search_role_mg)."""
    net = _net()
    monkeypatch.setenv("HM_SEARCH_GAMES_PER_WG", str(per_wg))
    kw = dict(games=16, nodes=100, seed=9, concurrent_games=16, max_macro_plies=50)
    res_a, rec_a, cnt_a = _selfplay(hm, net, **kw)
    assert res_a.persistent_searches > 6 and res_a.persistent_stalls == 0
    monkeypatch.setenv("HM_SEARCH_ABORT_EVERY", "3")
    monkeypatch.setenv("HM_SEARCH_ABORT_AFTER", "150")
    res_b, rec_b, cnt_b = _selfplay(hm, net, **kw)
    monkeypatch.delenv("HM_SEARCH_ABORT_EVERY")
    assert res_b.persistent_stalls > 0 and res_b.persistent_searches > 0
    assert cnt_b == cnt_a and rec_b == rec_a
    assert (res_b.samples, res_b.total_nodes) == (res_a.samples, res_a.total_nodes)
    # the direct call recovers the same way (SearchEngine.search_persistent)
    G = 12
    roots = O.random_positions(77, G * 5, 100)[::5][:G].copy()
    eng = hm.SearchEngine(G, 421)
    eng.set_games(roots)
    eng.begin_search(300, None, 0.0, 0.0)
    eng.run(net)
    want = eng.root_stats()
    monkeypatch.setenv("HM_SEARCH_ABORT_EVERY", "1")
    eng.set_games(roots)
    eng.begin_search(300, None, 0.0, 0.0)
    eng.search_persistent(net)
    monkeypatch.delenv("HM_SEARCH_ABORT_EVERY")
    assert getattr(eng, "stalls", 0) == 1
    _stats_equal(want, eng.root_stats(), G)
    eng.close()


@pytest.mark.parametrize("per_wg", [2, 4])
def test_several_games_per_search_workgroup_equal_lockstep(hm, monkeypatch, per_wg):
    """k_search_mg (the form hm_sp_search takes for a slow evaluator — the deployed 384-channel network —: one search workgroup serves
    several games in turn, tree walked in place, the freed CUs go to the evaluator), forced here onto the small network: root statistics
    of every game and whole self-play records equal the lockstep loop's."""
    monkeypatch.setenv("HM_SEARCH_GAMES_PER_WG", str(per_wg))
    net = _net()
    G, nodes = 30, 200
    roots = O.random_positions(4242, G * 7, 120)[::7][:G].copy()
    seeds = (np.arange(G, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(99)
    eng = hm.SearchEngine(G, 221)
    eng.set_games(roots)
    eng.begin_search(nodes, seeds, 0.3, 0.25)
    eng.run(net)
    want = eng.root_stats()
    eng.set_games(roots)
    eng.begin_search(nodes, seeds, 0.3, 0.25)
    assert eng.search_persistent(net) > 0.0
    _stats_equal(want, eng.root_stats(), G)
    eng.close()
    kw = dict(games=10, nodes=64, seed=21, concurrent_games=10, max_macro_plies=50)
    res_p, rec_p, cnt_p = _selfplay(hm, net, **kw)
    monkeypatch.setenv("HM_SELFPLAY_LOCKSTEP", "1")
    res_l, rec_l, cnt_l = _selfplay(hm, net, **kw)
    assert res_p.persistent_searches > 0 and res_l.persistent_searches == 0
    assert cnt_p == cnt_l and rec_p == rec_l


def test_single_launch_search_with_other_batch_sizes_equals_lockstep(hm):
    """hm_sp_set_batch_sizes (Engine::getBatchSize() leaves per iteration) through k_rollout: same results as the lockstep kernels"""
    net = _net()
    G = 10
    roots = O.random_positions(321, G * 9, 110)[::9][:G].copy()
    sizes = np.array([4, 8, 2, 6, 4, 1, 8, 5, 3, 7], np.uint8)
    eng = hm.SearchEngine(G, 421)
    eng.set_batch_sizes(sizes)
    eng.set_games(roots)
    eng.begin_search(300, None, 0.3, 0.25)
    eng.run(net)
    want = eng.root_stats()
    eng.set_games(roots)
    eng.begin_search(300, None, 0.3, 0.25)
    assert eng.search_persistent(net) > 0.0
    _stats_equal(want, eng.root_stats(), G)
    eng.close()


@pytest.mark.parametrize("G", [300, 600])
def test_many_game_slots_share_search_workgroups(hm, G):
    """more game slots than a quarter of the CUs: rollout_plan lets the live games share search workgroups (search_role_mg, up to 8 per
    workgroup) so that the launch stays resident at once; results equal the lockstep loop's.  With a mask that leaves few games alive
    the same engine goes back to one game per workgroup."""
    net = _net()
    roots = O.random_positions(808, G * 3, 100)[::3][:G].copy()
    eng = hm.SearchEngine(G, 100)
    assert eng.search_consumers() > 0
    for mask in (None, (np.arange(G) % 11 == 0).astype(np.uint8)):
        eng.set_games(roots)
        eng.begin_search(48, None, 0.0, 0.0, mask)
        eng.run(net)
        want = eng.root_stats()
        eng.set_games(roots)
        eng.begin_search(48, None, 0.0, 0.0, mask)
        assert eng.search_persistent(net) > 0.0
        assert getattr(eng, "stalls", 0) == 0
        _stats_equal(want, eng.root_stats(), G)
    eng.close()


def test_node_pool_of_the_headline_budget_fits_lds(hm):
    """BASELINE configs[2] / [3] (nodes 400: the self-play driver sizes the pool for 400 * 1.05 + 1 = 421 nodes): k_search must be able to keep the pool
    in LDS beside its static LDS.  A regression guard: one more LDS array in the kernel — or a helper the compiler stops inlining, whose
    LDS arguments then get allocated in every kernel that reaches it — silently sends the traversal back to global memory (-20 %)."""
    eng = hm.SearchEngine(64, 421)
    assert eng.search_lds_tree()
    eng.close()
