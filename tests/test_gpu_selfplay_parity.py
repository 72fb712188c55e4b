"""GPU self-play driver (hm_selfplay_run: lockstep search over G game slots, raw-policy opening, resignation, outcome
labelling, HVM4 serialisation) against the sequential CPU restatement of tools/selfplay.cc (oracle/selfplay.hpp) under
the shared deterministic hash evaluator: the records of every game must be IDENTICAL BYTES.  This also exercises what the
history-free search parity roots cannot: repetition keys / 3-fold / rule50 over a real game history on the device
(Board::is_draw, board.h:423-453), Dirichlet seeds per (game, ply), the per-game RNG draw order."""
import numpy as np
import pytest
import torch

import oracle_py as O

pytestmark = pytest.mark.gpu


class HashNet:
    """Callback evaluator (Engine seam): the oracle's hash evaluator on whatever rows the driver asks for."""
    native = False

    def __call__(self, planes):
        h = planes.cpu().numpy().view(np.uint16).reshape(-1, 4736)
        v, a, b, w, m = O.hash_evaluator(h)
        tc = lambda x: torch.from_numpy(x.view(np.float16)).cuda()
        return tc(v), tc(a), tc(b), tc(w), tc(m)


def _split_by_game(hm, tmp_path, rec, cnt, name):
    path = str(tmp_path / name)
    hm.write_chunk(path, rec, cnt)
    raw = open(path, "rb").read()[20:]
    # walk the records to cut per-game byte ranges
    import struct
    out, off = {}, 0
    while off < len(raw):
        start = off
        (gid,) = struct.unpack_from("<Q", raw, off)
        off += 24 + 4736
        for _ in range(2):
            (n,) = struct.unpack_from("<H", raw, off)
            off += 2 + 6 * n
        out[gid] = out.get(gid, b"") + raw[start:off]
    return out


@pytest.mark.parametrize("kw", [dict(games=4, nodes=100, seed=11, max_macro_plies=80, concurrent_games=4),
                                dict(games=5, nodes=40, seed=2, max_macro_plies=400, concurrent_games=3, resign_disable_fraction=0.0),
                                dict(games=3, nodes=64, seed=7, max_macro_plies=30, concurrent_games=8, raw_policy_mean_macro_plies=0.0)],
                         ids=["4x100", "5x40-resign-slots3", "3x64-no-raw-8slots"])
def test_selfplay_records_match_oracle_bytes(hm, tmp_path, kw):
    cfg = hm.default_selfplay_config(**kw)
    sp = hm.SelfPlay(cfg, HashNet())
    res = sp.run()
    rec, cnt = sp.records()
    sp.close()
    got = _split_by_game(hm, tmp_path, rec, cnt, "gpu.hvm")
    ocfg = O.selfplay_cfg(**kw)
    ora = O.SelfPlayOracle(ocfg, 1, 1)
    term = [0] * 5
    total = 0
    for g in range(kw["games"]):
        want, info, acts = ora.game(g)
        term[info["termination"]] += 1
        total += info["samples"]
        assert got.get(g, b"") == want, (g, info, len(got.get(g, b"")), len(want))
    assert list(res.terminations) == term and res.samples == total == cnt and res.games == kw["games"]


def test_game_history_repetition_on_device(hm):
    """Board::is_draw / repetition_count over a real game history (board.h:326-332, 423-453): knights shuffle out and back
    on both boards; GPU flags and repetition counts vs the oracle Board after every push, then a search from the
    repeated positions (game history in place) vs the oracle search."""
    G = 6
    eng = hm.SearchEngine(G, 300)
    boards = [O.Board() for _ in range(G)]
    eng.set_games(np.concatenate([b.compact(0, False) for b in boards]))
    cyc = [[(0, "g1f3"), (0, "g8f6"), (0, "f3g1"), (0, "f6g8")], [(1, "b1c3"), (1, "b8c6"), (1, "c3b1"), (1, "c6b8")]]
    for step in range(11):
        ma, mb = np.zeros(G, np.uint32), np.zeros(G, np.uint32)
        for g in range(G):
            brd, uci = cyc[(g + step // 4) % 2][step % 4] if g % 3 else cyc[g % 2][step % 4]
            if g == 5 and step >= 6:
                continue                                   # this game stops shuffling: double sits from here on
            m = boards[g].find_move(brd, uci)
            if not m:
                continue
            (ma if brd == 0 else mb)[g] = m
            boards[g].push(brd, m)
        eng.apply(ma, mb)
        out = eng.classify(0, 0, 0, 0)
        for g in range(G):
            assert bool(out[g, 1]) == boards[g].is_draw(0), (step, g)
            assert (out[g, 2], out[g, 3]) == (boards[g].repetition_count(0), boards[g].repetition_count(1)), (step, g)
        st, flags = eng.game_state()
        for g in range(G):
            assert bool(flags[g] & 2) == boards[g].is_draw(0)
            assert st["rep_count"][g].tolist() == [min(3, boards[g].repetition_count(0)), min(3, boards[g].repetition_count(1))]
    assert any(b.is_draw(0) for b in boards) and not all(b.is_draw(0) for b in boards)
    # in-search draw rule (ply > 0: one earlier occurrence suffices) with the game history in place
    st, _ = eng.game_state()
    team, adv = st["team"].astype(int), st["time_adv"].astype(int)
    eng.begin_search(120)

    def ev(planes):
        h = planes.cpu().numpy().view(np.uint16).reshape(-1, 4736)
        return tuple(torch.from_numpy(x.view(np.float16)).cuda() for x in O.hash_evaluator(h))
    eng.run(ev)
    rs = eng.root_stats()
    for g in range(G):
        s = O.Search(1, 1)
        ok = s.run(boards[g], int(team[g]), bool(adv[g]), 120)
        if not ok:
            assert rs["info"][g][0] == 4, (g, rs["info"][g])
            continue
        e = s.edges()
        n = rs["counts"][g]
        assert n == len(e["visits"]) and np.array_equal(rs["visits"][g, :n], e["visits"]) and np.array_equal(rs["q"][g, :n], e["q"]), g
        assert np.array_equal(rs["move_a"][g, :n], e["move_a"]) and np.array_equal(rs["move_b"][g, :n], e["move_b"])
        assert rs["info"][g][12] == s.best_move(), (g, rs["info"][g][12], s.best_move())
    eng.close()


def test_empty_game_slots_are_skipped(hm):
    """Fewer games than slots: dead slots hold no position and must not be touched by any kernel."""
    from hivemind_amd import net as N
    torch.manual_seed(0)
    net = N.FusedNet(N.rise_v3_small())
    cfg = hm.default_selfplay_config(games=2, nodes=24, seed=5, concurrent_games=8, max_macro_plies=24)
    sp = hm.SelfPlay(cfg, net)
    res = sp.run()
    rec, cnt = sp.records()
    sp.close()
    assert res.games == 2 and cnt == res.samples > 0
    cfg2 = hm.default_selfplay_config(games=2, nodes=24, seed=5, concurrent_games=2, max_macro_plies=24)
    sp2 = hm.SelfPlay(cfg2, net)
    sp2.run()
    rec2, cnt2 = sp2.records()
    sp2.close()
    assert cnt2 == cnt and rec2.tobytes() == rec.tobytes()


class DeviceHashNet:
    """The same stand-in network evaluated on the device (hm_hash_evaluator): no host round trip per lockstep iteration, so a
    run at BASELINE's full size finishes in seconds."""
    native = False

    def __init__(self, hm, salt=0):
        self.hm, self.salt = hm, salt

    def __call__(self, planes):
        n = planes.shape[0]
        f16 = dict(dtype=torch.float16, device=planes.device)
        out = (torch.empty(n, **f16), torch.empty((n, 4672), **f16), torch.empty((n, 4672), **f16), torch.empty((n, 3), **f16), torch.empty(n, **f16))
        self.hm.check(self.hm.lib.hm_hash_evaluator(planes.data_ptr(), n, self.salt, *[t.data_ptr() for t in out], None))
        return out


def test_device_hash_evaluator_equals_oracle(hm):
    boards = O.random_positions(99, 300, 90)
    planes = hm.board_to_planes(hm.to_device(boards), "f16")
    for salt in (0, 0x5EED, 2**63 + 12345):
        got = DeviceHashNet(hm, salt)(planes)
        want = O.hash_evaluator_salted(planes.cpu().numpy().view(np.uint16).reshape(-1, 4736), salt)
        for g, w in zip(got, want):
            assert np.array_equal(g.cpu().numpy().view(np.uint16).reshape(w.shape), w)


def test_selfplay_full_size_config2_sampled_games_match_oracle(hm, tmp_path):
    """BASELINE configs[2] at its full size — 64 games, nodes 400, 64 slots, every default of tools/selfplay.h — under the
    stand-in network: the records of sampled games (first, last, the longest and a short one) equal the sequential CPU
    restatement byte for byte, i.e. visit counts, sampled actions, outcomes and plane records of whole games match."""
    kw = dict(games=64, nodes=400, seed=20260, concurrent_games=64)
    sp = hm.SelfPlay(hm.default_selfplay_config(**kw), DeviceHashNet(hm))
    res = sp.run()
    rec, cnt = sp.records()
    sp.close()
    assert res.games == 64 and res.samples == cnt > 64 * 10
    got = _split_by_game(hm, tmp_path, rec, cnt, "full.hvm")
    sizes = sorted((len(v), g) for g, v in got.items())
    picks = sorted({0, 63, sizes[-1][1], sizes[len(sizes) // 4][1]})
    ora = O.SelfPlayOracle(O.selfplay_cfg(**kw), 1, 1)
    for g in picks:
        want, info, _ = ora.game(g)
        assert got.get(g, b"") == want, (g, info, len(got.get(g, b"")), len(want))


def test_selfplay_nodes1600_config4_sampled_games_match_oracle(hm, tmp_path):
    """BASELINE configs[4]'s search shape — nodes 1600, transposition-sharing MCGS, Dirichlet root noise (all defaults) — over
    whole games: sampled games of a 64-game run (the per-GPU game count of the configuration) equal the CPU restatement byte
    for byte."""
    kw = dict(games=64, nodes=1600, seed=404, concurrent_games=64, max_macro_plies=60)
    sp = hm.SelfPlay(hm.default_selfplay_config(**kw), DeviceHashNet(hm))
    res = sp.run()
    rec, cnt = sp.records()
    sp.close()
    assert res.games == 64
    got = _split_by_game(hm, tmp_path, rec, cnt, "n1600.hvm")
    ora = O.SelfPlayOracle(O.selfplay_cfg(**kw), 1, 1)
    for g in (3, 41):
        want, info, _ = ora.game(g)
        assert got.get(g, b"") == want, (g, info, len(got.get(g, b"")), len(want))
