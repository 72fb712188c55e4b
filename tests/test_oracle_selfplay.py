"""CPU suite for the self-play loop restatement (oracle/selfplay.hpp <- tools/selfplay.cc:160-232, 276-476, 558-748):
record layout as the reference reader expects it, determinism, per-game independence, and the replay property —
the joint actions the loop played, replayed on a fresh Board, reproduce every sample's planes, team and
time-advantage flag (so samples are paired with the right positions and the alternation of selfplay.cc:715-716
holds).  The GPU driver is compared with this restatement byte for byte in tests/test_gpu_selfplay_parity.py."""
import struct

import numpy as np

import oracle_py as O

HDR = "<QIHHBBbBf"


def parse(rec):
    out, off = [], 0
    while off < len(rec):
        gid, nodes, mply, mleft, team, adv, outcome, wdl, rootq = struct.unpack_from(HDR, rec, off)
        off += struct.calcsize(HDR)
        planes = np.frombuffer(rec, np.uint8, 4736, off)
        off += 4736
        pols = []
        for _ in range(2):
            (n,) = struct.unpack_from("<H", rec, off)
            off += 2
            pols.append(np.frombuffer(rec, np.dtype([("index", "<u2"), ("prob", "<f4")]), n, off))
            off += 6 * n
        out.append(dict(game_id=gid, nodes=nodes, macro_ply=mply, moves_left=mleft, team=team, time_adv=adv, outcome=outcome,
                        wdl=wdl, root_q=rootq, planes=planes, policy_a=pols[0], policy_b=pols[1]))
    assert off == len(rec)
    return out


def test_selfplay_oracle_records_and_replay():
    cfg = O.selfplay_cfg(games=3, nodes=48, seed=5, max_macro_plies=60)
    sp = O.SelfPlayOracle(cfg)
    for g in range(3):
        rec, info, acts = sp.game(g)
        samples = parse(rec)
        assert len(samples) == info["samples"] and sum(1 for a in acts if a[2]) == info["raw_plies"]
        assert [s["moves_left"] for s in samples] == list(range(len(samples), 0, -1))           # selfplay.cc:730-731
        assert all(s["game_id"] == g and s["wdl"] == s["outcome"] + 1 for s in samples)
        assert sum(s["nodes"] for s in samples) == info["nodes"]
        for s in samples:
            assert 1 <= s["nodes"] <= 48 * 1.05 + 16
            for pol in (s["policy_a"], s["policy_b"]):
                assert len(pol) >= 1 and abs(float(pol["prob"].sum()) - 1.0) < 1e-4 and np.all(np.diff(pol["index"].astype(int)) > 0)
        if info["winner"] >= 0:                                                                    # :726-729
            assert all(s["outcome"] == (1 if s["team"] == info["winner"] else -1) for s in samples)
        else:
            assert all(s["outcome"] == 0 for s in samples)
        # replay: searched plies are the actions without the raw flag, in order
        b = O.Board()
        rng_team = None
        k = 0
        team = samples[0]["team"] if not acts or not acts[0][2] else None
        # the starting team is not recorded: recover it from the first sample and the number of plies before it
        first_ply = samples[0]["macro_ply"] if samples else 0
        team = (samples[0]["team"] ^ (first_ply & 1)) if samples else 0
        adv = False
        for ply, (ma, mb, raw) in enumerate(acts):
            if not raw:
                s = samples[k]
                k += 1
                assert s["macro_ply"] == ply and s["team"] == team and s["time_adv"] == int(adv)
                want = O.planes(b.compact(team, adv), "u8")[0]
                assert np.array_equal(s["planes"], want), (g, ply)
            if ma:
                assert ma in b.legal_moves(0)
                b.push(0, ma)
            if mb:
                assert mb in b.legal_moves(1)
                b.push(1, mb)
            team ^= 1
            adv = not adv
        assert k == len(samples) or (k == len(samples) - 1 and info["termination"] in (3, 4))     # resignation: last sample has no action
        if info["termination"] == 1:
            assert b.is_checkmate(team, adv)
        elif info["termination"] == 2:
            assert b.is_draw(0)


def test_selfplay_oracle_is_deterministic_and_per_game():
    cfg = O.selfplay_cfg(games=4, nodes=32, seed=9, max_macro_plies=40)
    a = O.SelfPlayOracle(cfg)
    r2 = a.game(2)[0]
    r0 = a.game(0)[0]
    b = O.SelfPlayOracle(cfg)
    assert b.game(0)[0] == r0 and b.game(2)[0] == r2          # a game's records do not depend on what was played before it
    other = O.SelfPlayOracle(O.selfplay_cfg(games=4, nodes=32, seed=10, max_macro_plies=40))
    assert other.game(0)[0] != r0


def test_selfplay_oracle_config_switches():
    """No raw-policy opening when the mean is 0 (selfplay.cc:189-191); macro-ply limit termination."""
    cfg = O.selfplay_cfg(games=1, nodes=16, seed=3, max_macro_plies=6, raw_policy_mean_macro_plies=0.0)
    rec, info, acts = O.SelfPlayOracle(cfg).game(0)
    assert info["raw_plies"] == 0 and info["termination"] == 0 and info["samples"] == 6 and len(acts) == 6
    assert [s["macro_ply"] for s in parse(rec)] == list(range(6))
