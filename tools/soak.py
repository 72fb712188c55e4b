#!/usr/bin/env python3
"""Soak check: repeated self-play runs must be byte-identical per seed; prints one line per run."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hivemind_amd as hm
from hivemind_amd import net as N

hm.init(0)
torch.manual_seed(0)
fused = N.FusedNet(N.rise_v3_small())
digests = {}
RUNS = [(256, 100, 128, 1), (256, 100, 128, 1), (256, 100, 64, 1), (256, 100, 128, 2), (8, 1600, 8, 5), (8, 1600, 4, 5)]
RUNS += [(64, 400, c, 7) for c in (64, 64, 32, 64, 16, 64, 48, 64)]      # the bench configuration, repeated: races in the four-wave traversal would show as differing digests
for games, nodes, conc, seed in RUNS:
    cfg = hm.default_selfplay_config(games=games, nodes=nodes, seed=seed, concurrent_games=conc)
    sp = hm.SelfPlay(cfg, fused)
    t = time.time()
    res = sp.run()
    rec, cnt = sp.records()
    sp.close()
    # records are appended in finishing order, which depends on the slot count: digest per game, in game order
    path = os.path.join("/tmp", "soak.hvm")
    hm.write_chunk(path, rec, cnt)
    per_game = {}
    for smp in hm.read_hvm4(path):
        per_game.setdefault(smp["game_id"], []).append((smp["nodes"], smp["macro_ply"], smp["moves_left"], smp["team"], smp["time_adv"], smp["outcome"],
                                                         float(smp["root_q"]), smp["planes"].tobytes(), smp["policy_a"].tobytes(), smp["policy_b"].tobytes()))
    h = hashlib.md5()
    for gid in sorted(per_game):
        h.update(repr((gid, per_game[gid])).encode())
    d = h.hexdigest()
    print(games, nodes, conc, seed, "samples", res.samples, "nodes", res.total_nodes, "pos/s", round(res.samples / res.seconds, 1),
          "term", list(res.terminations), d, flush=True)
    key = (games, nodes, seed)
    if key in digests:
        assert digests[key] == d, "records differ between runs / slot counts for the same seed"
    digests[key] = d
print("soak OK")
