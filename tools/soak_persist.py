#!/usr/bin/env python3
"""Repeats one persistent-search self-play configuration in ONE process and prints the first give-up report (diagnostics of the
queue hand-off, hm_queue.hpp).  usage: soak_persist.py [--model full|small] [--runs N] [--games G] [--nodes K]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hivemind_amd as hm
from hivemind_amd import net as N

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="full")
ap.add_argument("--runs", type=int, default=4)
ap.add_argument("--games", type=int, default=64)
ap.add_argument("--nodes", type=int, default=400)
a = ap.parse_args()
hm.init(0)
torch.manual_seed(0)
net = N.FusedNet(N.rise_v33() if a.model == "full" else N.rise_v3_small())
for r in range(a.runs):
    t0 = time.perf_counter()
    sp = hm.SelfPlay(hm.default_selfplay_config(games=a.games, nodes=a.nodes, seed=6 + r, concurrent_games=a.games), net)
    try:
        res = sp.run()
        print(f"run {r}: ok {res.samples} samples, {res.persistent_searches} persistent searches, {res.persistent_stalls} stalls, {time.perf_counter() - t0:.1f}s", flush=True)
    except Exception as e:
        print(f"run {r}: FAILED after {time.perf_counter() - t0:.1f}s: {e}", flush=True)
        sys.exit(1)
    finally:
        sp.close()
