#!/usr/bin/env python3
"""Where a bench step spends its time outside hm_selfplay_run (engine creation, record copy, teardown)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hivemind_amd as hm
from hivemind_amd import net as N
hm.init(0)
torch.manual_seed(0)
net = N.FusedNet(N.rise_v3_small())
for k in range(3):
    t0 = time.perf_counter()
    cfg = hm.default_selfplay_config(games=64, nodes=400, seed=1 + k, concurrent_games=64)
    sp = hm.SelfPlay(cfg, net)
    t1 = time.perf_counter()
    res = sp.run()
    t2 = time.perf_counter()
    rec, cnt = sp.records()
    t3 = time.perf_counter()
    sp.close()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"create {1e3*(t1-t0):.1f} ms  run {1e3*(t2-t1):.1f} ms (engine clock {1e3*res.seconds:.1f})  records {1e3*(t3-t2):.1f} ms  close {1e3*(t4-t3):.1f} ms  samples {res.samples}")
