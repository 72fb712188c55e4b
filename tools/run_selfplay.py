#!/usr/bin/env python3
"""hivemind selfplay --games G --nodes N (engine/src/main.cc:122-141) on this GPU."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hivemind_amd as hm
from hivemind_amd import net as N

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=16)
ap.add_argument("--nodes", type=int, default=100)
ap.add_argument("--concurrent", type=int, default=16)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--model", default="small", choices=["small", "full"])
ap.add_argument("--max-macro-plies", type=int, default=400)
ap.add_argument("--out", default="gpurun_out/selfplay")
ap.add_argument("--graph", action="store_true")
ap.add_argument("--torch-net", action="store_true", help="library (MIOpen) path instead of the fused HIP forward")
a = ap.parse_args()
hm.init(0)
torch.manual_seed(0)
model = N.rise_v3_small() if a.model == "small" else N.rise_v33()
if a.torch_net:
    net = N.InferenceNet(model)
    if a.graph:
        net.capture(a.concurrent * 8)
else:
    net = N.FusedNet(model)
cfg = hm.default_selfplay_config(games=a.games, nodes=a.nodes, seed=a.seed, concurrent_games=a.concurrent, max_macro_plies=a.max_macro_plies)
sp = hm.SelfPlay(cfg, net)
t = time.time()
res = sp.run()
dt = time.time() - t
rec, cnt = sp.records()
os.makedirs(a.out, exist_ok=True)
path = os.path.join(a.out, f"chunk_{a.seed}_000000.hvm")
hm.write_chunk(path, rec, cnt)
samples = hm.read_hvm4(path)
for s in samples:
    for p in (s["policy_a"], s["policy_b"]):
        assert abs(float(p["prob"].sum()) - 1.0) < 1e-4
# algorithmic HBM bytes per launch of the lockstep kernels over THIS run (the formula of bench.py's lockstep roofline): what a PMC
# pass of this command is compared with (tools/pmc_summary.py --algo-log)
it_ = max(res.search_iterations, 1)
algo = None
if not res.persistent_searches:
    rows_ = res.eval_rows / it_
    algo = {"k_collect": (res.nodes_visited * (64 + 104) + res.edges_scanned * 40 + res.leaf_move_words * 4) / it_ + rows_ * 9472,
            "rise_forward": rows_ * (9472 + 2 * 9344 + 10) + 2.0e6,          # planes in, both policy planes + value heads out, the packed weights once
            "launches": it_, "rows_per_launch": rows_}
else:
    # single-launch search (k_rollout): bench.py's formula — 40 B per scanned / updated edge, 2 x 232 B per created node's position record,
    # 9472 B of planes + 12 B per legal move (list out, sorted moves / priors back) per network leaf — per launch (= per searched ply)
    n_ = max(res.persistent_searches, 1)
    algo = {"k_rollout": (res.edges_scanned * 40 + res.nodes_visited * 40 + res.total_nodes * 2 * 232 + res.eval_rows * 9472 + res.leaf_move_words * 12) / n_,
            "launches": n_, "rows_per_launch": res.eval_rows / n_}
print("ALGO_BYTES_PER_LAUNCH " + json.dumps(algo))
print(json.dumps(dict(games=res.games, samples=res.samples, searched=res.searched_positions, nodes=res.total_nodes,
                      eval_rows=res.eval_rows, eval_batches=res.eval_batches, iters=res.search_iterations, raw=res.raw_plies,
                      seconds=res.seconds, wall=dt, positions_per_s=res.samples / res.seconds, nodes_per_s=res.total_nodes / res.seconds,
                      term=list(res.terminations), bytes=res.record_bytes, chunk_samples=len(samples),
                      persistent_searches=res.persistent_searches, search_kernel_ms=res.search_kernel_ms, wait_ms=res.wait_ms,
                      leg_ms=dict(collect=res.collect_ms / max(res.search_iterations, 1), net=res.eval_ms / max(res.search_iterations, 1),
                                  process=res.process_ms / max(res.search_iterations, 1)),
                      iter_ms=res.seconds * 1e3 / max(res.eval_batches, 1), search_iter_ms=res.search_seconds * 1e3 / max(res.search_iterations, 1),
                      split_s=dict(search=res.search_seconds, prologue=res.prologue_seconds, raw=res.raw_seconds,
                                   host=res.seconds - res.search_seconds - res.prologue_seconds - res.raw_seconds))))
