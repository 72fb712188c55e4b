#!/usr/bin/env python3
"""bench.py — one JSON line per run (contract in the task statement).

Workload (round 1): BASELINE.json configs[1] — batched plane-encode of 65 536 random Bughouse
positions per GPU per step (fp16 planes, the dtype the reference feeds its net,
searchthread.cc:413-418).  A "step" = one pass of the encoder over the batch, inputs resident
in HBM.  N>1: every rank encodes its own batch (independent positions, no collective), weak
scaling.  The CPU baseline leg times the oracle's encoder (kind "port": the reference's
planes.cc needs CUDA headers this image lacks) on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POSITIONS = 65536
BYTES_OUT = 4736 * 2          # fp16 planes written per position
BYTES_IN = 208                # hm_board read per position
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--perft-depth", type=int, default=4, help="extra: joint perft depth reported beside the metric (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import hivemind_amd as hm
    hm.init(local)
    dev = torch.device("cuda", local)

    boards = hm.random_positions(N_POSITIONS, seed=42 + rank)
    out = torch.empty((N_POSITIONS, 74, 8, 8), dtype=torch.float16, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        hm.board_to_planes(boards, "f16", out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for s, e in ev:
        s.record()
        hm.board_to_planes(boards, "f16", out=out)
        e.record()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    value = world * N_POSITIONS * args.steps / dt
    algo_bytes = N_POSITIONS * (BYTES_OUT + BYTES_IN)
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9

    extra = {}
    if args.perft_depth > 0:
        # second metric of BASELINE.json: joint perft nodes/s; ply-2 frontier striped over ranks
        nodes, secs = hm.perft(args.perft_depth, shard=rank, nshards=world)
        tot = torch.tensor([float(nodes), secs], dtype=torch.float64, device=dev)
        if dist is not None:
            n_all = tot[0:1].clone(); s_all = tot[1:2].clone()
            dist.all_reduce(n_all, op=dist.ReduceOp.SUM)
            dist.all_reduce(s_all, op=dist.ReduceOp.MAX)
            nodes, secs = int(n_all.item()), float(s_all.item())
        extra["perft"] = {"depth": args.perft_depth, "nodes": int(nodes), "seconds": secs,
                          "nodes_per_s": nodes / secs if secs > 0 else None}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_py as O                     # checker / baseline leg only
        sample = 16384
        hb = boards[:sample].cpu().numpy().view(O.BOARD_DTYPE).reshape(-1)
        ob = np.zeros((sample, 4736), dtype=np.uint16)
        reps = 0
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < 10.0:
            O.lib.ora_time_planes(hb.ctypes.data, sample, 0, ob.ctypes.data, 8)
            reps += 8
        cdt = time.perf_counter() - t1
        got = out[:sample].cpu().numpy().view(np.uint16).reshape(sample, -1)
        assert np.array_equal(got, ob), "GPU planes differ from the oracle on the bench batch"
        cpu = {"value": sample * reps / cdt, "unit": "positions/s", "cores": 1, "kind": "port",
               "sample": f"{sample} of the {N_POSITIONS} bench positions x {reps} passes, oracle planes_f16, 1 thread"}

    if rank == 0:
        line = {
            "metric": "plane-encode positions/sec (BASELINE configs[1]; self-play positions/sec pending the GPU search)",
            "value": value, "unit": "positions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "batched plane-encode of 65536 random-playout Bughouse positions per GPU, fp16 out",
                       "positions_per_gpu": N_POSITIONS, "sharding": f"positions x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "encode_planes_kernel<f16>", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": algo_bytes},
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
