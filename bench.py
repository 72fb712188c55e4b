#!/usr/bin/env python3
"""bench.py — one JSON line per run (contract in the task statement).

Headline workload = BASELINE.json configs[2]: `selfplay --games 64 --nodes 400` with a random-init
RISEv3-small on one MI355X (synthetic: no dataset, torch.manual_seed(0) weights).  A "step" is one
complete self-play run of that configuration (fresh games, seed = base + step); `value` =
HVM4 samples (searched root positions) written per second, whole job.  N>1: every rank plays its own
64 games (games are the independent unit; the only exchange is the gather of finished records to
rank 0, inside the timed region) -> weak scaling.

`--workload planes` keeps the round-1 first bench (configs[1], batched plane-encode of 65 536
positions); its numbers are also reported under "extra" of the default run, with joint perft.

`--gpus N` without a torchrun environment (WORLD_SIZE unset): this process is only a launcher.  Before any
GPU / HIP call it starts N children of itself, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
the device-list shape of the reference's main.cc:154-173), forwards rank 0's JSON line and exits non-zero if
any child fails.  Under torchrun (WORLD_SIZE set) every process is a rank, as before.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
N_POSITIONS = 65536


def bench_planes(hm, dev, steps, warmup, rank):
    boards = hm.random_positions(N_POSITIONS, seed=42 + rank)
    out = torch.empty((N_POSITIONS, 74, 8, 8), dtype=torch.float16, device=dev)
    for _ in range(warmup):
        hm.board_to_planes(boards, "f16", out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, e in ev:
        s.record()
        hm.board_to_planes(boards, "f16", out=out)
        e.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
    algo = N_POSITIONS * (4736 * 2 + 208)
    return dict(positions_per_s=N_POSITIONS * steps / dt, kernel_ms=kern_ms, achieved_GBps=algo / (kern_ms * 1e-3) / 1e9,
                hbm_frac=algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, algorithmic_bytes_per_launch=algo), boards, out, dt


def cpu_planes_baseline(boards, out, seconds=8.0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O
    sample = 16384
    hb = boards[:sample].cpu().numpy().view(O.BOARD_DTYPE).reshape(-1)
    ob = np.zeros((sample, 4736), dtype=np.uint16)
    reps, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < seconds:
        O.lib.ora_time_planes(hb.ctypes.data, sample, 0, ob.ctypes.data, 8)
        reps += 8
    cdt = time.perf_counter() - t1
    got = out[:sample].cpu().numpy().view(np.uint16).reshape(sample, -1)
    assert np.array_equal(got, ob), "GPU planes differ from the oracle on the bench batch"
    return dict(value=sample * reps / cdt, unit="positions/s", cores=1, kind="port",
                sample=f"{sample} bench positions x {reps} passes, oracle planes_f16, 1 thread")


def cpu_selfplay_baseline(model, nodes, seconds=20.0):
    """Oracle single-thread reference-schedule search (oracle/search.hpp) with the same weights run by
    torch on the host cores: searched positions per second on a bounded sample of start-of-game roots."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))          # batch-8 evaluations do not scale past a few cores
    cpu_model = model.float().eval()

    def ev(planes_u16):
        x = torch.from_numpy(planes_u16.view(np.float16).astype(np.float32)).reshape(-1, 74, 8, 8)
        with torch.no_grad():
            v, a, b, w, m = cpu_model(x)
        h = lambda t: t.to(torch.float16).contiguous().numpy().view(np.uint16)
        return h(v.reshape(-1)), h(a), h(b), h(w), h(m.reshape(-1))
    roots = O.random_positions(1234, 40 * 11, 40)[::11]
    s = O.Search(1, 1)
    s.set_evaluator(ev)
    done, nodes_done, t0 = 0, 0, time.perf_counter()
    for i in range(len(roots)):
        b = O.Board()
        b.from_compact(roots[i:i + 1])
        if s.run(b, int(roots["team"][i]), bool(roots["time_adv"][i]), nodes):
            done += 1
            nodes_done += s.info()["nodes"]
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="positions/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{done} searches of {nodes} nodes from random-playout roots in {dt:.1f}s: oracle/search.hpp "
                       f"(single search thread, B=8) + the same RISEv3-small weights on torch CPU fp32 "
                       f"({torch.get_num_threads()} intra-op threads); {nodes_done / dt:.0f} nodes/s")


def _pmc_traffic(path):
    """bytes per launch by kernel = WRITE_SIZE + corrected FETCH_SIZE of a committed rocprofv3 PMC summary
    (tools/pmc_summary.py; per kernel and counter the grid size with the most launches).  Refused — empty result —
    unless the summary was taken on exactly the kernel sources this bench runs (sha256 over hivemind_amd/csrc + include):
    counters of an older kernel say nothing about this one."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    best = {}
    global _PMC_ALGO
    try:
        prof = json.load(open(os.path.join(ROOT, path)))
        if not isinstance(prof, dict) or prof.get("source_sha256") != pmc_summary.source_hash():
            return {}
        _PMC_ALGO[path] = prof.get("algorithmic_bytes_per_launch_same_phase")
        for r in prof["rows"]:
            key = (r["kernel"], r["counter"])
            if key not in best or r["launches"] > best[key]["launches"]:
                best[key] = r
    except (OSError, ValueError, KeyError):
        return {}
    out = {}
    for (kernel, _), r in best.items():
        out[kernel] = out.get(kernel, 0.0) + float(r["avg_bytes"])
    return out


_PMC_ALGO = {}                 # per summary file: algorithmic bytes per launch of the profiled command's own phase (tools/run_selfplay.py)
PMC_SELFPLAY = "profiles/r04_selfplay64_pmc_hbm.json"                 # the single-launch search (k_rollout) under rocprofv3 --pmc
PMC_LOCKSTEP = "profiles/r04_selfplay64_lockstep_pmc_hbm.json"       # the lockstep kernels (HM_SELFPLAY_LOCKSTEP=1)
PMC_PLANES = "profiles/r03_planes_pmc_hbm.json"


def launch_ranks(n, argv):
    """Parent of a `--gpus N` run started without torchrun: N child ranks of this script, one per GPU.  Nothing here touches
    the GPU (no HIP call, no torch.cuda query), so no process that initialised a device is ever replaced or forked."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.05)
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)), None)
    if failed is not None:                      # a dead rank leaves the others waiting in a collective: end exactly our children
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    if failed is not None or any(codes):
        print(f"bench.py: rank {failed[0] if failed else '?'} failed (exit codes {codes})", file=sys.stderr)
        return 1
    return 0


def bench_stub(steps, warmup, rank, world, dist):
    """Launcher / collective rehearsal without a GPU (tests/test_host_logic.py): the timing protocol of the real workloads
    (barrier, K timed steps, max over ranks, sum of the units) around a trivial CPU step."""
    import torch as T

    def allred(x, op):
        if dist is None:
            return x
        t = T.tensor([x], dtype=T.float64)
        dist.all_reduce(t, op=op)
        return float(t.item())
    if os.environ.get("HM_BENCH_STUB_FAIL_RANK") == str(rank):
        sys.exit(3)
    for _ in range(warmup):
        time.sleep(0.001)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    units = 0
    for _ in range(steps):
        time.sleep(0.002)
        units += 10 + rank
    if dist is not None:
        dist.barrier()
    dt = allred(time.perf_counter() - t0, dist.ReduceOp.MAX if dist else None)
    units = allred(units, dist.ReduceOp.SUM if dist else None)
    return dict(metric="stub units/sec (launcher rehearsal, no GPU)", value=units / dt, unit="units/s", steps=steps, warmup=warmup,
                ms_per_step=dt / steps * 1e3, dtype="none", config={"workload": "stub"}, roofline=None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="selfplay", choices=["selfplay", "planes", "stub"])
    ap.add_argument("--games", type=int, default=64)
    ap.add_argument("--nodes", type=int, default=400)
    ap.add_argument("--model", default="small", choices=["small", "full"])
    ap.add_argument("--perft-depth", type=int, default=5, help="extra: joint perft depth (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # rehearsal switches (not used by the driver): HM_BENCH_BACKEND=gloo and HM_BENCH_SHARE_GPU=1 run the N > 1 code path with
    # every rank on cuda:0 of a one-GPU box (the collectives then carry CPU tensors)
    backend = os.environ.get("HM_BENCH_BACKEND", "nccl")
    if os.environ.get("HM_BENCH_SHARE_GPU"):
        local = 0
    if args.workload == "stub":
        backend = "gloo"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.workload != "stub":
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if args.workload == "stub":
        line = bench_stub(args.steps, args.warmup, rank, world, dist)
        if rank == 0:
            line.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                         "cpu_baseline": None, "extra": {}})
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    import hivemind_amd as hm
    from hivemind_amd import net as N
    hm.init(local)
    dev = torch.device("cuda", local)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    extra, cpu, line = {}, None, None
    if args.workload == "planes":
        steps, warmup = max(args.steps, 50), max(args.warmup, 5)
        barrier()
        pl, boards, out, dt = bench_planes(hm, dev, steps, warmup, rank)
        dt = max_over_ranks(dt)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu = cpu_planes_baseline(boards, out)
        line = dict(metric="plane-encode positions/sec (BASELINE configs[1])", value=world * N_POSITIONS * steps / dt, unit="positions/s",
                    steps=steps, warmup=warmup, ms_per_step=dt / steps * 1e3, dtype="u64",
                    config={"workload": "batched plane-encode of 65536 random-playout Bughouse positions per GPU, fp16 out"},
                    roofline={"bound": "hbm", "achieved": pl["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pl["hbm_frac"],
                              "traffic": None, "kernel": "encode_planes_kernel<f16>", "kernel_ms": pl["kernel_ms"],
                              "algorithmic_bytes_per_launch": pl["algorithmic_bytes_per_launch"]})
    else:
        torch.manual_seed(0)
        model = N.rise_v3_small() if args.model == "small" else N.rise_v33()
        flops = N.flops_per_position(model)
        net = N.FusedNet(model, device=dev)        # single-launch HIP forward (hm_net_forward)

        def one_run(seed):
            cfg = hm.default_selfplay_config(games=args.games * world, nodes=args.nodes, seed=seed, concurrent_games=args.games,
                                             rank=rank, world=world)
            sp = hm.SelfPlay(cfg, net, device=dev)
            res = sp.run()
            rec, cnt = sp.records()
            rec, cnt = hm.gather_records(rec, cnt, dist)          # the only cross-GPU exchange
            sp.close()
            return res, rec, cnt
        for w in range(args.warmup):
            one_run(1000 + w)
        tot = dict(samples=0, nodes=0, eval_rows=0, iters=0, collect_ms=0.0, eval_ms=0.0, process_ms=0.0, nv=0, es=0, games=0, bytes=0, lw=0,
                   searches=0, search_kernel_ms=0.0, wait_ms=0.0, search_s=0.0, prologue_s=0.0, raw_s=0.0, seconds=0.0, positions=0,
                   stalls=0, tt_hits=0, tt_inserts=0)
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            res, rec, cnt = one_run(1 + k)
            tot["samples"] += res.samples; tot["nodes"] += res.total_nodes; tot["eval_rows"] += res.eval_rows
            tot["iters"] += res.search_iterations; tot["collect_ms"] += res.collect_ms; tot["eval_ms"] += res.eval_ms
            tot["process_ms"] += res.process_ms; tot["nv"] += res.nodes_visited; tot["es"] += res.edges_scanned
            tot["games"] += res.games; tot["bytes"] += rec.size; tot["lw"] += res.leaf_move_words
            tot["searches"] += res.persistent_searches; tot["search_kernel_ms"] += res.search_kernel_ms; tot["wait_ms"] += res.wait_ms
            tot["search_s"] += res.search_seconds; tot["prologue_s"] += res.prologue_seconds; tot["raw_s"] += res.raw_seconds
            tot["seconds"] += res.seconds; tot["positions"] += res.searched_positions
            tot["stalls"] += res.persistent_stalls; tot["tt_hits"] += res.tt_hits; tot["tt_inserts"] += res.tt_inserts
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        samples = sum_over_ranks(tot["samples"])
        nodes = sum_over_ranks(tot["nodes"])
        it = max(tot["iters"], 1)
        persistent = tot["searches"] > 0
        pmc = _pmc_traffic(PMC_SELFPLAY) if (args.games == 64 and args.nodes == 400 and args.model == "small") else {}
        if persistent:
            # Single-launch search (hm_sp_search): ONE launch of k_rollout per searched ply — its first workgroups are the games (one
            # workgroup per game for the whole search), the others the evaluator (one position per workgroup at a time); it lasts as long
            # as the ply's slowest game.  `kernel_ms` = average launch, HIP events on the stream it is launched on (hm_sp_search).
            # Legs are per game-iteration (device clock).
            launches = tot["searches"]
            ks_ms = tot["search_kernel_ms"] / launches
            legs = {"k_rollout search role: collect phase (tree traversal)": tot["collect_ms"] / it, "k_rollout search role: wait for the evaluator": tot["wait_ms"] / it,
                    "k_rollout search role: process phase (expand+backup)": tot["process_ms"] / it,
                    "k_rollout evaluator role: per position (one workgroup)": tot["eval_ms"] / max(tot["eval_rows"], 1)}
            dominant = "k_rollout"
            # HBM bytes the search needs per launch: the node pool stays in LDS for the whole search (loaded and written back once:
            # counted with the position records), so per visited path node only its edge record is scanned / updated in HBM (40 B each),
            # per created node a 232-byte position record is written and read back at the leaf, per network leaf the 9472-byte fp16
            # plane row and the legal move lists are written (4 B per move) and the evaluator's sorted move / prior arrays read (8 B)
            by = (tot["es"] * 40 + tot["nv"] * 40 + tot["nodes"] * 2 * 232 + tot["eval_rows"] * 9472 + tot["lw"] * 12) / launches
            ach = by / (ks_ms * 1e-3) / 1e9
            roof_tree = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_rollout (single-launch search: game workgroups + evaluator workgroups, hm_rollout.hip)", "kernel_ms": ks_ms,
                         "algorithmic_bytes_per_launch": by, "launches": launches,
                         "note": "latency-bound: a game's descents are a chain of dependent LDS / L2 reads on one wavefront; a launch lasts as long "
                                 "as its slowest game's search; the algorithmic bytes are the TREE's — the evaluator role's weight stream "
                                 "(2 MB per position out of L2 for RISEv3-small) is the counters' excess; see DESIGN.md"}
            tr = [v for k, v in pmc.items() if "k_rollout" in k]
            if tr:
                roof_tree["traffic"] = sum(tr)
                same = (_PMC_ALGO.get(PMC_SELFPLAY) or {}).get("k_rollout")
                if same:
                    roof_tree["algorithmic_bytes_per_launch_same_phase"] = same
                    roof_tree["traffic_over_algorithmic_same_phase"] = roof_tree["traffic"] / same
                roof_tree["traffic_source"] = PMC_SELFPLAY + " (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this workload's first 30 macro-plies on these kernel sources)"
            else:
                roof_tree["traffic_source"] = "no PMC summary for these kernel sources (tools/pmc_summary.py refuses stale profiles)"
            fl = tot["eval_rows"] * flops / launches
            ach = fl / (ks_ms * 1e-3) / 1e12
            busy_ms = tot["eval_ms"] / max(tot["eval_rows"], 1)
            roof_net = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS, "traffic": None,
                        "kernel": "k_rollout evaluator role (one position per workgroup at a time)", "kernel_ms": ks_ms,
                        "algorithmic_flops_per_launch": fl, "rows_per_launch": tot["eval_rows"] / launches, "busy_ms_per_position": busy_ms,
                        "TFLOPs_of_one_busy_workgroup": flops / (busy_ms * 1e-3) / 1e12,
                        "note": "launch duration = the search's; the evaluator idles whenever the games have no batch for it",
                        "traffic_source": roof_tree["traffic_source"]}
            lock = _pmc_traffic(PMC_LOCKSTEP) if pmc else {}
            if lock:
                same = _PMC_ALGO.get(PMC_LOCKSTEP) or {}
                cmp_ = {}
                for key in ("k_collect", "rise_forward"):
                    hit = [k for k in lock if key in k]
                    if hit and same.get(key):
                        trl = sum(lock[k] for k in hit)
                        cmp_[key] = {"traffic_bytes_per_launch": trl, "algorithmic_bytes_per_launch_same_phase": same[key], "ratio": trl / same[key]}
                extra["lockstep_pmc_traffic"] = {"source": PMC_LOCKSTEP, "bytes_per_launch": lock, "same_phase": cmp_,
                                                 "note": "counters and algorithmic bytes both over the profiled command (first 30 macro-plies, all games alive, lockstep kernels)"}
        else:
            pmc = _pmc_traffic(PMC_LOCKSTEP) if pmc is not None and (args.games == 64 and args.nodes == 400 and args.model == "small") else {}
            legs = {"k_collect (tree traversal)": tot["collect_ms"] / it, "RISEv3 forward (net)": tot["eval_ms"] / it,
                    "k_process (expand+backup)": tot["process_ms"] / it}
            dominant = max(legs, key=legs.get)
            # the forward evaluates only the rows that hold a leaf (ragged batch, hm_net_forward_groups)
            rows = tot["eval_rows"] / it
            net_ms = legs["RISEv3 forward (net)"]
            ach = rows * flops / (net_ms * 1e-3) / 1e12
            roof_net = {"bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                        "traffic": None, "kernel": "RISEv3 forward (net)", "kernel_ms": net_ms,
                        "algorithmic_flops_per_launch": rows * flops, "rows_per_launch": rows}
            # traversal: 64 B node header + 40 B per scanned edge read per visited node, 64+40 B written back per path node,
            # plus, per network leaf, the 9472-byte fp16 plane tensor and its legal move lists (4 B per move, counted by the kernel)
            tree_ms = legs["k_collect (tree traversal)"]
            by = (tot["nv"] * (64 + 104) + tot["es"] * 40 + tot["lw"] * 4) / it + rows * 9472
            ach = by / (tree_ms * 1e-3) / 1e9
            roof_tree = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_collect (tree traversal)", "kernel_ms": tree_ms, "algorithmic_bytes_per_launch": by,
                         "note": "latency-bound: one wavefront per game walks its tree with dependent loads; see DESIGN.md"}
            # HBM traffic per launch from the committed rocprofv3 PMC passes of this workload (WRITE_SIZE + 2 x FETCH_SIZE,
            # separate passes) - PMC counters cannot be read from inside the bench
            for r_, key in ((roof_tree, "k_collect"), (roof_net, "rise_forward")):
                hit = [k for k in pmc if key in k]
                if hit:
                    r_["traffic"] = sum(pmc[k] for k in hit)
                    same = (_PMC_ALGO.get(PMC_LOCKSTEP) or {}).get("k_collect" if key == "k_collect" else "rise_forward")
                    if same:                   # both numbers over the profiled command's own phase (the run's average launch is thinner)
                        r_["algorithmic_bytes_per_launch_same_phase"] = same
                        r_["traffic_over_algorithmic_same_phase"] = r_["traffic"] / same
                    r_["traffic_source"] = PMC_LOCKSTEP + " (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this workload on these kernel sources)"
                else:
                    r_["traffic_source"] = "no PMC summary for these kernel sources (tools/pmc_summary.py refuses stale profiles)"
        roof = roof_net if dominant.startswith("RISEv3") else roof_tree
        extra["search_mode"] = "single launch (k_rollout: game + evaluator workgroups, hm_queue.hpp)" if persistent else "lockstep (k_collect || forward -> k_process)"
        extra["rooflines"] = [roof_tree, roof_net]
        extra["selfplay"] = {"samples": samples, "nodes": nodes, "nodes_per_s": nodes / dt, "games": tot["games"] * world,
                             "iterations": tot["iters"], "iteration_unit": "game-iterations (per game: collect -> process)" if persistent else "lockstep iterations (all games)",
                             "leg_ms_per_iteration": legs,
                             "eval_rows": tot["eval_rows"], "record_bytes_rank0": tot["bytes"], "net_gflop_per_position": flops / 1e9,
                             "persistent_searches": tot["searches"], "search_kernel_ms_total": tot["search_kernel_ms"], "wait_ms_total": tot["wait_ms"],
                             "persistent_searches_repeated_after_a_stall": tot["stalls"],
                             "legs_mix_two_loops": bool(tot["stalls"]),     # a search the hang guard gave up was repeated on the lockstep loop: its iterations / leg times are lockstep ones
                             "transposition_table": {"lookups_that_hit": tot["tt_hits"], "inserts": tot["tt_inserts"],
                                                     "hit_rate": tot["tt_hits"] / max(tot["tt_hits"] + tot["tt_inserts"], 1)},
                             "wall_split_s": {"run": tot["seconds"], "search": tot["search_s"], "prologue": tot["prologue_s"], "raw_policy": tot["raw_s"]}}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu = cpu_selfplay_baseline(model, args.nodes)
        line = dict(metric=f"self-play positions/sec @ nodes={args.nodes}", value=samples / dt, unit="positions/s", steps=args.steps,
                    warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, dtype="fp16 net / u64 board / f32 tree",
                    config={"workload": f"selfplay --games {args.games} --nodes {args.nodes} per GPU, random-init RISEv3-{args.model} "
                                        f"(torch.manual_seed(0)), B=8 single-thread reference schedule per game",
                            "games_per_gpu": args.games, "nodes": args.nodes, "sharding": f"games x{world}, record gather only"},
                    roofline=roof)
        if not args.no_extra and world == 1:
            # the same engine with more games in flight (the 64-game configuration is latency-bound)
            cfg = hm.default_selfplay_config(games=256, nodes=args.nodes, seed=77, concurrent_games=256)
            sp = hm.SelfPlay(cfg, net, device=dev)
            r = sp.run()
            sp.close()
            it256 = max(r.search_iterations, 1)
            single = r.persistent_searches > 0
            extra["selfplay_256_concurrent_games"] = {
                "positions_per_s": r.samples / r.seconds, "nodes_per_s": r.total_nodes / r.seconds,
                "search_mode": "single launch (k_rollout; the live games share search workgroups while more than a quarter of the CUs would be games)" if single else "lockstep",
                "single_launch_searches": r.persistent_searches, "searches_repeated_after_a_stall": r.persistent_stalls,
                "leg_ms_per_iteration": ({"collect": r.collect_ms / it256, "wait_for_evaluator": r.wait_ms / it256, "process": r.process_ms / it256,
                                          "evaluator_per_position": r.eval_ms / max(r.eval_rows, 1)} if single else
                                         {"collect": r.collect_ms / it256, "net": r.eval_ms / it256, "process": r.process_ms / it256}),
                "iteration_unit": "game-iterations" if single else "lockstep iterations"}
            pl, boards, out, _ = bench_planes(hm, dev, 100, 10, rank)
            extra["plane_encode_64k"] = pl
            enc_all = _pmc_traffic(PMC_PLANES)
            enc = next((v for k, v in enc_all.items() if k.startswith("encode_planes_kernel<0>")), None)
            extra["rooflines"].append({"bound": "hbm", "achieved": pl["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": pl["hbm_frac"], "traffic": enc, "kernel": "encode_planes_kernel<f16>",
                                       "kernel_ms": pl["kernel_ms"], "algorithmic_bytes_per_launch": pl["algorithmic_bytes_per_launch"]})
            if rank == 0 and not args.no_cpu_baseline:
                extra["plane_encode_64k"]["cpu_baseline"] = cpu_planes_baseline(boards, out, 5.0)
        if not args.no_extra and args.perft_depth > 0:
            n, secs = hm.perft(args.perft_depth, shard=rank, nshards=world)
            n, secs = int(sum_over_ranks(n)), max_over_ranks(secs)
            extra["perft"] = {"depth": args.perft_depth, "nodes": n, "seconds": secs, "nodes_per_s": n / secs if secs > 0 else None}

    if rank == 0:
        line.update({"n_gpus": world, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                     "cpu_baseline": cpu, "extra": extra})
        order = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline", "extra"]
        print(json.dumps({k: line[k] for k in order}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
