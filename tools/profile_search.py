#!/usr/bin/env python3
"""Cycle accounting of k_collect / k_process for game slot 0 (diagnostic build, -DHM_SEARCH_PROF).

Build here:  python tools/profile_search.py --build     (writes hivemind_amd/csrc/libhivemind_amd_prof.so)
Run on GPU:  HIVEMIND_AMD_LIB=hivemind_amd/csrc/libhivemind_amd_prof.so python tools/profile_search.py
"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "hivemind_amd", "csrc", "libhivemind_amd_prof.so")
if "--build" in sys.argv:
    subprocess.check_call(["make", "-s", "-j6", "libhivemind_amd_prof.so"], cwd=os.path.join(ROOT, "hivemind_amd", "csrc"))   # -DHM_SEARCH_PROF -DHM_SINGLE_TU
    print("built", PROF)
    sys.exit(0)
os.environ.setdefault("HIVEMIND_AMD_LIB", PROF)
sys.path.insert(0, ROOT)
import numpy as np, torch
import hivemind_amd as hm
from hivemind_amd import net as N, _lib

games = int(os.environ.get("GAMES", 64)); nodes = int(os.environ.get("NODES", 400))
hm.init(0)
torch.manual_seed(0)
fn = _lib.lib.hm_sp_profile
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
buf = np.zeros(128, dtype=np.uint64)
cfg = hm.default_selfplay_config(games=games, nodes=nodes, seed=1, concurrent_games=games, max_macro_plies=int(os.environ.get("PLIES", 40)))
sp = hm.SelfPlay(cfg, N.FusedNet(N.rise_v3_small()))
fn(None, 1)
res = sp.run()
fn(buf.ctypes.data, 0)
names = ["select_and_expand(total)", "should_expand_new_child", "gen_next", "select_child(puct)", "jb_make", "canonicalize_child",
         "classify_terminal", "hash+store+planes", "ctx/traj store", "stage_table", "k_collect total", "process: expand phase", "process: total", "path_reset", "process: backup_batch",
         "classifier wave: load + history", "classify: checkmate x2", "classify: draw", "classify: waiting-board mate", "wait: classifier wave", "puct: arg-max loop",
         "wait: generator wave", "wait: classifier (join)", "classifier wave: request total", "classifier wave: until type ack", "k_collect: drain (block join)", "k_collect: write-back", "position_child: path_load", "position_child: path_store", "scan_edges",
         "(clock ticks)", "(clock ticks)",
         "k_search: collect phase", "k_search: collect_batch (wave 0)", "k_search: publish", "k_search: wait for evaluation", "k_search: process phase",
         "k_search: backups (wave 0)", "k_search: control", "k_search: drain + barrier", "expand: gather + softmax (wave 1)", "expand: sort + store (wave 1)",
         "expand: frontier + first child (wave 1)", "k_search: expansions (wave 1)", "-", "k_search: iteration tail", "wait: creation outcome (resolve_create)", "classifier wave: creation step"]
it = res.search_iterations
if res.persistent_searches:
    it = max(1, int(buf[64 + 32]))          # game slot 0 only: its own iterations
    print(f"persistent searches {res.persistent_searches}: per-iteration figures are for game slot 0 ({it} iterations); wait_ms {res.wait_ms:.1f} k_search ms {res.search_kernel_ms:.1f}")
print(f"samples {res.samples} iters {it} pos/s {res.samples / res.seconds:.1f}")
if int(buf[31]):
    print(f"k_collect shader clock: {int(buf[30]) / int(buf[31]) * 100:.0f} MHz (s_memtime / s_memrealtime x 100 MHz)")
for i, n in enumerate(names):
    cyc, cnt = int(buf[i]), int(buf[64 + i])
    print(f"{n:28s} cycles/iter {cyc / max(it, 1):10.0f}   calls/iter {cnt / max(it, 1):6.2f}   cycles/call {cyc / max(cnt, 1):9.0f}")

# straggler analysis: wave-0 traversal cycles per (k_collect launch, game slot); a launch lasts as long as its slowest game
L = 8192
dur = np.zeros((L, 64), dtype=np.uint32)
if _lib.lib.hm_sp_profile_launches(dur.ctypes.data, L) == 0:
    live = dur > 0
    rows = live.any(axis=1)
    d = dur[rows].astype(np.float64); lv = live[rows]
    mx = d.max(axis=1); mean = d.sum(axis=1) / lv.sum(axis=1)
    print(f"launches {rows.sum()}  live games/launch {lv.sum(axis=1).mean():.1f}  mean-of-max {mx.mean():.0f}  mean-of-mean {mean.mean():.0f}  ratio {mx.mean() / mean.mean():.2f}")
    for q in (16, 8, 4):
        # the same games split into groups of q slots, each group waiting only for its own slowest game
        gm = [np.where(lv[:, i:i + q].any(axis=1), d[:, i:i + q].max(axis=1), np.nan) for i in range(0, 64, q)]
        print(f"  groups of {q:2d}: mean group max {np.nanmean(np.stack(gm)):.0f}")
    print(f"leg ms/iter: collect {res.collect_ms / it:.4f} net {res.eval_ms / it:.4f} process {res.process_ms / it:.4f}; wall per iter {res.seconds / it * 1e3:.4f}")
