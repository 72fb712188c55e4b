#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into the per-kernel JSON bench.py reads for `roofline.traffic`.

usage (in the build container, after `gpurun` merged the pass directories back):
  python tools/pmc_summary.py --out profiles/r02_selfplay64_pmc_hbm.json --command "<what was profiled>" \
         gpurun_out/pmc_w gpurun_out/pmc_f

Each directory holds the *_counter_collection.csv of one `rocprofv3 --pmc <COUNTER> -- <cmd>` pass (separate passes, no
trace flags, as MI355X_MICROARCH.md §HBM prescribes).  Per kernel and counter: launches, average raw value (KB) and
bytes = raw * 1024 * correction, with FETCH_SIZE doubled on gfx950 (the guide's calibration for wide coalesced reads;
WRITE_SIZE is exact for 16-byte-per-lane stores).  The summary records the git commit and a hash of the kernel sources
(hivemind_amd/csrc + include): bench.py only trusts a summary whose source hash equals that of the tree it runs from."""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORRECTION = {"FETCH_SIZE": 2, "WRITE_SIZE": 1}
HOST_ONLY = {"hm_uci.hip"}          # host C++ without device code (the UCI front end): editing it cannot change a kernel


def source_hash(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "hivemind_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "hivemind_amd", "csrc", "*.hpp"))
                   + [os.path.join(root, "include", "hivemind_amd.h")])
    files = [f for f in files if os.path.basename(f) not in HOST_ONLY]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def short_name(k):
    m = re.match(r"_ZN3hmn19rise_forward_narrowILi(\d+)ELb(\d)EE", k.strip())     # rocprofv3 leaves this template mangled
    if m:
        return f"rise_forward_narrow<{m.group(1)},{'true' if m.group(2) == '1' else 'false'}>"
    m = re.match(r"_ZN3hms9k_rolloutILi(\d+)ELi(\d+)ELb(\d)EE", k.strip())
    if m:
        return f"k_rollout<{m.group(1)},{m.group(2)},{'true' if m.group(3) == '1' else 'false'}>"
    k = re.sub(r"\(.*\)$", "", k.strip())
    k = re.sub(r"^void ", "", k)
    return k.replace("hms::", "").replace("hmn::", "").replace("hmd::", "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default="")
    ap.add_argument("--algo-log", default=None, help="stdout of the profiled command (tools/run_selfplay.py prints ALGO_BYTES_PER_LAUNCH {...}): algorithmic bytes per launch of the same phase")
    ap.add_argument("--kernels", default="k_rollout,k_collect,k_process,k_begin,rise_forward,encode_planes_kernel,perft", help="substrings to keep")
    a = ap.parse_args()
    keep = [s for s in a.kernels.split(",") if s]
    acc = {}
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name, ctr = short_name(row["Kernel_Name"]), row["Counter_Name"]
                if not any(s in name for s in keep):
                    continue
                key = (name, ctr, int(row["Grid_Size"]))
                e = acc.setdefault(key, [0, 0.0])
                e[0] += 1
                e[1] += float(row["Counter_Value"])
    rows = []
    for (name, ctr, grid), (n, tot) in sorted(acc.items()):
        corr = CORRECTION.get(ctr, 1)
        rows.append(dict(kernel=name, counter=ctr, grid_size=grid, launches=n, avg_raw_KB=tot / n, gfx950_correction=corr,
                         avg_bytes=tot / n * 1024.0 * corr))
    try:
        sha = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
        dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "hivemind_amd/csrc", "include"], text=True).strip())
    except Exception:
        sha, dirty = None, None
    algo = None
    if a.algo_log and os.path.exists(a.algo_log):
        for line in open(a.algo_log, errors="replace"):
            if line.startswith("ALGO_BYTES_PER_LAUNCH "):
                algo = json.loads(line[len("ALGO_BYTES_PER_LAUNCH "):])
    json.dump(dict(command=a.command, git_commit=sha, git_dirty_sources=dirty, source_sha256=source_hash(), rows=rows,
                   algorithmic_bytes_per_launch_same_phase=algo), open(a.out, "w"), indent=1)
    print(f"wrote {len(rows)} rows to {a.out}")


if __name__ == "__main__":
    sys.exit(main())
