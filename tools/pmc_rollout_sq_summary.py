#!/usr/bin/env python3
"""Per-launch averages of the SQ counters of the single-launch search kernel (k_rollout) from a `rocprofv3 --pmc <SQ counters>` pass of
tools/run_selfplay.py (tools/profile_round.sh: no trace flags) -> profiles/rNN_rollout_pmc_sq.json.
usage: python tools/pmc_rollout_sq_summary.py --out profiles/r04_rollout_pmc_sq.json gpurun_out/pmc_sq"""
import argparse, csv, glob, json, os, re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--out", required=True)
a = ap.parse_args()
acc = defaultdict(lambda: [0.0, 0])
for d in a.dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_rollout" not in row["Kernel_Name"]:
                continue
            c = acc[row["Counter_Name"]]
            c[0] += float(row["Counter_Value"]); c[1] += 1
k = {c: v[0] / v[1] for c, v in acc.items()}
k["launches"] = max((v[1] for v in acc.values()), default=0)
if k.get("SQ_WAVE_CYCLES"):
    k["wait_any_frac_of_wave_cycles"] = k.get("SQ_WAIT_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
    k["issuing_frac_of_wave_cycles"] = k.get("SQ_ACTIVE_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
if k.get("SQ_INSTS_VALU"):
    k["salu_per_valu"] = k.get("SQ_INSTS_SALU", 0.0) / k["SQ_INSTS_VALU"]
out = {"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES "
                  "--output-format csv -- python3 tools/run_selfplay.py --games 64 --nodes 400 --concurrent 64 --max-macro-plies 30 (no trace flags); "
                  "per-launch averages over the k_rollout launches (both roles of the kernel together: 64 game workgroups + 192 evaluator workgroups)",
       "k_rollout": k}
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(k, indent=1))
