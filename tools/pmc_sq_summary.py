#!/usr/bin/env python3
"""Per-launch averages of the SQ counters of the fused forward (tools/profile_net_sq.sh passes) -> profiles/rNN_net_pmc_sq.json.
usage: python tools/pmc_sq_summary.py --out profiles/r02_net_pmc_sq.json gpurun_out/sq1 gpurun_out/sq2"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--out", required=True)
a = ap.parse_args()
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in a.dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.match(r"_ZN3hmn(19rise_forward_narrow|20rise_forward_narrow4)ILi(\d+)ELb(\d)EE", row["Kernel_Name"].strip())
            if not m:
                continue
            form = "4-wave form" if m.group(1).startswith("20") else "8-wave form"
            name = {"4": "small (RISEv3-small, 512 rows)", "12": "full (RISEv3.3, 512 rows)"}.get(m.group(2), "CTILES=" + m.group(2)) + ", " + form
            c = acc[name][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"]); c[1] += 1
out = {"command": "rocprofv3 --pmc <counters> --output-format csv -- python3 tools/bench_net_quick.py 512 (tools/profile_net_sq.sh: two passes, no trace "
                  "flags; the second pair of passes with HM_NET_WAVES=4); per-launch averages of the fused forward, both forms", "kernels": {}}
for name, cs in acc.items():
    k = {c: v[0] / v[1] for c, v in cs.items()}
    k["launches"] = max(v[1] for v in cs.values())
    if k.get("SQ_INSTS_MFMA"):
        k["valu_per_mfma"] = k.get("SQ_INSTS_VALU", 0.0) / k["SQ_INSTS_MFMA"]
    if k.get("SQ_WAVE_CYCLES"):
        k["wait_any_frac"] = k.get("SQ_WAIT_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
    out["kernels"][name] = k
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:1500])
