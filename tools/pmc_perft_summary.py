#!/usr/bin/env python3
"""SQ counters of the perft kernels (rocprofv3 --pmc pass of tools/run_perft.py) -> profiles/rNN_perft_pmc_sq.json.
VALU utilisation ("report nodes/s and VALU utilisation") = cycles the SIMDs spent issuing VALU instructions
(SQ_ACTIVE_INST_VALU, summed over SIMDs) / (SIMDs x kernel cycles); the kernel's duration comes from the kernel-trace pass.
usage: python tools/pmc_perft_summary.py --out profiles/r03_perft_pmc_sq.json --trace gpurun_out/perft_ks gpurun_out/perft_sq"""
import argparse, csv, glob, json, os
from collections import defaultdict
ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--trace", required=True)
ap.add_argument("--out", required=True)
a = ap.parse_args()
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(int)
for d in a.dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].strip()
            if "perft" not in k:
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row.get("Dispatch_Id"))
            if key not in seen:
                seen.add(key); launches[k] += 1
dur = {}
for f in glob.glob(os.path.join(a.trace, "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "perft" in row["Name"]:
            dur[row["Name"].strip()] = {"calls": int(row["Calls"]), "total_ns": float(row["TotalDurationNs"])}
SIMDS, CLK = 256 * 4, 2.4e9
out = {"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES -- "
                  "python3 tools/run_perft.py 5 (no trace flags); durations from a separate --kernel-trace --stats pass of the same command; sums over "
                  "all launches of each kernel", "kernels": {}}
for k, cs in acc.items():
    e = dict(cs)
    e["launches"] = launches[k]
    name = next((n for n in dur if n.split("(")[0] in k or k.split("(")[0] in n), None)
    if name:
        e["total_ms"] = dur[name]["total_ns"] * 1e-6
        cyc = dur[name]["total_ns"] * 1e-9 * CLK * SIMDS
        if "SQ_ACTIVE_INST_VALU" in e:
            e["valu_utilisation"] = e["SQ_ACTIVE_INST_VALU"] / cyc
        if "SQ_INSTS_VALU" in e:
            e["valu_instructions_per_simd_cycle"] = e["SQ_INSTS_VALU"] / cyc
    if e.get("SQ_WAVE_CYCLES"):
        e["wait_any_frac_of_wave_cycles"] = e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
        e["issue_frac_of_wave_cycles"] = e.get("SQ_ACTIVE_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
    out["kernels"][k] = e
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:2500])
