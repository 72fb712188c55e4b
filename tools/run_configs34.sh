# BASELINE configs[3] and configs[4] at their per-GPU size as bench lines (run from the repo root on the GPU box)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py --model full --steps 2 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/bench_line_full_net.json 2> gpurun_out/bench_full.err || { tail -3 gpurun_out/bench_full.err; exit 1; }
timeout -k 10 500 python3 bench.py --nodes 1600 --steps 1 --warmup 0 --no-extra --no-cpu-baseline > gpurun_out/bench_line_nodes1600.json 2> gpurun_out/bench_1600.err || { tail -3 gpurun_out/bench_1600.err; exit 1; }
python3 - <<'PY'
import json
for f in ("bench_line_full_net", "bench_line_nodes1600"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    sp = d["extra"]["selfplay"]
    print(f, "VALUE", round(d["value"], 1), "ms/step", round(d["ms_per_step"]), d["extra"]["search_mode"][:12], json.dumps(sp["leg_ms_per_iteration"]), sp["transposition_table"], sp["persistent_searches_repeated_after_a_stall"], sp["wall_split_s"])
    print("   rooflines:", [(r["kernel"][:12], round(r["frac"], 4)) for r in d["extra"]["rooflines"]])
PY
