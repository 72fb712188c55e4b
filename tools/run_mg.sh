# several games per search workgroup (k_search_mg): parity tests with the small net forced onto it, configs[3] test (default for the deployed net), bench lines
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
HM_SEARCH_GAMES_PER_WG=4 timeout -k 10 500 python -m pytest tests/test_gpu_persistent.py -x -q > gpurun_out/mg_pytest1.log 2>&1
rc=$?; echo "persistent tests with 4 games per workgroup rc=$rc"; tail -4 gpurun_out/mg_pytest1.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
HM_SEARCH_GAMES_PER_WG=3 timeout -k 10 300 python -m pytest tests/test_gpu_persistent.py -x -q -k "equals_lockstep_search" > gpurun_out/mg_pytest2.log 2>&1
rc=$?; echo "3 per workgroup rc=$rc"; tail -3 gpurun_out/mg_pytest2.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python -m pytest tests/test_gpu_selfplay.py -x -q -k "configs3" > gpurun_out/mg_pytest3.log 2>&1
rc=$?; echo "configs3 tests rc=$rc"; tail -4 gpurun_out/mg_pytest3.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --model full --steps 2 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/mg_bench_full.json 2> gpurun_out/mg_bench_full.err || { tail -3 gpurun_out/mg_bench_full.err; exit 1; }
HM_SEARCH_GAMES_PER_WG=1 timeout -k 10 300 python3 bench.py --model full --steps 2 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/mg_bench_full_k1.json 2> gpurun_out/mg_bench_full_k1.err || { tail -3 gpurun_out/mg_bench_full_k1.err; exit 1; }
python3 - <<'PY'
import json
for f in ("mg_bench_full", "mg_bench_full_k1"):
    d = json.load(open(f"gpurun_out/{f}.json")); sp = d["extra"]["selfplay"]
    print(f, "VALUE", round(d["value"], 1), json.dumps({k[:24]: round(v, 4) for k, v in sp["leg_ms_per_iteration"].items()}), sp["persistent_searches_repeated_after_a_stall"], round(d["roofline"]["kernel_ms"], 2))
PY
