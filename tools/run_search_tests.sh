# the search / self-play parity tests (GPU), then a short bench line
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_search.py tests/test_gpu_search_cases.py tests/test_gpu_persistent.py tests/test_gpu_selfplay_parity.py tests/test_gpu_selfplay.py tests/test_gpu_uci.py tests/test_gpu_tournament.py -x -q > gpurun_out/r3_search_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -12 gpurun_out/r3_search_pytest.log | cut -c1-400
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/r3_bench_quick.json 2> gpurun_out/r3_bench_quick.err
rc=$?
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3_bench_quick.json"))
print("VALUE", d["value"], d["ms_per_step"]); print(json.dumps(d["extra"]["selfplay"]["leg_ms_per_iteration"])); print(d["extra"]["selfplay"]["wall_split_s"], d["roofline"]["kernel_ms"])
PY
exit $rc
