# One gpurun call: search / self-play parity suites, then the bench line on the lockstep loop and on the persistent pair.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_search.py tests/test_gpu_search_cases.py tests/test_gpu_persistent.py tests/test_gpu_selfplay_parity.py tests/test_gpu_selfplay.py tests/test_gpu_uci.py tests/test_gpu_tournament.py -x -q > gpurun_out/r3_search_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/r3_search_pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
HM_SELFPLAY_LOCKSTEP=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/r3_bench_lockstep.json 2> gpurun_out/r3_bench_lockstep.err || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/r3_bench_quick.json 2> gpurun_out/r3_bench_quick.err || exit 1
python3 - <<'PY'
import json
for f in ("gpurun_out/r3_bench_lockstep.json","gpurun_out/r3_bench_quick.json"):
    d=json.load(open(f)); print(f, "VALUE", d["value"], json.dumps(d["extra"]["selfplay"]["leg_ms_per_iteration"]))
PY
