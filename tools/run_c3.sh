set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_selfplay.py -x -q -k "configs3" > gpurun_out/r3_c3_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -15 gpurun_out/r3_c3_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --model full --steps 1 --warmup 0 --no-cpu-baseline --no-extra > gpurun_out/r3_full_bench.json 2> gpurun_out/r3_full_bench.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3_full_bench.json"))
print("FULL", d["value"], d["ms_per_step"]); print(json.dumps(d["extra"]["selfplay"])[:900])
PY
