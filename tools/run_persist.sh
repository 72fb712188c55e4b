set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_persistent.py -x -q > gpurun_out/r3_persist_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"
tail -5 gpurun_out/r3_persist_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3_p_bench.json 2> gpurun_out/r3_p_bench.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3_p_bench.json"))
print(d["value"], d["ms_per_step"])
print(json.dumps(d["extra"]["selfplay"])[:700])
PY
