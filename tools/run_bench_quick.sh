# One gpurun call: the default bench line without the CPU baseline and the extra workloads; prints value and legs.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/r3_bench_quick.json 2> gpurun_out/r3_bench_quick.err
rc=$?
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r3_bench_quick.json"))
print("VALUE", d["value"], d["ms_per_step"]); print(json.dumps(d["extra"]["selfplay"]["leg_ms_per_iteration"])); print(d["extra"]["selfplay"]["wall_split_s"], d["roofline"]["kernel_ms"])
PY
exit $rc
