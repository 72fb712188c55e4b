# configs[3] at its per-GPU size (the full-size test), then the default bench line without the CPU baseline
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_selfplay.py -x -q -k "configs3" > gpurun_out/r3_c3_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -a "hivemind_amd error" gpurun_out/r3_c3_pytest.log | cut -c1-3000 | tail -2; tail -3 gpurun_out/r3_c3_pytest.log
[ $rc -eq 124 ] && exit 124
[ $rc -eq 137 ] && exit 137
timeout -k 10 400 python3 bench.py --no-cpu-baseline > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
rc2=$?
echo "bench rc=$rc2"; cut -c1-1200 gpurun_out/r3_bench.json; tail -3 gpurun_out/r3_bench.err
exit $rc
