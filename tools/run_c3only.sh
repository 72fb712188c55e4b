set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_selfplay.py -x -q -k "configs3" > gpurun_out/r3_c3_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -a "hivemind_amd error" gpurun_out/r3_c3_pytest.log | cut -c1-6000 | tail -1; tail -3 gpurun_out/r3_c3_pytest.log
exit $rc
