set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_persistent.py -x -q > gpurun_out/r3_persist_pytest.log 2>&1
echo "pytest rc=$?"
tail -30 gpurun_out/r3_persist_pytest.log
