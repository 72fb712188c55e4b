# One gpurun call: the whole -m gpu suite, then the default bench line (run from the repo root on the GPU box).
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r4_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -25 gpurun_out/r4_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err
rc=$?
echo "bench rc=$rc"; cut -c1-1500 gpurun_out/r4_bench.json; tail -3 gpurun_out/r4_bench.err
exit $rc
