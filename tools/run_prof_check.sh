set -o pipefail
cd $GRAFT_REPO_ROOT
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
echo "== kernel trace of the persistent pair"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ks_chk -o ks -- python3 $ROOT/tools/run_selfplay.py --games 16 --nodes 100 --concurrent 16 --out /tmp/sp1 > $ROOT/gpurun_out/ks_chk.log 2>&1
echo "rc=$?"; tail -2 $ROOT/gpurun_out/ks_chk.log | cut -c1-700
find $ROOT/gpurun_out/ks_chk -name '*kernel_stats.csv' | head -1 | xargs -r head -8
echo "== pmc pass without the lockstep switch (expects the 3 s give-up and the fallback)"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_chk -o w -- python3 $ROOT/tools/run_selfplay.py --games 8 --nodes 64 --concurrent 8 --max-macro-plies 12 --out /tmp/sp2 > $ROOT/gpurun_out/pmc_chk.log 2>&1
echo "rc=$?"; tail -2 $ROOT/gpurun_out/pmc_chk.log | cut -c1-700
find $ROOT/gpurun_out/ks_chk $ROOT/gpurun_out/pmc_chk -name '*.csv' -size +20M -delete
