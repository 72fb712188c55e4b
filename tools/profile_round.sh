#!/bin/bash
# One gpurun call's worth of round profiles (run from the repo root on the GPU box): the bench line, the kernel-trace statistics of
# the same command, and the separate PMC passes (no trace flags) of the lockstep kernels.  Everything lands under gpurun_out/;
# tools/pmc_summary.py turns the pass directories into profiles/rNN_*_pmc_hbm.json in the build container.
# Since round 4 the search is ONE kernel (k_rollout), so the PMC passes profile the timed path itself (round 3's two kernels could not
# run under --pmc, which runs kernels one at a time); a second pair of passes (HM_SELFPLAY_LOCKSTEP=1) keeps the lockstep kernels' counters.
set -e
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench.err
timeout -k 10 300 python3 bench.py --model full --steps 2 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/bench_line_full_net.json 2> gpurun_out/bench_full.err
timeout -k 10 300 python3 bench.py --nodes 1600 --steps 1 --warmup 0 --no-extra --no-cpu-baseline > gpurun_out/bench_line_nodes1600.json 2> gpurun_out/bench_1600.err
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ks -o ks -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $ROOT/gpurun_out/ks.log 2>&1
SP="python3 $ROOT/tools/run_selfplay.py --games 64 --nodes 400 --concurrent 64 --max-macro-plies 30 --out /tmp/sp_out"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_w -o w -- $SP > $ROOT/gpurun_out/pmc_w.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_f -o f -- $SP > $ROOT/gpurun_out/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $ROOT/gpurun_out/pmc_sq -o sq -- $SP > $ROOT/gpurun_out/pmc_sq.log 2>&1
export HM_SELFPLAY_LOCKSTEP=1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_lw -o w -- $SP > $ROOT/gpurun_out/pmc_lw.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_lf -o f -- $SP > $ROOT/gpurun_out/pmc_lf.log 2>&1
unset HM_SELFPLAY_LOCKSTEP
PL="python3 $ROOT/bench.py --workload planes --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_pw -o w -- $PL > $ROOT/gpurun_out/pmc_pw.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_pf -o f -- $PL > $ROOT/gpurun_out/pmc_pf.log 2>&1
PF="python3 $ROOT/tools/run_perft.py 5"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/perft_ks -o ks -- $PF > $ROOT/gpurun_out/perft_ks.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES --output-format csv -d $ROOT/gpurun_out/perft_sq -o sq -- $PF > $ROOT/gpurun_out/perft_sq.log 2>&1
cd $ROOT
timeout -k 10 300 python3 tools/cpu_baseline.py --json gpurun_out/cpu_baseline_gpuhost.json > gpurun_out/cpu_baseline_gpuhost.log 2>&1
find gpurun_out/ks gpurun_out/perft_ks -name '*kernel_trace.csv' -size +40M -delete
cat gpurun_out/bench_line.json
