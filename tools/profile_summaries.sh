#!/bin/bash
# build container, after tools/profile_round.sh has run under gpurun: turn gpurun_out/ into the round's profiles/ files
R=${1:-r04}
python tools/pmc_summary.py --out profiles/${R}_selfplay64_pmc_hbm.json --algo-log gpurun_out/pmc_w.log --command "rocprofv3 --pmc WRITE_SIZE|FETCH_SIZE (separate passes) -- python3 tools/run_selfplay.py --games 64 --nodes 400 --concurrent 64 --max-macro-plies 30 (single-launch search: k_rollout)" gpurun_out/pmc_w gpurun_out/pmc_f
python tools/pmc_summary.py --out profiles/${R}_selfplay64_lockstep_pmc_hbm.json --algo-log gpurun_out/pmc_lw.log --command "HM_SELFPLAY_LOCKSTEP=1 rocprofv3 --pmc WRITE_SIZE|FETCH_SIZE (separate passes) -- python3 tools/run_selfplay.py --games 64 --nodes 400 --concurrent 64 --max-macro-plies 30" gpurun_out/pmc_lw gpurun_out/pmc_lf
python tools/pmc_summary.py --out profiles/${R}_planes_pmc_hbm.json --command "rocprofv3 --pmc WRITE_SIZE|FETCH_SIZE (separate passes) -- python3 bench.py --workload planes --steps 3 --warmup 1 --no-cpu-baseline" gpurun_out/pmc_pw gpurun_out/pmc_pf
python tools/pmc_perft_summary.py --out profiles/${R}_perft_pmc_sq.json --trace gpurun_out/perft_ks gpurun_out/perft_sq > /dev/null
python tools/pmc_rollout_sq_summary.py --out profiles/${R}_rollout_pmc_sq.json gpurun_out/pmc_sq
cp gpurun_out/ks/ks_kernel_stats.csv profiles/${R}_bench_selfplay_kernel_stats.csv
cp gpurun_out/cpu_baseline_gpuhost.json profiles/${R}_cpu_baseline_gpuhost.json
cp gpurun_out/bench_line.json profiles/${R}_bench_line.json
cp gpurun_out/bench_line_full_net.json profiles/${R}_bench_line_full_net.json
cp gpurun_out/bench_line_nodes1600.json profiles/${R}_bench_line_nodes1600.json
