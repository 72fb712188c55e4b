#!/bin/bash
# SQ counters of the fused forward at 512 rows (both networks), two rocprofv3 --pmc passes without trace flags; run from the repo
# root on the GPU box, summarise with tools/pmc_sq_summary.py in the build container.
set -e
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/sq1 -o a -- python3 $R/tools/bench_net_quick.py 512 > $R/gpurun_out/sq1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/sq2 -o b -- python3 $R/tools/bench_net_quick.py 512 > $R/gpurun_out/sq2.log 2>&1
tail -2 $R/gpurun_out/sq2.log
# the 4-wave form (the evaluator role of the four-wave k_rollout): the same two passes with HM_NET_WAVES=4
export HM_NET_WAVES=4
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/sq3 -o a -- python3 $R/tools/bench_net_quick.py 512 > $R/gpurun_out/sq3.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/sq4 -o b -- python3 $R/tools/bench_net_quick.py 512 > $R/gpurun_out/sq4.log 2>&1
tail -2 $R/gpurun_out/sq4.log
