#!/bin/bash
# Raw rocprofv3 inputs of one round's committed profile summaries.  Run on the GPU box from the repo root:
#     bash profiles/collect.sh r03
# Writes under gpurun_out/<tag>_* (scratch); `python profiles/summarize.py <tag>` then turns them into profiles/<tag>_*.
# Counter passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass; no other trace
# domain next to --pmc).  The program under the profiler is python3 itself (no wrapper process).
set -u
TAG=${1:-r03}
R=$(pwd)
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_graph -o p -- $B --no-replay --no-timeline --steps 400 --warmup 20 > $OUT/${TAG}_graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_eager -o p -- $B --eager --no-replay --no-timeline --steps 50 --warmup 5 > $OUT/${TAG}_eager.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_replay -o p -- $B --eager --no-timeline --steps 5 --warmup 2 > $OUT/${TAG}_replay.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o p -- $B --eager --no-replay --no-timeline --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o p -- $B --eager --no-replay --no-timeline --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_sq -o p -- $B --eager --no-replay --no-timeline --steps 10 --warmup 3 > /dev/null 2>&1
# calibration of the MFMA-utilisation formula on a kernel of known efficiency (4096^3 NT GEMM, ~90 % of peak by its clock)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_sqcal -o p -- $R/tests/native/bench_kernels gemm4k 5 > $OUT/${TAG}_sqcal.log 2>&1
# counters of the N = 1e6 sweep's kernels on a short sweep (8 tiles: the per-launch figures do not depend on N)
BS="python3 $R/bench.py --workload stress --stress-n 65536 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_stfetch -o p -- $BS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_stwrite -o p -- $BS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_stsq -o p -- $BS > /dev/null 2>&1
# the round's bench line: the default command (its `secondary` object carries the other BASELINE configs and the stress case)
python3 $R/bench.py > $OUT/${TAG}_line_smnist.log 2>&1
ls $OUT | grep ${TAG}_ | head -40
