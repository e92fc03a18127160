#!/bin/bash
# VALU issue budget of one tracking step per kernel: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / SQ_WAVE_CYCLES of every kernel of a short bench run
# (one counter pass; run through gpurun from the repo root). Output: gpurun_out/<tag>_pmc_sq_step.txt
set -o pipefail
TAG=${1:-r02_b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq_step -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $OUT/${TAG}_pmc_sq_step > $OUT/${TAG}_pmc_sq_step.txt
rm -rf $OUT/${TAG}_pmc_sq_step
echo done
