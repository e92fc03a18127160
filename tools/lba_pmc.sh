#!/bin/bash
# Counters of the lock-step window-solve batch (bench.py --config local_ba under rocprofv3 --pmc): gpurun_out/<tag>_pmc_lba_sq.txt
set -o pipefail
TAG=${1:-r03_f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--config local_ba --no-cpu-baseline --steps 1 --warmup 1"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_lba_sq -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || { echo "sq pass failed"; exit 1; }
python3 $R/tools/pmc_summary.py $OUT/${TAG}_lba_sq > $OUT/${TAG}_pmc_lba_sq.txt; rm -rf $OUT/${TAG}_lba_sq
# (a second pass with the TA_* counters did not finish on this pool — killed at its limit — and is not repeated here)
grep -A9 "k_bab_schur\|k_bab_f_hpp\|k_bab_f_lin" $OUT/${TAG}_pmc_lba_sq.txt | head -40
