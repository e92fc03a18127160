#!/bin/bash
# MFMA counters of the window-solve kernels (tools/lba_time.py under rocprofv3 --pmc): gpurun_out/<tag>_pmc_mfma_local_ba.txt
set -o pipefail
TAG=${1:-r02_c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_mfma -- python3 $R/tools/chol_probe.py > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $OUT/${TAG}_mfma > $OUT/${TAG}_pmc_mfma_local_ba.txt
rm -rf $OUT/${TAG}_mfma
grep -A7 "k_ba_chol_solve\|k_ba_schur" $OUT/${TAG}_pmc_mfma_local_ba.txt
