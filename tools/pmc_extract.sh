#!/bin/bash
# SQ counters of the extractor kernels alone (tools/extract_times.py B): issue, wait and LDS-conflict counters in two passes.
# bash tools/pmc_extract.sh TAG [B]   (env such as VIORB_FAST_V2=1 is inherited)
set -o pipefail
TAG=${1:-r03_x}; B=${2:-256}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/${TAG}_p1 -- python3 $R/tools/extract_times.py $B > /dev/null 2>&1 || { echo "pass 1 failed"; exit 1; }
python3 $R/tools/pmc_summary.py $OUT/${TAG}_p1 > $OUT/${TAG}_pmc_sq_ex$B.txt
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/${TAG}_p2 -- python3 $R/tools/extract_times.py $B > /dev/null 2>&1 || { echo "pass 2 failed"; exit 1; }
python3 $R/tools/pmc_summary.py $OUT/${TAG}_p2 > $OUT/${TAG}_pmc_lds_ex$B.txt
rm -rf $OUT/${TAG}_p1 $OUT/${TAG}_p2
echo ok
