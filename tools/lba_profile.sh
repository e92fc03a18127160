#!/bin/bash
# kernel stats of single-window / batched window solves (tools/lba_time.py): gpurun_out/<tag>_lba_kernel_stats.csv
set -o pipefail
TAG=${1:-r02_b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_lba -- python3 $R/tools/lba_time.py > $OUT/${TAG}_lba_time.txt 2>&1
find $OUT/${TAG}_lba -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_lba_kernel_stats.csv \;
rm -rf $OUT/${TAG}_lba
grep -v amdgpu $OUT/${TAG}_lba_time.txt
