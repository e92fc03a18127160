#!/bin/bash
# kernel timeline of a few steps (rocprofv3 --kernel-trace): gpurun_out/<tag>_trace.csv. Usage: tools/trace_step.sh <tag> [bench args]
set -o pipefail
TAG=${1:-trace}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events "$@" > $OUT/${TAG}_trace.json 2> /dev/null
find $OUT/${TAG}_trace -name "*kernel_trace.csv" -exec cp {} $OUT/${TAG}_trace.csv \;
rm -rf $OUT/${TAG}_trace
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/${TAG}_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
# keep the last ~3 steps
import collections
out=open("$OUT/${TAG}_timeline.txt","w")
n=len(rows)
for r in rows[int(n*0.72):int(n*0.9)]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    out.write("%10.1f %10.1f %8.1f q%s %s\n"%(s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"].split("(")[0][:40]))
out.close()
PY
echo done
