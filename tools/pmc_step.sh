#!/bin/bash
# SQ / TCC counters of EVERY kernel of the tracking step (bench.py, default 512 streams) — one rocprofv3 --pmc pass per counter group.
# Run through gpurun from the repo root:  bash tools/pmc_step.sh r03_a [streams]
# bench.py must not fork under --pmc (the profiler's preloaded library has initialised the GPU before Python starts; forked pool workers of such a
# process never exit): --gen-procs 1 generates the synthetic streams in-process, --distinct 16 keeps that to ~12 s.
set -o pipefail
TAG=${1:-r03_a}
S=${2:-512}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--streams $S --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-host-input-pass --distinct 16 --gen-procs 1"
date
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq_step -- python3 $R/bench.py $ARGS > $OUT/${TAG}_pmc_sq_step.json 2> $OUT/${TAG}_pmc_sq_step.err || { echo "sq pass failed rc=$?"; tail -5 $OUT/${TAG}_pmc_sq_step.err; exit 1; }
python3 $R/tools/pmc_summary.py $OUT/${TAG}_pmc_sq_step > $OUT/${TAG}_pmc_sq_step${S}.txt
rm -rf $OUT/${TAG}_pmc_sq_step
echo "sq done"; date
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_step_$C -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_pmc_step_$C.err || { echo "$C pass failed"; exit 1; }
  python3 $R/tools/pmc_summary.py $OUT/${TAG}_pmc_step_$C > $OUT/${TAG}_pmc_${C}_step${S}.txt
  rm -rf $OUT/${TAG}_pmc_step_$C
done
echo "tcc done"; date
