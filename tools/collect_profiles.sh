#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02_a      -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# kernel-trace/--stats and every --pmc pass are separate runs (the pool refuses --pmc combined with the trace domains that crash nodes).
set -o pipefail
TAG=${1:-r03_c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $OUT/${TAG}_bench256.json 2> $OUT/${TAG}_bench256.err
echo "bench done"; date
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --no-extra-passes > $OUT/${TAG}_bench256_under_rocprof.json 2> /dev/null
find $OUT/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bench256_kernel_stats.csv \;
echo "kernel stats done"; date
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$C -- python3 $R/tools/extract_times.py 256 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $OUT/${TAG}_pmc_$C > $OUT/${TAG}_pmc_${C}_ex256.txt
done
if [ "$2" == "sq" ]; then   # the extractor alone; the whole step's counters come from tools/pmc_step.sh below
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 $R/tools/extract_times.py 256 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $OUT/${TAG}_pmc_sq > $OUT/${TAG}_pmc_sq_ex256.txt
  echo "sq done"
fi
echo "pmc done"; date
python3 $R/tools/kernel_times_serial.py 256 > $OUT/${TAG}_serial_kernel_times.txt 2>&1
python3 $R/tools/extract_times.py 256 > $OUT/${TAG}_extract_times.txt 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_bf -- python3 $R/tools/bruteforce_bench.py 256 1000 50 > $OUT/${TAG}_bruteforce.txt 2>&1
find $OUT/${TAG}_bf -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bruteforce_kernel_stats.csv \;
python3 $R/tools/make_traffic_table.py $TAG > /dev/null 2>&1
timeout -k 10 300 python3 $R/bench.py --host-input --no-cpu-baseline > $OUT/${TAG}_bench_host_input.json 2> $OUT/${TAG}_bench_host_input.err
for c in synth720p kitti_stereo local_ba dropin; do echo "bench $c"; timeout -k 10 400 python3 $R/bench.py --config $c > $OUT/${TAG}_bench_$c.json 2> $OUT/${TAG}_bench_$c.err; done
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_bf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_step_FETCH_SIZE $OUT/${TAG}_pmc_step_WRITE_SIZE
timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --steps 60 --all-kernel-events --timeline $OUT/${TAG}_timeline.txt > $OUT/${TAG}_bench256_all_events.json 2> /dev/null
bash $R/tools/lba_profile.sh $TAG > /dev/null 2>&1
bash $R/tools/pmc_step.sh $TAG 1024 > /dev/null 2>&1
ls $OUT | grep ${TAG}
