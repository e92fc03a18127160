#!/bin/bash
# A/B of the pose-solver instantiations (VIORB_POSE_MP="P,WPP"): parity tests, then the bench line. usage: tools/pose_mp_ab.sh OUTDIR "cfg cfg ..."
out=${1:-gpurun_out/pose_mp}; cfgs=${2:-"1,4 2,2 4,2 4,1 2,4"}; mkdir -p $out
for c in $cfgs; do
  n=$(echo $c | tr ',' '_')
  VIORB_POSE_MP=$c timeout -k 10 600 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_tracker.py tests/test_gpu_native_tracker.py -x -q -m gpu -k "pose_opt or tracker or batched" > $out/test_$n.txt 2>&1
  echo "cfg $c tests: $(tail -1 $out/test_$n.txt)"
done
for c in $cfgs; do
  n=$(echo $c | tr ',' '_')
  VIORB_POSE_MP=$c timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra-passes > $out/bench_$n.json 2> $out/bench_$n.err
  python - "$out/bench_$n.json" "$c" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print("cfg", sys.argv[2], "frames/s", d["value"], "ms/step", d["ms_per_step"], "pose us", d.get("roofline_pose",{}).get("avg_launch_us"), "fast us", d["roofline"]["avg_launch_us"])
except Exception as e: print("cfg", sys.argv[2], "bench failed", e)
PY
done
