#!/bin/bash
# Same-box comparison of SEVERAL values of one environment switch on the default bench step:  tools/ab_multi.sh VAR rounds v1 v2 v3 ...
# (BENCH_ARGS="--batch-per-gpu 32": extra bench arguments, e.g. another per-GPU batch)
var=$1; rounds=$2; shift 2
out=gpurun_out/ab_${var}.log
mkdir -p gpurun_out; : > $out
for r in $(seq $rounds); do
  for v in "$@"; do
    line=$(env $var=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-f16-leg $BENCH_ARGS 2>/dev/null | tail -1)
    echo "$var=$v $BENCH_ARGS $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", round(d["ms_per_step"],3), "median", round(d["ms_per_step_median"],3), "fc1_ms", round(d["roofline"]["avg_launch_ms"],4))')" | tee -a $out
  done
done
