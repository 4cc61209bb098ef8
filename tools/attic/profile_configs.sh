#!/bin/bash
# rocprofv3 kernel statistics for the BASELINE configurations other than the headline:
#   tools/profile_configs.sh <tag>   ->  gpurun_out/<tag>_cfg/{S2,XL2_b64,XL2_sample,B2_b32}_kernel_stats.csv + the bench lines
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r04}
out=gpurun_out/${tag}_cfg
mkdir -p $out
run() {   # name, command...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o $name -- "$@" > $out/$name.json 2> $out/$name.err
  cp $out/prof_$name/*/${name}_kernel_stats.csv $out/${name}_kernel_stats.csv 2>/dev/null || cp $out/prof_$name/${name}_kernel_stats.csv $out/${name}_kernel_stats.csv
  echo "$name: $(grep -o '"ms_per_step[^,]*' $out/$name.json | head -2 | tr '\n' ' ')"
}
run S2 python3 bench.py --steps 10 --warmup 3 --model DiT-S/2 --no-cpu-baseline --no-parity --no-f16-leg
run XL2_b64 python3 bench.py --steps 6 --warmup 2 --model DiT-XL/2 --batch-per-gpu 64 --no-cpu-baseline --no-parity --no-f16-leg
run XL2_sample python3 tools/sample_bench.py --model DiT-XL/2 --n 128 --steps 10
run B2_b32 python3 bench.py --steps 10 --warmup 3 --batch-per-gpu 32 --no-cpu-baseline --no-parity --no-f16-leg
ls $out
