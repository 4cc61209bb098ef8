#!/bin/bash
# rocprofv3 kernel trace of a short default bench run, reduced by tools/trace_step.py:  tools/prof_step.sh <tag> [bench args]
# (environment switches of the library are inherited; output gpurun_out/<tag>/ + gpurun_out/<tag>_step.txt)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-parity --no-f16-leg "$@" > $out/bench.json 2> $out/rocprof.err
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_step.py $f > gpurun_out/${tag}_step.txt
python3 tools/trace_step.py $f --full > gpurun_out/${tag}_full.txt
head -3 gpurun_out/${tag}_step.txt
