#!/bin/bash
# HBM-side counters of the fc1 GEMM INSIDE the training step (the isolated collection of tools/collect_profiles.sh finds the launch's A
# operand in the Infinity Cache; in the step it was written by the kernel before):  tools/pmc_in_step.sh <tag> [band]
# -> gpurun_out/<tag>_instep[_band<b>]/pmc_<counters>/step_counter_collection.csv ; summarised by tools/pmc_in_step_summary.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05}; band=${2:-}
out=gpurun_out/${tag}_instep${band:+_band$band}
mkdir -p $out
[ -n "$band" ] && export MAPDIT_GEMM_BAND=$band
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  t=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$t -o step -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-f16-leg > /dev/null 2>> $out/rocprof.err
  echo "$t done"
done
