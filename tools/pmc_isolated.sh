#!/bin/bash
# The isolated half of tools/collect_profiles.sh alone: PMC passes over tools/gemm_one.py fc1 -> gpurun_out/<tag>_final/pmc_*
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r04}
out=gpurun_out/${tag}_final
mkdir -p $out
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_TCC_HIT_sum_TCC_MISS_sum $out/pmc_mfma
sha256sum map-dit_amd/csrc/gemm.hip > $out/gemm_hip.sha256
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$t -o fc1 -- python3 tools/gemm_one.py fc1 > /dev/null 2>> $out/rocprof.err
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/pmc_mfma -o fc1 -- python3 tools/gemm_one.py fc1 > /dev/null 2>> $out/rocprof.err
echo "isolated pmc done"; cat $out/gemm_hip.sha256; find $out/pmc_mfma -name "*counter_collection.csv" | head -2
