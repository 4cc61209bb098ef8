#!/bin/bash
# Re-collect only the PMC passes of the roofline kernel (isolated + in the step) after gemm.hip changed:  tools/pmc_only.sh <tag>
# then locally: python tools/summarise_profiles.py <tag> && python tools/merge_pmc_in_step.py <tag>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05}
out=gpurun_out/${tag}_final
mkdir -p $out
sha256sum map-dit_amd/csrc/gemm.hip > $out/gemm_hip.sha256
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo $c | tr ' ' '_')
  rm -rf $out/pmc_$t
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$t -o fc1 -- python3 tools/gemm_one.py fc1 > /dev/null 2>> $out/rocprof.err
done
rm -rf $out/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/pmc_mfma -o fc1 -- python3 tools/gemm_one.py fc1 > /dev/null 2>> $out/rocprof.err
echo "isolated pmc done"
rm -rf gpurun_out/${tag}_instep gpurun_out/${tag}_instep_band6
bash tools/pmc_in_step.sh $tag
bash tools/pmc_in_step.sh $tag 6
echo "in-step pmc done"
