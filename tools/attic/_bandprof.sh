cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 3 12 1; do
  MAPDIT_GEMM_BAND=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_band/b$v -o bench -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity --no-f16-leg > /dev/null 2>&1
  f=$(ls gpurun_out/r04_band/b$v/*/bench_kernel_stats.csv gpurun_out/r04_band/b$v/bench_kernel_stats.csv 2>/dev/null | head -1)
  echo "band=$v: $(grep -E 'EpiQkvHeads|EpiSilu2GradT|EpiMulAux' $f | awk -F'","' '{gsub(/"/,"",$0); split($0,a,","); printf "%s ", $0}' | grep -o 'Epi[A-Za-z0-9<>]*[^,]*,[0-9]*,[0-9]*,[0-9.]*' | awk -F, '{printf "%s avg %.1f us | ", $1, $4/1000}')"
done
