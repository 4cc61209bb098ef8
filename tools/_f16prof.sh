cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pr in bf16 f16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_f16prof/$pr -o s -- python3 tools/sample_bench.py --model DiT-XL/2 --n 128 --steps 10 --precision $pr > /dev/null 2>&1
  f=$(ls gpurun_out/r04_f16prof/$pr/*/s_kernel_stats.csv gpurun_out/r04_f16prof/$pr/s_kernel_stats.csv 2>/dev/null | head -1)
  echo "== $pr"; head -7 $f | cut -c1-200 | awk -F'","' '{print $1, $2, $4}'
done
for pr in bf16 f16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_f16prof/t$pr -o s -- python3 bench.py --steps 5 --warmup 3 --precision $pr --no-cpu-baseline --no-parity --no-f16-leg > /dev/null 2>&1
  f=$(ls gpurun_out/r04_f16prof/t$pr/*/s_kernel_stats.csv gpurun_out/r04_f16prof/t$pr/s_kernel_stats.csv 2>/dev/null | head -1)
  echo "== train $pr"; head -14 $f | cut -c1-200 | awk -F'","' '{print $1, $2, $4}'
done
