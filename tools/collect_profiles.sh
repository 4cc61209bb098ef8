#!/bin/bash
# Round-end measurements on the GPU box: bench JSON lines, rocprofv3 kernel stats of the bench command, PMC passes for the roofline
# kernel (fc1 GEMM).  Everything lands under gpurun_out/<tag>_final/ (tag = $1, default r03); tools/summarise_profiles.py <tag>
# turns it into profiles/<tag>_*.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r05}
out=gpurun_out/${tag}_final
mkdir -p $out
sha256sum map-dit_amd/csrc/gemm.hip > $out/gemm_hip.sha256
python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench default done: $(tail -c 300 $out/bench_default.json | head -c 200)"
for b in 128 64 32; do python bench.py --steps 30 --warmup 5 --batch-per-gpu $b --no-cpu-baseline --no-parity --no-f16-leg > $out/bench_b$b.json 2>/dev/null; done
# round 5: the compute side of sharded weight passes at the same per-GPU batches (one process = rank 0 of 256 / b ranks, no collectives)
for b in 128 64 32; do python bench.py --steps 30 --warmup 5 --batch-per-gpu $b --no-cpu-baseline --no-parity --no-f16-leg --emulate-world $((256 / b)) > $out/bench_b${b}_emulated_zero1w.json 2>/dev/null; done
python bench.py --steps 20 --warmup 5 --mp-off mp_silu,mp_residual,mp_pos_enc,mp_embedding > $out/bench_mp_off.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --rotation-modulation > $out/bench_rotation.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --model DiT-S/2 --no-cpu-baseline > $out/bench_S2.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --model DiT-XL/2 --batch-per-gpu 64 --no-cpu-baseline > $out/bench_XL2_b64.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --precision f16 --no-cpu-baseline > $out/bench_f16.json 2>/dev/null
python tools/sample_bench.py --model DiT-XL/2 --n 128 --steps 20 > $out/sample_XL2_bf16.json 2>/dev/null
python tools/sample_bench.py --model DiT-XL/2 --n 128 --steps 20 --precision f16 > $out/sample_XL2_f16.json 2>/dev/null
python tools/sample_bench.py --model DiT-B/2 --n 64 --steps 30 --precision f16 > $out/sample_B2_f16.json 2>/dev/null
echo "benches done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bench -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-parity --no-f16-leg > $out/bench_under_rocprof.json 2> $out/rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof32 -o bench32 -- python3 bench.py --steps 10 --warmup 3 --batch-per-gpu 32 --no-cpu-baseline --no-parity --no-f16-leg > $out/bench32_under_rocprof.json 2>> $out/rocprof.err
for cfg in "S2 bench.py --steps 10 --warmup 3 --model DiT-S/2 --no-cpu-baseline --no-parity --no-f16-leg" \
           "XL2_b64 bench.py --steps 6 --warmup 2 --model DiT-XL/2 --batch-per-gpu 64 --no-cpu-baseline --no-parity --no-f16-leg" \
           "XL2_sample tools/sample_bench.py --model DiT-XL/2 --n 128 --steps 10" \
           "B2_rotation bench.py --steps 8 --warmup 3 --rotation-modulation --no-cpu-baseline --no-f16-leg"; do
  set -- $cfg; name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o $name -- python3 "$@" > $out/${name}_under_rocprof.json 2>> $out/rocprof.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof32e -o bench32e -- python3 bench.py --steps 10 --warmup 3 --batch-per-gpu 32 --emulate-world 8 --no-cpu-baseline --no-parity --no-f16-leg > $out/bench32e_under_rocprof.json 2>> $out/rocprof.err
python3 tools/trace_step.py $out/prof/bench_kernel_trace.csv > $out/step_by_dispatch.txt 2>> $out/rocprof.err
python3 tools/trace_step.py $out/prof/bench_kernel_trace.csv --full > $out/step_by_dispatch_full.txt 2>> $out/rocprof.err
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$tag -o fc1 -- python3 tools/gemm_one.py fc1 > /dev/null 2>> $out/rocprof.err
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $out/pmc_mfma -o fc1 -- python3 tools/gemm_one.py fc1 > /dev/null 2>> $out/rocprof.err
echo "pmc done"; ls -R $out | head -60
