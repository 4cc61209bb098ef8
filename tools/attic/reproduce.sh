#!/usr/bin/env bash
# Re-creates every measurement quoted in DESIGN.md §5 on a 1-GPU MI355X box (run from the repo root; ~10 GPU-minutes).
# Outputs go to gpurun_out/; the summaries the repo tracks are copies of these files under profiles/.
set -euo pipefail
export PYTHONPATH=.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
python -m pytest tests -q -m gpu                                                   # parity through the C ABI
python bench.py                                   > gpurun_out/bench_default.json  # headline: DiT-B/2, 256 / GPU
(cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD" && rocprofv3 --kernel-trace --stats --output-format csv \
    -d gpurun_out/prof -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do                     # fc1 HBM traffic, one pass per counter group
    (cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD" && rocprofv3 --pmc $c --kernel-trace --output-format csv \
        -d "gpurun_out/pmc_$(echo $c | tr ' ' '_')" -o p -- python3 tools/gemm_one.py fc1 > /dev/null 2>&1)
done
python tools/gemm_bench.py                        > gpurun_out/gemm_bench.log      # GEMM shapes of one block, kernel variants
python tools/vendor_gemm_ref.py                   > gpurun_out/vendor_gemm.log     # reference point: torch.matmul on the same shapes
python tools/gemm_stamps.py --build && python tools/gemm_stamps.py > gpurun_out/gemm_stamps.log   # K-loop timeline (instrumented build)
python tools/sample_bench.py --steps 20           > gpurun_out/sample_xl2.log      # BASELINE config 5: DiT-XL/2 sampling step
for m in "DiT-S/2 256" "DiT-L/2 128" "DiT-XL/2 64" "DiT-XL/2 128" "DiT-B/2 32" "DiT-B/2 64" "DiT-B/2 128"; do
    set -- $m
    python bench.py --model "$1" --batch-per-gpu "$2" --steps 10 --warmup 3 --no-cpu-baseline >> gpurun_out/other_models.jsonl
done
python bench.py --precision bf16x3 --batch-per-gpu 64 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_bf16x3.json   # parity engine's speed
python tools/precision_trajectory.py --model DiT-S/2 --batch 64 --steps 300 > gpurun_out/precision_trajectory.log  # bf16 vs bf16x3, 300 steps
python -m mapdit_amd.train --synthetic --model DiT-B/2 --num-steps 300 --batch-size 256 --log-every 25 --ckpt-every 1000 \
    --ema-snapshot-every 150 --results-dir /tmp/mapdit_res > gpurun_out/train300.log
# round 2
bash tools/collect_profiles.sh && python tools/summarise_profiles.py                # bench lines at 256/128/64/32, kernel stats, fc1 PMC traffic
python tools/precision_rank.py                    > gpurun_out/precision_rank.log   # which bf16 roundings carry the logits error
python tools/gemm_ablate.py --build && python tools/gemm_ablate.py > gpurun_out/gemm_ablate.log   # fill / K loop / epilogue per tile
for t in mfma_issue store_rate load_rate; do hipcc --offload-arch=gfx950 -O3 tools/$t.hip -o tools/_stamps/$t && tools/_stamps/$t > gpurun_out/$t.log; done
python tools/step_stress.py 2000 2                > gpurun_out/step_stress.log      # bit-reproducibility beside a second process on the GPU
python tools/dp_repeat.py 4                       > gpurun_out/dp_repeat.log        # two-rank runs, bit for bit
echo "done: see gpurun_out/"
