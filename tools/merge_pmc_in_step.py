#!/usr/bin/env python3
"""profiles/<tag>_fc1_pmc_traffic.json: the isolated collection of tools/summarise_profiles.py merged with the in-step collection of
tools/pmc_in_step.sh (top-level figures = the launch inside the training step, which is what bench.py times).
    python tools/merge_pmc_in_step.py [tag = r04]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
path = os.path.join(ROOT, "profiles", f"{tag}_fc1_pmc_traffic.json")
iso = json.load(open(path))
if "isolated" in iso:
    iso = {**iso["isolated"], **{k: iso[k] for k in ("kernel", "model", "per_gpu_batch", "gemm_hip_sha256", "corrections", "algorithmic_bytes_per_launch")}}
def load(p):
    return json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_in_step_summary.py"), p], capture_output=True, text=True).stdout)
b3, b6 = load(os.path.join(ROOT, "gpurun_out", f"{tag}_instep")), load(os.path.join(ROOT, "gpurun_out", f"{tag}_instep_band6"))
d = {k: iso[k] for k in ("kernel", "model", "per_gpu_batch", "gemm_hip_sha256", "corrections", "algorithmic_bytes_per_launch")}
d["isolated"] = {k: v for k, v in iso.items() if k not in d}
d["in_step"] = {"command": f"tools/pmc_in_step.sh {tag}: rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 3 --warmup 2 ... (one pass per counter "
                            "group; mean over the 60 fc1 launches of the run), summarised by tools/pmc_in_step_summary.py", **b3}
d["in_step_band6"] = {"command": "the same with MAPDIT_GEMM_BAND=6 (round 3's band width)", **b6}
for k in ("read_bytes_corrected", "write_bytes", "l2_hit_rate", "mfma_busy_over_sq_busy", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "TCC_HIT_sum", "TCC_MISS_sum"):
    d[k] = b3[k]
d["FETCH_SIZE_KiB"], d["WRITE_SIZE_KiB"] = b3["FETCH_SIZE"], b3["WRITE_SIZE"]
d["hbm_bytes_per_launch"] = b3["hbm_side_bytes_per_launch"]
d["command"] = d["in_step"]["command"]
d["note"] = ("top-level figures = the launch INSIDE the training step (what bench.py times); 'isolated' = five back-to-back launches of tools/gemm_one.py "
             "as in rounds 1-3.  FETCH_SIZE counts the L2's fabric-side requests, Infinity-Cache hits included (MI355X_MICROARCH.md): the A operand "
             "was written by the kernel before with plain stores and is re-read once per band (4 bands of 3 column tiles).")
json.dump(d, open(path, "w"), indent=1)
print(json.dumps({k: d[k] for k in ("hbm_bytes_per_launch", "l2_hit_rate", "mfma_busy_over_sq_busy")}), "isolated:", json.dumps({k: d["isolated"].get(k) for k in ("hbm_bytes_per_launch", "l2_hit_rate", "mfma_busy_over_sq_busy")}))
