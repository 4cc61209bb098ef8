#!/usr/bin/env python3
"""Turn what tools/collect_profiles.sh <tag> left under gpurun_out/<tag>_final/ into the committed summaries under profiles/:
<tag>_kernel_stats_bench_steps5.csv (+ _batch32), <tag>_fc1_pmc_traffic.json (HBM bytes per launch of the roofline kernel, corrected as
MI355X_MICROARCH.md prescribes, tied to the sha256 of the gemm.hip it was collected on), <tag>_bench_lines.jsonl.
    python tools/summarise_profiles.py [tag = r03]"""
import csv
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
SRC = os.path.join(ROOT, "gpurun_out", TAG + "_final")
DST = os.path.join(ROOT, "profiles")


def counter(tag, names):
    rows = list(csv.DictReader(open(os.path.join(SRC, f"pmc_{tag}", "fc1_counter_collection.csv"))))
    out = {}
    for n in names:
        vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == n and ("gemm_mfma256_kernel" in r["Kernel_Name"] or "gemm_mfma256w_kernel" in r["Kernel_Name"])]
        out[n] = sum(vals) / max(len(vals), 1)
        out[n + "_launches"] = len(vals)
    return out


shutil.copy(os.path.join(SRC, "prof", "bench_kernel_stats.csv"), os.path.join(DST, TAG + "_kernel_stats_bench_steps5.csv"))
shutil.copy(os.path.join(SRC, "prof32", "bench32_kernel_stats.csv"), os.path.join(DST, TAG + "_kernel_stats_bench_batch32_steps10.csv"))
fetch = counter("FETCH_SIZE", ["FETCH_SIZE"])
write = counter("WRITE_SIZE", ["WRITE_SIZE"])
tcc = counter("TCC_HIT_sum_TCC_MISS_sum", ["TCC_HIT_sum", "TCC_MISS_sum"])
mf = counter("mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"])
sha_file = os.path.join(SRC, "gemm_hip.sha256")
sha = open(sha_file).read().split()[0] if os.path.exists(sha_file) else hashlib.sha256(open(os.path.join(ROOT, "map-dit_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()
M, D, H = 65536, 768, 3072
read_b = fetch["FETCH_SIZE"] * 1024 * 2          # KiB -> bytes; gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads
write_b = write["WRITE_SIZE"] * 1024
d = {
    "kernel": "gemm_mfma256w_kernel<0,0,EpiSilu2GradT<true>> NT [65536,768]x[3072,768]^T (fc1: activation + derivative factor, both bf16)",
    "command": "rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 tools/gemm_one.py fc1  (one pass per counter "
               "group, 5 launches each, mean; tools/collect_profiles.sh)",
    "model": "DiT-B/2", "per_gpu_batch": 256, "gemm_hip_sha256": sha,
    "FETCH_SIZE_KiB": fetch["FETCH_SIZE"], "WRITE_SIZE_KiB": write["WRITE_SIZE"],
    "TCC_HIT_sum": tcc["TCC_HIT_sum"], "TCC_MISS_sum": tcc["TCC_MISS_sum"],
    "corrections": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM); "
                   "WRITE_SIZE exact for 16-B/lane stores",
    "read_bytes_corrected": read_b, "write_bytes": write_b, "hbm_bytes_per_launch": read_b + write_b,
    "algorithmic_bytes_per_launch": 2 * (M * D + H * D + 2 * M * H),
    "l2_hit_rate": tcc["TCC_HIT_sum"] / max(tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"], 1),
    "mfma_busy_over_sq_busy": mf["SQ_VALU_MFMA_BUSY_CYCLES"] / max(32 * mf["SQ_BUSY_CYCLES"], 1),
    "SQ_VALU_MFMA_BUSY_CYCLES": mf["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_BUSY_CYCLES": mf["SQ_BUSY_CYCLES"], "SQ_WAVE_CYCLES": mf["SQ_WAVE_CYCLES"],
}
json.dump(d, open(os.path.join(DST, TAG + "_fc1_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: d[k] for k in ("hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "l2_hit_rate", "mfma_busy_over_sq_busy")}))
with open(os.path.join(DST, TAG + "_bench_lines.jsonl"), "w") as f:
    for name in ("bench_default", "bench_b128", "bench_b64", "bench_b32", "bench_b128_emulated_zero1w", "bench_b64_emulated_zero1w",
                 "bench_b32_emulated_zero1w", "bench_mp_off", "bench_rotation", "bench_S2", "bench_XL2_b64", "bench_f16", "bench_under_rocprof",
                 "sample_XL2_bf16", "sample_XL2_f16", "sample_B2_f16"):
        p = os.path.join(SRC, name + ".json")
        if os.path.exists(p) and os.path.getsize(p) > 10:
            line = [ln for ln in open(p) if ln.startswith("{")]
            if line:
                f.write(json.dumps({"run": name, **json.loads(line[-1])}) + "\n")
for name in ("S2", "XL2_b64", "XL2_sample", "B2_rotation"):              # kernel statistics of the other BASELINE configurations
    for cand in (os.path.join(SRC, f"prof_{name}", f"{name}_kernel_stats.csv"),):
        if os.path.exists(cand):
            shutil.copy(cand, os.path.join(DST, f"{TAG}_kernel_stats_{name}.csv"))
for src, dst in (("prof32e/bench32e_kernel_stats.csv", "_kernel_stats_bench_batch32_emulated_zero1w.csv"), ("step_by_dispatch.txt", "_step_by_dispatch.txt")):
    if os.path.exists(os.path.join(SRC, src)):
        shutil.copy(os.path.join(SRC, src), os.path.join(DST, TAG + dst))
print(f"wrote profiles/{TAG}_*")
