#!/bin/bash
# A/B of two builds of the library on the block's GEMM shapes, interleaved on one box:  tools/ab_gemm.sh <other .so> [rounds]
other=$1
for r in 1 2; do
  MAPDIT_LIB=$other timeout -k 10 300 python tools/gemm_bench.py --rounds ${2:-3} > gpurun_out/ab_other_$r.log 2>&1
  timeout -k 10 300 python tools/gemm_bench.py --rounds ${2:-3} > gpurun_out/ab_this_$r.log 2>&1
done
python - <<'PY'
import re
def load(f):
    d = {}
    for ln in open(f):
        m = re.match(r"(.{20})\s+(\S+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m: d[(m.group(1).strip(), m.group(2))] = float(m.group(4))
    return d
o = [load(f"gpurun_out/ab_other_{r}.log") for r in (1, 2)]
t = [load(f"gpurun_out/ab_this_{r}.log") for r in (1, 2)]
print(f"{'case':22s} {'kernel':>8s} {'other':>7s} {'this':>7s} {'other':>7s} {'this':>7s}   TFLOP/s")
for k in o[0]:
    print(f"{k[0]:22s} {k[1]:>8s} {o[0][k]:7.0f} {t[0].get(k, 0):7.0f} {o[1].get(k, 0):7.0f} {t[1].get(k, 0):7.0f}")
PY
