#!/usr/bin/env python3
"""One training step out of a rocprofv3 --kernel-trace CSV, dispatch by dispatch:
    python tools/trace_step.py <kernel_trace.csv> [--step K] [--full]
Picks the K-th last interval between two adam_ema_kernel dispatches (default: the last complete one) and prints
  * per kernel family AND launch geometry (grid size tells proj from fc2 etc.): launches, average / total duration;
  * the time the GPU had no kernel in flight (gaps), and the time two kernels were in flight together (side-stream overlap);
  * with --full every dispatch: start offset, duration, gap to the previous end, queue, short name.
"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"(\w+)(<.*>)?\(", n)
    if m:
        t = m.group(2) or ""
        t = t.replace("false", "0").replace("true", "1").replace(" ", "")
        return m.group(1) + t
    return n[:60]


def main():
    path = sys.argv[1]
    k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 1
    full = "--full" in sys.argv
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
                     r["Queue_Id"], int(r["VGPR_Count"])))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adam_ema_kernel" in r[2] or "adam_ema_ranges_kernel" in r[2]]
    # optimiser launches of one step may be several (EMA ranges): step boundary = gap of > 50 dispatches between adam launches
    bounds = [adam[0]] + [adam[j] for j in range(1, len(adam)) if adam[j] - adam[j - 1] > 50]
    if len(bounds) < k + 1:
        sys.exit("not enough steps in the trace")
    lo, hi = bounds[-k - 1], bounds[-k]
    step = rows[lo:hi]
    t0 = step[0][0]
    fam = defaultdict(lambda: [0, 0])
    busy_end = t0
    gaps = 0
    overlap = 0
    for i, (s, e, n, g, q, v) in enumerate(step):
        key = (short(n), g, v)
        fam[key][0] += 1
        fam[key][1] += e - s
        if s > busy_end:
            gaps += s - busy_end
        else:
            overlap += min(e, busy_end) - s
        if full:
            print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - busy_end) / 1e3:7.1f}  q{q}  grid {g:6d}  {short(n)[:90]}")
        busy_end = max(busy_end, e)
    wall = busy_end - t0
    tot = sum(v[1] for v in fam.values())
    print(f"step wall {wall / 1e6:.3f} ms, sum of kernel durations {tot / 1e6:.3f} ms, idle gaps {gaps / 1e6:.3f} ms, "
          f"time with two kernels in flight {overlap / 1e6:.3f} ms, {len(step)} dispatches")
    for (n, g, v), (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{d / 1e6:8.3f} ms  x{c:3d}  avg {d / c / 1e3:8.1f} us  grid {g:6d} vgpr {v:3d}  {n[:100]}")


if __name__ == "__main__":
    main()
