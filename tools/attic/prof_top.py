#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run (its rocpd sqlite output): python tools/prof_top.py results.db [steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(top_kernels)")]
rows = [dict(zip(cols, r)) for r in cur.execute("select * from top_kernels")]
tot = sum(r["total_duration"] for r in rows)
print(f"total kernel time {tot / 1e6:.3f} ms over {steps} steps = {tot / 1e6 / steps:.3f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    n = re.sub(r"\(.*", "", n)[:78]
    print(f"{n:80s} calls/step {r['total_calls'] / steps:6.1f}  {r['total_duration'] / 1e6 / steps:8.3f} ms/step  avg {r['average'] / 1e3:8.1f} us  {r['percentage']:5.1f}%")
