#!/usr/bin/env python3
"""Per-kernel durations of the dispatches that did NOT overlap any other dispatch (the per-stage timing passes of bench.py run
on one stream with nothing beside them), from a rocprofv3 --kernel-trace rocpd sqlite database.
usage: tools/rocpd_isolated.py <results.db> [title]"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
cur = db.cursor()
rows = None
for q in ("select name, start, end from kernels", "select kernel_name, start, end from kernels", "select name, start_timestamp, end_timestamp from kernels"):
    try:
        rows = list(cur.execute(q)); break
    except sqlite3.Error:
        continue
if rows is None:
    print("schema not recognised; tables / views:")
    for r in cur.execute("select type, name, sql from sqlite_master"): print(r[0], r[1], (r[2] or "")[:300].replace("\n", " "))
    sys.exit(1)
rows.sort(key=lambda r: r[1])
n = len(rows)
iso = defaultdict(list); allk = defaultdict(list)
max_end_before = 0
for i, (name, s, e) in enumerate(rows):
    short = name.split("(")[0].replace("void ", "")[:60]
    allk[short].append(e - s)
    nxt = rows[i + 1][1] if i + 1 < n else 1 << 62
    if s >= max_end_before and e <= nxt: iso[short].append(e - s)
    max_end_before = max(max_end_before, e)
print(f"# dispatches without any overlap: {title}\n")
print("| kernel | isolated calls | mean (us) | min (us) | all calls | mean of all (us) |")
print("|---|---:|---:|---:|---:|---:|")
for k in sorted(allk, key=lambda k: -sum(iso.get(k, [0]))):
    a = iso.get(k, [])
    if not a: continue
    print(f"| `{k}` | {len(a)} | {sum(a)/len(a)/1e3:.1f} | {min(a)/1e3:.1f} | {len(allk[k])} | {sum(allk[k])/len(allk[k])/1e3:.1f} |")
