#!/usr/bin/env python3
"""Dump the per-kernel statistics of a rocprofv3 (--kernel-trace --stats) rocpd sqlite database as
markdown.  usage: tools/rocpd_summary.py <results.db> [title] > profiles/<name>.md"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(f"# rocprofv3 --kernel-trace --stats: {title}\n")
print("| kernel | calls | total (us) | average (us) | % |")
print("|---|---:|---:|---:|---:|")
for name, calls, tot, avg, pct in rows:
    short = name.split("(")[0]
    if len(short) > 90:
        short = short[:87] + "..."
    print(f"| `{short}` | {calls} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")
