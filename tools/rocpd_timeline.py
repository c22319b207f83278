#!/usr/bin/env python3
"""Timeline of the overlapped steps from a rocprofv3 --kernel-trace rocpd database: every dispatch of the timed steps (the LAST `--last-ms`
milliseconds before the profiling passes begin is not known here, so: the window is chosen by dispatch count -- the step launches
k_fast_cells exactly once, so step i spans [i-th k_fast_cells start, (i+1)-th k_fast_cells start)) with queue, start and duration, plus per
1-ms slice how many kernels were resident and which.  Reading it answers what the step's critical path is (which stream is the long one,
what runs beside k_lsd_grow, where the GPU idles).
usage: tools/rocpd_timeline.py <results.db> [step index, default 2] > timeline.md"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
step = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_c = "name" if "name" in cols else "kernel_name"
s_c = "start" if "start" in cols else "start_timestamp"
e_c = "end" if "end" in cols else "end_timestamp"
q_c = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
rows = list(cur.execute(f"select {name_c}, {s_c}, {e_c}, {q_c or 0} from kernels order by {s_c}"))
short = lambda n: n.split("(")[0].replace("void ", "").replace("plp::", "").split("<")[0][:28]
fc = [r[1] for r in rows if "k_fast_cells" in r[0]]
if len(fc) <= step + 1:
    print("not enough steps in the trace", len(fc)); sys.exit(1)
t0, t1 = fc[step], fc[step + 1]
win = [r for r in rows if r[2] > t0 and r[1] < t1]
queues = sorted({r[3] for r in win})
print(f"# step {step}: {(t1 - t0) / 1e6:.2f} ms between two k_fast_cells launches; {len(win)} dispatches on {len(queues)} queues (columns: table `kernels` = {cols})\n")
# per queue: how much of the step some kernel of that queue was executing (union of its dispatches' intervals), and the sum of their durations
print("| queue | busy (ms) | busy / step | sum of kernel durations (ms) | kernels |")
print("|---|---:|---:|---:|---|")
for qi, q in enumerate(queues):
    iv = sorted((max(s, t0), min(e, t1)) for n, s, e, qq in win if qq == q)
    busy, cur_s, cur_e = 0.0, None, None
    for a, b in iv:
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    if cur_e is not None:
        busy += cur_e - cur_s
    names = {}
    for n, s, e, qq in win:
        if qq == q:
            names[short(n)] = names.get(short(n), 0.0) + (min(e, t1) - max(s, t0)) / 1e6
    top = ", ".join(f"{k} {v:.2f}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:6])
    print(f"| {qi} | {busy / 1e6:.2f} | {busy / (t1 - t0):.2f} | {sum(names.values()):.2f} | {top} |")
print()
print("| start (ms) | dur (ms) | queue | kernel |")
print("|---:|---:|---|---|")
for n, s, e, q in win:
    if (e - s) > 30000:      # > 30 us
        print(f"| {(s - t0) / 1e6:8.3f} | {(e - s) / 1e6:7.3f} | {queues.index(q)} | {short(n)} |")
print("\n## resident kernels per millisecond\n")
for ms in range(int((t1 - t0) / 1e6) + 1):
    a, b = t0 + ms * 1e6, t0 + (ms + 1) * 1e6
    res = {}
    for n, s, e, q in win:
        ov = min(e, b) - max(s, a)
        if ov > 0:
            res[short(n)] = res.get(short(n), 0) + ov / 1e6
    print(f"- {ms:3d} ms: " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(res.items(), key=lambda kv: -kv[1])))
