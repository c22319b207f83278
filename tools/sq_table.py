#!/usr/bin/env python3
"""Per-kernel SQ counter table from a rocprofv3 --pmc rocpd database (see tools/run_pmc_sq.sh)."""
import re, sqlite3, sys
from collections import defaultdict
cur = sqlite3.connect(sys.argv[1]).cursor()
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0   # bench steps inside the profile (warmup + timed + 3 profiling passes are separate launches)
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for name, cn, val, dur in cur.execute("select kernel_name,counter_name,value,duration from counters_collection"):
    s = re.sub(r"\(.*", "", name).replace("void ", "").strip()[:40]
    acc[s][cn] += val
    if cn == 'SQ_WAVES': cnt[s] += 1; acc[s]['dur'] += dur
print("| kernel | launches | ms/launch | waves | VALU/wave | SALU/wave | LDS/wave | cycles/wave | issuing % | waiting % | issue-stalled % | VALU issue ms/launch (1024 SIMDs x 2.4 GHz, 2 cycles per wave64 instruction: MI355X_MICROARCH.md, v_fma_f32) |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
tot = 0
for s, a in sorted(acc.items(), key=lambda kv: -kv[1]['dur']):
    if not s.startswith('plp'): continue
    n = cnt[s]; w = a['SQ_WAVES'] or 1; wc = a['SQ_WAVE_CYCLES'] or 1
    valu_ms = a['SQ_INSTS_VALU'] / n * 2 / 1024 / 2.4e9 * 1e3   # a LOWER bound of the VALU time: f64, transcendental and 32-bit integer multiplies take longer
    print(f"| `{s}` | {n} | {a['dur']/n/1e6:.3f} | {w/n:.0f} | {a['SQ_INSTS_VALU']/w:.0f} | {a['SQ_INSTS_SALU']/w:.0f} | {a['SQ_INSTS_LDS']/w:.0f} | {4*wc/w:.0f} | {100*a['SQ_ACTIVE_INST_ANY']/wc:.1f} | {100*a['SQ_WAIT_ANY']/wc:.1f} | {100*a['SQ_WAIT_INST_ANY']/wc:.1f} | {valu_ms:.2f} |")
