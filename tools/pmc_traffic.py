#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (rocpd sqlite output).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir_f> -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir_w> -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py <dir_f>/f_results.db <dir_w>/w_results.db --batch 2048 --md profiles/X.md --json profiles/pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read
(MI355X_MICROARCH.md, HBM section; visible in the same run: the 629 MB torch copy of the frame batch reads 307 MB), so
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch.
"""
import argparse, json, re, sqlite3
from collections import defaultdict


def per_kernel(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for name, value, dur in cur.execute("select kernel_name, value, duration from counters_collection where counter_name = ?", (counter,)):
        short = re.sub(r"\(.*", "", name).replace("void ", "").strip()
        if len(short) > 60:
            short = short[:57] + "..."
        a = acc[short]
        a[0] += 1; a[1] += value; a[2] += dur
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db"); ap.add_argument("write_db")
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--md"); ap.add_argument("--json")
    ap.add_argument("--source", default=None, help="the committed file this json is a digest of (bench.py quotes it as roofline.traffic_source)")
    a = ap.parse_args()
    f, w = per_kernel(a.fetch_db, "FETCH_SIZE"), per_kernel(a.write_db, "WRITE_SIZE")
    rows = []
    for k in f:
        nf, vf, _ = f[k]
        nw, vw, _ = w.get(k, [1, 0.0, 0.0])
        fetch, write = vf / nf * 1024, vw / max(nw, 1) * 1024
        rows.append((k, nf, fetch, write, 2 * fetch + write))
    rows.sort(key=lambda r: -r[4] * r[1])
    lines = ["| kernel | launches | FETCH_SIZE/launch (MB, raw) | WRITE_SIZE/launch (MB) | traffic/launch = 2F+W (MB) | per frame (KB) |", "|---|---|---|---|---|---|"]
    for k, n, fe, wr, tr in rows:
        lines.append(f"| `{k}` | {n} | {fe / 1e6:.2f} | {wr / 1e6:.2f} | {tr / 1e6:.2f} | {tr / a.batch / 1e3:.1f} |")
    text = "\n".join(lines)
    print(text)
    if a.md:
        open(a.md, "w").write(f"# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes), batch = {a.batch} frames\n\n"
                              "traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B: FETCH_SIZE on gfx950 counts half of a wide coalesced read "
                              "(guide: MI355X_MICROARCH.md, HBM); gather-heavy kernels may be over-corrected by up to 2x on the read side.\n\n" + text + "\n")
    if a.json:
        json.dump({"batch": a.batch, "source": a.source or a.md, "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024", "traffic_bytes_per_launch": {k: tr for k, _, _, _, tr in rows}},
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
