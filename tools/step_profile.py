#!/usr/bin/env python3
"""Digest of a round profile for bench.py's `roofline` object: what the dominant kernel takes INSIDE the overlapped step, and the step against the
one resource the round-4 model found binding -- register-time.

    python tools/step_profile.py <kernel-trace results.db> <SQ-counter results.db> --bench <bench.json of the same tree> --out profiles/step_profile.json --source profiles/<tag>_*.md

in_step: durations of `plp::k_lsd_grow` from the kernel trace of `bench.py --steps 3 --warmup 1`, split by launch size (the step launches it per line sub-block of
1024 frames, two in flight; the isolated stage passes of the same run launch it once over 2048 frames).
occupancy_bound: sum over the step's kernels of waves x cycles per wave x allocated VGPRs (SQ_WAVES, SQ_WAVE_CYCLES of the counter pass, one launch = the whole batch;
VGPRs from the dispatch records of the counter pass -- arch + accumulation registers as allocated; the build's resource remarks where a database lacks them -- rounded up
to the allocation granule of 8) x launches per step = register-cycles per step; the chip offers 1024 SIMDs x 512 VGPRs x 2.4e9 cycles/s; ideal_ms = the step if the register
file were packed perfectly, packing = ideal_ms / measured ms per step.  occupancy_bound.lds: the same sum with the LDS a wave's workgroup holds (lds_block_size / waves per
workgroup) against 256 CUs x 160 KB: the second resource a resident wave occupies while it waits.
(Until the end of round 6 the remarks were matched by SUBSTRING and the maximum taken: k_lsd_grow was booked with k_lsd_grow_mw's 170 registers instead of its own 111,
which put the ideal at 17.6 ms and the packing at 0.77; with each kernel's own allocation: 15.1 ms, 0.67.)"""
import argparse, glob, json, os, re, sqlite3, sys
from collections import defaultdict

LAUNCHES_PER_STEP = {"plp::k_resize_linear": 7}          # every other kernel: 1 (the counter pass runs the line path unsplit); the matcher kernels: 2 (below)
CHIP = 1024 * 512 * 2.4e9
CHIP_LDS = 256 * 160 * 1024 * 2.4e9   # bytes of LDS x cycles per second


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


def vgprs():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "structure-plp-slam_amd", "csrc", "build")
    out = {}
    for path in glob.glob(os.path.join(here, "*.remarks")):
        name = None
        for line in open(path, errors="replace"):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1); continue
            m = re.search(r"\sVGPRs: (\d+)", line)
            if m and name:
                out[name] = int(m.group(1))
    return out


def bounds(per_kernel, ms):
    """register-time and LDS-time of a step against what the chip offers"""
    total = sum(v["register_cycles"] for v in per_kernel.values())
    total_lds = sum(v.get("lds_byte_cycles", 0.0) for v in per_kernel.values())
    ideal = total / CHIP * 1e3
    ob = {"register_cycles": total, "ideal_ms": round(ideal, 3), "ms_per_step_of_that_run": ms, "packing": round(ideal / ms, 4),
          "shares": {k: round(v["register_cycles"] / total, 4) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1]["register_cycles"])[:8]}}
    if total_lds:   # the same sum for the other resource a resident wave holds while it waits: LDS (256 CUs x 160 KB)
        ideal_lds = total_lds / CHIP_LDS * 1e3
        ob["lds"] = {"lds_byte_cycles": total_lds, "ideal_ms": round(ideal_lds, 3), "packing": round(ideal_lds / ms, 4),
                     "shares": {k: round(v.get("lds_byte_cycles", 0.0) / total_lds, 4) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1].get("lds_byte_cycles", 0.0))[:6]}}
    return ob


def recompute(path):
    """--recompute <step_profile.json>: the bounds of a stored table again, with the registers the CURRENT build's remarks give each kernel"""
    d = json.load(open(path)); V = vgprs()
    for k, rec in d["per_kernel"].items():
        v = vg_of(V, k)
        if v is not None:
            rec["vgprs"] = v
        rec["register_cycles"] = rec["waves"] * rec["cycles_per_wave"] * rec["vgprs"] * rec["launches_per_step"]
    d["occupancy_bound"] = bounds(d["per_kernel"], d["occupancy_bound"]["ms_per_step_of_that_run"])
    json.dump(d, open(path, "w"), indent=1)
    print(json.dumps(d["occupancy_bound"], indent=1))


def vg_of(V, kernel):
    # the kernel's own entry of the resource remarks, by its MANGLED name: <length><identifier> (+ IL<type><value>E for a template argument).  (Until the end of round 6 this
    # matched substrings and took the maximum: k_lsd_grow got k_lsd_grow_mw's 170 registers instead of its 111, the matchers' <1> variants the largest instantiation's.)
    base = kernel.split("<")[0].split("::")[-1]
    key = f"{len(base)}{base}"
    targ = re.search(r"<(\d+)>", kernel)
    pat = re.compile(r"(?<!\d)" + key + (r"IL[a-z]" + targ.group(1) + "E" if targ else r"E"))
    cands = [v for k, v in V.items() if pat.search(k) and ("ss_thr" in k) == ("ss_thr" in kernel) and ("ss_lat" in k) == ("ss_lat" in kernel) and "debug" not in k]
    assert len(cands) <= 1, (kernel, cands)
    return (cands[0] + 7) // 8 * 8 if cands else None


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--recompute":
        return recompute(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("kt_db"); ap.add_argument("sq_db"); ap.add_argument("--bench", required=True); ap.add_argument("--out", required=True); ap.add_argument("--source", default=None)
    ap.add_argument("--dominant", default="plp::k_lsd_grow")
    a = ap.parse_args()
    bench = json.load(open(a.bench))
    B = bench["config"]["frames_per_rank_per_step"]
    # ---- in-step launches of the dominant kernel, by launch size
    cur = sqlite3.connect(a.kt_db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    size_col = next((c for c in ("grid_size", "grid_size_x", "grid_x", "grid") if c in cols), None)
    name_col = "name" if "name" in cols else "kernel_name"
    s_col, e_col = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = list(cur.execute(f"select {name_col}, {s_col}, {e_col}" + (f", {size_col}" if size_col else "") + " from kernels"))
    by_size = defaultdict(list)
    for r in rows:
        if short(r[0]) == a.dominant:
            by_size[int(r[3]) if size_col else 0].append((r[2] - r[1]) / 1e6)
    in_step = {}
    if by_size:
        sizes = sorted(by_size)
        full = sizes[-1]                                   # the isolated passes launch the whole batch
        for sz in sizes:
            frames = B * sz // full if full else B
            in_step[str(frames)] = {"launches": len(by_size[sz]), "mean_ms": round(sum(by_size[sz]) / len(by_size[sz]), 4), "isolated": sz == full and len(sizes) > 1}
    # ---- register-time of a step
    V = vgprs()
    vg = lambda kernel: vg_of(V, kernel)
    cur = sqlite3.connect(a.sq_db).cursor()
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int); disp = {}
    ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    has_disp = all(c in ccols for c in ("workgroup_size", "lds_block_size", "vgpr_count", "accum_vgpr_count"))
    q = "select kernel_name, counter_name, value" + (", workgroup_size, lds_block_size, vgpr_count, accum_vgpr_count" if has_disp else "") + " from counters_collection"
    for row in cur.execute(q):
        name, cn, val = row[:3]
        s = short(name)
        acc[s][cn] += val
        if cn == "SQ_WAVES":
            cnt[s] += 1
        if has_disp:   # what the dispatch itself was given: registers per lane (arch + accumulation, allocated), LDS per workgroup
            disp[s] = {"wg": row[3], "lds": row[4], "vgprs": row[5] + row[6]}
    per_kernel = {}
    for s, c in acc.items():
        if not s.startswith("plp::") or s.endswith("k_lsd_order") or not cnt[s]:
            continue
        waves = c["SQ_WAVES"] / cnt[s]
        cyc_per_wave = 4 * c["SQ_WAVE_CYCLES"] / max(c["SQ_WAVES"], 1)
        v = vg(s)
        # (the dispatch records' vgpr_count is NOT used: on this target it comes back as about half of what the kernel descriptor allots -- 56 for k_lsd_grow's 111 -- except
        # where it does not (k_match_prep: 56 for 55); the build's resource remarks are the compiler's own statement of the allocation)
        if v is None:
            continue
        launches = LAUNCHES_PER_STEP.get(s, 2 if "k_match_" in s else 1)
        rc = waves * cyc_per_wave * v * launches
        per_kernel[s] = {"waves": round(waves), "cycles_per_wave": round(cyc_per_wave), "vgprs": v, "launches_per_step": launches, "register_cycles": rc}
        if s in disp:
            lds_per_wave = disp[s]["lds"] / max(1, (disp[s]["wg"] + 63) // 64)
            lc = waves * cyc_per_wave * lds_per_wave * launches
            per_kernel[s].update({"lds_bytes_per_workgroup": disp[s]["lds"], "waves_per_workgroup": (disp[s]["wg"] + 63) // 64, "lds_byte_cycles": lc})
    out = {"batch": B, "source": a.source, "dominant": a.dominant, "in_step_launches": in_step, "per_kernel": per_kernel}
    out["occupancy_bound"] = bounds(per_kernel, bench["ms_per_step"])
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("in_step_launches", "occupancy_bound")}, indent=1))


if __name__ == "__main__":
    main()
