#!/usr/bin/env python3
"""bench.py — frames/s of the front-end hot path on synthetic 640x480 replay (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of `--batch` frames that are already resident in HBM:
  ORB extract (8-level pyramid, FAST cells, quadtree, orientation, blur, rBRIEF)   stream A
  LSD + LBD line extract, in PLP_BENCH_LINE_SPLIT (2) sub-blocks                   streams B1, B2 (concurrent, as the
                                                                                   reference runs two threads per frame)
  halo exchange (RCCL all-gather of the two-frame feature tails), then             stream C
  match_current_and_last_frames  (frame b against frame b-1, margin 20, orientation check)
  match_frame_and_landmarks      (frame b against the key points of frames b-1 and b-2 as ~2K local landmarks, margin 10)
  match_current_and_last_frames_line (frame b's key lines against frame b-1's, margin 20)
  match_frame_and_landmarks_line (frame b's key lines against those of frames b-1 and b-2 as local line landmarks, margin 10)
The step itself lives in structure-plp-slam_amd/replay_step.py (class tracker_step) so that tests/test_gpu_bench_step.py checks, frame by
frame against the oracle, exactly the code that is timed here; `--verify N` repeats that check on N frames of the last timed step.
Steps are software-pipelined (stream C matches step n while A/B extract step n+1); every step's work is inside the timed
region.  Frames shard across ranks in contiguous blocks, the only exchange being that halo (weak scaling: every rank
owns its own batch); the timed region is bracketed by a barrier + synchronize and the max over ranks is used.  Rank 0
prints ONE JSON line.  `roofline` = the dominant kernel's algorithmic bytes per launch / its HIP-event duration
measured on its launch stream; `cpu_baseline` = the oracle restatement of the same work on this box's host cores.
"""
import argparse
import importlib
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# A hardware queue per stream (the runtime's default is four; the PCIe-inclusive pass drives seven streams, the resident step four: the resident
# number does not move with this setting, profiles/r03_scheduling_experiments.md).  Read by the HIP runtime when it initialises: set before torch.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# multi-process runs (one rank per GPU over RCCL): the host driver supports dmabuf IPC only; exported on the GPU boxes already, kept here for any other launcher
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
SHIFT_X = -3.0          # the replay window slides +3 px per frame, so scene points move -3 px


def level_pixels(rows, cols, sf, n_levels):
    s, out = np.float32(1.0), []
    for l in range(n_levels):
        if l:
            s = np.float32(np.float32(sf) * s)
        out.append((int(round(cols * 1.0 / float(s))) if l else cols) * (int(round(rows * 1.0 / float(s))) if l else rows))
    return out


def algorithmic_bytes(rows, cols, n_levels, mean_kp, mean_cand, mean_lines, mean_len, mean_raw):
    """SURVEY.md §8(d) per-frame figures, split per stage (bytes the algorithm inherently moves once)."""
    px = level_pixels(rows, cols, 1.2, n_levels)
    P, P0 = sum(px), rows * cols
    return {
        "pyramid": (P - px[-1]) + (P - px[0]),          # read every source level once, write every resized level
        "fast_cells": P + 4 * mean_cand,                 # read each level once, write packed candidates
        "blur7": 2 * P,                                  # read + write every level
        "quadtree": 3 * 4 * mean_cand + 4 * mean_kp,     # candidates in, keys/indices once, selection out
        "orient_rbrief": mean_kp * (749 + 512 + 28 + 32),
        "lsd_blur11_resize": 2 * P0 + 1.25 * P0,         # 11x11 blur r/w + half-res resize
        "lsd_gradient_bins": 0.25 * P0 * (1 + 8),        # read u8, write f32 angle + f32 norm
        # SURVEY 8(d) gives "ordering / growing" ONE figure, 0.25 P0 (8 + 1): each pixel read once + its used flag.  It is charged to region growing
        # (the dominant kernel); the seed ordering -- stable counting sort or the replay of std::sort -- has no bytes of its own in that budget
        # (its time is in stage_ms_per_batch, its traffic in profiles/), so that the per-stage figures add up to the 8(d) total.
        "lsd_grow": 0.25 * P0 * (8 + 1) + 16 * mean_raw, # each pixel's angle once + used flag, segments out
        "keylines": mean_raw * (16 + 68),
        "lbd_blur5_sobel": 2 * P0 + P0 * (1 + 4),        # 5x5 blur r/w, Sobel read u8 write 2 x s16
        "lbd": mean_lines * (63 * mean_len * 4 + 68 + 32),
        "line_finalize": mean_lines * (68 + 32 + 24) * 2,
    }


def step_profile_fields(bytes_per_frame, B, dominant):
    """in-step launch time of the dominant kernel and the register-time bound of the step, from profiles/step_profile.json (committed; same command, same batch)"""
    out = {"in_step_launch_ms": None, "in_step_frac": None, "occupancy_bound": None, "step_profile_source": None}
    try:
        sp = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "step_profile.json")))
    except (OSError, ValueError):
        return out
    if sp.get("batch") != B or sp.get("dominant", "").split("::")[-1] != {"lsd_grow": "k_lsd_grow"}.get(dominant, dominant):
        return out
    ins = {int(k): v for k, v in sp.get("in_step_launches", {}).items() if not v.get("isolated")}
    if ins:
        frames, v = max(ins.items(), key=lambda kv: kv[1]["launches"])
        out["in_step_launch_ms"] = {"frames_per_launch": frames, "mean_ms": v["mean_ms"], "launches_in_the_trace": v["launches"]}
        out["in_step_frac"] = round(bytes_per_frame * frames / (v["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)
    ob = sp.get("occupancy_bound")
    if ob:
        out["occupancy_bound"] = {k: ob[k] for k in ("register_cycles", "ideal_ms", "ms_per_step_of_that_run", "packing", "shares", "lds") if k in ob}
    out["step_profile_source"] = sp.get("source")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=2048, help="frames per rank per step (region growing is one wave per frame: 2048 frames put two of them on every SIMD; larger batches give the same throughput)")
    ap.add_argument("--keypoints", type=int, default=1000, help="Feature.max_num_keypoints (TUM RGB-D YAML: 1000)")
    ap.add_argument("--distinct", type=int, default=64, help="distinct frames of the synthetic pan that fill the batch (repeated): 64 is what every round measured; "
                    "--distinct 2048 makes every frame of a 2048-frame step its own (the pan crosses its 2 561-pixel canvas 2.4 times)")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive pass and the single-frame latency pass (profiling runs)")
    ap.add_argument("--no-latency", action="store_true", help="two calls instead of 256 in the single-frame latency pass")
    ap.add_argument("--orb-only", action="store_true", help="time the ORB extractor alone (config 1 shape)")
    ap.add_argument("--seed-order", choices=("libstdcxx", "stable"), default="libstdcxx", help="LSD seed order: the reference's (std::sort as libstdc++ implements it, replayed on the "
                    "device; the library's default) or the cheaper stable order (PLP_SEED_ORDER_STABLE); the other one is timed beside it (`other_seed_order`)")
    ap.add_argument("--verify", type=int, default=64, help="after the timed region: re-derive N frames of the last step (features and the four matcher "
                    "results) with the CPU oracle and compare (0 = off; rank 0, N = 1 only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("PLP_BENCH_SHARE_GPU"):      # diagnostic: all ranks on GPU 0 over gloo, to exercise the N > 1 plumbing on a 1-GPU box
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    plp = importlib.import_module("structure-plp-slam_amd")
    synth = importlib.import_module("structure-plp-slam_amd.synth")

    B, K = args.batch, args.keypoints
    uniq = max(3, min(B, args.distinct))   # distinct frames of the pan, repeated to fill the batch (64 by default: every round's figure is on that workload)
    frames_np = synth.replay(1234 + rank, uniq, args.rows, args.cols)
    d_frames = torch.from_numpy(frames_np).to(dev)
    if uniq < B:
        d_frames = d_frames.repeat((B + uniq - 1) // uniq, 1, 1)[:B].contiguous()
    rs = importlib.import_module("structure-plp-slam_amd.replay_step")
    HALO = rs.HALO
    serial = bool(os.environ.get("PLP_BENCH_SERIAL"))                            # diagnostic: one stream for everything
    ts = rs.tracker_step(plp, B, K, args.rows, args.cols, device_index=local_rank, orb_only=args.orb_only,
                         n_line=int(os.environ.get("PLP_BENCH_LINE_SPLIT", "2")), nbuf=int(os.environ.get("PLP_BENCH_NBUF", "2")), serial=serial,
                         shift=(SHIFT_X, 0.0), parts=os.environ.get("PLP_BENCH_PARTS", "orb,lines,match"),   # PLP_BENCH_PARTS: diagnostic, time a subset of the step
                         seed_order=plp.SEED_ORDER_STABLE if args.seed_order == "stable" else plp.SEED_ORDER_LIBSTDCXX,
                         line_depth=int(os.environ.get("PLP_BENCH_LINE_DEPTH", "1")),
                         halo_mode=os.environ.get("PLP_BENCH_HALO", "ring"))      # PLP_BENCH_HALO=allgather: the step's one exchange as an all-gather (the collective north_star names)
    cap, lcap, NBUF = ts.cap, ts.lcap, ts.NBUF
    kps2, desc2, cnt2, kl2, lbd2, fn2, lcnt2 = ts.kps2, ts.desc2, ts.cnt2, ts.kl2, ts.lbd2, ts.fn2, ts.lcnt2
    d_kps, d_desc, d_cnt = kps2[0][HALO:], desc2[0][HALO:], cnt2[0][HALO:]
    d_kl, d_lbd, d_fn, d_lcnt = kl2[0][HALO:], lbd2[0][HALO:], fn2[0], lcnt2[0][HALO:]
    m1, n1, m2, n2, m3, n3, m4, n4 = ts.m1, ts.n1, ts.m2, ts.n2, ts.m3, ts.n3, ts.m4, ts.n4
    ex, lts = ts.ex, ts.lts
    lt = lts[0] if lts else None
    mt_last, mt_lm = ts.mt_last, ts.mt_lm
    grid, sf, cur, sA, sBs = ts.grid, ts.sf, ts.cur, ts.sA, ts.sBs
    match_stage = ts.match_stage

    def step():
        return ts.step(d_frames)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    last_buf = 0
    for _ in range(args.steps):
        last_buf = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ts.last_batch_status()
    # ---- parity of the timed step itself: N frames of the LAST timed step (features + the four matcher results) against the CPU oracle.
    # After the timed region; the oracle is the checker here, never part of what is measured (tests/bench_step_check.py).
    # At N > 1 EVERY rank checks its own block (at most 8 frames each: the oracle runs on the host cores all ranks share) -- frames 0 and 1, whose
    # predecessors came over the halo exchange, among them -- and the halo rows themselves against the predecessor rank's last two frames, which it
    # regenerates from that rank's seed; the mismatch counts are all-reduced, so one bad rank fails the line.
    verified = verified_halo = None
    if args.verify > 0 and uniq >= 3:
        import bench_step_check as BC
        n_ver = min(args.verify, B) if world == 1 else min(args.verify, B, 8)
        ids = np.unique(np.concatenate([[0, 1], np.linspace(0, B - 1, n_ver).astype(np.int64)]))[:max(n_ver, 2)]   # frames 0 and 1 read the halo
        h = BC.fetch(ts, last_buf)
        g6 = BC.O.grid6(ts.grid)
        stable = args.seed_order == "stable"
        bad = []
        for b in ids:
            bad += BC.check_frame(h, int(b), K, g6, ts.shift, sf, frames_np[int(b) % uniq], args.orb_only, stable_order=stable)
        prev = (rank - 1) % world
        prev_np = frames_np if prev == rank else synth.replay(1234 + prev, uniq, args.rows, args.cols)
        bad += BC.check_halo(h, K, [prev_np[(B - HALO + j) % uniq] for j in range(HALO)], args.orb_only, stable_order=stable)
        counts = [len(bad), len(ids), HALO]
        if dist is not None:
            t = torch.tensor(counts, dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            counts = [int(v) for v in t.tolist()]
        if bad:
            print(json.dumps({"error": "bench.py --verify: the timed step differs from the oracle", "rank": rank, "mismatches": bad[:8]}), file=sys.stderr)
        if counts[0]:
            sys.exit(3)
        verified, verified_halo = counts[1], counts[2]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fps = world * B * args.steps / elapsed
    # ---- the same steps with the OTHER seed order, for the side-by-side figure: same buffers, same streams
    other = None
    if not args.orb_only and ts.lts:
        o_name = "stable" if args.seed_order == "libstdcxx" else "libstdcxx"
        for l_ in ts.lts:
            l_.set_seed_order(plp.SEED_ORDER_STABLE if o_name == "stable" else plp.SEED_ORDER_LIBSTDCXX)
        k2 = max(2, args.steps // 2)
        for _ in range(2):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        barrier()
        e2 = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([e2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e2 = float(t.item())
        other = {"seed_order": o_name, "value": round(world * B * k2 / e2, 1), "ms_per_step": round(1e3 * e2 / k2, 4), "steps": k2}
        for l_ in ts.lts:
            l_.set_seed_order(plp.SEED_ORDER_STABLE if args.seed_order == "stable" else plp.SEED_ORDER_LIBSTDCXX)
        step(); barrier()      # the feature buffers hold the headline mode's results again (the passes below read them)

    # ---- per-kernel HIP-event timing (separate, synchronous pass) -> roofline of the dominant kernel
    mean_kp = float(d_cnt.float().mean().item())
    n_prof = 3
    ex.set_profiling(True)
    for _ in range(n_prof):
        ex.extract_batch(d_frames, d_kps, d_desc, d_cnt, stream=cur)
    stage_ms, _ = ex.stage_times_ms()
    ex.set_profiling(False)
    stage_ms.pop("batch_total"); stage_ms.pop("l0_copy")
    mean_lines = mean_raw = mean_len = 0.0
    if not args.orb_only:
        lt.set_profiling(True)
        for _ in range(n_prof):
            lt.extract_batch(d_frames, d_kl, d_lbd, d_fn, d_lcnt, stream=cur)
        lms, _ = lt.stage_times_ms()
        lt.set_profiling(False)
        lms.pop("batch_total")
        stage_ms.update(lms)
        mean_lines = float(d_lcnt.float().mean().item())
        mean_raw = float(np.mean([len(lt.debug_read(lt.DBG_RAW, f)) for f in range(min(B, 8))]))
        kl0 = d_kl[0, :max(int(d_lcnt[0].item()), 1)].cpu().numpy().view(plp.KL_DTYPE)
        mean_len = float(kl0["numOfPixels"].mean()) if len(kl0) else 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(sA):
            e0.record(sA)
            for _ in range(n_prof):
                match_stage()
            e1.record(sA)
        torch.cuda.synchronize(dev)
        stage_ms["match_4x"] = e0.elapsed_time(e1) / n_prof
        match_dbg = {"last_frame": mt_last.debug_counters()[:2].tolist(), "landmarks": mt_lm.debug_counters()[:2].tolist()}
    nl = ex.get_num_scale_levels()
    mean_cand = float(sum(len(ex.debug_read(ex.DBG_CANDIDATES, l, 0)) for l in range(nl)))
    per_frame = algorithmic_bytes(args.rows, args.cols, nl, mean_kp, mean_cand, mean_lines, mean_len, mean_raw)
    n_match_q = 3 * mean_kp + 3 * mean_lines            # last-frame + 2 x landmark queries, points and lines
    # 32 M + 60 C bytes, C = candidates whose descriptor distance is computed.  C is COUNTED here (it was assumed to be 15): for eight
    # frames of the step, every query of the two point matchers against the key points of its target frame -- inside the query's square window (margin x the scale
    # factor of its level, data/common.cc get_keypoints_in_cell) and inside the matcher's level range -- on the host from the step's own arrays.
    cand_per_query = None
    if not args.orb_only:
        hk = kps2[last_buf][:HALO + 8].cpu().numpy().view(plp.KP_DTYPE).reshape(HALO + 8, cap)
        hc = cnt2[last_buf][:HALO + 8].cpu().numpy()
        sf_np = np.asarray(sf, np.float32)
        tot_c = tot_q = 0
        for b in range(8):
            t = hk[HALO + b][:hc[HALO + b]]
            tx, ty, tl = t["x"].astype(np.float64), t["y"].astype(np.float64), t["octave"].astype(np.int64)
            for back, margin, lo, hi in ((1, 20.0, -1, 1), (1, 10.0, -1, 0), (2, 10.0, -1, 0)):     # last-frame matcher (levels l-1 .. l+1); landmarks of frames b-1, b-2 (l-1 .. l)
                q = hk[HALO + b - back][:hc[HALO + b - back]]
                qx, qy, ql = q["x"].astype(np.float64) + back * SHIFT_X, q["y"].astype(np.float64), q["octave"].astype(np.int64)
                r = margin * sf_np[ql].astype(np.float64)
                for i0 in range(0, len(q), 256):
                    sl = slice(i0, i0 + 256)
                    inside = (np.abs(tx[None, :] - qx[sl, None]) < r[sl, None]) & (np.abs(ty[None, :] - qy[sl, None]) < r[sl, None])
                    inside &= (tl[None, :] >= ql[sl, None] + lo) & (tl[None, :] <= ql[sl, None] + hi)
                    tot_c += int(inside.sum())
                tot_q += len(q)
        cand_per_query = tot_c / max(tot_q, 1)
    per_frame["match_4x"] = n_match_q * 32 + n_match_q * (cand_per_query if cand_per_query is not None else 15) * (32 + 28)
    kern = {k: v for k, v in stage_ms.items() if k in per_frame and v > 0}
    dominant = max(kern, key=kern.get)
    launches = {"pyramid": 7, "lsd_blur11_resize": 2, "lsd_gradient_bins": 2, "lbd_blur5_sobel": 2, "match_4x": 8}.get(dominant, 1)
    dom_bytes = per_frame[dominant] * B
    achieved = dom_bytes / (kern[dominant] * 1e-3) / 1e9
    # HBM-side bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.py; separate FETCH_SIZE /
    # WRITE_SIZE runs of this same command at the same batch).  Only quoted when the batch matches.
    traffic = traffic_source = None
    stage_kernel = {"lsd_grow": "plp::k_lsd_grow", "fast_cells": "plp::k_fast_cells", "quadtree": "plp::k_quadtree", "lbd": "plp::k_lbd",
                    "orient_rbrief": "plp::k_orient_rbrief", "blur7": "plp::k_blur7", "pyramid": "plp::k_resize_linear",
                    "match_4x": "plp::k_match_topk_cells"}
    try:
        pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")))
        if pmc.get("batch") == B and dominant in stage_kernel:
            traffic = pmc["traffic_bytes_per_launch"].get(stage_kernel[dominant])
            traffic_source = pmc.get("source", "profiles/pmc_traffic.json") if traffic is not None else None
    except (OSError, ValueError):
        pass
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                # `traffic` is NOT measured by this run: it is the PMC figure of the committed profile named here (same command, same batch)
                "traffic_source": traffic_source,
                # whole path: algorithmic bytes of one frame through all stages x frames/s / peak
                "path_bytes_per_frame": int(sum(per_frame.values())), "path_frac": round(sum(per_frame.values()) * fps / world / 1e9 / HBM_PEAK_GBS, 6),
                "launch_ms": round(kern[dominant] / launches, 4), "bytes_per_launch": int(dom_bytes / launches),
                # The figures above describe an ISOLATED, synchronous pass.  Inside the overlapped step the dominant kernel runs per line sub-block beside the other
                # streams' kernels: its launch duration there and the fraction that follows come from the committed kernel trace of this command (profiles/step_profile.json,
                # tools/step_profile.py); `occupancy_bound` = the step against the two resources a resident wave holds while it waits -- registers (waves x cycles x VGPRs of all
                # its kernels against 1024 SIMDs x 512 registers x 2.4 GHz) and, under `lds`, LDS bytes against 256 CUs x 160 KB.  Quoted from the committed profile, not measured by this run (as `traffic`).
                **step_profile_fields(per_frame[dominant], B, dominant),
                "match_candidates_per_query_counted": None if cand_per_query is None else round(cand_per_query, 2),
                "stage_ms_per_batch": {k: round(v, 4) for k, v in stage_ms.items()},
                "stage_GBps": {k: round(per_frame[k] * B / (kern[k] * 1e-3) / 1e9, 1) for k in kern}}

    # ---- beside the headline (HBM-resident) number: (b) the same steps fed from pinned host memory with the results brought
    # back, copies on their own streams inside the timed region; (c) latency of the synchronous single-frame entry points, the
    # calls tracking_module makes once per frame (timing boundary of run_tum_rgbd_slam_with_line.cc:95-105)
    extras = {}
    if not args.no_extras and not args.orb_only and world == 1:
        h_frames = torch.empty((B, args.rows, args.cols), dtype=torch.uint8, pin_memory=True)
        h_frames.copy_(d_frames.cpu())
        n_stage = max(2, int(os.environ.get("PLP_BENCH_STAGES", "2")))
        stage = [torch.empty_like(d_frames) for _ in range(n_stage)]
        # What comes back: the features that exist, not the capacity they were allotted (the reference's extract() hands back vectors of exactly
        # that many entries).  Per step the live rows of every padded array are packed on the device (plp_pack_rows_device: offsets = prefix
        # sum of the per-frame counts) and the device-to-host copies move exactly offsets[B] rows; their sizes are known on the host one
        # step later (the offsets come back first), so the bulk copy of step n is issued after step n + 1 has been enqueued.
        rp = ts.replay
        pt_arrays = lambda buf: [(kps2[buf][HALO:], 28), (desc2[buf][HALO:], 32), (m1.view(B, cap, 1), 4), (m2.view(B, cap, 1), 4)]
        ln_arrays = lambda buf: [(kl2[buf][HALO:], 68), (lbd2[buf][HALO:], 32), (fn2[buf], 24), (m3.view(B, lcap, 1), 4), (m4.view(B, lcap, 1), 4)]
        flat = lambda nbytes, **kw: torch.empty(nbytes, dtype=torch.uint8, **kw)
        d_pk = [[flat(B * (cap if i < 4 else lcap) * rb, device=dev) for i, (_, rb) in enumerate(pt_arrays(0) + ln_arrays(0))] for _ in range(NBUF)]
        h_pk = [[flat(t.numel(), pin_memory=True) for t in d_pk[0]] for _ in range(NBUF)]
        d_off = [torch.empty((2, B + 1), dtype=torch.int64, device=dev) for _ in range(NBUF)]
        h_off = [torch.empty((2, B + 1), dtype=torch.int64, pin_memory=True) for _ in range(NBUF)]
        small = lambda buf: [cnt2[buf][HALO:], lcnt2[buf][HALO:], n1, n2, n3, n4]
        h_small = [[torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in small(0)] for _ in range(NBUF)]
        sH, sD, sD2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        stage_free = [None] * n_stage; down_done = [None] * NBUF; ev_off = [None] * NBUF
        frames_default = d_frames
        d2h_bytes = []

        def finish_download(buf):
            """the bulk device-to-host copy of the step that filled set `buf`: its sizes are on the host once the offsets have arrived"""
            ev_off[buf].synchronize()
            tot_pt, tot_ln = int(h_off[buf][0, B]), int(h_off[buf][1, B])
            nbytes = 0
            with torch.cuda.stream(sD2):
                sD2.wait_event(ev_off[buf])
                for i, (_, rb) in enumerate(pt_arrays(buf) + ln_arrays(buf)):
                    nb = (tot_pt if i < 4 else tot_ln) * rb
                    if nb:
                        h_pk[buf][i][:nb].copy_(d_pk[buf][i][:nb], non_blocking=True)
                    nbytes += nb
                down_done[buf] = torch.cuda.Event(); down_done[buf].record(sD2)
            d2h_bytes.append(nbytes + h_off[buf].numel() * 8 + sum(t.numel() * t.element_size() for t in h_small[buf]))

        pending = []
        pack_done = [None]

        # The next step's upload is enqueued before the host waits for this step's offsets (PLP_BENCH_PREFETCH=0: at the start of its own step).
        # This pass drives seven streams: on the runtime's default of four hardware queues an 11 ms copy queued early sits in front of another
        # stream's kernels and the early upload LOSES (31.8 against 28.7 ms per step); with a queue per stream (GPU_MAX_HW_QUEUES=8, set at the top of
        # this file) it wins (27.7 ms).  profiles/r03_pcie.md
        prefetch = os.environ.get("PLP_BENCH_PREFETCH", "1") == "1"
        uploaded = {}

        def upload(n):
            """enqueue the host-to-device copy of step n's frames (once): it starts as soon as the extractors that last read this staging buffer are through"""
            if n in uploaded:
                return
            sb = n % n_stage
            for ev_ in stage_free[sb] or ():
                sH.wait_event(ev_)                                           # (the matchers never read pixels)
            with torch.cuda.stream(sH):
                stage[sb].copy_(h_frames, non_blocking=True)
                uploaded[n] = torch.cuda.Event(); uploaded[n].record(sH)

        def host_step(n, last=False):
            nonlocal d_frames
            sb = n % n_stage
            upload(n)
            ev = uploaded.pop(n)
            for s_ in [sA] + sBs:
                s_.wait_event(ev)
            buf = ts.step_no % NBUF
            if down_done[buf] is not None:
                sA.wait_event(down_done[buf])
                for s_ in sBs:
                    s_.wait_event(down_done[buf])
            d_frames = stage[sb]
            if pack_done[0] is not None:
                ts.sC.wait_event(pack_done[0])                              # m1..m4 / n1..n4 are ONE set of buffers: this step's matchers must not overwrite them
                                                                             # before the previous step's pack kernels and count copies have read them
            step()
            d_frames = frames_default
            sD.wait_event(ts.done_match[buf])
            with torch.cuda.stream(sD):                                      # pack this step's live rows; offsets, counts and match counts go first
                for i, (t, rb) in enumerate(pt_arrays(buf)):
                    rp.pack_rows(plp, t, cnt2[buf][HALO:], d_pk[buf][i], d_off[buf][0], i == 0, sD)
                for i, (t, rb) in enumerate(ln_arrays(buf)):
                    rp.pack_rows(plp, t, lcnt2[buf][HALO:], d_pk[buf][4 + i], d_off[buf][1], i == 0, sD)
                h_off[buf].copy_(d_off[buf], non_blocking=True)
                for h, t in zip(h_small[buf], small(buf)):
                    h.copy_(t, non_blocking=True)
                ev_off[buf] = torch.cuda.Event(); ev_off[buf].record(sD)
                pack_done[0] = ev_off[buf]
            stage_free[sb] = list(ts.extract_events)
            if prefetch and not last:
                upload(n + 1)                                                # before the host blocks below: the next step's frames travel while this step computes
            while pending:                                                   # the previous step's bulk copy, now that this step keeps the GPU busy
                finish_download(pending.pop(0))
            pending.append(buf)
        for n in range(2):
            host_step(n, last=n == 1)
        while pending:
            finish_download(pending.pop(0))
        barrier()
        d2h_bytes.clear()
        t0 = time.perf_counter()
        for n in range(args.steps):
            host_step(n, last=n == args.steps - 1)                           # exactly args.steps uploads inside the timed region
        while pending:
            finish_download(pending.pop(0))
        barrier()
        el = time.perf_counter() - t0
        extras["pcie_inclusive_value"] = round(B * args.steps / el, 1)
        extras["pcie_inclusive_ms_per_step"] = round(1e3 * el / args.steps, 4)
        extras["pcie_bytes_per_step"] = {"h2d": int(h_frames.numel()), "d2h": int(np.mean(d2h_bytes)) if d2h_bytes else 0,
                                         "d2h_padded_arrays_would_be": int(sum(t.numel() for t in d_pk[0]) + 6 * B * 4)}
        # the link itself: one step's frames host-to-device and one step's packed results device-to-host, each alone on an idle GPU
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(sH):
            e0.record(sH)
            for _ in range(3):
                stage[0].copy_(h_frames, non_blocking=True)
            e1.record(sH)
            for _ in range(3):
                for i in range(len(h_pk[0])):
                    h_pk[0][i].copy_(d_pk[0][i], non_blocking=True)
            e2.record(sH)
        torch.cuda.synchronize(dev)
        h2d_ms = e0.elapsed_time(e1) / 3
        extras["pcie_link"] = {"h2d_GBps": round(h_frames.numel() / h2d_ms / 1e6, 1), "d2h_GBps": round(sum(t.numel() for t in d_pk[0]) / (e1.elapsed_time(e2) / 3) / 1e6, 1),
                               "h2d_ms_per_step_alone": round(h2d_ms, 3),
                               "note": "pinned host memory; the frames of one step take h2d_ms_per_step_alone on an idle link -- the floor of the PCIe-inclusive step"}
        # the packed copy of the last step against the padded arrays (frame 0 and the last frame): the packing moves what it should
        lb = (ts.step_no - 1) % NBUF
        o = h_off[lb][0]
        k_first = kps2[lb][HALO].cpu().numpy()[:int(o[1])]
        assert np.array_equal(h_pk[lb][0][:int(o[1]) * 28].numpy().reshape(-1, 28), k_first), "packed key points differ from the padded array"
        d_last = desc2[lb][HALO + B - 1].cpu().numpy()[:int(o[B] - o[B - 1])]
        assert np.array_equal(h_pk[lb][1][int(o[B - 1]) * 32:int(o[B]) * 32].numpy().reshape(-1, 32), d_last), "packed descriptors differ from the padded array"
        # ... and one match array (the buffers every step shares: the pass's last step wrote m1, nothing has overwritten it since)
        m_first = m1[0].cpu().numpy()[:int(o[1])]
        assert np.array_equal(h_pk[lb][2][:int(o[1]) * 4].numpy().view(np.int32), m_first), "packed matches differ from the padded array"
        del h_frames, stage, h_pk, d_pk
        # single-frame latency, 256 calls each, the host-pointer entry points on one frame at a time
        lat = {}
        n_lat = 2 if args.no_latency else 256      # --no-latency: a token pass (A/B runs of the PCIe arrangement)
        imgs = [np.ascontiguousarray(frames_np[i % uniq]) for i in range(n_lat)]

        def timed(fn):
            ts = []
            for i in range(n_lat):
                t_ = time.perf_counter(); fn(i); ts.append(time.perf_counter() - t_)
            return round(1e3 * float(np.median(ts)), 4), round(1e3 * float(np.mean(ts)), 4)
        feats = {}

        def run_orb(i):
            feats["kd"] = ex.extract(imgs[i])
        ex.extract(imgs[0]); lt.extract_LSD_LBD(imgs[0])
        lat["orb_extract"] = timed(run_orb)
        lat["line_extract"] = timed(lambda i: lt.extract_LSD_LBD(imgs[i]))
        k0, d0 = ex.extract(imgs[0]); k1, d1 = ex.extract(imgs[1])
        q = dict(q_valid=np.ones(len(k0), np.uint8), q_reproj=np.stack([k0["x"] + np.float32(SHIFT_X), k0["y"]], 1).astype(np.float32),
                 q_x_right=np.full(len(k0), -1, np.float32), q_level=k0["octave"].astype(np.int32), q_angle=k0["angle"].astype(np.float32), q_desc=d0,
                 q_has_obs=np.ones(len(k0), np.uint8))
        t_ = dict(t_kps=k1, t_desc=d1, t_x_right=np.full(len(k1), -1, np.float32), t_occupied=np.zeros(len(k1), np.uint8))
        lat["match_current_and_last_frames"] = timed(lambda i: mt_last.match_host(plp.MODE_LAST_FRAME, len(k1), len(k0), {**t_, **q}, margin=20.0, direction=0, scale_factors=sf, grid=grid))
        lat["match_frame_and_landmarks"] = timed(lambda i: mt_lm.match_host(plp.MODE_LANDMARKS, len(k1), len(k0), {**t_, **q}, margin=10.0, scale_factors=sf, grid=grid))
        import threading

        def run_pair(i):            # ORB || LSD in two threads, as data/frame.cc:691-694
            th = threading.Thread(target=lambda: lt.extract_LSD_LBD(imgs[i]))
            th.start(); ex.extract(imgs[i]); th.join()
        lat["orb_par_line_extract"] = timed(run_pair)
        extras["latency_ms_median_mean"] = lat
        extras["latency_note"] = f"{n_lat} synchronous single-frame calls each through the host-pointer C ABI (plp_orb_extract, plp_line_extract, plp_match_host), {uniq} distinct frames"

    what = ("ORB extract only" if args.orb_only else
            "ORB extract || LSD+LBD extract, then match_current_and_last_frames + match_frame_and_landmarks (~2K landmarks) + match_current_and_last_frames_line + match_frame_and_landmarks_line")
    out = {
        "metric": "frames/sec ORB+LSD extract+match, 640x480 TUM-RGBD, 1/2/4/8 GPU",
        "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"TUM-RGBD-shaped replay {args.cols}x{args.rows} (BASELINE configs[1]): {what}; K={K}, 8 levels, 1.2; "
                               "BoW matchers not included",
                   "frames_per_rank_per_step": B, "distinct_frames": uniq, "keypoints_mean": round(mean_kp, 1), "lines_mean": round(mean_lines, 1),
                   "matches_mean": [round(float(n1.float().mean().item()), 1), round(float(n2.float().mean().item()), 1)] + ([] if args.orb_only else [round(float(n3.float().mean().item()), 1), round(float(n4.float().mean().item()), 1)]),
                   "match_rescans_rounds": (match_dbg if not args.orb_only else None),
                   "sharding": f"contiguous frame blocks per rank; ONE exchange per step and rank of the 2-frame feature halo the matchers read -- key points, descriptors, key lines, LBD "
                               f"rows and both count arrays packed into one record per frame ({ts.halo.record_bytes} bytes), moved by "
                               + ("one send to the successor + one receive from the predecessor (ring)" if ts.halo_mode == "ring" else "one all_gather_into_tensor of the packed tails")
                               + (" over RCCL" if world > 1 and not os.environ.get("PLP_BENCH_SHARE_GPU") else (" over gloo (diagnostic: all ranks on one GPU)" if world > 1 else "; a single rank copies its own tail")),
                   "halo_mode": ts.halo_mode, "halo_bytes_per_rank_per_step": ts.halo.bytes_per_step},
        "roofline": roofline,
        # LSD seed order of the headline number (the reference's: std::sort as libstdc++ implements it) and the same steps in the other order
        "seed_order": args.seed_order, "other_seed_order": other,
        # frames of the LAST TIMED step whose features and four matcher results were recomputed by the CPU oracle and found identical
        # (at N > 1: summed over the ranks, every rank checks its own block; `verified_halo_rows` = the rows that came over the halo exchange, checked against
        # the oracle's extraction of the predecessor rank's last two frames)
        "verified_frames": verified, "verified_halo_rows": verified_halo,
    }
    out.update(extras)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU baseline is reported at N = 1 only
        import ctypes as C
        import oracle_lib as O
        cores = os.cpu_count() or 1
        try:
            import psutil
            phys = psutil.cpu_count(logical=False) or cores
        except Exception:
            phys = cores
        tot = (C.c_long * 3)()
        if args.orb_only:
            n_cpu = max(cores * 3, 24)
            sample = np.ascontiguousarray(np.tile(frames_np, ((n_cpu + uniq - 1) // uniq, 1, 1))[:n_cpu])
            sec = O.lib().oracle_orb_time_frames(sample.ctypes.data_as(C.c_void_p), n_cpu, args.rows, args.cols, K, cores, tot)
            out["cpu_baseline"] = {"value": round(n_cpu / sec, 1), "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": f"{n_cpu} frames of the same replay on {cores} threads (oracle restatement, g++ -O2 strict FP)"}
        else:
            # three thread configurations x two builds of the oracle restatement (strict FP = the parity build;
            # -O3 -ffast-math -mtune=native = the reference's Release flags, CMakeLists.txt:57-58, timing only)
            libs = {"strict_O2": O.lib()}
            fast = pathlib.Path(O.ORACLE_DIR) / "liboracle_fast.so"
            if fast.exists():
                libs["ref_flags_O3_fastmath"] = C.CDLL(str(fast))
            cfgs = {}
            for tag, L in libs.items():
                fn = L.oracle_front_time_frames2
                fn.restype = C.c_double
                fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
                # one worker per PHYSICAL core is the fair many-core figure (one per SMT thread oversubscribes the FP units: its per-thread
                # ORB time is ~15x the single-thread one and the total swings by 30 % between runs); both are reported
                runs = [("1_thread", 1, 0, 8), ("ref_2_threads_per_frame", 1, 1, 8), ("physical_cores", phys, 0, max(phys * 3, 24))]
                if cores != phys:
                    runs.append(("all_threads", cores, 0, max(cores * 3, 24)))
                for name, threads, pair, n_cpu in runs:
                    sample = np.ascontiguousarray(np.tile(frames_np, ((n_cpu + uniq - 1) // uniq, 1, 1))[:n_cpu])
                    st3 = (C.c_double * 3)()
                    # the many-core figures swing by +- 20 % between runs: three runs, the MEDIAN is reported (all three kept in `runs_frames_per_s`)
                    reps = 3 if threads > 1 else 1
                    secs = sorted(fn(sample.ctypes.data_as(C.c_void_p), n_cpu, args.rows, args.cols, K, threads, SHIFT_X, pair, st3, tot) for _ in range(reps))
                    sec = secs[len(secs) // 2]
                    cfgs[f"{tag}/{name}"] = {"frames_per_s": round(n_cpu / sec, 2), "runs_frames_per_s": [round(n_cpu / t, 2) for t in secs], "frames": n_cpu, "threads": threads * (2 if pair else 1),
                                             "thread_ms_per_frame": {"orb": round(1e3 * st3[0] / n_cpu, 2), "lines" if not pair else "orb_par_lines_wall": round(1e3 * st3[1] / n_cpu, 2),
                                                                     "match": round(1e3 * st3[2] / n_cpu, 2)}}
            best = max((k for k in cfgs if k.endswith("/physical_cores") or k.endswith("/all_threads")), key=lambda k: cfgs[k]["frames_per_s"])
            out["cpu_baseline"] = {"value": cfgs[best]["frames_per_s"], "unit": "frames/s", "cores": cfgs[best]["threads"], "physical_cores": phys, "hardware_threads": cores, "kind": "port",
                                   "sample": f"{cfgs[best]['frames']} frames of the same replay, contiguous blocks on {cfgs[best]['threads']} threads ({best.split('/')[1]}: the faster many-core configuration, median of three runs), same stages incl. all four matcher calls: the oracle RESTATEMENT "
                                             f"({best.split('/')[0]}), not the reference's OpenCV build, which is not available here", "configs": cfgs}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
