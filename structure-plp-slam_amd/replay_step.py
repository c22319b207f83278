"""One step of the batched replay: the unit `bench.py` times and `tests/test_gpu_bench_step.py` pins to the oracle.

A step takes B frames that are resident in HBM through the per-frame sequence of the tracker (data/frame.cc:1125-1167 for the
extraction; module/frame_tracker.cc:66-87 motion_based_track and tracking_module.cc:908-1064 search_local_landmarks[_line] for the
matcher calls), for all frames of the batch at once:

  stream A        ORB extract                                           plp_orb_extract_batch_device
  streams B1..Bn  LSD + LBD extract, the batch cut into n sub-blocks    plp_line_extract_batch_device
  stream C        ONE packed halo exchange of the two-frame tails of all feature arrays (replay.halo_exchanger), then for every frame b
                    m1  match_current_and_last_frames       frame b-1's key points moved by the pan -> frame b   margin 20, ratio 0.9, orientation check
                    m2  match_frame_and_landmarks           key points of frames b-2, b-1 as ~2K local landmarks  margin 10, ratio 0.8
                    m3  match_current_and_last_frames_line  frame b-1's key lines moved by the pan -> frame b    margin 20
                    m4  match_frame_and_landmarks_line      key lines of frames b-2, b-1 as local line landmarks  margin 10, ratio 0.8

The replay has no map: a camera that pans by `shift` pixels per frame stands in for the motion model, so the "reprojection" of a
feature of frame b-1 (b-2) is its position moved by 1 x (2 x) the shift (csrc/replay_kernels.hip builds all queries on the device;
descriptors are read in place through q_desc_stride).  Steps are software-pipelined: stream C works on step n while A and B already
extract step n + 1 into the other of the NBUF feature sets.
"""
import importlib
import os

import numpy as np

HALO = 2    # the matchers of frame b read frames b-1 and b-2: rows 0..HALO-1 of every feature array hold the predecessor rank's tail
LCAP = 512  # key lines kept per frame (a 640x480 frame yields ~50 after the >= 60 px filter)


class tracker_step:
    def __init__(self, plp, B, K, rows, cols, device_index=0, orb_only=False, n_line=2, nbuf=2, serial=False, shift=(-3.0, 0.0), parts="orb,lines,match", line_grow_waves=0, seed_order=None,
                 line_depth=1, halo_mode="ring"):
        import torch
        self.torch = torch
        self.plp, self.B, self.K, self.rows, self.cols = plp, B, K, rows, cols
        self.replay = importlib.import_module((__package__ or "structure-plp-slam_amd") + ".replay")
        self.dev = dev = torch.device("cuda", device_index)
        self.orb_only, self.shift, self.parts = orb_only, shift, parts
        self.cap, self.lcap = 2 * K + 64, LCAP
        cap, lcap = self.cap, self.lcap
        # line_depth d > 1 (experiment): d sets of line contexts and streams, step n uses set n % d -- the line chains (seed sort -> region growing ->
        # LBD: one long dependent chain per sub-block) of d consecutive steps are in flight together.  Needs a feature set more than steps in flight.
        self.line_depth = line_depth = max(1, line_depth)
        self.NBUF = NBUF = max(2, nbuf, line_depth + 1 if line_depth > 1 else 2)
        full = lambda shape, dt, zero=False: (torch.zeros if zero else torch.empty)((HALO + B,) + shape, dtype=dt, device=dev)
        # NBUF sets of outputs: the matchers of step n read set n % NBUF while the extractors of the next steps fill the others
        self.kps2 = [full((cap, 28), torch.uint8) for _ in range(NBUF)]
        self.desc2 = [full((cap, 32), torch.uint8) for _ in range(NBUF)]
        self.cnt2 = [full((), torch.int32, True) for _ in range(NBUF)]
        self.kl2 = [full((lcap, 68), torch.uint8, True) for _ in range(NBUF)]
        self.lbd2 = [full((lcap, 32), torch.uint8, True) for _ in range(NBUF)]
        self.fn2 = [torch.empty((B, lcap, 3), dtype=torch.float64, device=dev) for _ in range(NBUF)]
        self.lcnt2 = [full((), torch.int32, True) for _ in range(NBUF)]
        mk = lambda c: (torch.empty((B, c), dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
        (self.m1, self.n1), (self.m2, self.n2), (self.m3, self.n3), (self.m4, self.n4) = mk(cap), mk(cap), mk(lcap), mk(lcap)
        self.ex = plp.orb_extractor(K, device=device_index)
        # The line path is one long dependent chain per launch (region growing is a single wave per frame), so
        # the batch is cut into n_line contiguous sub-blocks, each with its own context (scratch planes) and HIP stream.
        n_line = max(1, n_line)
        while B % n_line:
            n_line -= 1
        self.n_line = n_line
        self.lts = [] if orb_only else [plp.LineFeatureTracker(device=device_index) for _ in range(n_line * line_depth)]
        for lt in self.lts:
            lt.set_grow_waves(line_grow_waves)      # 0 = automatic (several waves per frame only for batches of at most 256 frames)
            if seed_order is not None:
                lt.set_seed_order(seed_order)       # None = the library's default: the reference's std::sort order (PLP_SEED_ORDER_LIBSTDCXX)
        self.mt_last = plp.matcher(0.9, True, device=device_index)      # motion_based_track: match::projection(0.9, true)
        self.mt_lm = plp.matcher(0.8, True, device=device_index)        # search_local_landmarks: match::projection(0.8)
        self.mt_line = plp.matcher(0.9, True, device=device_index)      # motion_based_track, lines: match_current_and_last_frames_line
        self.mt_lm_line = plp.matcher(0.8, True, device=device_index)   # search_local_landmarks_line: match::projection(0.8)
        self.sf_lsd = np.ones(1, np.float32)                            # LineFeatureTracker: one LSD level (line_extractor.cc:34-35)
        self.grid = plp.make_grid(cols, rows)
        self.sf = self.ex.get_scale_factors()
        self.cur = torch.cuda.current_stream(dev)
        # Four streams = the runtime's four hardware queues (GPU_MAX_HW_QUEUES): a fifth would share a queue with one of these and its
        # kernels would wait behind that stream's.  The line streams carry the step's longest dependent chain (region growing is ~15 ms
        # however few frames it gets): line_prio < 0 gives them the higher stream priority, so their kernels are dispatched first.
        line_prio = int(os.environ.get("PLP_BENCH_LINE_PRIO", "0"))
        self.sA = torch.cuda.Stream(dev)
        self.sBs = [self.sA if serial else torch.cuda.Stream(dev, priority=line_prio) for _ in range(n_line * line_depth)]
        self.sC = self.sA if serial else torch.cuda.Stream(dev)
        self.pq = self.replay.point_queries(plp, B, cap, dev)
        self.lq = None if orb_only else self.replay.line_queries(plp, B, lcap, dev, landmarks=True)
        # ONE exchange per step: the tails of every feature array the matchers read, packed into one record per frame (replay.halo_exchanger)
        self.halo_mode = halo_mode
        self.halo = self.replay.halo_exchanger(self._halo_arrays(0), halo=HALO, mode=halo_mode)
        self.done_match = [None] * NBUF
        self.extract_events = []
        self.step_no = 0
        self.sA.wait_stream(self.cur)
        for s in self.sBs:
            s.wait_stream(self.cur)
        # The line streams only meet the rest of the step at the matchers (which wait for them) and at the reuse of a feature set NBUF steps later:
        # a stream that starts late STAYS late.  PLP_BENCH_LINE_PHASE_MS delays line stream i by i x that many ms once, before the first step, so
        # that one sub-block's front (blur, gradient, seed sort: throughput kernels) runs beside the other's region growing (latency-bound) instead
        # of beside its front (profiles/r04_line_phase.md).
        shift_ms = float(os.environ.get("PLP_BENCH_LINE_PHASE_MS", "0"))
        if shift_ms > 0 and not serial:
            for i, s in enumerate(self.sBs):
                if i:
                    with torch.cuda.stream(s):
                        torch.cuda._sleep(int(i * shift_ms * 1e-3 * 2.1e9))

    def _halo_arrays(self, buf):
        pts = [self.kps2[buf], self.desc2[buf], self.cnt2[buf]]
        return pts if self.orb_only else pts + [self.kl2[buf], self.lbd2[buf], self.lcnt2[buf]]

    def match_stage(self, buf=0, st=None, before_lines=None):
        """the tracker's four matcher calls for every frame of the step held in feature set `buf`, on stream st"""
        plp, replay, B, cap, lcap = self.plp, self.replay, self.B, self.cap, self.lcap
        st = st or self.sA
        kps, desc, cnt = self.kps2[buf], self.desc2[buf], self.cnt2[buf]
        # The two frames preceding this rank's block come from the previous rank: ONE packed exchange per step into rows 0..HALO-1 of all six arrays (round 6; until
        # then the points went first and the lines in a second exchange once their streams were done).  It needs every extractor of the step, so the matchers start
        # when the line streams are done too -- they run on a stream of their own beside the NEXT step's extractors, which is where the time goes either way.
        if before_lines is not None:
            before_lines()
        self.halo(self._halo_arrays(buf))
        self.pq.build(kps, cnt, HALO, self.shift, st)       # reprojections / levels / angles / validity of the queries: one launch
        pq = self.pq
        t = dict(t_kps=kps[HALO:], t_desc=desc[HALO:], t_counts=cnt[HALO:])
        # descriptors are read in place: the queries of frame b are rows (b + HALO - 1) resp. (b + HALO - 2 .. b + HALO - 1) of `desc`
        # the feature arrays are strided by cap = 2 K + 64, a frame holds about K key points: the matchers size their LDS for the expected count
        t["t_count_hint"] = self.K + max(64, self.K // 8)
        q1 = dict(q_reproj=pq.q1_reproj, q_level=pq.q1_level, q_angle=pq.q1_angle, q_counts=pq.q1_counts, q_desc=desc[HALO - 1:], q_desc_stride=cap)
        self.mt_last.match_device(plp.MODE_LAST_FRAME, cap, cap, {**t, **q1}, self.m1, self.n1, margin=20.0, direction=0, scale_factors=self.sf, grid=self.grid, B=B, stream=st)
        q2 = dict(q_reproj=pq.q2_reproj, q_level=pq.q2_level, q_valid=pq.q2_valid, q_desc=desc[HALO - 2:], q_desc_stride=cap)
        self.mt_lm.match_device(plp.MODE_LANDMARKS, cap, 2 * cap, {**t, **q2}, self.m2, self.n2, margin=10.0, scale_factors=self.sf, grid=self.grid, B=B, stream=st)
        if self.orb_only:
            return
        kl, lbd, lcnt = self.kl2[buf], self.lbd2[buf], self.lcnt2[buf]
        lq = self.lq
        lq.build(kl, lcnt, HALO, self.shift, st, feat_kps=kps, feat_kp_counts=cnt)   # key lines of the previous frames, both end points moved by the pan
        tl = dict(t_kl=kl[HALO:], t_desc=lbd[HALO:], t_counts=lcnt[HALO:])
        q3 = dict(q_reproj=lq.q_sp, q_reproj2=lq.q_ep, q_level=lq.q_level, q_counts=lq.q_counts, q_desc=lbd[HALO - 1:], q_desc_stride=lcap, is_rgbd=0, num_levels_lsd=1)
        self.mt_line.match_device(plp.MODE_LAST_FRAME_LINE, lcap, lcap, {**tl, **q3}, self.m3, self.n3, margin=20.0, direction=0, scale_factors=self.sf_lsd, B=B, stream=st)
        q4 = dict(q_reproj=lq.q2_sp, q_reproj2=lq.q2_ep, q_level=lq.q2_level, q_valid=lq.q2_valid, q_desc=lbd[HALO - 2:], q_desc_stride=lcap, t_kp_octave=lq.t_kp_octave)
        self.mt_lm_line.match_device(plp.MODE_LANDMARKS_LINE, lcap, 2 * lcap, {**tl, **q4}, self.m4, self.n4, margin=10.0, scale_factors=self.sf_lsd, B=B, stream=st)

    def step(self, d_frames):
        """enqueue one step over d_frames (uint8 [B, rows, cols] in HBM); returns the index of the feature set it fills"""
        torch, B, parts = self.torch, self.B, self.parts
        sA, sC = self.sA, self.sC
        n = self.step_no; self.step_no += 1
        buf = n % self.NBUF
        done = self.done_match[buf]
        if done is not None:
            sA.wait_event(done)          # the matchers of step n - NBUF have read this set
        if "orb" in parts:
            self.ex.extract_batch(d_frames, self.kps2[buf][HALO:], self.desc2[buf][HALO:], self.cnt2[buf][HALO:], stream=sA)
        ready = torch.cuda.Event(); ready.record(sA)
        line_ready = []
        if not self.orb_only:
            bs = B // self.n_line
            g0 = (n % self.line_depth) * self.n_line
            for i, (lti, sbi) in enumerate(zip(self.lts[g0:g0 + self.n_line], self.sBs[g0:g0 + self.n_line])):
                sl = slice(i * bs, (i + 1) * bs)
                if done is not None:
                    sbi.wait_event(done)     # the line matchers of step n - NBUF have read this set
                if "lines" in parts:
                    lti.extract_batch(d_frames[sl], self.kl2[buf][HALO:][sl], self.lbd2[buf][HALO:][sl], self.fn2[buf][sl], self.lcnt2[buf][HALO:][sl], stream=sbi)
                ev = torch.cuda.Event(); ev.record(sbi); line_ready.append(ev)
        self.extract_events = [ready] + line_ready      # once these have passed, d_frames has been read (the matchers never look at pixels)
        if "match" not in parts:
            return buf
        sC.wait_event(ready)
        with torch.cuda.stream(sC):
            self.match_stage(buf, sC, before_lines=lambda: [sC.wait_event(ev) for ev in line_ready])
            self.done_match[buf] = torch.cuda.Event(); self.done_match[buf].record(sC)
        return buf

    def last_batch_status(self):
        self.ex.last_batch_status()
        for lti in self.lts:
            lti.last_batch_status()

    def feature_set(self, buf):
        """views of feature set `buf` without the halo rows: what the extractors of that step wrote"""
        return dict(kps=self.kps2[buf][HALO:], desc=self.desc2[buf][HALO:], cnt=self.cnt2[buf][HALO:], kl=self.kl2[buf][HALO:], lbd=self.lbd2[buf][HALO:],
                    fn=self.fn2[buf], lcnt=self.lcnt2[buf][HALO:])
