"""Batched-replay sharding across GPUs: one process per GPU, contiguous frame blocks per rank.

Extraction needs no communication.  Matching frame f against frames f-1 and f-2 does at block
boundaries: the first `halo` frames of rank r need the features of the last `halo` frames of rank r-1.
`exchange_halo` is that one exchange step: the tail features packed into one record per frame and shifted to the
successor rank (or all-gathered) in a single collective -- RCCL over xGMI on GPUs (backend "nccl"), gloo on CPU in the
tests -- after which every rank holds its predecessor's tail (rank 0 takes the last rank's: the replay is circular).
~2 x 2064 x 60 B = 250 KB per rank and step.
"""
import torch
import torch.distributed as dist


def frame_block(rank, world, n_frames):
    """[start, stop) of the contiguous block of the sequence owned by `rank`."""
    per, extra = divmod(n_frames, world)
    start = rank * per + min(rank, extra)
    return start, start + per + (1 if rank < extra else 0)


def exchange_halo(tensors, halo=2, group=None, mode="ring"):
    """tensors: list of per-frame feature tensors [B, ...] of this rank (key points, descriptors, counts...).
    Returns the list of [halo, ...] tensors holding the predecessor rank's last `halo` frames.

    ONE exchange per call: the tails of all tensors are packed into a single byte record per frame
    ({count, key points, descriptors, ...} back to back) and moved by
      mode "ring"       one send to the successor + one receive from the predecessor (batch_isend_irecv): only the bytes that are
                        needed cross a link -- xGMI is point-to-point, a neighbour shift is its natural pattern;
      mode "allgather"  one all_gather_into_tensor of the packed tails (every rank sees every tail), the form north_star names."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [t[-halo:].clone() for t in tensors]          # single rank: circular replay inside the block
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = tensors[0].device
    tails = [t[-halo:].contiguous() for t in tensors]
    sizes = [tl[0].numel() * tl.element_size() for tl in tails]          # bytes per frame of each tensor
    packed = torch.cat([tl.view(torch.uint8).reshape(halo, -1) for tl in tails], 1).contiguous()      # [halo, record bytes]
    bounce = packed.is_cuda and dist.get_backend(group) == "gloo"        # gloo moves host tensors only (diagnostic runs on one GPU)
    src = packed.cpu() if bounce else packed
    if mode == "allgather":
        gathered = torch.empty((world * halo, src.shape[1]), dtype=torch.uint8, device=src.device)
        dist.all_gather_into_tensor(gathered, src, group=group)
        prev = (rank - 1) % world
        got = gathered[prev * halo:(prev + 1) * halo]
    else:
        got = torch.empty_like(src)
        ops = [dist.P2POp(dist.isend, src, (rank + 1) % world, group), dist.P2POp(dist.irecv, got, (rank - 1) % world, group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    got = got.to(dev) if bounce else got
    out, off = [], 0
    for tl, nbytes in zip(tails, sizes):
        out.append(got[:, off:off + nbytes].contiguous().view(tl.dtype).reshape(tl.shape))
        off += nbytes
    return out


def exchange_halo_into(buffers, halo=2, group=None, mode="ring"):
    """buffers: feature tensors laid out [halo + B, ...] -- rows halo.. hold this rank's block (the extractors write there), rows
    0..halo-1 receive the predecessor's tail.  Same single exchange as exchange_halo, without re-concatenating the block: the
    matchers then read `buffer[halo - 1 + b]` as frame b's predecessor straight from the array the extractor filled.
    Stream order is the CALLER's: the exchange runs on the current stream and waits for nothing else, so the current stream must already
    be ordered after the extractor streams that fill `buffers` (bench.py / replay_step: stream C waits for the extractors' events first)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        for t in buffers:
            # single rank: circular replay inside the block.  With fewer than `halo` frames in the block the tail overlaps the
            # head rows (rows B..B+halo-1 vs 0..halo-1): copy through a temporary, copy_ on overlapping views is undefined.
            t[:halo].copy_(t[-halo:].clone() if t.shape[0] < 2 * halo else t[-halo:])
        return
    tails = exchange_halo([t[halo:] for t in buffers], halo=halo, group=group, mode=mode)
    for t, tail in zip(buffers, tails):
        t[:halo].copy_(tail)


class halo_exchanger:
    """The ONE packed exchange of a step: the two-frame tails of ALL feature arrays of the step -- key points, descriptors, key lines, LBD rows
    and both count arrays -- travel as one record per frame in one collective, through buffers that are allocated once (exchange_halo builds its record with
    torch.cat: an allocation inside the timed step).  mode "ring": one send to the successor and one receive from the predecessor (xGMI is point-to-point: only the
    bytes that are needed cross a link); "allgather": one all_gather_into_tensor of the packed tails, the collective north_star names.  A single rank copies its
    own tail (the replay is circular).  Stream order is the caller's, as for exchange_halo_into."""

    def __init__(self, buffers, halo=2, mode="ring", group=None):
        assert mode in ("ring", "allgather"), mode
        assert all(t.is_contiguous() for t in buffers), "the head rows are written through a view: the arrays must be contiguous"
        self.halo, self.mode, self.group = halo, mode, group
        self.sizes = [int(t[0].numel() * t.element_size()) for t in buffers]          # bytes per frame of every array
        self.record_bytes = sum(self.sizes)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.bytes_per_step = halo * self.record_bytes if self.world > 1 else 0
        if self.world > 1:
            dev = buffers[0].device
            self.rank = dist.get_rank(group)
            self.bounce = dev.type == "cuda" and dist.get_backend(group) == "gloo"      # gloo moves host tensors only (diagnostic runs on one GPU)
            self.send = torch.empty((halo, self.record_bytes), dtype=torch.uint8, device=dev)
            self.recv = torch.empty((halo, self.record_bytes), dtype=torch.uint8, device=dev)
            xdev = "cpu" if self.bounce else dev
            if self.bounce:
                self.h_send, self.h_recv = torch.empty_like(self.send, device="cpu"), torch.empty_like(self.recv, device="cpu")
            if mode == "allgather":
                self.gathered = torch.empty((self.world * halo, self.record_bytes), dtype=torch.uint8, device=xdev)

    def __call__(self, buffers):
        """buffers laid out [halo + B, ...]: rows halo.. are this rank's block, rows 0..halo-1 receive the predecessor's tail"""
        halo = self.halo
        if self.world == 1:
            for t in buffers:
                t[:halo].copy_(t[-halo:].clone() if t.shape[0] < 2 * halo else t[-halo:])
            return
        rows = lambda t, sl: t[sl].contiguous().view(torch.uint8).reshape(halo, -1)    # (a slice of whole frames of a contiguous array: no copy)
        off = 0
        for t, nb in zip(buffers, self.sizes):
            self.send[:, off:off + nb].copy_(rows(t, slice(t.shape[0] - halo, t.shape[0])))
            off += nb
        src, dst = self.send, self.recv
        if self.bounce:
            self.h_send.copy_(self.send); src, dst = self.h_send, self.h_recv
        if self.mode == "allgather":
            dist.all_gather_into_tensor(self.gathered, src, group=self.group)
            prev = (self.rank - 1) % self.world
            dst = self.gathered[prev * halo:(prev + 1) * halo]
        else:
            ops = [dist.P2POp(dist.isend, src, (self.rank + 1) % self.world, self.group), dist.P2POp(dist.irecv, dst, (self.rank - 1) % self.world, self.group)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.bounce or self.mode == "allgather":
            self.recv.copy_(dst)
        off = 0
        for t, nb in zip(buffers, self.sizes):
            rows(t, slice(0, halo)).copy_(self.recv[:, off:off + nb])
            off += nb


class point_queries:
    """Device-side builder of the tracker's per-frame queries in a replay (plp_replay_point_queries_device): outputs are allocated
    once and rewritten every step."""

    def __init__(self, plp, B, cap, device):
        import torch
        self.plp, self.B, self.cap = plp, B, cap
        z = lambda shape, dt: torch.empty(shape, dtype=dt, device=device)
        self.q1_reproj, self.q1_level, self.q1_angle, self.q1_counts = z((B, cap, 2), torch.float32), z((B, cap), torch.int32), z((B, cap), torch.float32), z((B,), torch.int32)
        self.q2_reproj, self.q2_level, self.q2_valid = z((B, 2 * cap, 2), torch.float32), z((B, 2 * cap), torch.int32), z((B, 2 * cap), torch.uint8)

    def build(self, feat_kps, feat_counts, halo, shift, stream):
        """feat_kps: uint8 [halo + B, cap, 28]; feat_counts: int32 [halo + B]"""
        self.plp._check(self.plp.lib().plp_replay_point_queries_device(feat_kps.data_ptr(), feat_counts.data_ptr(), halo, self.B, self.cap, float(shift[0]), float(shift[1]),
                                                                       self.q1_reproj.data_ptr(), self.q1_level.data_ptr(), self.q1_angle.data_ptr(), self.q1_counts.data_ptr(),
                                                                       self.q2_reproj.data_ptr(), self.q2_level.data_ptr(), self.q2_valid.data_ptr(), stream.cuda_stream))


class line_queries:
    """Device-side builder of the line queries (plp_replay_line_queries_device): the previous frame's key lines for
    match_current_and_last_frames_line and -- with landmarks=True -- the key lines of the two previous frames as local line landmarks for
    match_frame_and_landmarks_line, plus the key-point octaves that matcher reads with a line index."""

    def __init__(self, plp, B, cap, device, landmarks=False):
        import torch
        self.plp, self.B, self.cap, self.landmarks = plp, B, cap, landmarks
        z = lambda shape, dt: torch.empty(shape, dtype=dt, device=device)
        self.q_sp, self.q_ep, self.q_level, self.q_counts = z((B, cap, 2), torch.float32), z((B, cap, 2), torch.float32), z((B, cap), torch.int32), z((B,), torch.int32)
        if landmarks:
            self.q2_sp, self.q2_ep, self.q2_level, self.q2_valid = z((B, 2 * cap, 2), torch.float32), z((B, 2 * cap, 2), torch.float32), z((B, 2 * cap), torch.int32), z((B, 2 * cap), torch.uint8)
            self.t_kp_octave = z((B, cap), torch.int32)

    def build(self, feat_kl, feat_counts, halo, shift, stream, feat_kps=None, feat_kp_counts=None):
        """feat_kl: uint8 [halo + B, cap, 68]; feat_counts: int32 [halo + B]; with landmarks also feat_kps uint8 [halo + B, kp_cap, 28] and
        feat_kp_counts int32 [halo + B] of the same frames"""
        p = lambda t: t.data_ptr() if t is not None else None
        lm = self.landmarks
        kp_cap = feat_kps.shape[1] if lm else 0
        self.plp._check(self.plp.lib().plp_replay_line_queries_device(feat_kl.data_ptr(), feat_counts.data_ptr(), halo, self.B, self.cap, float(shift[0]), float(shift[1]),
                                                                      self.q_sp.data_ptr(), self.q_ep.data_ptr(), self.q_level.data_ptr(), self.q_counts.data_ptr(),
                                                                      p(self.q2_sp) if lm else None, p(self.q2_ep) if lm else None, p(self.q2_level) if lm else None,
                                                                      p(self.q2_valid) if lm else None, p(feat_kps) if lm else None, p(feat_kp_counts) if lm else None,
                                                                      kp_cap, p(self.t_kp_octave) if lm else None, stream.cuda_stream))


def pack_rows(plp, src, counts, dst, offsets, compute_offsets, stream):
    """src: torch tensor [B, cap, ...] (rows of a multiple of 4 bytes); counts int32 [B]; dst: flat uint8 tensor with room for B * cap rows;
    offsets int64 [B + 1].  dst gets the live rows back to back (plp_pack_rows_device); offsets[B] = number of rows packed."""
    B, cap = src.shape[0], src.shape[1]
    row_bytes = src[0, 0].numel() * src.element_size()
    plp._check(plp.lib().plp_pack_rows_device(src.data_ptr(), counts.data_ptr(), B, cap, row_bytes, dst.data_ptr(), offsets.data_ptr(), int(bool(compute_offsets)),
                                              stream.cuda_stream))
    return row_bytes


def with_halo(t, halo_t):
    """[halo + B, ...]: predecessor frames in front, so frame b's predecessors are rows b+halo-1, b+halo-2."""
    return torch.cat([halo_t, t], 0)
