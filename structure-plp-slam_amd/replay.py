"""Batched-replay sharding across GPUs: one process per GPU, contiguous frame blocks per rank.

Extraction needs no communication.  Matching frame f against frames f-1 and f-2 does at block
boundaries: the first `halo` frames of rank r need the features of the last `halo` frames of rank r-1.
`exchange_halo` is that one exchange step — an all-gather of the ranks' tail features (RCCL over xGMI on
GPUs: backend "nccl"; gloo on CPU in the tests), after which every rank keeps its predecessor's tail
(rank 0 takes the last rank's: the replay is circular).  ~2 x 2064 x 60 B = 250 KB per rank.
"""
import torch
import torch.distributed as dist


def frame_block(rank, world, n_frames):
    """[start, stop) of the contiguous block of the sequence owned by `rank`."""
    per, extra = divmod(n_frames, world)
    start = rank * per + min(rank, extra)
    return start, start + per + (1 if rank < extra else 0)


def exchange_halo(tensors, halo=2, group=None):
    """tensors: list of per-frame feature tensors [B, ...] of this rank (key points, descriptors, counts...).
    Returns the list of [halo, ...] tensors holding the predecessor rank's last `halo` frames."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [t[-halo:].clone() for t in tensors]          # single rank: circular replay inside the block
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out = []
    for t in tensors:
        tail = t[-halo:].contiguous()
        bounce = tail.is_cuda and dist.get_backend(group) == "gloo"     # gloo gathers host tensors only (diagnostic runs)
        src = tail.cpu() if bounce else tail
        gathered = torch.empty((world * halo,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(gathered, src, group=group)     # concatenated along dim 0 (nccl and gloo agree on this form)
        prev = (rank - 1) % world
        out.append(gathered[prev * halo:(prev + 1) * halo].to(tail.device, copy=True))
    return out


def with_halo(t, halo_t):
    """[halo + B, ...]: predecessor frames in front, so frame b's predecessors are rows b+halo-1, b+halo-2."""
    return torch.cat([halo_t, t], 0)
