"""Multi-GPU sample partitioning.

The reference is single-GPU.  Samples are independent given the (pixel, frameID) seeding (CUDATracerProg.cu:373-377),
so the path shards with no data-path collective: the scene is replicated, rank r of G renders frame indices
r, r+G, r+2G, ... into a private SUM buffer, and one RCCL reduce (sum, fp32) of the accumulation buffers combines
them at the end; the root scales by 1/N and tonemaps.  One process per GPU (torch.distributed, backend nccl = RCCL).
"""


def frames_for_rank(first_frame, n_frames, rank, world):
    """(first, count, stride) of the frame indices rank `rank` renders out of [first_frame, first_frame+n_frames)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    count = (n_frames - rank + world - 1) // world if n_frames > rank else 0
    return first_frame + rank, count, world


def reduce_accumulation(acc, n_frames_total, dst=0, group=None):
    """Sum-reduce per-rank SUM buffers onto rank `dst` and turn them into the mean image (alpha forced to 1).

    `acc` is a torch tensor [..., 4] living wherever the process group's backend wants it (GPU for nccl/RCCL, CPU for
    gloo).  Returns the mean on dst, None elsewhere.
    """
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank(group) != dst:
            return None
    out = acc * (1.0 / float(n_frames_total))
    out[..., 3] = 1.0
    return out
