"""Multi-GPU sample partitioning.

The reference is single-GPU.  Samples are independent given the (pixel, frameID) seeding (CUDATracerProg.cu:373-377),
so the path shards with no data-path collective: the scene is replicated, rank r of G renders frame indices
r, r+G, r+2G, ... into a private SUM buffer, and one RCCL reduce (sum, fp32) of the accumulation buffers combines
them at the end; the root scales by 1/N and tonemaps.  One process per GPU (torch.distributed, backend nccl = RCCL).

Alternative ("interleaved row tiles" for interactive latency): every rank renders EVERY frame but only its
share of the image -- tiles of `tile_rows` rows dealt round-robin (`mipt_set_tile_partition`) -- so a frame is
finished G times sooner; the buffers hold zeros outside a rank's tiles and the same reduce(SUM) assembles the image.
"""


def tiles_for_rank(height, tile_rows, rank, world):
    """Row ranges [(y0, y1), ...] rank `rank` of `world` renders: tiles rank, rank + world, ... of tile_rows rows."""
    if world < 1 or not 0 <= rank < world or tile_rows < 1:
        raise ValueError("bad rank/world/tile_rows")
    n_tiles = (height + tile_rows - 1) // tile_rows
    return [(k * tile_rows, min((k + 1) * tile_rows, height)) for k in range(rank, n_tiles, world)]


def assemble_tiles(acc, dst=0, group=None):
    """Tile-partitioned ranks hold the running MEAN in their own tiles and zeros elsewhere: one reduce(SUM) assembles the
    image on `dst` (alpha forced to 1).  Returns it on dst, None elsewhere."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if dist.get_rank(group) != dst:
            return None
    acc[..., 3] = 1.0
    return acc


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


def render_progressive(render_frames, acc, first_frame, n_frames, rank, world, preview_every=0, on_preview=None, dst=0,
                       group=None, to_reduce=lambda t: t):
    """Sample-strided render with an optional progressive preview ("periodic reduce every K frames").

    render_frames(first, count, stride) must ADD `count` frames (indices first, first+stride, ...) to this rank's SUM
    buffer `acc` and return only when they are in it -- with libmipt: `tracer.render(scene, settings, first, count,
    acc_ptr, None, stride, 1); tracer.synchronize()`.  The job is walked in rounds of `preview_every` global frames
    (0 = one round); after every round but the last a COPY of the sum buffers is reduced onto `dst`, which receives
    on_preview(mean_image, frames_done); the private sums keep growing, so the final reduce_accumulation is exact and
    the preview costs one extra reduce per round.  `to_reduce` moves a tensor to where the backend wants it (e.g.
    `.cpu()` for gloo).  Returns the final mean on dst, None elsewhere.
    """
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    step = preview_every if preview_every and preview_every > 0 else max(n_frames, 1)
    done = 0
    while done < n_frames:
        chunk = min(step, n_frames - done)
        first, count, stride = frames_for_rank(first_frame + done, chunk, rank, world)
        if count:
            render_frames(first, count, stride)
        done += chunk
        if done < n_frames and on_preview is not None:
            snap = to_reduce(acc).clone()
            if multi:
                dist.reduce(snap, dst=dst, op=dist.ReduceOp.SUM, group=group)
            if rank == dst:
                snap = snap * (1.0 / float(done))
                snap[..., 3] = 1.0
                on_preview(snap, done)
    return reduce_accumulation(to_reduce(acc), n_frames, dst=dst, group=group)
