"""Multi-GPU sharding logic on CPU: world_size-2 gloo processes render disjoint frame subsets (sample striding,
SURVEY 8e) into SUM buffers -- with the oracle standing in for the GPU -- and one reduce combines them."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_frames_for_rank_partition():
    from cudapathtracer_amd.dist import frames_for_rank
    for n in (0, 1, 5, 8, 4096):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, count, stride = frames_for_rank(10, n, r, world)
                seen += [first + i * stride for i in range(count)]
            assert sorted(seen) == list(range(10, 10 + n))
    with pytest.raises(ValueError):
        frames_for_rank(0, 4, 2, 2)


WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import cudapathtracer_amd as pt
from cudapathtracer_amd.dist import frames_for_rank, reduce_accumulation
from oracle import orc
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
data = pt.builtin_scene(0)
o = orc.Scene(data)
W, H, N = 48, 32, 6
s = pt.make_settings(pt.make_camera(**pt.CORNELL_CAMERA), W, H, 3)
first, count, stride = frames_for_rank(0, N, rank, world)
acc = np.zeros((H, W, 4), np.float32)
for i in range(count):          # sum mode: acc += sample of frame first + i*stride
    r = o.render(s, orc.ISECT_BRUTE, orc.RNG_NODE, orc.MATH_DET, first + i * stride, 1, acc=np.zeros((H, W, 4), np.float32), want_ldr=False, nthreads=2)
    acc[..., :3] += r["sample"]
mean = reduce_accumulation(torch.from_numpy(acc), N, dst=0)
if rank == 0:
    np.save(os.environ["OUT"], mean.numpy())
else:
    assert mean is None
dist.barrier(); dist.destroy_process_group()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sample_striding_matches_single_rank(tmp_path, ptlib, oracle):
    out = str(tmp_path / "mean.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO=ROOT, OUT=out, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    o = oracle.Scene(ptlib.builtin_scene(0))
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), 48, 32, 3)
    ref = o.render(s, oracle.ISECT_BRUTE, oracle.RNG_NODE, oracle.MATH_DET, 0, 6, want_ldr=False)["acc"]   # running mean
    assert (got[..., 3] == 1).all()
    assert np.allclose(got[..., :3], ref[..., :3], rtol=1e-5, atol=1e-6)      # summation order differs


GPU_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
import cudapathtracer_amd as pt
from cudapathtracer_amd.dist import frames_for_rank, reduce_accumulation
t = pt.Tracer(0)                                   # every rank on the one GPU of the box: the tracer before the process group
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sc = pt.Scene(t, pt.builtin_scene(0))
W, H, N = 96, 64, 7
s = pt.make_settings(pt.make_camera(**pt.CORNELL_CAMERA), W, H, 4)
first, count, stride = frames_for_rank(0, N, rank, world)
acc = t.malloc(W * H * 16)
t.memset(acc, 0, W * H * 16)
t.render(sc, s, first, count, acc, None, frame_stride=stride, sum_mode=1)      # the integrator adds straight into acc
t.synchronize()
rays = t.counters().rays
mean = reduce_accumulation(torch.from_numpy(t.download(acc, (H, W, 4), np.float32)), N, dst=0)
tot = torch.tensor([rays], dtype=torch.int64)
dist.reduce(tot, 0)
if rank == 0:
    np.save(os.environ["OUT"], mean.numpy())
    open(os.environ["OUT"] + ".rays", "w").write(str(int(tot[0])))
else:
    assert mean is None
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_libmipt_ranks_render_the_frames_one_rank_renders(tmp_path, tracer, ptlib):
    """two processes, each with its own libmipt tracer (both on the box's one GPU), sample striding + reduce through
    dist.py: the mean equals the single-tracer running mean, the rays add up to the same total"""
    out = str(tmp_path / "mean.npy")
    script = tmp_path / "worker.py"
    script.write_text(GPU_WORKER)
    env = dict(os.environ, REPO=ROOT, OUT=out, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    W, H, N = 96, 64, 7
    sc = ptlib.Scene(tracer, ptlib.builtin_scene(0))
    s = ptlib.make_settings(ptlib.make_camera(**ptlib.CORNELL_CAMERA), W, H, 4)
    acc = tracer.malloc(W * H * 16)
    tracer.memset(acc, 0, W * H * 16)
    tracer.reset_counters()
    tracer.render(sc, s, 0, N, acc, None)
    tracer.synchronize()
    ref = tracer.download(acc, (H, W, 4), np.float32)
    assert int(open(out + ".rays").read()) == tracer.counters().rays
    assert (got[..., 3] == 1).all()
    assert np.allclose(got[..., :3], ref[..., :3], rtol=1e-4, atol=1e-5) and ref[..., :3].mean() > 0.01
    tracer.free(acc)


PROGRESSIVE_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from cudapathtracer_amd.dist import render_progressive
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
acc = torch.zeros((4, 6, 4))
rendered = []
def render_frames(first, count, stride):          # stand-in for tracer.render(..., sum_mode=1): frame f adds f+1 everywhere
    for i in range(count):
        f = first + i * stride
        rendered.append(f)
        acc[..., :3] += float(f + 1)
previews = []
mean = render_progressive(render_frames, acc, 3, 11, rank, world, preview_every=4, on_preview=lambda img, n: previews.append((n, img.clone())))
lists = [None] * world
dist.all_gather_object(lists, rendered)
if rank == 0:
    frames = sorted(f for l in lists for f in l)
    assert frames == list(range(3, 14)), frames                        # every frame exactly once across the ranks
    assert [n for n, _ in previews] == [4, 8]                          # previews after 4 and 8 of 11 frames, none at the end
    for n, img in previews:
        assert torch.allclose(img[..., :3], torch.full((4, 6, 3), sum(range(4, 4 + n)) / n)) and (img[..., 3] == 1).all()
    assert torch.allclose(mean[..., :3], torch.full((4, 6, 3), sum(range(4, 15)) / 11.0)) and (mean[..., 3] == 1).all()
    open(os.environ["OUT"], "w").write("ok")
else:
    assert mean is None and previews == []
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [1, 2, 3])
def test_progressive_preview_reduce(tmp_path, world):
    out = str(tmp_path / "ok.txt")
    script = tmp_path / "worker.py"
    script.write_text(PROGRESSIVE_WORKER)
    env = dict(os.environ, REPO=ROOT, OUT=out, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(out).read() == "ok"


def test_tiles_for_rank_partition():
    from cudapathtracer_amd.dist import tiles_for_rank
    for height, tile_rows, world in ((1080, 16, 8), (2160, 64, 3), (7, 3, 2), (5, 8, 4), (1, 1, 1)):
        rows = np.zeros(height, int)
        for r in range(world):
            for y0, y1 in tiles_for_rank(height, tile_rows, r, world):
                assert 0 <= y0 < y1 <= height and y0 % tile_rows == 0 and (y0 // tile_rows) % world == r
                rows[y0:y1] += 1
        assert (rows == 1).all()                                  # every row belongs to exactly one rank
    with pytest.raises(ValueError):
        tiles_for_rank(10, 0, 0, 1)


TILE_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO"])
from cudapathtracer_amd.dist import tiles_for_rank, assemble_tiles
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
H, W, T = 23, 6, 4
acc = torch.zeros((H, W, 4))
for frame in range(5):                     # stand-in for tracer.trace with a tile partition: running mean in own tiles only
    w = 1.0 / (frame + 1)
    for y0, y1 in tiles_for_rank(H, T, rank, world):
        ys = torch.arange(y0, y1, dtype=torch.float32)[:, None, None]
        acc[y0:y1, :, :3] = acc[y0:y1, :, :3] * (1 - w) + (ys + frame) * w
        acc[y0:y1, :, 3] = 1.0
img = assemble_tiles(acc)
if rank == 0:
    want = torch.arange(H, dtype=torch.float32)[:, None, None] + 2.0       # mean of y + frame over frames 0..4
    assert torch.allclose(img[..., :3], want.expand(H, W, 3)) and (img[..., 3] == 1).all()
    open(os.environ["OUT"], "w").write("ok")
else:
    assert img is None
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [1, 2, 3])
def test_tile_partition_assembles_by_reduce(tmp_path, world):
    out = str(tmp_path / "ok.txt")
    script = tmp_path / "worker.py"
    script.write_text(TILE_WORKER)
    env = dict(os.environ, REPO=ROOT, OUT=out, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(out).read() == "ok"
