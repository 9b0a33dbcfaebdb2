#!/usr/bin/env python3
"""Benchmark of the hot path: PathTracer::Trace (one sample per pixel per step) on N MI355X.

  python bench.py --gpus 1 --steps 8 --warmup 2
  python bench.py --gpus N ...            (no launcher: bench.py starts the N ranks itself, see spawn_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one frame (1 spp) of the workload per GPU.  The headline workload (`metric`, `value`, `ms_per_step`, `roofline`,
`cpu_baseline`) is BASELINE.json configs[1]: Cornell box, 1920x1080, 8 bounces.  BASELINE's metric is quoted on
"Cornell + HDRI scene", so the same run then measures, the same way, configs[2] (DefaultScene + synthetic HDRI, 1080p, 12
bounces) and the configs[4] stand-in (10 M-triangle displaced grid, 3840x2160, 4 bounces: the HBM-stressing case) and
attaches them under `workloads` in the ONE JSON line.  `--workload cornell|default|grid` runs just that one as the headline.

Timed region: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, MAX over ranks.
Because K frames of a cache-resident scene take ~0.1 s, that K-step measurement is repeated until the repeats add up to
`--min-seconds` (5 s: long enough for a 5 s utilisation sampler to see the device busy) and the MEDIAN repeat is reported (`repeats`, `timed_region_s` say how many and how long).
With N > 1 the scene is replicated, rank r renders frame indices r, r+N, ... into a private sum buffer (weak scaling:
one frame per GPU per step) and ONE RCCL reduce of the accumulation buffers closes every timed repeat.

Roofline (per workload, dominant kernel k_wavestack).  Both bounds are always computed and attached: `valu` = ALGORITHMIC
lane-operations per launch (per node visit, per triangle test, per shade, counts from the CPU oracle on the same
flat BVH) / the launch's HIP-event duration against the guide's fp32 vector rate; `hbm` = SURVEY 8d's algorithmic bytes per
ray against 8 TB/s, with the MEASURED traffic (`traffic_gb_per_launch`, `traffic_frac`: rocprofv3 PMC passes, 2 x FETCH_SIZE
+ WRITE_SIZE) beside it.  Which of the two is the headline is read from counters, not from the scene's size: `valu` when
the PMC passes of that workload show the un-packed VALU issue slots >= 0.8 busy (`bound_from` spells out the figures and
the file they come from); a configuration without counters falls back to working set vs L2.
`trace_loop` is the same workload driven through the reference's own call sequence (Trace + blocking bitmap read-back per
frame): the per-frame figure an interactive caller sees, PCIe included, beside the batched, pipelined `value`.
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
# fp32 vector peak of the guide: 157.3 TFLOP/s = 78.65 T fused multiply-adds per second = lane-operations per second when
# every instruction is a packed FMA (v_pk_fma_f32).  Plain wave64 VALU instructions issue at a quarter of a SIMD's cycles
# (profiles/r2/issue_calibration.jsonl: 0.22-0.25 v_fma_f32 per SIMD-cycle at 1..8 waves per SIMD), i.e. half of this.
VALU_PEAK_TLOPS = 78.65
VALU_ISSUE_PEAK_TLOPS = 39.3  # 1024 SIMDs x 2.4 GHz / 4 cycles x 64 lanes: what un-packed instructions can reach (measured 35-39)
# algorithmic VALU lane-operations: BVH2 node visit 46 (two slab tests: 12 FMA + 2 x (3 min + 3 max) + compares
# and selects); Moeller-Trumbore 45; one shade 150.  A quantised 4-wide visit by the same counting: 9 (exponent scales,
# origin) + 4 x (6 byte->float + 6 FMA + 3 min + 3 max + 2 clamps + 2 compare/select) + 10 (5-exchange sort on keys) = 107
OPS_NODE_BVH2, OPS_NODE_Q4, OPS_TRI, OPS_SHADE = 46.0, 107.0, 45.0, 150.0

WORKLOADS = {          # name -> (BASELINE.json config, width, height, maxDepth)
    "cornell": ("configs[1]", 1920, 1080, 8),
    "default": ("configs[2]", 1920, 1080, 12),
    "grid": ("configs[4] stand-in", 3840, 2160, 4),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=32)
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--workload", default="all", choices=["all", "cornell", "default", "grid"],
                   help="all (default): cornell is the headline, default and grid are attached under `workloads`")
    p.add_argument("--min-seconds", type=float, default=5.0, help="the K-step measurement is repeated until the repeats add up to this")
    p.add_argument("--share-gpus", action="store_true", help="rehearsal: allow more ranks than visible GPUs (ranks are placed round-robin; needs --backend gloo, RCCL refuses two ranks on one GPU)")
    p.add_argument("--rehearse-launch", action="store_true", help="CPU rehearsal of the N-rank control flow only (launch, rendezvous, frame partition, closing reduce, rank-0 line): traces NOTHING, reports value null")
    p.add_argument("--trace-loop-frames", type=int, default=48, help="frames of the reference's own call sequence (Trace + blocking LDR read-back per frame) timed for `trace_loop`; 0 disables")
    p.add_argument("--width", type=int, default=0, help="0 = 1920 (cornell, default) or 3840 (grid)")
    p.add_argument("--height", type=int, default=0, help="0 = 1080 (cornell, default) or 2160 (grid)")
    p.add_argument("--depth", type=int, default=0, help="0 = 8 for cornell, 12 for default, 4 for grid (BASELINE.json)")
    p.add_argument("--grid-n", type=int, default=2236, help="grid workload: n x n quads = 2 n^2 triangles (2236 -> 10.0 M)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU oracle budget of the headline workload (0 disables cpu_baseline); the attached workloads get half")
    p.add_argument("--tile-pixels", type=int, default=0)
    p.add_argument("--integrator", type=int, default=0, help="0 persistent wave stacks (default), 1 level-synchronous queues")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI (default); gloo = rehearsal of the N>1 flow through host memory")
    p.add_argument("--node-width", type=int, default=0, help="BVH node layout: 0 library default, 2 fp32 BVH2, 4 quantised 4-wide")
    p.add_argument("--tri-cost", type=float, default=0.0, help="SAH triangle cost of the BVH build (0 = library default; tuning experiments)")
    p.add_argument("--max-leaf", type=int, default=0, help="largest leaf of the BVH build (0 = library default 4; tuning experiments)")
    p.add_argument("--pipeline", type=int, default=1, help="1 (default): consecutive frames alternate between two streams so a frame's tail overlaps the next frame; 0: one stream")
    p.add_argument("--estimator", type=int, default=0, help="0 the reference's estimator (default, the parity configuration); 1 env-map / emitter sampling with occlusion rays; 2 VNDF-sampled specular + cosine-weighted diffuse lobes; 3 both")
    p.add_argument("--option", action="append", default=[], help="key=value passed to mipt_set_option (tuning experiments)")
    p.add_argument("--closest-hit-only", type=int, default=1, help="1 (default): a scene with an emitters' tree is also timed without it, reported as closest_hit_only; 0: skip (counter passes: one kernel variant per run)")
    return p.parse_args()


def oracle_sample(pt, data, bvh, width, settings, seconds, threads, estimator=0, emit_tree=None):
    """Times the CPU oracle (same flat BVH, node-keyed RNG) on a bounded sample of the workload and returns its rate and
    the mean node visits / triangle tests per ray (the figures the algorithmic bytes and operations are computed from)."""
    from oracle import orc                      # cpu_baseline leg only: checker / reported baseline
    import ctypes as C
    o = orc.Scene(data)
    o.set_flat_bvh(*bvh, width)
    o.set_estimator(estimator)
    o.set_anyhit_last_level(True)       # count the work of the product's algorithm (last-level rays need only hit-or-miss)
    if emit_tree and emit_tree[0]:      # ... and in a scene with emitters they go through the emitters' tree first
        o.set_emitter_tree(*emit_tree)
    W, H = settings.viewportWidth, settings.viewportHeight

    def run(rows, depth):
        s = pt.Settings()
        C.memmove(C.addressof(s), C.addressof(settings), C.sizeof(pt.Settings))
        s.maxDepth = depth
        tot = dict(rays=0, nodes=0, tris=0, hits=0, anyhit=0, emitfirst=0, sec=0.0)
        acc = np.zeros((H, W, 4), np.float32)
        for y in rows:      # bands of BAND rows: enough 64-pixel work items per call to keep 16 threads busy
            t0 = time.perf_counter()
            r = o.render(s, orc.ISECT_FLAT, orc.RNG_NODE, orc.MATH_DET, 0, 1, rect=(0, y, W, min(y + BAND, H)), acc=acc, want_ldr=False, nthreads=threads)
            tot["sec"] += time.perf_counter() - t0
            st = r["stats"]
            tot["rays"] += st.rays; tot["nodes"] += st.node_visits; tot["tris"] += st.tri_tests; tot["hits"] += st.hits; tot["anyhit"] += st.anyhit_rays; tot["emitfirst"] += st.emitfirst_rays
        return tot

    BAND = 8
    probe_rows = [H // 3, H // 2]
    probe = run(probe_rows, settings.maxDepth)
    per_row = max(probe["sec"] / (len(probe_rows) * BAND), 1e-5)
    n_rows = int(min(H, max(BAND, seconds / per_row)))
    if n_rows >= H:
        # the whole frame fits the budget: time whole frames (one call each, no per-row call overhead), as many as fit
        n_frames = int(min(16, max(1, seconds / (per_row * H))))
        s = pt.Settings()
        C.memmove(C.addressof(s), C.addressof(settings), C.sizeof(pt.Settings))
        t0 = time.perf_counter()
        r = o.render(s, orc.ISECT_FLAT, orc.RNG_NODE, orc.MATH_DET, 0, n_frames, want_ldr=False, nthreads=threads)
        st = r["stats"]
        full = dict(rays=st.rays, nodes=st.node_visits, tris=st.tri_tests, hits=st.hits, anyhit=st.anyhit_rays, emitfirst=st.emitfirst_rays, sec=time.perf_counter() - t0)
        desc = f"frames 0..{n_frames - 1} of the same workload, whole"
    else:
        n_bands = max(1, n_rows // BAND)
        rows = sorted(set(int(round(i * (H - BAND) / max(n_bands - 1, 1))) // BAND * BAND for i in range(n_bands)))
        full = run(rows, settings.maxDepth)
        desc = f"{len(rows)} evenly spaced bands of {BAND} rows of frame 0 of the same workload"
    rays = max(full["rays"], 1)
    return dict(desc=desc, rays=full["rays"], sec=full["sec"], mrays=full["rays"] / full["sec"] / 1e6,
                n_node=full["nodes"] / rays, n_tri=full["tris"] / rays, miss=1.0 - full["hits"] / rays, anyhit=full["anyhit"] / rays, emitfirst=full["emitfirst"] / rays)


def dbg(msg):
    if os.environ.get("BENCH_DEBUG"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def source_sha16():
    """Hash of the kernel + host sources: PMC figures measured on another build are not attached to this one."""
    h = hashlib.sha256()
    for f in ("kernels.hip", "wavestack_kernel.inc", "kernels.h", "device_math.h", "api.hip", "bvh_builder.cpp"):
        with open(os.path.join(ROOT, "cudapathtracer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_for(workload, default_config):
    """(figures, label): HBM bytes per launch etc. from the committed rocprofv3 --pmc passes of this same command, tagged with
    where they come from.  `figures` is None when they were measured on other sources or another configuration (never
    silently stale); `label` is the newest committed entry of this workload at its default configuration whatever the
    sources -- used ONLY to name what binds (valu_busy), with its origin spelled out beside it."""
    if not default_config:
        return None, None
    label = None
    for rnd in ("r3", "r2", "r1"):
        path = os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")
        try:
            pj = json.load(open(path))
        except Exception:
            continue
        meta = pj.get("_meta", {})
        if workload not in pj:
            continue
        e = dict(pj[workload])
        same = meta.get("source_sha16") == source_sha16()
        e["source"] = f"profiles/{rnd}/pmc_traffic.json ({meta.get('collected', 'rocprofv3 --pmc')}; sources {meta.get('source_sha16')}{'' if same else ', NOT the sources of this build'})"
        if label is None:
            label = e
        if same:
            return e, e
    return None, label


def build_scene(pt, a, name, t, rank, multi, dist):
    depth = a.depth or WORKLOADS[name][3]
    W = a.width or WORKLOADS[name][1]
    H = a.height or WORKLOADS[name][2]
    if name == "cornell":
        data = pt.builtin_scene(0)
        cam = pt.make_camera(**pt.CORNELL_CAMERA)
    elif name == "default":
        data = pt.builtin_scene(1, pt.synthetic_sky(2048, 1024, 0x5EED))
        cam = pt.make_camera(**pt.DEFAULT_CAMERA)
    else:
        # BASELINE.json configs[4] stand-in: displaced grid, one diffuse material, synthetic HDRI
        v, i = pt.grid_mesh(a.grid_n, 8.0, 1.0, 1)
        sky = pt.SkyLight(); sky.color[:] = (0.98, 0.97, 0.99); sky.dir[:] = (-0.577, -0.577, -0.577); sky.intensity = 5
        data = pt.SceneData([(v, i, pt.make_material(color=(0.8, 0.8, 0.8), roughness=0.5, metallic=0.0, opacity=1.0, n=2.5))],
                            sky, pt.synthetic_sky(2048, 1024, 0x5EED))
        cam = pt.make_camera((-7.0, 4.0, 0.0), 0.0, 2.0, float(np.float32(np.pi / 3)))
    settings = pt.make_settings(cam, W, H, depth)
    if name == "grid" and multi:
        # 10 M triangles: rank 0 builds the BVH and writes the cache file, the others load it instead of building eight times
        cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mipt_bench_grid{a.grid_n}_w{a.node_width}.bvh")
        sc = pt.Scene(t, data, node_width=a.node_width, bvh_cache=cache, tri_cost=a.tri_cost, max_leaf=a.max_leaf) if rank == 0 else None
        dist.barrier()
        if rank != 0:
            sc = pt.Scene(t, data, node_width=a.node_width, bvh_cache=cache, tri_cost=a.tri_cost, max_leaf=a.max_leaf)
    else:
        sc = pt.Scene(t, data, node_width=a.node_width, tri_cost=a.tri_cost, max_leaf=a.max_leaf)
    return data, settings, sc, W, H, depth


def trace_loop(pt, t, sc, settings, W, H, frames):
    """The reference's own frame loop, call for call (Common/RendererWindow.hpp:100-148): two CUDABuffer work-alikes for the
    accumulation image and the bitmap, then per frame Trace -> bitmap->cpu_data() (a BLOCKING device-to-host copy of the
    w*h*4 LDR bytes into the pinned mirror) -> frameID++.  Frames cannot overlap in this sequence, and the copy rides on
    PCIe: this is what an interactive RendererWindow caller gets per frame.  It is reported beside the batched figure and
    is never `value` (the boundary of the path hands over device pointers)."""
    import ctypes as C
    accb, bitmap = pt.Buffer(t, W * H * 16), pt.Buffer(t, W * H * 4)
    s = pt.Settings()
    C.memmove(C.addressof(s), C.addressof(settings), C.sizeof(pt.Settings))

    def frame(i):
        s.frameID = i
        t.trace(sc, s, accb.mutable_gpu_data(), bitmap.mutable_gpu_data())
        return bitmap.cpu_data()

    for i in range(4):
        frame(i)
    c0 = t.counters()
    t0 = time.perf_counter()
    for i in range(4, 4 + frames):
        frame(i)
    dt = time.perf_counter() - t0
    c1 = t.counters()
    accb.close(); bitmap.close()
    return {"ms_per_frame": round(dt / frames * 1e3, 3), "mrays_per_s": round((c1.rays - c0.rays) / dt / 1e6, 1), "frames": frames,
            "sequence": "per frame: mipt_trace (1 spp, fold + tonemap) then a blocking read-back of the RGBA8 bitmap into pinned host memory, frameID++ "
                        "(RendererWindow::render, Common/RendererWindow.hpp:128-148); PCIe-inclusive, frames do not overlap",
            "ldr_bytes_per_frame": W * H * 4}


def run_workload(name, a, ctx, cpu_seconds):
    """Measures one workload; returns the dict of its figures (rank 0) or None (other ranks)."""
    pt, ptdist, torch, dist, t, dev = ctx["pt"], ctx["ptdist"], ctx["torch"], ctx["dist"], ctx["t"], ctx["dev"]
    rank, world, multi = ctx["rank"], ctx["world"], ctx["multi"]
    data, settings, sc, W, H, depth = build_scene(pt, a, name, t, rank, multi, dist)     # BVH built before any timed region
    dbg(f"{name}: scene uploaded")
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    ldr = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    sum_mode = 1 if multi else 0
    cdev = dev if a.backend == "nccl" else torch.device("cpu")

    def step(i):   # global step index -> this rank's frame
        t.render(sc, settings, i * world + rank, 1, acc.data_ptr(), None if sum_mode else ldr.data_ptr(), 1, sum_mode)

    def barrier():
        t.synchronize()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()

    def close(n_steps):   # the path's one exchange step: reduce the sum buffers onto rank 0, scale, tonemap
        mean = ptdist.reduce_accumulation(acc if a.backend == "nccl" else acc.cpu(), n_steps * world, dst=0)
        if rank == 0:
            mean = mean.to(dev)
            torch.cuda.synchronize()        # torch's stream produced `mean`; the tracer reads it on its own stream
            t.tonemap(mean.data_ptr(), 1.0, W * H, ldr.data_ptr())

    for i in range(a.warmup):
        step(i)
    if multi:   # the whole closing sequence once, untimed (first use of a torch kernel or an RCCL communicator costs tens of ms)
        t.synchronize()
        close(max(a.warmup, 1))
        acc.zero_()
    barrier()
    t.reset_counters()
    t.set_option("profile_levels", 1)

    def timed_repeat(first_step):
        if multi:
            acc.zero_()
        rays0 = t.counters()
        barrier()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(first_step + i)
        t.synchronize()
        if multi:
            close(a.steps)
        barrier()
        dt = time.perf_counter() - t0
        c = t.counters()
        v = torch.tensor([dt, 0.0, 0.0], dtype=torch.float64, device=cdev)
        r = torch.tensor([float(c.rays - rays0.rays), float(c.shadow_rays - rays0.shadow_rays)], dtype=torch.float64, device=cdev)
        if multi:
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
        return float(v[0].item()), float(r[0].item()), float(r[1].item())

    reps = [timed_repeat(a.warmup)]
    # every rank sees the same (max-reduced) first duration, so every rank runs the same number of repeats
    n_rep = int(min(64, max(1, math.ceil(a.min_seconds / max(reps[0][0], 1e-6)))))
    for k in range(1, n_rep):
        reps.append(timed_repeat(a.warmup + k * a.steps))
    t.set_option("profile_levels", 0)
    cnt = t.counters()                 # (read before anything else is rendered)
    prof = t.level_profile()
    # Beside the headline, for a scene with an emitters' tree: the same frames with every ray traced to its closest hit (the scene
    # built without the second tree), so that the line shows what the "emitters first" rule is worth and what the rate is without it
    closest_only = None
    if a.closest_hit_only and not multi and a.integrator == 0 and a.estimator in (0, 2) and sc.emitter_tree()[2] and not os.environ.get("MIPT_EMITTER_TREE"):
        os.environ["MIPT_EMITTER_TREE"] = "0"
        try:
            plain = pt.Scene(t, data, node_width=a.node_width, tri_cost=a.tri_cost, max_leaf=a.max_leaf)
        finally:
            del os.environ["MIPT_EMITTER_TREE"]
        for i in range(a.warmup):
            t.render(plain, settings, i, 1, acc.data_ptr(), ldr.data_ptr(), 1, 0)
        t.synchronize()
        times, r0 = [], t.counters().rays
        while sum(times) < min(a.min_seconds, 1.0) or not times:
            t0 = time.perf_counter()
            for i in range(a.steps):
                t.render(plain, settings, a.warmup + len(times) * a.steps + i, 1, acc.data_ptr(), ldr.data_ptr(), 1, 0)
            t.synchronize()
            times.append(time.perf_counter() - t0)
        rays_plain = (t.counters().rays - r0) / (len(times) * a.steps)
        med = sorted(times)[len(times) // 2]
        closest_only = {"value": round(rays_plain * a.steps / med / 1e6, 1), "unit": "Mrays/s", "ms_per_step": round(med / a.steps * 1e3, 3),
                        "note": "same frames, scene built without the emitters' tree ($MIPT_EMITTER_TREE=0): every ray traced to its closest hit"}
        plain.close()
    order = sorted(range(len(reps)), key=lambda k: reps[k][0])
    dt, total_rays, total_shadow = reps[order[len(order) // 2]]            # the median repeat
    timed_s = sum(r[0] for r in reps)
    out = None
    if rank == 0:
        nodes, refs = sc.bvh()
        e_nodes, e_refs, _ = sc.emitter_tree()            # behind the main tree on the device (scenes with emitters)
        nn, nr = len(nodes) + len(e_nodes), len(refs) + len(e_refs)
        launches_total = n_rep * a.steps
        if a.integrator == 0:    # one k_wavestack launch per frame traces every ray of the frame
            lds = sc.bvh_width == 2 and nn * 4 + nr * 3 <= 192          # <LDS scene, 4-wide nodes, light sampling, estimator-2 lobes>
            emit_variant = bool(sc.emitter_tree()[2]) and a.estimator in (0, 2)      # <..., EMIT>: the scene has an emitters' tree
            kernel_name = "mipt::k_wavestack<%s, %s, %s, %s, %s>" % tuple("true" if x else "false" for x in (lds, sc.bvh_width == 4, a.estimator in (1, 3), a.estimator >= 2, emit_variant))
            sec_rays, sec_ms, sec_launches = float(cnt.rays), prof.ms[0], prof.launches[0]
        else:                    # k_level<false,*> launches carry the non-primary levels
            kernel_name = "mipt::k_level<false,*>"
            sec_rays = sum(prof.rays[k] for k in range(1, 32))
            sec_ms = sum(prof.ms[k] for k in range(1, 32))
            sec_launches = sum(prof.launches[k] for k in range(1, 32))
        cpu = None
        if cpu_seconds > 0:
            threads = min(os.cpu_count() or 1, 16)
            # the oracle walks the arrays the device holds: the main tree and, behind it, the emitters' tree (if the scene has one)
            fn, fr, _, eroot, nmain = sc.flat_with_emitters()
            cpu = oracle_sample(pt, data, (fn, fr), sc.bvh_width, settings, cpu_seconds if world == 1 else min(cpu_seconds, 3.0), threads, a.estimator, (eroot, nmain))
        default_config = (W, H, depth) == WORKLOADS[name][1:] and (name != "grid" or a.grid_n == 2236) and not a.node_width and not a.tri_cost and not a.max_leaf and not a.estimator and not a.option and a.integrator == 0 and a.pipeline == 1
        pmc, pmc_label = pmc_for(name, default_config)
        working_set = nn * 64 + nr * 48
        roof = None
        if cpu is not None and sec_ms > 0 and sec_launches > 0:
            n_node, n_tri, miss = cpu["n_node"], cpu["n_tri"], cpu["miss"]
            rays_launch = sec_rays / sec_launches
            # Duration of one launch.  Pipelined launches overlap pairwise (two streams), so a launch's own event window
            # -- and rocprofv3's kernel duration -- includes the time it shares the GPU with, or waits for CUs behind, its
            # neighbour: the windows add up to more than the time the kernel occupied the GPU.  The effective duration is
            # the HIP-event span of the timed launches (first start to last end, per timed repeat) over their number;
            # scripts/kt_steady.py computes the same two figures from the rocprofv3 kernel trace.
            window_s = sec_ms / sec_launches * 1e-3
            launch_s = (cnt.kernel_span_ms / sec_launches * 1e-3) if (a.integrator == 0 and cnt.kernel_span_ms > 0) else window_s
            wall_s = dt / a.steps
            # ---- HBM: SURVEY 8(d): B_ray = 64*N_node + 48*N_tri + 64 (+12 per env-map miss)
            b_ray = 64.0 * n_node + 48.0 * n_tri + 64.0 + (12.0 * miss if data.env is not None else 0.0)
            hbm = {"achieved": round(rays_launch * b_ray / launch_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(rays_launch * b_ray / launch_s / 1e9 / HBM_PEAK_GBS, 4),
                   "achieved_wall": round(rays_launch * b_ray / wall_s / 1e9, 1),
                   "bytes_per_ray": round(b_ray, 1), "algorithmic_gb_per_launch": round(rays_launch * b_ray / 1e9, 2),
                   "traffic_gb_per_launch": None if not pmc else round(pmc["hbm_gb_per_launch"], 2),
                   "traffic_frac": None if not pmc else round(pmc["hbm_gb_per_launch"] / launch_s / HBM_PEAK_GBS, 4)}
            # ---- VALU: algorithmic lane-operations per ray, shades = rays that spawned children = (rays - primaries) / 2
            shades = max(0.0, (rays_launch - W * H) / 2.0) / rays_launch
            c_node = OPS_NODE_Q4 if sc.bvh_width == 4 else OPS_NODE_BVH2
            ops_ray = c_node * n_node + OPS_TRI * n_tri + OPS_SHADE * shades
            valu_ach = rays_launch * ops_ray / launch_s / 1e12
            valu = {"achieved": round(valu_ach, 2), "peak": VALU_PEAK_TLOPS, "unit": "T lane-op/s",
                    "frac": round(valu_ach / VALU_PEAK_TLOPS, 4),
                    "achieved_wall": round(rays_launch * ops_ray / wall_s / 1e12, 2),
                    "frac_of_unpacked_issue_peak": round(valu_ach / VALU_ISSUE_PEAK_TLOPS, 4), "unpacked_issue_peak": VALU_ISSUE_PEAK_TLOPS,
                    "ops_per_ray": round(ops_ray, 1), "ops_per_node_visit": c_node, "ops_per_triangle_test": OPS_TRI, "ops_per_shade": OPS_SHADE,
                    "shades_per_ray": round(shades, 4),
                    "peak_source": "MI355X_MICROARCH.md: 157.3 TFLOP/s fp32 vector = 78.65 T FMA/s (packed); un-packed issue measured by k_test_issue, profiles/r2/issue_calibration.jsonl"}
            if pmc:
                valu["pmc"] = {k: pmc[k] for k in ("valu_busy", "lane_utilisation", "wait_frac", "l2_hit") if k in pmc}
                if "executed_lane_ops_per_launch" in pmc:      # what the kernel executes per ray against what the algorithm needs
                    valu["pmc"]["executed_lane_ops_per_ray"] = round(pmc["executed_lane_ops_per_launch"] / rays_launch, 1)
            # What binds is read from counters, not from a size threshold: VALU issue when the un-packed issue slots are
            # >= 0.8 busy (PMC valu_busy of this workload's default configuration), HBM when the measured traffic is the
            # larger share of its peak.  Without counters for this configuration: a working set inside L2 cannot be HBM
            # bound; beyond L2 the algorithmic bytes decide as long as they stay below the peak.
            if pmc_label is not None and "valu_busy" in pmc_label:
                traffic_frac_label = pmc_label.get("hbm_gb_per_launch", 0.0) / launch_s / HBM_PEAK_GBS
                bound = "valu" if (pmc_label["valu_busy"] >= 0.8 or pmc_label["valu_busy"] >= traffic_frac_label) else "hbm"
                bound_from = f"counters: valu_busy {pmc_label['valu_busy']:.3f}, measured HBM traffic {traffic_frac_label:.2f} of peak; {pmc_label['source']}"
            else:
                bound = "hbm" if (working_set > (32 << 20) and hbm["frac"] <= 1.0) else "valu"
                bound_from = "no counters for this configuration: working set against the 32 MiB of L2"
            head = hbm if bound == "hbm" else valu
            roof = {"bound": bound, "achieved": head["achieved"], "peak": head["peak"], "unit": head["unit"], "frac": head["frac"],
                    "traffic": hbm["traffic_gb_per_launch"], "traffic_unit": "GB per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)",
                    "traffic_source": pmc["source"] if pmc else "none for this build/configuration (scripts/profile_all.sh regenerates profiles/r2/pmc_traffic.json)",
                    "kernel": kernel_name, "n_node_per_ray": round(n_node, 3), "n_tri_per_ray": round(n_tri, 3),
                    "rays_per_launch": round(rays_launch, 1), "avg_launch_ms": round(launch_s * 1e3, 4), "avg_launch_window_ms": round(window_s * 1e3, 4), "launches": int(sec_launches),
                    "avg_launch_ms_is": "HIP-event span of the timed launches / launches (launches overlap pairwise when pipelined); avg_launch_window_ms = mean of the launches' own event windows",
                    # consecutive frames run on two streams, so a launch's event window includes the time it shares the GPU
                    # with (or queues behind) its neighbour; the wall figures divide the timed region by the launches instead
                    "frames_pipelined": bool(a.pipeline and a.integrator == 0), "wall_ms_per_launch": round(wall_s * 1e3, 4),
                    "working_set_mb": round(working_set / 1e6, 2),
                    "bound_from": bound_from,
                    "valu": valu, "hbm": hbm}
        out = {
            "value": round(total_rays / dt / 1e6, 1), "unit": "Mrays/s", "ms_per_step": round(dt / a.steps * 1e3, 3),
            "repeats": n_rep, "timed_region_s": round(timed_s, 3), "ms_per_step_min_max": [round(reps[order[0]][0] / a.steps * 1e3, 3), round(reps[order[-1]][0] / a.steps * 1e3, 3)],
            "config": {"workload": f"{name}_{W}x{H}_{depth}bounce_1spp_per_gpu_step" + (f"_n{a.grid_n}" if name == "grid" else ""), "baseline_config": WORKLOADS[name][0],
                       "width": W, "height": H, "max_depth": depth,
                       "triangles": data.triangle_count, "bvh": "sbvh" + ("-q4" if sc.bvh_width == 4 else ""), "rng": "node", "integrator": "wave-stacks" if a.integrator == 0 else "level-queues", "parallelism": f"sample-striding x{world}",
                       "env_map": None if data.env is None else "synthetic 2048x1024",
                       # how last-level rays (which can only add emission or the sky) are traced; the image and the per-pixel ray
                       # counts are those of closest-hit tracing in every case
                       "last_level_rays": ("emitters first, then the scene up to the first non-emitter (closest_hit_only gives the rate without it)" if sc.emitter_tree()[2] and a.estimator in (0, 2) and a.integrator == 0
                                           else "first hit ends them (no mesh emits)" if a.integrator == 0 and a.estimator in (0, 2) and not any(m[2].emissiveStrength != 0 and any(c != 0 for c in m[2].emissiveColor) for m in data.meshes)
                                           else "closest hit")},
            "spp_per_s": round(a.steps * world / dt, 3), "rays_per_step_per_gpu": round(total_rays / a.steps / world, 1),
            "shadow_rays": round(total_shadow / a.steps / world, 1),
            # kernel_window_ms: mean of the launches' own HIP-event windows (= rocprofv3's average kernel duration); pipelined
            # launches overlap pairwise, so the windows add up to about twice the time the kernels held the GPU, which is
            # kernel_span_ms_per_step (first start to last end of a batch / launches)
            "kernel_window_ms": round(cnt.kernel_ms / max(launches_total, 1), 3),
            "kernel_span_ms_per_step": round(cnt.kernel_span_ms / max(launches_total, 1), 3) if cnt.kernel_span_ms > 0 else None,
            # share of the counted rays that are last-level rays traced hit-or-miss (scenes without emitting meshes: such a
            # ray adds nothing if it hits anything, so it stops at its first hit; same image, cheaper ray), from the CPU sample
            "anyhit_ray_frac": None if cpu is None else round(cpu["anyhit"], 4),
            # share of the counted rays that are last-level rays in a scene WITH emitters: intersected with the emitting triangles
            # first, then walked through the scene only as far as it takes to know whether that emitter is the closest hit (same
            # image; a ray that can add nothing costs two triangle tests on the Cornell box)
            "emitter_first_ray_frac": None if cpu is None else round(cpu["emitfirst"], 4),
            "closest_hit_only": closest_only,
            "accumulate_ms": round(prof.accumulate_ms / max(launches_total, 1), 3),
            "roofline": roof,
        }
        if a.integrator:
            out["level_ms"] = [round(prof.ms[k] / launches_total, 3) for k in range(depth)]
            out["level_mrays"] = [round(prof.rays[k] / launches_total / 1e6, 2) for k in range(depth)]
        if world == 1 and not multi and a.trace_loop_frames > 0 and a.integrator == 0:
            out["trace_loop"] = trace_loop(pt, t, sc, settings, W, H, a.trace_loop_frames)
        if cpu is not None and world == 1:
            out["cpu_baseline"] = {"value": round(cpu["mrays"], 2), "unit": "Mrays/s", "cores": min(os.cpu_count() or 1, 16), "kind": "port",
                                   "sample": f"{cpu['desc']} ({cpu['rays']} rays, {cpu['sec']:.1f} s), same flat BVH"}
    del acc, ldr
    sc.close()
    torch.cuda.empty_cache()
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` with no launcher in the environment: this process becomes the launcher.  It makes no
    GPU call (torch.cuda.device_count() does not initialise the device), starts N copies of this script -- one rank per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run would -- relays rank 0's
    JSON line to its own stdout and exits with the first non-zero exit code of a rank (the others are stopped by PID).
    More ranks than visible GPUs is an error, never a silent shrink; --share-gpus lifts it for rehearsals."""
    import socket
    import subprocess
    import threading
    if not a.rehearse_launch:
        import torch
        visible = torch.cuda.device_count()
        if visible == 0:
            print(f"bench.py: --gpus {a.gpus} requested but no GPU is visible", file=sys.stderr)
            return 2
        if a.gpus > visible and not a.share_gpus:
            print(f"bench.py: --gpus {a.gpus} requested but only {visible} GPU(s) are visible (a rehearsal with several ranks per GPU needs --share-gpus --backend gloo)", file=sys.stderr)
            return 2
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_SELF_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def relay():      # rank 0's JSON line goes to stdout; anything else it prints there (library banners) to stderr
        for line in procs[0].stdout:
            dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            dst.write(line); dst.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc = 0
    live = set(range(a.gpus))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    th.join(timeout=5)
    return rc


def rehearse_rank(a, rank, world):
    """--rehearse-launch: the N-rank control flow on the CPU and nothing else.  Every rank joins the gloo group, takes
    its frames by the product's partition rule (dist.frames_for_rank), adds a frame-dependent constant per frame into a
    small sum buffer instead of rendering, and the closing reduce + scale of the real flow (dist.reduce_accumulation)
    must return the mean over all frames exactly once each.  No ray is traced and `value` is null: this proves that
    `bench.py --gpus N` yields N cooperating ranks, not a throughput."""
    import torch
    import torch.distributed as dist
    from cudapathtracer_amd.dist import frames_for_rank, reduce_accumulation
    dist.init_process_group("gloo")
    assert dist.get_rank() == rank and dist.get_world_size() == world
    n_frames = a.steps * world
    first, count, stride = frames_for_rank(0, n_frames, rank, world)
    acc = torch.zeros((4, 8, 4))
    t0 = time.perf_counter()
    for i in range(count):
        acc[..., :3] += float(first + i * stride + 1)
    mean = reduce_accumulation(acc, n_frames, dst=0)
    v = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    ranks = [None] * world
    dist.all_gather_object(ranks, (rank, os.getpid(), count))
    if rank == 0:
        want = sum(range(1, n_frames + 1)) / n_frames
        assert torch.allclose(mean[..., :3], torch.full((4, 8, 3), want)) and (mean[..., 3] == 1).all(), "rehearsal reduce"
        assert sorted(r for r, _, _ in ranks) == list(range(world)) and len({p for _, p, _ in ranks}) == world
        print(json.dumps({"metric": "Mrays/s (extension+shadow)", "value": None, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none",
                          "rehearsal": "launch only: N ranks, gloo rendezvous, frame partition and closing reduce on the CPU; no ray traced, nothing measured",
                          "frames_per_rank": [c for _, _, c in sorted(ranks)], "config": {"workload": "none (rehearse-launch)", "parallelism": f"sample-striding x{world}"}}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    a = parse()
    if os.environ.get("BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get("BENCH_DEBUG_DUMP_AFTER", "40")), exit=False)
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.rehearse_launch):
        sys.exit(spawn_ranks(a))            # no launcher: become one (before any GPU call)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        # never shrink or grow silently: the line would carry an n_gpus nobody asked for
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} rank(s); start `python bench.py --gpus N` "
                  f"without a launcher (it spawns its own ranks) or pass the same N to both", file=sys.stderr)
        sys.exit(2)
    if a.rehearse_launch:
        return rehearse_rank(a, rank, world)
    import torch
    import torch.distributed as dist
    visible = torch.cuda.device_count()
    if visible == 0:
        print("bench.py: no GPU visible (there is no CPU fallback; --rehearse-launch rehearses the rank launch alone)", file=sys.stderr)
        sys.exit(2)
    if local >= visible and not a.share_gpus:
        print(f"bench.py: rank {rank} has LOCAL_RANK {local} but only {visible} GPU(s) are visible (--share-gpus allows a rehearsal with several ranks per GPU)", file=sys.stderr)
        sys.exit(2)
    if a.share_gpus and world > visible and a.backend == "nccl":
        print("bench.py: --share-gpus needs --backend gloo (RCCL refuses two ranks on one GPU)", file=sys.stderr)
        sys.exit(2)
    local = local % visible      # --share-gpus rehearsals put several ranks on one GPU
    # BENCH_FORCE_COLLECTIVES=1: run the N > 1 flow (process group, warm-up reduce, sum buffers, reduce, all_reduce) with
    # whatever world size there is -- a one-rank rehearsal of the RCCL calls on a one-GPU box
    multi = world > 1 or bool(os.environ.get("BENCH_FORCE_COLLECTIVES"))
    # order matters twice: torch initialises the GPU before libmipt is loaded (two HIP clients in one process), and the
    # tracer takes its hardware queues (mipt_create) before RCCL creates its streams -- a process has only
    # GPU_MAX_HW_QUEUES of them and the frame pipeline needs two that are not shared
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.zeros(1, device=dev)
    import cudapathtracer_amd as pt
    from cudapathtracer_amd import dist as ptdist
    t = pt.Tracer(local)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    if a.tile_pixels:
        t.set_option("tile_pixels", a.tile_pixels)
    t.set_option("integrator", a.integrator)
    t.set_option("estimator", a.estimator)
    t.set_option("pipeline_frames", a.pipeline)
    for kv in a.option:
        k, v = kv.split("=")
        t.set_option(k, int(v))
    ctx = dict(pt=pt, ptdist=ptdist, torch=torch, dist=dist, t=t, dev=dev, rank=rank, world=world, multi=multi)
    names = ["cornell", "default", "grid"] if a.workload == "all" else [a.workload]
    head = run_workload(names[0], a, ctx, a.cpu_seconds)
    extra = {}
    for n in names[1:]:
        extra[n] = run_workload(n, a, ctx, a.cpu_seconds / 2)
    if rank == 0:
        out = {"metric": "Mrays/s (extension+shadow)", "value": head["value"], "unit": "Mrays/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "estimator": ("reference", "light sampling (env map + emissive triangles)", "VNDF specular + cosine diffuse lobes", "VNDF + cosine lobes with light sampling")[a.estimator]}
        out.update({k: v for k, v in head.items() if k not in ("value", "unit", "ms_per_step")})
        if extra:
            out["workloads"] = extra
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
