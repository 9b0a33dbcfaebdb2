#!/usr/bin/env python3
"""Benchmark of the hot path: PathTracer::Trace (one sample per pixel per step) on N MI355X.

  python bench.py --gpus 1 --steps 8 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one frame (1 spp) of the workload per GPU.  Default workload = BASELINE.json configs[1]: Cornell box,
1920x1080, 8 bounces (`--workload default` = configs[2]: DefaultScene + synthetic HDRI, 12 bounces; `--workload grid`
= the configs[4] stand-in: 10 M-triangle displaced grid, 3840x2160, 4 bounces -- the HBM-stressing case).  With N > 1 the
scene is replicated, rank r renders frame indices r, r+N, ... into a private sum buffer (weak scaling: one frame per
GPU per step) and ONE RCCL reduce of the accumulation buffers closes the timed region.

Prints ONE JSON line on rank 0: whole-job Mrays/s (closest-hit extension rays; the reference traces no shadow rays)
plus spp/s, a roofline object for the dominant kernel (k_wavestack, the persistent extend+shade kernel) and, at
N = 1, the CPU oracle timed on this host as a reported baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=32)
    p.add_argument("--warmup", type=int, default=4)
    p.add_argument("--workload", default="cornell", choices=["cornell", "default", "grid"])
    p.add_argument("--width", type=int, default=0, help="0 = 1920 (cornell, default) or 3840 (grid)")
    p.add_argument("--height", type=int, default=0, help="0 = 1080 (cornell, default) or 2160 (grid)")
    p.add_argument("--depth", type=int, default=0, help="0 = 8 for cornell, 12 for default, 4 for grid (BASELINE.json)")
    p.add_argument("--grid-n", type=int, default=2236, help="grid workload: n x n quads = 2 n^2 triangles (2236 -> 10.0 M)")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU oracle budget (0 disables cpu_baseline)")
    p.add_argument("--tile-pixels", type=int, default=0)
    p.add_argument("--integrator", type=int, default=0, help="0 persistent wave stacks (default), 1 level-synchronous queues")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI (default); gloo = rehearsal of the N>1 flow through host memory")
    p.add_argument("--node-width", type=int, default=0, help="BVH node layout: 0 library default, 2 fp32 BVH2, 4 quantised 4-wide")
    p.add_argument("--pipeline", type=int, default=1, help="1 (default): consecutive frames alternate between two streams so a frame's tail overlaps the next frame; 0: one stream")
    p.add_argument("--estimator", type=int, default=0, help="0 the reference's estimator (default, the parity configuration); 1 env-map / emitter sampling with occlusion rays; 2 VNDF-sampled specular + cosine-weighted diffuse lobes; 3 both")
    p.add_argument("--traffic-gb", type=float, default=None, help="HBM bytes per dominant-kernel launch from a PMC run (GB)")
    return p.parse_args()


def oracle_sample(pt, data, bvh, width, settings, seconds, threads, estimator=0):
    """Times the CPU oracle (same flat BVH, node-keyed RNG) on evenly spaced rows of frame 0 and returns its rate
    and the mean node visits / triangle tests per non-primary ray (the figures B_ray is computed from)."""
    from oracle import orc                      # cpu_baseline leg only: checker / reported baseline
    import ctypes as C
    o = orc.Scene(data)
    o.set_flat_bvh(*bvh, width)
    o.set_estimator(estimator)
    W, H = settings.viewportWidth, settings.viewportHeight

    def run(rows, depth):
        s = pt.Settings()
        C.memmove(C.addressof(s), C.addressof(settings), C.sizeof(pt.Settings))
        s.maxDepth = depth
        tot = dict(rays=0, nodes=0, tris=0, hits=0, sec=0.0)
        acc = np.zeros((H, W, 4), np.float32)
        for y in rows:
            t0 = time.perf_counter()
            r = o.render(s, orc.ISECT_FLAT, orc.RNG_NODE, orc.MATH_DET, 0, 1, rect=(0, y, W, y + 1), acc=acc, want_ldr=False, nthreads=threads)
            tot["sec"] += time.perf_counter() - t0
            st = r["stats"]
            tot["rays"] += st.rays; tot["nodes"] += st.node_visits; tot["tris"] += st.tri_tests; tot["hits"] += st.hits
        return tot

    probe_rows = [H // 3, H // 2]
    probe = run(probe_rows, settings.maxDepth)
    per_row = max(probe["sec"] / len(probe_rows), 1e-4)
    n_rows = int(min(H, max(4, seconds / per_row)))
    if n_rows >= H:
        # the whole frame fits the budget: time whole frames (one call each, no per-row call overhead), as many as fit
        n_frames = int(min(16, max(1, seconds / (per_row * H))))

        def run_frames(depth):
            s = pt.Settings()
            C.memmove(C.addressof(s), C.addressof(settings), C.sizeof(pt.Settings))
            s.maxDepth = depth
            t0 = time.perf_counter()
            r = o.render(s, orc.ISECT_FLAT, orc.RNG_NODE, orc.MATH_DET, 0, n_frames, want_ldr=False, nthreads=threads)
            st = r["stats"]
            return dict(rays=st.rays, nodes=st.node_visits, tris=st.tri_tests, hits=st.hits, sec=time.perf_counter() - t0)
        full, prim = run_frames(settings.maxDepth), run_frames(1)
        desc = f"frames 0..{n_frames - 1} of the same workload, whole"
    else:
        rows = sorted(set(int(round(i * (H - 1) / max(n_rows - 1, 1))) for i in range(n_rows)))
        full = run(rows, settings.maxDepth)
        prim = run(rows, 1)
        desc = f"{len(rows)} evenly spaced rows of frame 0 of the same workload"
    sec_rays = max(full["rays"] - prim["rays"], 1)
    return dict(desc=desc, rays=full["rays"], sec=full["sec"], mrays=full["rays"] / full["sec"] / 1e6,
                n_node=(full["nodes"] - prim["nodes"]) / sec_rays, n_tri=(full["tris"] - prim["tris"]) / sec_rays,
                miss_frac=1.0 - (full["hits"] - prim["hits"]) / sec_rays,
                n_node_all=full["nodes"] / max(full["rays"], 1), n_tri_all=full["tris"] / max(full["rays"], 1),
                miss_all=1.0 - full["hits"] / max(full["rays"], 1))


def dbg(msg):
    if os.environ.get("BENCH_DEBUG"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    a = parse()
    if os.environ.get("BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get("BENCH_DEBUG_DUMP_AFTER", "40")), exit=False)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    local = local % max(torch.cuda.device_count(), 1)      # rehearsals may put several ranks on one GPU
    # BENCH_FORCE_COLLECTIVES=1: run the N > 1 flow (process group, warm-up reduce, sum buffers, reduce, all_reduce) with
    # whatever world size there is -- a one-rank rehearsal of the RCCL calls on a one-GPU box
    multi = world > 1 or bool(os.environ.get("BENCH_FORCE_COLLECTIVES"))
    if a.gpus != world and rank == 0:
        print(f"note: --gpus {a.gpus} but WORLD_SIZE {world}; using {world} process(es)", file=sys.stderr)
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

    depth = a.depth or {"cornell": 8, "default": 12, "grid": 4}[a.workload]
    W = a.width or (3840 if a.workload == "grid" else 1920)
    H = a.height or (2160 if a.workload == "grid" else 1080)
    if a.workload == "cornell":
        data = pt.builtin_scene(0)
        cam = pt.make_camera(**pt.CORNELL_CAMERA)
    elif a.workload == "default":
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

    dbg("scene data ready")
    if a.tile_pixels:
        t.set_option("tile_pixels", a.tile_pixels)
    t.set_option("integrator", a.integrator)
    t.set_option("estimator", a.estimator)
    t.set_option("pipeline_frames", a.pipeline)
    if a.workload == "grid" and multi:
        # 10 M triangles: rank 0 builds the BVH and writes the cache file, the others load it instead of building eight times
        cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"mipt_bench_grid{a.grid_n}_w{a.node_width}.bvh")
        sc = pt.Scene(t, data, node_width=a.node_width, bvh_cache=cache) if rank == 0 else None
        dist.barrier()
        if rank != 0:
            sc = pt.Scene(t, data, node_width=a.node_width, bvh_cache=cache)
    else:
        sc = pt.Scene(t, data, node_width=a.node_width)
    dbg("scene uploaded")
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    ldr = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    sum_mode = 1 if multi else 0

    def step(i):   # global step index -> this rank's frame
        frame = i * world + rank
        t.render(sc, settings, frame, 1, acc.data_ptr(), None if sum_mode else ldr.data_ptr(), 1, sum_mode)

    def barrier():
        t.synchronize()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()

    for i in range(a.warmup):
        step(i)
    dbg("warm-up frames queued")
    if multi:   # warm the RCCL path too, then start the timed run from a clean sum buffer
        t.synchronize()
        dbg("warm-up frames done")
        # the whole closing sequence once, untimed: reduce, scale, tonemap (first use of a torch kernel or an RCCL
        # communicator costs tens of milliseconds)
        warm = ptdist.reduce_accumulation(acc.clone() if a.backend == "nccl" else acc.cpu(), max(a.warmup, 1) * world, dst=0)
        if rank == 0:
            warm = warm.to(dev)
            torch.cuda.synchronize()
            t.tonemap(warm.data_ptr(), 1.0, W * H, ldr.data_ptr())
        dbg("warm-up reduce done")
        acc.zero_()
    dbg("warm-up done")
    barrier()
    t.reset_counters()
    t.set_option("profile_levels", 1)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    t.synchronize()
    t_frames = time.perf_counter() - t0
    if multi:   # the path's one exchange step: reduce the sum buffers onto rank 0, scale, tonemap
        mean = ptdist.reduce_accumulation(acc if a.backend == "nccl" else acc.cpu(), a.steps * world, dst=0)
        if rank == 0:
            mean = mean.to(dev)
            torch.cuda.synchronize()        # torch's stream produced `mean`; the tracer reads it on its own stream
            t.tonemap(mean.data_ptr(), 1.0, W * H, ldr.data_ptr())
    barrier()
    dt = time.perf_counter() - t0
    dbg(f"timed region done: frames {t_frames * 1e3:.2f} ms, closing reduce + tonemap + barrier {(dt - t_frames) * 1e3:.2f} ms")
    t.set_option("profile_levels", 0)

    cnt = t.counters()
    prof = t.level_profile()
    cdev = dev if a.backend == "nccl" else torch.device("cpu")
    rays = torch.tensor([float(cnt.rays), float(cnt.shadow_rays)], dtype=torch.float64, device=cdev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if multi:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_rays, total_shadow, dt = float(rays[0].item()), float(rays[1].item()), float(tmax.item())

    if rank == 0:
        nn, nr = (len(x) for x in sc.bvh())
        if a.integrator == 0:    # one k_wavestack launch per frame traces every ray of the frame
            lds = sc.bvh_width == 2 and nn * 4 + nr * 3 <= 256          # <LDS scene, 4-wide nodes, light sampling>
            kernel_name = "mipt::k_wavestack<%s, %s, %s, %s>" % tuple("true" if x else "false" for x in (lds, sc.bvh_width == 4, a.estimator in (1, 3), a.estimator >= 2))
            sec_rays, sec_ms, sec_launches = float(cnt.rays), prof.ms[0], prof.launches[0]
        else:                    # k_level<false,*> launches carry the non-primary levels
            kernel_name = "mipt::k_level<false,*>"
            sec_rays = sum(prof.rays[k] for k in range(1, 32))
            sec_ms = sum(prof.ms[k] for k in range(1, 32))
            sec_launches = sum(prof.launches[k] for k in range(1, 32))
        cpu = None
        n_node = n_tri = miss = None
        if a.cpu_seconds > 0:
            threads = min(os.cpu_count() or 1, 16)
            cpu = oracle_sample(pt, data, sc.bvh(), sc.bvh_width, settings, a.cpu_seconds if world == 1 else min(a.cpu_seconds, 3.0), threads, a.estimator)
            n_node, n_tri, miss = cpu["n_node"], cpu["n_tri"], cpu["miss_frac"]
            if a.integrator == 0:
                n_node, n_tri, miss = cpu["n_node_all"], cpu["n_tri_all"], cpu["miss_all"]
        traffic = a.traffic_gb
        pmc = {}
        default_size = (W, H) == ((3840, 2160) if a.workload == "grid" else (1920, 1080)) and (a.workload != "grid" or a.grid_n == 2236)
        if traffic is None and a.integrator == 0 and default_size and not a.depth and not a.node_width and not a.estimator:
            # HBM bytes per launch from the committed PMC passes of this same command (scripts/pmc_passes.sh;
            # FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md, plus WRITE_SIZE)
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "r1", "pmc_traffic.json")))
                traffic = round(pj[a.workload]["hbm_gb_per_launch"], 2)
                pmc = {k: pj[a.workload][k] for k in ("valu_busy", "lane_utilisation", "wait_frac", "l2_hit") if k in pj[a.workload]}
            except Exception:
                traffic = None
        roof = None
        if n_node is not None and sec_ms > 0:
            # B_ray = 64*N_node + 48*N_tri + 64 (+12 per env-map miss); units per launch = rays of that launch
            b_ray = 64.0 * n_node + 48.0 * n_tri + 64.0 + (12.0 * miss if data.env is not None else 0.0)
            achieved = sec_rays * b_ray / (sec_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "GB per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)",
                    "algorithmic_gb_per_launch": round(sec_rays / max(sec_launches, 1) * b_ray / 1e9, 2),
                    "kernel": kernel_name, "bytes_per_ray": round(b_ray, 1), "n_node_per_ray": round(n_node, 3),
                    "n_tri_per_ray": round(n_tri, 3), "rays_per_launch": round(sec_rays / max(sec_launches, 1), 1),
                    "avg_launch_ms": round(sec_ms / max(sec_launches, 1), 4), "launches": int(sec_launches),
                    # consecutive frames run on two streams, so a launch's event window includes the time it shares the GPU
                    # with (or queues behind) its neighbour; the wall figures divide the timed region by the launches instead
                    "frames_pipelined": bool(a.pipeline and a.integrator == 0),
                    "wall_ms_per_launch": round(dt * 1e3 / a.steps, 4),
                    "achieved_wall": round(sec_rays / max(sec_launches, 1) * b_ray / (dt / a.steps) / 1e9, 1),
                    "frac_wall": round(sec_rays / max(sec_launches, 1) * b_ray / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, 4),
                    "working_set_mb": round((nn * 64 + nr * 48) / 1e6, 2),
                    # issue-side PMC figures of the same command (profiles/r1, scripts/pmc_to_json.py): what binds when HBM does not
                    "pmc": pmc or None,
                    "note": ("working set (BVH+triangles) is cache/LDS resident for this workload: algorithmic bytes are served on-chip, so frac is not an HBM utilisation"
                             if (nn * 64 + nr * 48) < 32e6 else
                             "working set exceeds the 32 MB of L2; algorithmic bytes above `traffic` are re-reads served by L2 / Infinity Cache")}
        out = {
            "metric": "Mrays/s (extension+shadow)", "value": round(total_rays / dt / 1e6, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.workload}_{W}x{H}_{depth}bounce_1spp_per_gpu_step" + (f"_n{a.grid_n}" if a.workload == "grid" else ""), "width": W, "height": H, "max_depth": depth,
                       "triangles": data.triangle_count, "bvh": "sbvh" + ("-q4" if sc.bvh_width == 4 else ""), "rng": "node", "integrator": "wave-stacks" if a.integrator == 0 else "level-queues", "parallelism": f"sample-striding x{world}",
                       "env_map": None if data.env is None else "synthetic 2048x1024"},
            "spp_per_s": round(a.steps * world / dt, 3), "rays_per_step_per_gpu": round(total_rays / a.steps / world, 1),
            "shadow_rays": round(total_shadow / a.steps / world, 1), "estimator": ("reference", "light sampling (env map + emissive triangles)", "VNDF specular + cosine diffuse lobes", "VNDF + cosine lobes with light sampling")[a.estimator], "kernel_ms_per_step": round(cnt.kernel_ms / a.steps, 3),
            "level_ms": [round(prof.ms[k] / a.steps, 3) for k in range(depth)] if a.integrator else None,
            "level_mrays": [round(prof.rays[k] / a.steps / 1e6, 2) for k in range(depth)] if a.integrator else None,
            "accumulate_ms": round(prof.accumulate_ms / a.steps, 3),
            "roofline": roof,
        }
        if cpu is not None and world == 1:
            out["cpu_baseline"] = {"value": round(cpu["mrays"], 2), "unit": "Mrays/s", "cores": min(os.cpu_count() or 1, 16), "kind": "port",
                                   "sample": f"{cpu['desc']} ({cpu['rays']} rays, {cpu['sec']:.1f} s), same flat BVH"}
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
