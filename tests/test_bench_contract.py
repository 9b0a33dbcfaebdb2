"""bench.py's output contract (one JSON line with the driver's keys plus roofline and cpu_baseline), on a small frame."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def check_roofline(rf):
    """the roofline object is a fraction of a bound that can bind: never above 1, whichever bound the workload gets"""
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "valu", "hbm"):
        assert k in rf, k
    assert rf["bound"] in ("valu", "hbm") and rf["bound_from"]
    assert "traffic_frac" in rf["hbm"] and "traffic_gb_per_launch" in rf["hbm"]          # measured traffic rides beside the algorithmic figure
    assert 0.0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-3
    if rf["bound"] == "hbm":
        assert rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    else:
        assert rf["unit"] == "T lane-op/s" and rf["peak"] == 78.65
    assert 0.0 < rf["valu"]["frac"] <= 1.0 and 0.0 < rf["valu"]["frac_of_unpacked_issue_peak"] <= 1.0
    assert rf["hbm"]["frac"] > 0.0
    # PMC traffic is either absent or says where it was measured
    assert rf["traffic"] is None or "profiles/" in rf["traffic_source"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--workload", "cornell"], ["--workload", "default", "--estimator", "1"], ["--workload", "cornell", "--integrator", "1", "--depth", "4"]])
def test_bench_prints_one_json_line_with_the_contract_keys(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--width", "320", "--height", "180",
           "--cpu-seconds", "1", "--min-seconds", "0.05", "--trace-loop-frames", "6"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    check_roofline(d["roofline"])
    assert d["roofline"]["bound"] == "valu"             # both scenes are cache resident
    assert d["repeats"] >= 1 and d["timed_region_s"] > 0
    assert "kernel_ms_per_step" not in d and d["kernel_window_ms"] > 0 and "kernel_span_ms_per_step" in d
    assert 0.0 <= d["anyhit_ray_frac"] < 1.0
    if "--integrator" not in extra:
        assert d["kernel_span_ms_per_step"] > 0
        tl = d["trace_loop"]          # the reference's call sequence: Trace + blocking bitmap read-back per frame
        assert tl["ms_per_frame"] > 0 and tl["mrays_per_s"] > 0 and tl["frames"] == 6 and tl["ldr_bytes_per_frame"] == 320 * 180 * 4
        assert (d["anyhit_ray_frac"] > 0.2) == (extra[1] == "default")     # Default has no emitting mesh: its last-level rays are hit-or-miss
        # Cornell has a lamp: its last-level rays go through the emitters' tree first, and the line says so and carries the rate of
        # the same frames with every ray traced to its closest hit
        if extra[1] == "cornell":
            assert d["emitter_first_ray_frac"] > 0.2 and d["roofline"]["kernel"].endswith(", true>")
            ch = d["closest_hit_only"]
            assert ch["value"] > 0 and ch["ms_per_step"] > 0 and "closest hit" in ch["note"]
        else:
            assert d["emitter_first_ray_frac"] == 0 and d["closest_hit_only"] is None
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mrays/s" and cb["value"] > 0 and 1 <= cb["cores"] <= 16 and cb["sample"]
    if "--estimator" in extra:
        assert d["shadow_rays"] > 0
    else:
        assert d["shadow_rays"] == 0


@pytest.mark.gpu
def test_bench_default_command_carries_all_three_workloads():
    """no --workload: Cornell is the headline, DefaultScene + HDRI and the grid ride along under `workloads`, each with
    its own roofline fraction in (0, 1] (small frames and a small grid here; the driver runs the full sizes)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--width", "256", "--height", "144",
           "--cpu-seconds", "1", "--min-seconds", "0.05", "--grid-n", "96", "--trace-loop-frames", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["workload"].startswith("cornell_") and set(d["workloads"]) == {"default", "grid"}
    check_roofline(d["roofline"])
    for name, w in d["workloads"].items():
        assert w["value"] > 0 and w["ms_per_step"] > 0 and w["config"]["workload"].startswith(name + "_")
        check_roofline(w["roofline"])
        assert w["cpu_baseline"]["value"] > 0 and w["trace_loop"]["ms_per_frame"] > 0
    assert d["workloads"]["grid"]["roofline"]["bound"] == "valu"


@pytest.mark.gpu
def test_bench_multi_gpu_flow_runs_over_rccl_with_one_rank():
    """BENCH_FORCE_COLLECTIVES=1: the N > 1 flow (RCCL process group, sum buffers, warm-up and closing reduce, all_reduce
    of the counters) on the one GPU of the box; the image it tonemaps is the mean of the frames it rendered."""
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, BENCH_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--width", "320", "--height", "180",
           "--cpu-seconds", "1", "--min-seconds", "0", "--workload", "cornell"]       # one repeat: the same frames in both runs
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"] is not None and d["config"]["parallelism"] == "sample-striding x1"
    # same frames, same scene: the ray count per step equals the plain run's
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    d2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert d["rays_per_step_per_gpu"] == d2["rays_per_step_per_gpu"]


# ---- `python bench.py --gpus N` with no launcher: bench.py starts its own N ranks ---------------------------------------
def run_bench(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)
    return r, [l for l in r.stdout.splitlines() if l.strip()]


@pytest.mark.parametrize("n", [2, 3])
def test_bench_gpus_n_without_a_launcher_starts_n_ranks(n):
    """CPU: the launch, rendezvous, frame partition, closing reduce and rank-0 line of `bench.py --gpus N` with nothing but
    the plain command (the kernels need a GPU, so --rehearse-launch traces nothing and says so: value null)."""
    r, lines = run_bench(["--gpus", str(n), "--backend", "gloo", "--rehearse-launch", "--steps", "5"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["frames_per_rank"] == [5] * n and d["value"] is None and "no ray traced" in d["rehearsal"]
    assert d["config"]["parallelism"] == f"sample-striding x{n}"


def test_bench_never_shrinks_the_rank_count_silently():
    """a launcher that started another number of ranks than --gpus asks for is an error, not a note"""
    r, lines = run_bench(["--gpus", "2", "--rehearse-launch"], env=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode == 2 and not lines and "WORLD_SIZE=1" in r.stderr


def visible_gpus():
    """device count from the HIP runtime libmipt is linked against -- NOT through torch: importing torch into the test
    process brings its own copy of the runtime along, and the raw-stream tests that follow then talk to the wrong one"""
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")
    except OSError:
        return 0
    n = C.c_int(0)
    return n.value if hip.hipGetDeviceCount(C.byref(n)) == 0 else 0


def test_bench_without_a_gpu_fails_loudly():
    if visible_gpus() > 0:
        pytest.skip("a GPU is visible")
    for args in (["--gpus", "2"], ["--gpus", "1"]):
        r, lines = run_bench(args)
        assert r.returncode != 0 and not lines and "no GPU" in r.stderr


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_are_visible():
    n = visible_gpus() + 1
    r, lines = run_bench(["--gpus", str(n), "--workload", "cornell", "--width", "64", "--height", "64"])
    assert r.returncode != 0 and not lines and "visible" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_runs_two_ranks_on_the_gpu():
    """the plain command, two ranks sharing the box's one GPU (--share-gpus, gloo because RCCL refuses two ranks on one
    device): n_gpus 2, each rank renders its own frames (same rays per step per GPU as a single rank), one closing reduce"""
    common = ["--steps", "4", "--warmup", "2", "--width", "320", "--height", "180", "--cpu-seconds", "1", "--min-seconds", "0", "--workload", "cornell"]
    r, lines = run_bench(["--gpus", "2", "--share-gpus", "--backend", "gloo"] + common)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "sample-striding x2" and d["scaling"] == "weak"
    check_roofline(d["roofline"])
    r1, lines1 = run_bench(["--gpus", "1"] + common)
    d1 = json.loads(lines1[0])
    # frames 2..9 over two ranks against frames 2..5 on one: per-frame ray counts differ by well under a percent
    assert abs(d["rays_per_step_per_gpu"] / d1["rays_per_step_per_gpu"] - 1) < 0.01
