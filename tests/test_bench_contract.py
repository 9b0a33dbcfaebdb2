"""bench.py's output contract (one JSON line with the driver's keys plus roofline and cpu_baseline), on a small frame."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--workload", "default", "--estimator", "1"], ["--integrator", "1", "--depth", "4"]])
def test_bench_prints_one_json_line_with_the_contract_keys(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--width", "320", "--height", "180",
           "--cpu-seconds", "1"] + extra
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
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mrays/s" and cb["value"] > 0 and 1 <= cb["cores"] <= 16 and cb["sample"]
    if "--estimator" in extra:
        assert d["shadow_rays"] > 0
    else:
        assert d["shadow_rays"] == 0


@pytest.mark.gpu
def test_bench_multi_gpu_flow_runs_over_rccl_with_one_rank():
    """BENCH_FORCE_COLLECTIVES=1: the N > 1 flow (RCCL process group, sum buffers, warm-up and closing reduce, all_reduce
    of the counters) on the one GPU of the box; the image it tonemaps is the mean of the frames it rendered."""
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ, BENCH_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--width", "320", "--height", "180",
           "--cpu-seconds", "1"]
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
