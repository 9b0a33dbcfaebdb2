#!/usr/bin/env python3
"""Round-2 experiment driver (GPU box): runs bench.py once per variant and workload and prints one table.

  python3 scripts/exp_r2.py [--workloads cornell,default,grid] [--steps 16] name[:lib.so][:ENV=VAL[:ENV=VAL...]] ...

A variant is a name, optionally a library (path relative to cudapathtracer_amd/, loaded through MIPT_LIB; built with
`make -C cudapathtracer_amd/csrc variant NAME=x EXTRA=-DMIPT_EXP_...`), optionally environment settings and bench
options (opt.key=value -> bench.py --option key=value).  Every run is its own process with a time limit.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
workloads, steps, minsec = ["cornell", "default", "grid"], 16, 0.6
while args and args[0].startswith("--"):
    k = args.pop(0)
    v = args.pop(0)
    if k == "--workloads":
        workloads = v.split(",")
    elif k == "--steps":
        steps = int(v)
    elif k == "--min-seconds":
        minsec = float(v)
rows = []
for spec in args:
    parts = spec.split(":")
    name, env, opts, extra = parts[0], dict(os.environ), [], []
    for p in parts[1:]:
        if p.endswith(".so"):
            env["MIPT_LIB"] = os.path.join(ROOT, "cudapathtracer_amd", p)
        elif p.startswith("opt."):
            opts += ["--option", p[4:]]
        elif p.startswith("ARG"):
            extra += p[3:].split("=")
        else:
            k, v = p.split("=", 1)
            env[k] = v
    res = {}
    for w in workloads:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", str(steps), "--warmup", "3", "--cpu-seconds", "0",
               "--min-seconds", str(minsec), "--trace-loop-frames", "0"] + opts + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                res[w] = "FAILED rc=%d %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:120] if r.stderr.strip() else "")
                if r.returncode in (-6, -11, 124, 137, 134, 139):
                    print(f"{name} {w}: {res[w]} -- stopping", flush=True)
                    sys.exit(1)
                continue
            d = json.loads(line[-1])
            res[w] = (d["value"], d["ms_per_step"], d["rays_per_step_per_gpu"])
        except subprocess.TimeoutExpired:
            print(f"{name} {w}: TIMEOUT -- stopping", flush=True)
            sys.exit(1)
    rows.append((name, res))
    print(name, " | ".join(f"{w}: {res[w][0]:9.1f} Mrays/s {res[w][1]:7.3f} ms" if isinstance(res[w], tuple) else f"{w}: {res[w]}" for w in workloads), flush=True)
base = rows[0][1] if rows else {}
print("\nrelative to", rows[0][0] if rows else "-")
for name, res in rows:
    print(f"{name:28s}", "  ".join(f"{w} {res[w][0] / base[w][0]:6.3f}x" if isinstance(res[w], tuple) and isinstance(base.get(w), tuple) else f"{w}   n/a" for w in workloads))
