"""Turns the passes of scripts/pmc_passes.sh into profiles/<round>/pmc_traffic.json: per workload, for the dominant kernel
(k_wavestack), HBM bytes per launch and the issue-side figures bench.py reports next to the HBM roofline.

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM
section), so fetched bytes are doubled before they are added to the written bytes.
valu_busy = 4 * SQ_ACTIVE_INST_VALU / (SIMDs * cycles), cycles = GRBM_GUI_ACTIVE / 8 XCDs, 1024 SIMDs: an un-packed wave64
VALU instruction occupies its SIMD for four cycles and SQ_ACTIVE_INST_VALU reads 1.000 per instruction -- both measured with
mipt::k_test_issue (profiles/r2/issue_calibration.jsonl, issue_calibration_pmc.txt); lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU);
wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS).
usage: pmc_to_json.py out.json tag=pass_root [tag=pass_root ...]"""
import csv, glob, hashlib, json, os, sys, time
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def source_sha16():      # same hash as bench.py: the figures are attached only to the sources they were measured on
    h = hashlib.sha256()
    for f in ("kernels.hip", "wavestack_kernel.inc", "kernels.h", "device_math.h", "api.hip", "bvh_builder.cpp"):
        h.update(open(os.path.join(ROOT, "cudapathtracer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]
out = {"_meta": {"source_sha16": source_sha16(), "collected": "rocprofv3 --pmc passes of `bench.py --workload W --steps 2 --warmup 1`, scripts/pmc_passes.sh, " + time.strftime("%Y-%m-%d")}}
for tag, root in (a.split("=") for a in sys.argv[2:]):
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "k_wavestack" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    per = lambda c: tot[c] / max(n[c], 1)          # per launch
    e = {"kernel": "mipt::k_wavestack", "launches": max(n["FETCH_SIZE"], 1),
         "fetch_size_kib_per_launch": per("FETCH_SIZE"), "write_size_kib_per_launch": per("WRITE_SIZE"),
         "hbm_gb_per_launch": (2 * per("FETCH_SIZE") + per("WRITE_SIZE")) * 1024 / 1e9}
    if n["SQ_ACTIVE_INST_VALU"] and n["GRBM_GUI_ACTIVE"]:
        cycles = per("GRBM_GUI_ACTIVE") / 8
        e["valu_busy"] = round(4 * per("SQ_ACTIVE_INST_VALU") / (1024 * cycles), 3)
        e["lane_utilisation"] = round(per("SQ_THREAD_CYCLES_VALU") / (64 * per("SQ_ACTIVE_INST_VALU")), 3)
        e["valu_wave_instructions_per_launch"] = per("SQ_INSTS_VALU")
        e["executed_lane_ops_per_launch"] = per("SQ_THREAD_CYCLES_VALU")      # one count per active lane per VALU instruction (checked on k_test_issue)
    if n["SQ_WAVE_CYCLES"]:
        e["wait_frac"] = round(per("SQ_WAIT_ANY") / per("SQ_WAVE_CYCLES"), 3)
    if n["TCC_HIT_sum"] and n["TCC_MISS_sum"]:
        e["l2_hit"] = round(per("TCC_HIT_sum") / (per("TCC_HIT_sum") + per("TCC_MISS_sum")), 3)
    out[tag] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
