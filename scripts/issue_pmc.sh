#!/bin/bash
# SQ counters of the issue-calibration launches (one rocprofv3 --pmc pass, counters only), next to the kernel's own clock.
# usage (GPU box): scripts/issue_pmc.sh <round>    -> gpurun_out/issue_<round>/{calibration.jsonl,pmc_summary.txt}
set -u
ROUND=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/issue_$ROUND
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export PYTHONPATH=$R
timeout -k 10 120 python3 $R/scripts/issue_calibration.py > $OUT/calibration.jsonl || exit 1
cat $OUT/calibration.jsonl
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc -o p -- python3 $R/scripts/issue_calibration.py > $OUT/calibration_under_pmc.jsonl 2> $OUT/pmc.err || { echo "pmc pass failed"; tail -5 $OUT/pmc.err; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
rows = defaultdict(dict)
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_test_issue" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[int(r["Dispatch_Id"])]["grid"] = int(r["Grid_Size"])
with open(os.path.join(out, "pmc_summary.txt"), "w") as fo:
    names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"]
    hdr = "dispatch grid " + " ".join(names) + "  ACTIVE_VALU/INSTS_VALU  INSTS_VALU/(1024*GUI/8)"
    print(hdr); fo.write(hdr + "\n")
    for d in sorted(rows):
        c = rows[d]
        line = f"{d} {c['grid']} " + " ".join(f"{c.get(n, 0):.0f}" for n in names)
        if c.get("SQ_INSTS_VALU") and c.get("GRBM_GUI_ACTIVE"):
            line += f"  {c['SQ_ACTIVE_INST_VALU'] / c['SQ_INSTS_VALU']:.3f}  {c['SQ_INSTS_VALU'] / (1024 * c['GRBM_GUI_ACTIVE'] / 8):.4f}"
        print(line); fo.write(line + "\n")
PY
rm -rf $OUT/pmc
echo ISSUE_PMC_DONE
