#!/usr/bin/env python3
"""Per-kernel resource table of libmipt.so's gfx950 code object (VGPR / SGPR / spills / LDS / scratch / occupancy).

  python scripts/code_object.py [path/to/libmipt.so] > profiles/rN/code_object.txt

Runs on the CPU: unbundles the .hip_fatbin with llvm-objdump --offloading into a temp dir and reads the
amdhsa.kernels metadata with llvm-readelf --notes.  waves/SIMD is the register-file limit of MI355X_MICROARCH.md
(512 VGPR+AGPR per lane per SIMD, allocation granule 8), capped by LDS (160 KiB / CU) for 256-thread blocks.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "cudapathtracer_amd", "libmipt.so")
    tmp = tempfile.mkdtemp(prefix="mipt_co_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
        co = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)          # one code object per .hip translation unit
        assert co, "no gfx950 code object found"
        notes = "".join(subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout for f in co)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*(-\s+)?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        dash, key, val = m.groups()
        if key == "agpr_count" and dash:
            cur = {}
            kernels.append(cur)
        if cur is not None and key in ("agpr_count", "group_segment_fixed_size", "name", "private_segment_fixed_size", "sgpr_count",
                                       "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "max_flat_workgroup_size"):
            cur[key] = val.strip().strip("'")
    kernels = [k for k in kernels if "name" in k]
    names = demangle([k["name"] for k in kernels])
    print(f"# code object of {os.path.relpath(lib, ROOT)} ({co[0].split('--')[-1]}), llvm-readelf --notes; waves/SIMD = min(8, 512 // alloc_vgpr, LDS limit at 256 threads/block)")
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'v_spill':>8} {'s_spill':>8} {'lds_B':>7} {'scratch_B':>10} {'waves/SIMD':>11}  kernel")
    for k, n in sorted(zip(kernels, names), key=lambda x: x[1]):
        v, a = int(k["vgpr_count"]), int(k["agpr_count"])
        alloc = (v + a + 7) // 8 * 8
        lds = int(k["group_segment_fixed_size"])
        occ = min(8, 512 // max(alloc, 8))
        if lds:
            occ = min(occ, (160 * 1024) // lds)      # 256-thread blocks: one wave per SIMD per block
        n = re.sub(r"\(.*\)$", "", n)
        print(f"{v:>5} {a:>5} {int(k['sgpr_count']):>5} {int(k['vgpr_spill_count']):>8} {int(k['sgpr_spill_count']):>8} {lds:>7} "
              f"{int(k['private_segment_fixed_size']):>10} {occ:>11}  {n}")


if __name__ == "__main__":
    main()
