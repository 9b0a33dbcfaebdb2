#!/usr/bin/env python3
"""VALU issue calibration on the GPU box: what one gfx950 SIMD issues per cycle.

  python3 scripts/issue_calibration.py [iters]            -> one JSON line per occupancy (1, 2, 4, 8 waves per SIMD)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -- python3 scripts/issue_calibration.py

Every wave of mipt::k_test_issue (csrc/microbench.hip) issues iters x 64 independent v_fma_f32; the kernel stamps its
own shader clock, so  wave-instructions per SIMD-cycle = waves_per_simd * iters * 64 / median_cycles  needs no
assumption about the clock the chip holds.  lane-ops/s = wave-instructions x 64 / HIP-event seconds.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cudapathtracer_amd as pt

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t = pt.Tracer(0)
for packed, wps in [(0, w) for w in (1, 2, 4, 8)] + [(1, w) for w in (1, 2, 4, 8)]:
    best = None
    for rep in range(3):
        r = t.test_issue(wps, iters, packed)
        if best is None or r.ms < best.ms:
            best = r
    r = best
    per_simd_cycle = wps * iters * 64 / r.median_cycles
    clock_ghz = r.clock_ghz
    print(json.dumps({"instruction": "v_pk_fma_f32" if packed else "v_fma_f32", "fma_per_lane_per_instruction": 2 if packed else 1,
                      "fma_lane_ops_per_s_T": round(r.wave_instructions * 64 * (2 if packed else 1) / (r.ms * 1e-3) / 1e12, 2),
                      "waves_per_simd": wps, "iters": iters, "waves": r.waves, "simds": r.simds,
                      "wave_instructions": r.wave_instructions, "event_ms": round(r.ms, 4),
                      "median_cycles_per_wave": r.median_cycles, "max_cycles_per_wave": r.max_cycles,
                      "cycles_per_wave_instruction_one_wave": round(r.median_cycles / (iters * 64), 3),
                      "wave_instructions_per_simd_cycle": round(per_simd_cycle, 4),
                      "in_kernel_clock_ghz": round(clock_ghz, 3),
                      "wave_instructions_per_simd_cycle_by_event_time": round(r.wave_instructions / r.simds / (r.ms * 1e-3 * clock_ghz * 1e9), 4)}))
