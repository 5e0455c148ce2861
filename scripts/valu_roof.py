#!/usr/bin/env python3
"""Calibrates the VALU roof of the sketch's issue-bound kernels: cycles per wave-instruction per SIMD
for the integer instructions ntHash is made of, at 1, 2, 4 and 8 waves per SIMD.

  python scripts/valu_roof.py > profiles/r02_valu_roof.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ntsynt_amd.device import Context
    ctx = Context(0)
    rows = []
    for kind in Context.VALU_KINDS:
        for wps in (1, 2, 4, 8):
            rows.append(ctx.bench_valu(kind, wps, 40000))
    best = {}
    for r in rows:
        b = best.setdefault(r["instruction"], r)
        if r["cycles_per_wave_instr_per_simd"] < b["cycles_per_wave_instr_per_simd"]:
            best[r["instruction"]] = r
    out = {"what": "integer VALU issue rate on gfx950, eight independent chains per lane, s_memtime inside the kernel",
           "guide": "MI355X_MICROARCH.md: v_fma_f32 (wave64) 2 cycles per SIMD",
           "best_cycles_per_wave_instr_per_simd": {k: v["cycles_per_wave_instr_per_simd"] for k, v in best.items()},
           "rows": rows}
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
