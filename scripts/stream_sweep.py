#!/usr/bin/env python3
"""Persistent GEMV: sweep (workgroups per CU, rows per row group, waves, ring depth) against the row-block kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from overlap_exp import mk, timeit, L, capi
def main():
    for (segs, K) in [([11008, 11008], 4096), ([12288], 4096), ([4096], 4096), ([4096], 11008), ([14336, 14336], 4096), ([128256], 4096)]:
        sets = mk(segs, K, reps=24 if sum(segs) < 100000 else 5)
        row = {"segs": segs, "K": K}
        capi.set_gemv_config(); row["rowblock"] = round(timeit(sets, len(segs), 1), 2)
        best = None
        for bpc, nw in ((1, 16), (2, 8), (2, 12), (3, 8), (1, 12), (2, 16)):
            for rows in (1, 2):
                for depth in (2, 3):
                    if rows == 4 and depth == 3: continue
                    try:
                        capi.set_gemv_config(10 * bpc + rows, nw, 0, depth)
                        us = round(timeit(sets, len(segs), 1), 2)
                    except Exception as e:
                        us = None
                    row[f"{bpc}x{nw} r{rows} d{depth}"] = us
                    if us and (best is None or us < best[0]): best = (us, f"{bpc}x{nw} r{rows} d{depth}")
        row["best"] = best
        capi.set_gemv_config()
        print(json.dumps(row), flush=True)
if __name__ == "__main__":
    main()
