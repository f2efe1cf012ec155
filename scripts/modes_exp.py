#!/usr/bin/env python3
"""Row-block GEMV: the same launch with pieces removed (normal / stream-only / compute-only) per variant."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from overlap_exp import mk, timeit, L, capi
def main():
    cases = [([11008, 11008], 4096), ([4096], 11008), ([4096], 4096)]
    for (segs, K) in cases:
        sets = mk(segs, K, reps=24)
        for cfg in [None, (4, 8, 1, 1), (4, 4, 1, 1), (2, 4, 1, 2), (2, 8, 1, 2), (1, 4, 1, 2)]:
            row = {"segs": segs, "K": K, "cfg": cfg}
            for mode, name in ((0, "normal"), (3, "xfirst"), (1, "stream"), (4, "compute")):
                try:
                    if cfg is None: capi.set_gemv_config()
                    else: capi.set_gemv_config(*cfg)
                    capi.check(L.tce_w4a16_set_debug_mode(mode))
                    row[name] = round(timeit(sets, len(segs), 1), 2)
                except Exception as e:
                    row[name] = str(e)[:60]
            print(json.dumps(row), flush=True)
    L.tce_w4a16_set_debug_mode(0); capi.set_gemv_config()
if __name__ == "__main__":
    main()
