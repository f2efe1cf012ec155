#!/usr/bin/env python3
"""Chained vs stream-ordered plans: time per launch for repeated decode-shaped launches (independent weights per launch)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from overlap_exp import mk, L, capi
def time_plan(plan, reps=5):
    s = torch.cuda.current_stream().cuda_stream
    plan.launch(s); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): plan.launch(s)
    e1.record(); torch.cuda.synchronize()
    plan.status()
    return e0.elapsed_time(e1) * 1e3 / (reps * plan.n_launches)
def main():
    for (segs, K) in [([11008, 11008], 4096), ([4096], 11008), ([4096], 4096), ([12288], 4096)]:
        sets = mk(segs, K, reps=24)
        launches = [[sets[i % len(sets)][0][j] for j in range(len(segs))] for i in range(48)]
        row = {"segs": segs, "K": K}
        for name, cfg, chained in (("rowblock", None, False), ("persist", (0, 8, 0, 0), False), ("persist16", (0, 16, 0, 0), False), ("chained", None, True),
                                   ("chained r2d3", (2, 16, 0, 3), True), ("chained r4d2", (4, 16, 0, 2), True), ("chained r2d2", (2, 16, 0, 2), True), ("chained r1d3", (1, 16, 0, 3), True),
                                   ("chained 2x8 r2d3", (22, 8, 0, 3), True), ("chained 2x8 r1d3", (21, 8, 0, 3), True)):
            try:
                if cfg is None: capi.set_gemv_config()
                else: capi.set_gemv_config(*cfg)
                if chained and cfg is not None: pass
                plan = capi.Plan(launches, chained=chained)
                row[name] = round(time_plan(plan), 2)
                if chained and not plan.chained: row[name] = "not chained"
                plan.close()
            except Exception as e:
                row[name] = str(e)[:80]
        capi.set_gemv_config()
        print(json.dumps(row), flush=True)
if __name__ == "__main__":
    main()
