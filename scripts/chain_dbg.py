#!/usr/bin/env python3
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from overlap_exp import mk, L, capi
sets = mk([4096], 4096, reps=8)
for n in (2, 3, 8):
    launches = [[sets[i % len(sets)][0][0]] for i in range(n)]
    for eager in (False,):
        plan = capi.Plan(launches, chained=True)
        t0 = time.time()
        try:
            plan.launch(torch.cuda.current_stream().cuda_stream); plan.status(); r = "ok"
        except Exception as e:
            r = str(e)[-60:]
        print(json.dumps({"n": n, "eager": eager, "chained": plan.chained, "geo": plan.geometry() if plan.chained else None, "result": r, "s": round(time.time() - t0, 3)}), flush=True)
        plan.close()
