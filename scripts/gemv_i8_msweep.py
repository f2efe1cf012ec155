#!/usr/bin/env python3
"""Batched decode on the packed copies: the token plan (129 launches) at M = 1..4 rows -- the int8-contraction kernel spends its sixteen output columns on rows x planes,
so M = 2 / 4 sequences read the weights once -- against the paths the same descriptors take with that kernel off (fp16 GEMV for M <= 2, the small-batch kernel for 3 / 4)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.decode import SHAPES, DecodeLinears
dev = torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "baseline-named"
for m in (1, 2, 3, 4):
    dl = DecodeLinears(SHAPES[model], device=dev, group_size=128, m=m, prepack=True)
    row = {"model": model, "M": m}
    for name, mode, tiles in (("i8", 0, 0), ("i8_two_tiles_per_wave", 0, 2), ("i8_off", 1, 0)):
        if tiles and not os.environ.get("MSWEEP_TWO_TILES"): continue
        capi.set_gemv_i8(mode, tiles)
        plan = dl.make_plan()
        capi.set_gemv_i8()
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(5): plan.launch(s)
        torch.cuda.synchronize()
        ts = []
        for rnd in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): plan.launch(s)
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 20)
        row[name] = {"ms_per_step": round(min(ts), 4), "sequence_tokens_per_s": round(m * 1e3 / min(ts), 1)}
        del plan
    print(json.dumps(row), flush=True)
    del dl
    torch.cuda.empty_cache()
