#!/usr/bin/env python3
"""One decode token's linears (129 launches) issued by tce_plan_launch as a graph replay vs launch by launch from C (debug mode 15001): ms per token, alternating."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.decode import DecodeLinears, SHAPES
L = capi.lib()
dev = torch.device("cuda:0")
for wl in ("llama3-8b", "baseline-named"):
    dl = DecodeLinears(SHAPES[wl], dev, prepack=True)
    plan = dl.make_plan()
    s = torch.cuda.Stream()
    def run(mode, n=200):
        capi.check(L.tce_w4a16_set_debug_mode(mode))
        for _ in range(10): plan.launch(s.cuda_stream)
        s.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record(s)
        for _ in range(n): plan.launch(s.cuda_stream)
        b.record(s)
        t_issue = time.perf_counter() - t0
        s.synchronize()
        return a.elapsed_time(b) / n, t_issue / n * 1e3
    res = {"workload": wl, "launches": plan.n_launches}
    for rnd in range(3):
        for name, mode in (("graph", 15000), ("eager", 15001)):
            ms, host = run(mode)
            res.setdefault(name, []).append([round(ms, 4), round(host, 4)])
    capi.check(L.tce_w4a16_set_debug_mode(15000))
    print(json.dumps(res), flush=True)
    del plan, dl
    torch.cuda.empty_cache()
