#!/usr/bin/env python3
"""Persistent stream kernel: stream-only time and per-wave timestamps (start, x staged, end)."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinychatengine_amd import capi, quantize
dev = torch.device("cuda:0"); L = capi.lib()
def mk(segs, K, reps=8):
    G = 128; zw = quantize.calculate_zeros_width(K, G)
    x = torch.randn(1, K, device=dev).to(torch.float16); sets = []
    for rep in range(reps):
        ds, keep = [], []
        for n in segs:
            qw = torch.randint(-2**31, 2**31 - 1, (n, K // 8), dtype=torch.int32, device=dev)
            sc = (torch.rand((n, zw * 8), device=dev) * 0.01 + 0.001).to(torch.float16)
            zp = torch.full((n, zw), -2004318072, dtype=torch.int32, device=dev)
            out = torch.empty(1, n, dtype=torch.float16, device=dev); keep += [qw, sc, zp, out, x]
            ds.append(capi.W4A16Desc(M=1, N=n, K=K, group_size=G, A=x.data_ptr(), qweight=qw.data_ptr(), scales=sc.data_ptr(), zeros=zp.data_ptr(), C=out.data_ptr()))
        sets.append(((capi.W4A16Desc * len(ds))(*ds), keep))
    return sets
def timeit(sets, nseg, launches=64):
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sp = C.c_void_p(s.cuda_stream)
        with torch.cuda.graph(g, stream=s):
            for i in range(launches): capi.check(L.tce_w4a16_forward_group(sets[i % len(sets)][0], nseg, sp))
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [g.replay() for _ in range(3)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * launches)
for (segs, K) in [([11008, 11008], 4096), ([4096], 11008)]:
    sets = mk(segs, K, reps=24 if sum(segs) < 100000 else 5)
    for cfg in [(2, 16, 2), (2, 8, 2)]:
        capi.set_gemv_config(cfg[0], cfg[1], 0, cfg[2])
        for mode, name in ((0, "normal"), (1, "stream-only"), (4, "compute-only")):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            us = timeit(sets, len(segs))
            print(json.dumps({"segs": segs, "cfg": cfg, "mode": name, "us": round(us, 2)}), flush=True)
        nw = 256 * cfg[1]
        buf = torch.zeros(nw * 4, dtype=torch.int64, device=dev)
        capi.check(L.tce_w4a16_set_debug_buffer(C.c_void_p(buf.data_ptr()))); capi.check(L.tce_w4a16_set_debug_mode(2))
        for i in range(4): capi.check(L.tce_w4a16_forward_group(sets[i][0], len(segs), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        raw = buf.cpu().numpy().reshape(nw, 4).astype(np.float64); raw = raw[raw[:, 0] > 0]
        ghz = float(np.median(raw[:, 3] / ((raw[:, 2] - raw[:, 0]) * 10.0)))
        t = raw * 10.0; t -= t[:, 0].min()
        q = lambda a: [round(float(np.percentile(a, p)) / 1e3, 2) for p in (0, 10, 50, 90, 100)]
        print(json.dumps({"segs": segs, "cfg": cfg, "waves": len(t), "start_us": q(t[:, 0]), "x_ready_us": q(t[:, 1]), "end_us": q(t[:, 2]), "shader_GHz": round(ghz, 3)}), flush=True)
        L.tce_w4a16_set_debug_mode(0); L.tce_w4a16_set_debug_buffer(None)
capi.set_gemv_config()
