#!/usr/bin/env python3
"""Per-wave timeline of one GEMV launch (debug mode 2): when do waves start, get x, finish the math, end?"""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinychatengine_amd import lab; lab.use_lab()  # (the decode kernel's timestamp form: the diagnostics build)
from tinychatengine_amd import capi, quantize
dev = torch.device("cuda:0")
L = capi.lib()
def run(segs, K, variant):
    G = 128; zw = quantize.calculate_zeros_width(K, G)
    x = torch.randn(1, K, device=dev).to(torch.float16)
    sets = []
    for rep in range(6):
        ds = []
        keep = []
        for n in segs:
            qw = torch.randint(-2**31, 2**31 - 1, (n, K // 8), dtype=torch.int32, device=dev)
            sc = (torch.rand((n, zw * 8), device=dev) * 0.01 + 0.001).to(torch.float16)
            zp = torch.full((n, zw), -2004318072, dtype=torch.int32, device=dev)
            out = torch.empty(1, n, dtype=torch.float16, device=dev)
            keep += [qw, sc, zp, out]
            ds.append(capi.W4A16Desc(M=1, N=n, K=K, group_size=G, A=x.data_ptr(), qweight=qw.data_ptr(), scales=sc.data_ptr(), zeros=zp.data_ptr(), C=out.data_ptr()))
        sets.append(((capi.W4A16Desc * len(ds))(*ds), keep))
    capi.set_gemv_config(*variant)
    rows_per_block = variant[0] * variant[1]
    nblocks = sum((n + rows_per_block - 1) // rows_per_block for n in segs)
    nw = nblocks * variant[1] * variant[2]
    buf = torch.zeros(nw * 4, dtype=torch.int64, device=dev)
    capi.check(L.tce_w4a16_set_debug_buffer(C.c_void_p(buf.data_ptr())))
    capi.check(L.tce_w4a16_set_debug_mode(2))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i in range(6):
        capi.check(L.tce_w4a16_forward_group(sets[i][0], len(segs), st))
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(nw, 4).astype(np.float64) * 10.0  # ns
    L.tce_w4a16_set_debug_mode(0); capi.set_gemv_config()
    t0 = t[:, 0].min()
    t -= t0
    q = lambda a: [round(float(np.percentile(a, p)) / 1e3, 2) for p in (0, 10, 50, 90, 100)]
    rec = {"segs": segs, "K": K, "variant": variant, "waves": nw, "start_us p0/10/50/90/100": q(t[:, 0]), "x_ready_us": q(t[:, 1]),
           "math_done_us": q(t[:, 2]), "end_us": q(t[:, 3]), "lifetime_us": q(t[:, 3] - t[:, 0]), "stage_us": q(t[:, 1] - t[:, 0]),
           "compute_us": q(t[:, 2] - t[:, 1])}
    # concurrency profile: waves alive per microsecond
    end = t[:, 3].max()
    edges = np.arange(0, end + 500, 500)
    alive = [int(((t[:, 0] <= e) & (t[:, 3] > e)).sum()) for e in edges]
    rec["alive_every_0.5us"] = alive
    print(json.dumps(rec), flush=True)
run([11008, 11008], 4096, (4, 4, 1, 1))
run([11008, 11008], 4096, (4, 8, 1, 1))
run([11008, 11008], 4096, (2, 4, 1, 2))
run([4096], 4096, (2, 4, 1, 2))
run([4096], 11008, (2, 4, 1, 2))
