#!/usr/bin/env python3
"""Persistent GEMV: per-wave timestamps (start, x staged, end) broken down by XCD, by wave slot and by workgroup."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinychatengine_amd import lab; lab.use_lab()  # (the persistent kernel's timestamp form: the diagnostics build)
from overlap_exp import mk, timeit, L, capi
def main():
    for (segs, K) in [([11008, 11008], 4096), ([128256], 4096)]:
        sets = mk(segs, K, reps=24 if sum(segs) < 100000 else 5)
        for cfg in [(12, 16, 2), (22, 8, 2), (14, 16, 2)]:
            bpc, rows, nw = cfg[0] // 10, cfg[0] % 10, cfg[1]
            capi.set_gemv_config(cfg[0], nw, 0, cfg[2])
            capi.check(L.tce_w4a16_set_debug_mode(0))
            us = timeit(sets, len(segs), 1)
            nwaves = 256 * bpc * nw
            buf = torch.zeros(nwaves * 4, dtype=torch.int64, device="cuda")
            capi.check(L.tce_w4a16_set_debug_buffer(C.c_void_p(buf.data_ptr()))); capi.check(L.tce_w4a16_set_debug_mode(2))
            for i in range(4): capi.check(L.tce_w4a16_forward_group(sets[i][0], len(segs), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            raw = buf.cpu().numpy().reshape(nwaves, 4).astype(np.float64)
            L.tce_w4a16_set_debug_mode(0); L.tce_w4a16_set_debug_buffer(None)
            ok = raw[:, 0] > 0
            t = raw * 10.0 / 1e3; t0 = t[ok, 0].min(); end = t[:, 2] - t0; xr = t[:, 1] - t0
            blk = np.arange(nwaves) // nw; slot = np.arange(nwaves) % nw; xcd = blk % 8
            q = lambda a: [round(float(np.percentile(a, p)), 2) for p in (0, 10, 50, 90, 100)]
            out = {"segs": segs, "cfg": cfg, "us": round(us, 2), "end_us": q(end[ok]), "x_ready_us": q(xr[ok]),
                   "end_by_xcd": [round(float(end[ok & (xcd == x)].mean()), 2) for x in range(8)],
                   "xready_by_xcd": [round(float(xr[ok & (xcd == x)].mean()), 2) for x in range(8)],
                   "end_by_slot": [round(float(end[ok & (slot == s)].mean()), 2) for s in range(nw)],
                   "block_mean_end_pct": q(np.array([end[ok & (blk == b)].mean() for b in range(256 * bpc) if (ok & (blk == b)).any()])),
                   "within_block_spread_pct": q(np.array([np.ptp(end[ok & (blk == b)]) for b in range(256 * bpc) if (ok & (blk == b)).any()])),
                   "shader_GHz": round(float(np.median(raw[ok, 3] / ((raw[ok, 2] - raw[ok, 0]) * 10.0))), 3)}
            print(json.dumps(out), flush=True)
    capi.set_gemv_config()
if __name__ == "__main__":
    main()
