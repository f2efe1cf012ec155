#!/usr/bin/env python3
"""Where the time of tce_layernorm_q_w8a8_group goes at OPT-1.3B / 6.7B widths (m = 1): the launch as it runs, without the sums (83: wrong results, timing only), without the output rows (84), and the
workgroup-per-8-rows form of the OPT-125M sizes (81).  hipGraph of 32 launches rotating over weight sets of > 400 MB; us per launch.
    python scripts/lnq_wide_phases.py > gpurun_out/lnq_wide_phases.jsonl"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tune import dev, time_graph, capi
L = capi.lib()
g = torch.Generator(device=dev).manual_seed(3)
ri = lambda *s: torch.randint(-127, 128, s, device=dev, generator=g, dtype=torch.int32).to(torch.int8)
for name, K, ns in (("OPT-6.7B q,k,v", 4096, (4096, 4096, 4096)), ("OPT-6.7B fc1", 4096, (16384,)), ("OPT-1.3B q,k,v", 2048, (2048, 2048, 2048)), ("OPT-1.3B fc1", 2048, (8192,))):
    nsets = max(2, int(4.5e8 // (sum(ns) * K)))
    x = torch.randn(1, K, device=dev); lw = torch.randn(K, device=dev); lb = torch.randn(K, device=dev)
    sets = []
    for s in range(nsets):
        Ws = [ri(n, K) for n in ns]
        bs = [ri(n) for n in ns]
        outs = [torch.empty(1, n, dtype=torch.int8, device=dev) for n in ns]
        descs = [capi.W8A8Desc(M=1, N=n, K=K, batch=1, A=0, B=W.data_ptr(), bias=b.data_ptr(), C=o.data_ptr(), alpha=0.0005, beta=0.02, q_min=-128, q_max=127,
                               bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8) for W, b, o, n in zip(Ws, bs, outs, ns)]
        sets.append(((capi.W8A8Desc * len(descs))(*descs), Ws, bs, outs))
    row = {"launch": name, "k": K, "rows": sum(ns), "weight_MB": round(sum(ns) * K / 1e6, 1), "weight_sets": nsets}
    for label, mode in (("as_it_runs", 80), ("no_sums", 83), ("no_output_rows", 84), ("workgroup_per_8_rows", 81)):
        capi.check(L.tce_w4a16_set_debug_mode(mode))
        fn = lambda i, sp: capi.check(L.tce_layernorm_q_w8a8_group(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1, K, sets[i % nsets][0], len(ns), None, sp))
        ts = sorted(time_graph(fn, 32) for _ in range(3))
        row[label + "_us"] = round(ts[0], 2)
    # workgroup 0's own clock (debug mode 85; 100 MHz): start | x in LDS | first sum (its chains' end, all chains' end, walk's end) | second sum (same) | int8 row | rows done
    stamps = torch.zeros(32, dtype=torch.int64, device=dev)
    capi.check(L.tce_w4a16_set_debug_buffer(C.c_void_p(stamps.data_ptr())))
    capi.check(L.tce_w4a16_set_debug_mode(85))
    per = []
    for rep in range(6):
        capi.check(L.tce_layernorm_q_w8a8_group(x.data_ptr(), lw.data_ptr(), lb.data_ptr(), 1, K, sets[rep % nsets][0], len(ns), None, None))
        torch.cuda.synchronize()
        t = stamps.cpu().numpy()
        t0 = int(t[0])
        per.append({"x_in_lds": (int(t[1]) - t0) / 100, "sum1_chain_wave0": (int(t[8]) - t0) / 100, "sum1_chains_all": (int(t[9]) - t0) / 100, "sum1_walk": (int(t[10]) - t0) / 100, "sum1_done": (int(t[2]) - t0) / 100,
                    "sum2_chain_wave0": (int(t[12]) - t0) / 100, "sum2_chains_all": (int(t[13]) - t0) / 100, "sum2_walk": (int(t[14]) - t0) / 100, "sum2_done": (int(t[3]) - t0) / 100,
                    "int8_row": (int(t[4]) - t0) / 100, "rows_done": (int(t[5]) - t0) / 100})
    row["workgroup0_timeline_us"] = per[-1]
    row["workgroup0_timeline_us_previous"] = per[-2]
    capi.check(L.tce_w4a16_set_debug_buffer(None))
    capi.check(L.tce_w4a16_set_debug_mode(80))
    row["weights_GBps_as_it_runs"] = round(sum(ns) * K / row["as_it_runs_us"] / 1e3, 1)
    print(json.dumps(row), flush=True)
