#!/usr/bin/env python3
"""tce_w8a8_matmul (int8 out, int8 bias): for every (M, N, K) the dispatcher's choice against every forced kernel family (tce_w8a8_set_tuning) -- 64x64 tiles with
1 / 2 / 4 quartets, the deep-pipeline tile with 1 / 2 / 4 quartets, the 128-row tiles (128 / 64 columns, one / two quartets) -- us per launch, graphs of 32 launches over
weight sets rotating through HBM.  usage: w8a8_form_sweep.py [MxNxK ...]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tune import dev, time_graph, capi
L = capi.lib()
g = torch.Generator(device=dev).manual_seed(3)
ri = lambda *s: torch.randint(-127, 128, s, device=dev, generator=g, dtype=torch.int32).to(torch.int8)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
if not shapes:
    shapes = [(M, N, K) for (N, K) in ((768, 768), (3072, 768), (768, 3072), (2048, 2048), (8192, 2048), (2048, 8192), (4096, 4096), (16384, 4096), (4096, 16384)) for M in (16, 108, 512, 2048)]
forms = [("auto", {}), ("q1", dict(quartets_per_tile=1, big_tiles=9, deep_pipeline=9)), ("q2", dict(quartets_per_tile=2, big_tiles=9, deep_pipeline=9)), ("q4", dict(quartets_per_tile=4, big_tiles=9, deep_pipeline=9)),
         ("deep1", dict(deep_pipeline=1, big_tiles=9)), ("deep2", dict(deep_pipeline=2, big_tiles=9)), ("deep4", dict(deep_pipeline=4, big_tiles=9)),
         ("big128", dict(big_tiles=1)), ("big64", dict(big_tiles=2)), ("big128_2q", dict(big_tiles=3)), ("big64_2q", dict(big_tiles=4))]
for (M, N, K) in shapes:
    nsets = max(2, min(48, int(3e8 // (N * K))))
    A = ri(M, K)
    sets = []
    for _ in range(nsets):
        W, b, o = ri(N, K), ri(N), torch.empty(M, N, dtype=torch.int8, device=dev)
        d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=A.data_ptr(), B=W.data_ptr(), bias=b.data_ptr(), C=o.data_ptr(), alpha=0.0005, beta=0.02, q_min=-128, q_max=127,
                          bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8)
        sets.append((d, W, b, o))
    row = {"M": M, "N": N, "K": K}
    for name, tune in forms:
        capi.w8a8_set_tuning(**tune)
        try:
            row[name] = round(min(time_graph(lambda i, sp: capi.check(L.tce_w8a8_matmul(C.byref(sets[i % nsets][0]), sp)), 32) for _ in range(2)), 2)
        except Exception:  # noqa: BLE001
            row[name] = None; L.tce_reset_last_error()
    capi.w8a8_set_tuning()
    best = min((v, k) for k, v in row.items() if isinstance(v, float) and k != "auto")
    row["best"] = best[1]; row["auto_over_best"] = round(row["auto"] / best[0], 3)
    print(json.dumps(row), flush=True)
    del sets
    torch.cuda.empty_cache()
