#!/usr/bin/env python3
"""W4A16 at 5 <= M < 192 (between the decode kernel and the 128-row GEMM): the dispatcher's choice against the small-batch kernel forced off / on
(tce_w4a16_set_debug_mode(1000 + largest M it takes)), with and without a packed copy.  us per launch, weights rotating over three copies."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0"); L = capi.lib(); st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1000.0 / reps)
    return min(ts)
g = torch.Generator(device=dev).manual_seed(1)
for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (6144, 4096), (14336, 4096)):
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128) for _ in range(3)]
    for M in (5, 8, 16, 32, 48, 64, 96, 128, 160, 191):
        x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        descs = [l.desc(x, out) for l in lins]; it = [0]
        def run():
            capi.check(capi.w4a16_forward(descs[it[0] % 3], st)); it[0] += 1
        row = {"M": M, "N": N, "K": K}
        for name, mode in (("auto", 1128), ("small_batch_off", 1004), ("small_batch_on", 1256)):
            L.tce_w4a16_set_debug_mode(mode)
            try:
                row[name] = round(timed(run), 2)
                row[name + "_is"] = " ".join(capi.describe_dispatch(descs[0]).split()[:3])
            except Exception:  # noqa: BLE001
                row[name] = None; L.tce_reset_last_error()
        L.tce_w4a16_set_debug_mode(1128)
        print(json.dumps(row), flush=True)
    del lins
    torch.cuda.empty_cache()
