#!/usr/bin/env python3
"""Round 5 probe: the pre-packed prefill GEMM with the two waves of a SIMD at different priorities for the whole kernel (tce_w4a16_set_debug_mode(697): the first quartet of a
two-quartet workgroup / the workgroups of a CU's first dispatch round at s_setprio 2) against equal priorities (696), same process, three weight sets in rotation."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0"); L = capi.lib(); st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=20, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1000.0 / reps)
    return min(ts)
g = torch.Generator(device=dev).manual_seed(1)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2048, 4096, 4096), (2048, 11008, 4096), (4096, 4096, 4096), (512, 11008, 4096), (2048, 4096, 11008), (512, 4096, 4096)]
for (M, N, K) in shapes:
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    descs = [l.desc(x, out) for l in lins]
    for d in descs:  # (NOZ8=1: the narrow forms, what linears with real zero points run)
        d.flags = (d.flags & ~capi.TCE_W4_ZERO_POINT_IS_8) if os.environ.get("NOZ8") else (d.flags | capi.TCE_W4_ZERO_POINT_IS_8)
    it = [0]
    def run():
        capi.check(capi.w4a16_forward(descs[it[0] % 3], st)); it[0] += 1
    row = {"M": M, "N": N, "K": K, "is": capi.describe_dispatch(descs[0])}
    for rnd in range(3):
        for mode, name in ((696, "equal"), (697, "prio2"), (6974, "prio2_by_slot_parity")):
            L.tce_w4a16_set_debug_mode(mode)
            row.setdefault(name, []).append(round(timed(run), 2))
    L.tce_w4a16_set_debug_mode(696)
    print(json.dumps(row), flush=True)
    del lins, descs
    torch.cuda.empty_cache()
