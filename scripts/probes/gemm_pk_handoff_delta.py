import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0"); L = capi.lib(); st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=30, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1000.0 / reps)
    return min(ts)
g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K) in ((512, 4096, 4096), (512, 4096, 11008), (384, 4096, 4096)):
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    descs = [l.desc(x, out) for l in lins]; it = [0]
    def run():
        capi.check(capi.w4a16_forward(descs[it[0] % 3], st)); it[0] += 1
    row = {"M": M, "N": N, "K": K}
    L.tce_w4a16_set_debug_mode(642)
    for d in (0, 1, 2, 3, 4, 6, 8, 2):
        L.tce_w4a16_set_debug_mode(6950 + d)
        row.setdefault(f"delta{d}", []).append(round(timed(run), 2))
    L.tce_w4a16_set_debug_mode(6952); L.tce_w4a16_set_debug_mode(60)
    print(json.dumps(row), flush=True)
