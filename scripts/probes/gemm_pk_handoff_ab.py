import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
for (M, N, K) in ((512, 4096, 4096), (512, 4096, 11008), (384, 4096, 4096), (512, 5120, 5120), (256, 4096, 4096)):
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    descs = [l.desc(x, out) for l in lins]; it = [0]
    def run():
        capi.check(capi.w4a16_forward(descs[it[0] % 3], st)); it[0] += 1
    row = {"M": M, "N": N, "K": K}
    for name, sw, mode in (("auto_last_arriver", 694, 60), ("auto_handoff", 695, 60), ("cut2_last_arriver", 694, 642), ("cut2_handoff", 695, 642), ("cut4", 695, 644), ("auto_last_arriver_again", 694, 60), ("auto_handoff_again", 695, 60)):
        L.tce_w4a16_set_debug_mode(sw); L.tce_w4a16_set_debug_mode(mode)
        row[name] = round(timed(run), 2)
        if name.startswith("auto"): row[name + "_is"] = " ".join(capi.describe_dispatch(descs[0]).split()[1:4])
        L.tce_w4a16_set_debug_mode(60)
    L.tce_w4a16_set_debug_mode(695)
    row["TF_handoff"] = round(2.0 * M * N * K / min(row["auto_handoff"], row["auto_handoff_again"]) / 1e6, 1)
    row["TF_last_arriver"] = round(2.0 * M * N * K / min(row["auto_last_arriver"], row["auto_last_arriver_again"]) / 1e6, 1)
    print(json.dumps(row), flush=True)
