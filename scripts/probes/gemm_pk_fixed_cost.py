"""Where a k-range-cut prefill launch's time goes: 512 x 4096 x K for K = 512 .. 8192 under forced forms (61: whole tiles, one quartet -- no exchange; 642: cut in two,
hand-off; 2676: form 16; 62: two quartets, whole tiles) -- the intercept of us over k-blocks is what a launch pays around its loop, the slope is its k-block."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0"); L = capi.lib(); st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=40, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1000.0 / reps)
    return min(ts)
g = torch.Generator(device=dev).manual_seed(1)
M, N = int(os.environ.get("PK_M", 512)), int(os.environ.get("PK_N", 4096))
for K in (512, 1024, 2048, 4096, 8192):
    nset = max(3, int(300e6 // (N * K // 2)))
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(nset)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    descs = [l.desc(x, out) for l in lins]; it = [0]
    def run():
        capi.check(capi.w4a16_forward(descs[it[0] % nset], st)); it[0] += 1
    row = {"M": M, "N": N, "K": K, "k_blocks": K // 128}
    for name, mode in (("whole_1q", 61), ("whole_2q", 62), ("cut2_1q", 642), ("cut2_2q_form16", 2676), ("auto", 60)):
        L.tce_w4a16_set_debug_mode(60); L.tce_w4a16_set_debug_mode(mode)
        row[name] = round(timed(run), 2)
        row[name + "_is"] = " ".join(capi.describe_dispatch(descs[0]).split()[1:4])
        L.tce_w4a16_set_debug_mode(60)
    print(json.dumps(row), flush=True)
    del lins, descs
    torch.cuda.empty_cache()
