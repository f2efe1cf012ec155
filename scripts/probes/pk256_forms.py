import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0"); L = capi.lib(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
def timed(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1000.0 / reps)
    return min(ts)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 4096, 4096), (2048, 4096, 4096), (2048, 11008, 4096), (2048, 4096, 11008), (4096, 11008, 4096), (4096, 4096, 11008), (1024, 4096, 4096), (1024, 11008, 4096)]
for (M, N, K) in shapes:
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    descs = [l.desc(x, out) for l in lins]; it = [0]
    def run():
        capi.check(capi.w4a16_forward(descs[it[0] % 3], st)); it[0] += 1
    row = {"M": M, "N": N, "K": K}
    ref = None
    for name, modes in (("rows128_auto", (690, 60)), ("x2_256x128", (68,)), ("w_256x256", (2669,)), ("rows128_auto_again", (690, 60))):
        for m_ in modes: L.tce_w4a16_set_debug_mode(m_)
        out.fill_(float("nan")); run(); torch.cuda.synchronize()
        o = out.float().cpu().numpy()
        if ref is None: ref = o
        else: row[name + "_bad"] = int((~np.isclose(ref, o, rtol=2e-3, atol=1e-3)).sum())
        row[name] = round(timed(run), 2)
        L.tce_w4a16_set_debug_mode(60); L.tce_w4a16_set_debug_mode(691)
    fl = 2.0 * M * N * K
    for k in ("rows128_auto", "x2_256x128", "w_256x256"): row["TF_" + k] = round(fl / row[k] / 1e6, 1)
    print(json.dumps(row), flush=True)
    del lins, descs
