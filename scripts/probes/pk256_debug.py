import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tinychatengine_amd import lab; lab.use_lab()  # (loop parts switched off: the diagnostics build)
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0"); L = capi.lib(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K) in ((256, 256, 512), (256, 128, 256), (512, 256, 1024), (300, 200, 768), (1024, 4096, 4096)):
    lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack()
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    outs = {}
    for mode in (61, 66, 68):
        L.tce_w4a16_set_debug_mode(mode)
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
        d = lin.desc(x, out)
        print(mode, capi.describe_dispatch(d))
        capi.check(capi.w4a16_forward(d, st)); torch.cuda.synchronize()
        outs[mode] = out.float().cpu().numpy()
    L.tce_w4a16_set_debug_mode(60)
    for mm in (66, 68):
      a, b = outs[61], outs[mm]
      print('mode', mm)
      nan = np.isnan(b)
      bad = ~np.isclose(a, b, rtol=2e-3, atol=1e-3) | nan
      print(M, N, K, "nan", int(nan.sum()), "bad", int(bad.sum()), "of", a.size)
      if bad.any():
        rows = np.where(bad.any(axis=1))[0]; cols = np.where(bad.any(axis=0))[0]
        print(" bad rows", rows[:40], "...", len(rows)); print(" bad cols", cols[:40], "...", len(cols))
        r, c = np.argwhere(bad)[0]; print(" first", r, c, a[r, c], b[r, c])
# ablations at M=2048 4096x4096
M, N, K = 2048, 4096, 4096
lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
out = torch.empty(M, N, dtype=torch.float16, device=dev)
descs = [l.desc(x, out) for l in lins]; it = [0]
def run():
    capi.check(capi.w4a16_forward(descs[it[0] % 3], st)); it[0] += 1
def timed(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1000.0 / reps)
    return min(ts)
row = {}
L.tce_w4a16_set_debug_mode(66); row["full_256"] = round(timed(run), 2)
for name, bits in (("no_rescale", 1), ("no_unpack", 2), ("no_frag_reads", 4), ("no_mfma", 8), ("no_dma", 16), ("no_barriers", 32), ("mfma_dma_barriers_only", 7), ("mfma_only", 55)):
    L.tce_w4a16_set_debug_mode(2600 + bits)
    try:
        row[name] = round(timed(run), 2)
    except Exception as e:
        row[name] = str(e)[:80]; L.tce_reset_last_error()
L.tce_w4a16_set_debug_mode(2600); L.tce_w4a16_set_debug_mode(60)
L.tce_w4a16_set_debug_mode(61); row["full_128_1q"] = round(timed(run), 2); L.tce_w4a16_set_debug_mode(60)
L.tce_w4a16_set_debug_mode(690); row["auto_128"] = round(timed(run), 2)
L.tce_w4a16_set_debug_mode(68); row["full_256x2"] = round(timed(run), 2); L.tce_w4a16_set_debug_mode(60); L.tce_w4a16_set_debug_mode(691)
print(json.dumps(row))
