"""Round 6: form 16 of the packed prefill GEMM (128 x 128 tiles, two quartets alternating a run's k-blocks, every tile's k range handed off between two workgroups)
against the dispatcher's choice and the forms it combines (2: two quartets, whole k range; cut in 2 / 4 with one quartet), same process, weights in rotation.
us per launch, TFLOP/s, and the worst error against the dispatcher's output in units of the W4A16 tolerance."""
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
SHAPES = [(512, 4096, 4096, 128), (512, 4096, 11008, 128), (512, 4096, 14336, 128), (384, 4096, 4096, 128), (256, 4096, 4096, 128), (512, 5120, 5120, 128), (640, 4096, 4096, 128), (512, 4096, 4096, 64), (512, 2048, 8192, 128), (1024, 4096, 4096, 128)]
for (M, N, K, G) in SHAPES:
    nset = max(3, int(400e6 // (N * K // 2)))
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), G).prepack() for _ in range(nset)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    descs = [l.desc(x, out) for l in lins]; it = [0]
    def run():
        capi.check(capi.w4a16_forward(descs[it[0] % nset], st)); it[0] += 1
    row = {"M": M, "N": N, "K": K, "G": G, "weight_sets": nset}
    ref = None
    for name, mode in (("auto", 60), ("form16", 2676), ("form2", 62), ("cut2", 642), ("cut4", 644), ("auto_again", 60), ("form16_again", 2676)):
        L.tce_w4a16_set_debug_mode(60); L.tce_w4a16_set_debug_mode(mode)
        row[name] = round(timed(run), 2)
        row[name + "_is"] = " ".join(capi.describe_dispatch(descs[0]).split()[1:4])
        out.fill_(float("nan")); capi.check(capi.w4a16_forward(descs[0], st)); torch.cuda.synchronize()
        o = out.float().clone()
        if ref is None: ref = o
        else:
            tol = 1e-3 * torch.maximum(ref.abs(), ref.pow(2).mean().sqrt() / 64)
            row[name + "_err_over_tol_vs_auto"] = round(float(((o - ref).abs() / tol).max()), 3)
        L.tce_w4a16_set_debug_mode(60)
    fl = 2.0 * M * N * K
    row["TF_auto"] = round(fl / min(row["auto"], row["auto_again"]) / 1e6, 1)
    row["TF_form16"] = round(fl / min(row["form16"], row["form16_again"]) / 1e6, 1)
    print(json.dumps(row), flush=True)
    del lins, descs
    torch.cuda.empty_cache()
