"""Round 6: LOADER WAVES on the narrow one-quartet form (debug modes 52002 / 52004 of profiles/r6/gemm_pk_loader_waves_experiment.patch -- a measured negative, not in the tree: apply the patch to run this) beside the forms without them.  Derived from gemm_pk_form16_ab.py:
Round 6: form 16 of the packed prefill GEMM (128 x 128 tiles, two quartets alternating a run's k-blocks, every tile's k range handed off between two workgroups)
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
SHAPES = [(512, 4096, 4096, 128), (512, 4096, 11008, 128), (512, 4096, 14336, 128), (384, 4096, 4096, 128), (256, 4096, 4096, 128), (512, 5120, 5120, 128), (512, 2048, 8192, 128), (128, 4096, 4096, 128), (200, 4096, 11008, 128)]
if os.environ.get("PK_SHAPES"): SHAPES = [tuple(int(v) for v in x.split("x")) + (128,) for x in os.environ["PK_SHAPES"].split(",")]
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
    for name, mode, ldr in (("auto", 60, 0), ("form16", 2676, 0), ("cut2", 642, 0), ("cut2_ldr2", 642, 2), ("cut2_ldr4", 642, 4), ("cut4", 644, 0), ("cut4_ldr2", 644, 2), ("cut4_ldr4", 644, 4), ("cut3_ldr2", 643, 2), ("auto_again", 60, 0), ("cut2_ldr2_again", 642, 2)):
        L.tce_w4a16_set_debug_mode(60); L.tce_w4a16_set_debug_mode(mode); L.tce_w4a16_set_debug_mode(52000 + ldr)
        row[name] = round(timed(run), 2)
        row[name + "_is"] = " ".join(capi.describe_dispatch(descs[0]).split()[1:4])
        out.fill_(float("nan")); capi.check(capi.w4a16_forward(descs[0], st)); torch.cuda.synchronize()
        o = out.float().clone()
        if ref is None: ref = o
        else:
            tol = 1e-3 * torch.maximum(ref.abs(), ref.pow(2).mean().sqrt() / 64)
            row[name + "_err_over_tol_vs_auto"] = round(float(((o - ref).abs() / tol).max()), 3)
        L.tce_w4a16_set_debug_mode(60); L.tce_w4a16_set_debug_mode(52000)
    fl = 2.0 * M * N * K
    row["TF_auto"] = round(fl / min(row["auto"], row["auto_again"]) / 1e6, 1)
    row["TF_cut2_ldr2"] = round(fl / min(row["cut2_ldr2"], row["cut2_ldr2_again"]) / 1e6, 1)
    print(json.dumps(row), flush=True)
    del lins, descs
    torch.cuda.empty_cache()
