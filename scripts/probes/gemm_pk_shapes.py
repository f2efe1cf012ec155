"""The packed prefill GEMM at the dispatcher's choice on a list of shapes (weights in rotation, > 256 MiB per shape): us per launch and TFLOP/s.  For A/Bs of two BUILDS
in one session (TCE_LIB_PATH=... per process, alternating).  PK_SHAPES=MxNxK,... overrides the list."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0"); L = capi.lib(); st = torch.cuda.current_stream().cuda_stream
def timed(fn, reps=30, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1000.0 / reps)
    return min(ts)
g = torch.Generator(device=dev).manual_seed(1)
SHAPES = [(512, 4096, 4096), (512, 11008, 4096), (512, 4096, 11008), (512, 14336, 4096), (512, 4096, 14336), (2048, 4096, 4096), (2048, 11008, 4096), (2048, 4096, 11008), (128, 4096, 4096), (256, 4096, 4096)]
if os.environ.get("PK_SHAPES"): SHAPES = [tuple(int(v) for v in x.split("x")) for x in os.environ["PK_SHAPES"].split(",")]
rows = {}
for (M, N, K) in SHAPES:
    nset = max(3, int(400e6 // (N * K // 2)))
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(nset)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    pairs = os.environ.get("PK_PAIRS", "0") == "1"  # the gate / up rows interleaved, SiLU * mul in the epilogue: C is [M][N / 2]
    out = torch.empty(M, N // 2 if pairs else N, dtype=torch.float16, device=dev)
    fl = (capi.TCE_W4_ZERO_POINT_IS_8 if os.environ.get("PK_Z8", "1") == "1" else 0) | (capi.TCE_W4_SILU_MUL_PAIRS if pairs else 0)
    descs = [l.desc(x, out, flags=fl) for l in lins]; it = [0]
    def run():
        capi.check(capi.w4a16_forward(descs[it[0] % nset], st)); it[0] += 1
    us = timed(run)
    rows[f"{M}x{N}x{K}"] = [round(us, 2), round(2.0 * M * N * K / us / 1e6, 1)]
    del lins, descs
    torch.cuda.empty_cache()
print(json.dumps({"lib": os.path.basename(os.environ.get("TCE_LIB_PATH", "in-tree")), "us_TF": rows}), flush=True)
