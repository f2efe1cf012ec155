#!/usr/bin/env python3
"""Prefill GEMM on the packed copy, round 5: the WIDE form (128 rows x 64 columns per wave: tce_w4a16_set_debug_mode 2670 one quartet per 128 x 256 tile, 2671 two quartets
alternating its k-blocks, 2672 / 2682..2684 every tile's k range cut across workgroups) against the narrow forms (the dispatcher's choice without / with the 256-row tiles,
form 1, form 8), per (M, N, K): us per launch and TFLOP/s, weights rotating over three copies, the forms alternating inside one process (boxes differ by 4-8 % on MFMA work).
usage: gemm_pkw_sweep.py [--abl] [MxNxK ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinychatengine_amd import lab; lab.use_lab()  # (loop parts switched off: the diagnostics build)
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
args = [a for a in sys.argv[1:] if not a.startswith("--")]
ABL = "--abl" in sys.argv
shapes = [tuple(int(v) for v in a.split("x")) for a in args]
if not shapes:
    shapes = [(M, N, K) for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (14336, 4096)) for M in (512, 1024, 2048, 4096)]
cache = {}
for (M, N, K) in shapes:
    if (N, K) not in cache:
        cache.clear(); torch.cuda.empty_cache()
        cache[(N, K)] = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
    lins = cache[(N, K)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    descs = [l.desc(x, out) for l in lins]; it = [0]
    def run():
        capi.check(capi.w4a16_forward(descs[it[0] % 3], st)); it[0] += 1
    row = {"M": M, "N": N, "K": K}
    fl = 2.0 * M * N * K
    forms = [("narrow_auto", 60), ("form1", 61), ("form8", 68), ("wide1", 2670), ("wide2", 2671), ("wide_cut2", 2682), ("wide_cut3", 2683), ("wide_cut4", 2684), ("wide3_1", 2673), ("wide3_2", 2674), ("wide512", 2675), ("narrow_auto_again", 60)]
    if ABL:  # parts of the wide loop switched off (results meaningless): 1 no rescale, 2 no unpack, 4 no fragment reads, 8 no MFMAs, 16 no activation DMAs, 32 no barriers
        forms = [("wide1", 2670)] + [(f"wide1_abl{b}", 26000 + b) for b in (1, 2, 4, 8, 16, 32, 7, 55)] + [("wide1_again", 2670), ("wide1_dma_in_phase1", 26128), ("wide1_dma_in_front", 26256), ("wide1_third", 2670), ("wide1_dma_in_phase1_again", 26128)]
    for name, mode in forms:
        L.tce_w4a16_set_debug_mode(692 if mode == 60 else 693)
        L.tce_w4a16_set_debug_mode(mode)
        try:
            row[name] = round(timed(run), 2)
            if not name.startswith("narrow") and "abl" not in name:
                what = capi.describe_dispatch(descs[0])
                row[name + "_is"] = " ".join(what.split()[1:5])
        except Exception as e:  # noqa: BLE001
            row[name] = None; L.tce_reset_last_error()
        L.tce_w4a16_set_debug_mode(26000); L.tce_w4a16_set_debug_mode(60)
    if not ABL:
        L.tce_w4a16_set_debug_mode(693)
        row["auto_with_wide"] = round(timed(run), 2); row["auto_with_wide_is"] = " ".join(capi.describe_dispatch(descs[0]).split()[:5])
        L.tce_w4a16_set_debug_mode(692)
        bn = min(row["narrow_auto"], row["narrow_auto_again"])
        bw = min(v for k, v in row.items() if k.startswith("wide") and isinstance(v, float))
        row["TF_narrow"] = round(fl / bn / 1e6, 1); row["TF_wide_best"] = round(fl / bw / 1e6, 1); row["TF_auto_with_wide"] = round(fl / row["auto_with_wide"] / 1e6, 1); row["wide_over_narrow"] = round(bn / bw, 3)
    print(json.dumps(row), flush=True)
