#!/usr/bin/env python3
"""RMSNorm prologue + one linear of N rows (K = 4096) on the packed copy: one against two tiles per wave (tce_w4a16_set_gemv_i8(0, 1 / 2)) over N -- where the rule's
threshold (1024 tiles) sits, and what lm_head-sized launches want."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
dev = torch.device("cuda:0"); L = capi.lib()
g = torch.Generator(device=dev).manual_seed(5)
K = 4096
x = torch.randn(1, K, device=dev).to(torch.float16)
gamma = (1 + 0.1 * torch.randn(K, device=dev)).float()
for N in (8192, 12288, 14336, 16384, 20480, 24576, 28672, 32000, 65536, 128256):
    nsets = max(2, min(24, int(4e8 // (N * K // 2))))
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(nsets)]
    outs = torch.empty(1, N, dtype=torch.float16, device=dev)
    descs = [l.desc(x, outs, gamma=gamma, eps=1e-6) for l in lins]
    row = {"N": N, "tiles": N // 16}
    for tiles in (1, 2, 0):
        capi.set_gemv_i8(0, tiles)
        gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            sp = C.c_void_p(s.cuda_stream)
            with torch.cuda.graph(gr, stream=s):
                for i in range(4 * nsets): capi.check(L.tce_w4a16_forward(C.byref(descs[i % nsets]), sp))
        ts = []
        for _ in range(4):
            gr.replay(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3): gr.replay()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / (3 * 4 * nsets))
        row[{1: "one_tile_us", 2: "two_tiles_us", 0: "rule_us"}[tiles]] = round(min(ts), 2)
    capi.set_gemv_i8()
    print(json.dumps(row), flush=True)
    del lins, descs
    torch.cuda.empty_cache()
