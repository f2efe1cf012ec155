"""Where does a k-block of the pre-packed GEMM spend its time?  The loop with parts switched off (tce_w4a16_set_debug_mode(600 + bits)).
   gpurun -- 'python scripts/gemm_pk_ablation.py > gpurun_out/gemm_pk_ablation.jsonl'"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tinychatengine_amd import lab; lab.use_lab()  # (loop parts switched off: the diagnostics build)
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4

dev = torch.device("cuda:0")
L = capi.lib()
st = torch.cuda.current_stream().cuda_stream
NAMES = {0: "full", 1: "-rescale", 2: "-unpack", 4: "-fragment reads", 8: "-MFMA", 16: "-activation DMA", 32: "-barriers", 6: "-unpack -reads",
         7: "-rescale -unpack -reads", 23: "MFMA + barriers only", 55: "MFMA only", 47: "activation DMA only (+ waits)", 48: "-DMA -barriers"}
g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K) in ((2048, 4096, 4096), (512, 4096, 4096), (512, 11008, 4096)):
    lins = [Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack() for _ in range(3)]
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    descs = [l.desc(x, out) for l in lins]
    for abl, name in NAMES.items():
        L.tce_w4a16_set_debug_mode(600 + abl if abl else 61)
        for i in range(6):
            capi.check(capi.w4a16_forward(descs[i % 3], st))
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20):
                capi.check(capi.w4a16_forward(descs[i % 3], st))
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 50.0)
        print(json.dumps({"M": M, "N": N, "K": K, "ablation": abl, "what": name, "us": round(best, 2), "us_per_kblock": round(best / (K / 128), 3)}), flush=True)
    L.tce_w4a16_set_debug_mode(60)
