"""The launches of one SmoothQuant OPT-125M decoder layer, issued eagerly so that `rocprofv3 --kernel-trace --stats` names and times each kernel
(DESIGN 3.3): 12 layers x N decode tokens at 512 keys, then 12 layers x a few 512-row prefills.

    rocprofv3 --kernel-trace --stats -d gpurun_out/opt_prof -o opt -- python scripts/opt_layer_profile.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinychatengine_amd.opt_layer import Int8OPTDecoderLayer  # noqa: E402

dev = torch.device("cuda:0")
E, H, F, NL = 768, 12, 3072, 12
for m, pos, reps in ((1, 511, 40), (512, 0, 5)):
    tgz = pos + m
    layers = [Int8OPTDecoderLayer(E, H, F, 512, m, dev, seed=7 + i) for i in range(NL)]
    hid0 = torch.randn(m, E, device=dev)
    hid = hid0.clone()
    mask = torch.zeros((m, tgz), device=dev)
    if m > 1:
        mask.masked_fill_(torch.triu(torch.ones(m, tgz, dtype=torch.bool, device=dev), diagonal=pos + 1), torch.finfo(torch.float32).min)
    for _ in range(reps):
        hid.copy_(hid0)
        for l in layers:
            l.step(hid, pos, mask)
    torch.cuda.synchronize()
    print(f"rows {m}: {reps} x {NL} layers issued, finite = {bool(torch.isfinite(hid).all().item())}")
