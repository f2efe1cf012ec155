"""The launches of a prompt through decoder layers (DecoderBlock.prefill: 8 per layer), issued eagerly so that `rocprofv3 --kernel-trace` names and times each
kernel: Llama-2-7B-shaped layers (hidden 4096, 32 heads, ffn 11008), 512 and 2048 rows.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prefill_prof -o prefill -- python scripts/prefill_profile.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinychatengine_amd.decoder_block import DecoderBlock  # noqa: E402

dev = torch.device("cuda:0")
hidden, heads, ffn, layers, ctx = 4096, 32, 11008, 4, 2048
ang = np.random.default_rng(0).uniform(0, 2 * np.pi, (ctx, 64))
cos = torch.from_numpy(np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)).to(dev)
sin = torch.from_numpy(np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)).to(dev)
blocks = [DecoderBlock(hidden, heads, ffn, ctx, dev, cos, sin, seed=100 + i).prepare_prefill() for i in range(layers)]
for m, reps in ((512, 6), (2048, 3)):
    rows0 = torch.randn(m, hidden, device=dev).to(torch.float16)
    rows = rows0.clone()
    for _ in range(reps):
        rows.copy_(rows0)
        for b in blocks:
            b.prefill(rows, 0)
    torch.cuda.synchronize()
    print(f"rows {m}: {reps} x {layers} layers issued, finite = {bool(torch.isfinite(rows.float()).all().item())}")
