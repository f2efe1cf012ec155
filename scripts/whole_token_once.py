#!/usr/bin/env python3
"""One decode token through 32 complete decoder layers (bench.py's `decode_with_attention` leg), issued eagerly a few times at one context
length, for the profiler: rocprofv3 --kernel-trace --stats then names every kernel of the token and its average duration.
   gpurun -- 'cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_token/kt -o kt -- \
              python $GRAFT_REPO_ROOT/scripts/whole_token_once.py 512'"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.decode import SHAPES, DecodeLinears
from tinychatengine_amd.decoder_block import DecoderBlock

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
shape = SHAPES["baseline-named"]
heads, hd, ctx_max = shape.hidden // 128, 128, 2048
ang = np.random.default_rng(0).uniform(0, 2 * np.pi, (ctx_max, hd // 2))
cos = torch.from_numpy(np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)).to(dev)
sin = torch.from_numpy(np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)).to(dev)
blocks = [DecoderBlock(shape.hidden, heads, shape.ffn, ctx_max, dev, cos, sin, seed=100 + i) for i in range(shape.layers)]
for b in blocks:
    b.attention.k_cache.normal_(0, 0.8)
    b.attention.v_cache.normal_(0, 0.8)
dl = DecodeLinears(shape, device=dev, group_size=128, m=1, layers=1)
hid0 = torch.randn(1, shape.hidden, device=dev).to(torch.float16)
hid = hid0.clone()
for _ in range(reps):
    hid.copy_(hid0)
    for b in blocks:
        b.step(hid, ctx - 1)
    capi.check(capi.w4a16_forward(dl.lm_head.desc(hid, dl.logits), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("finite", bool(torch.isfinite(hid.float()).all().item()))
