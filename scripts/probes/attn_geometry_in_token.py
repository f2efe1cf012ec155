#!/usr/bin/env python3
"""Round 6: the fast attention step's geometry (waves per workgroup, workgroups the key range is cut for, query heads per workgroup: tce_attention_set_tuning) swept INSIDE the
whole token (graphs captured per setting, alternating) -- in situ the step reads q / k / v that the previous launch just wrote.  Derived from:
Round 5, decode experiment (b): what would the whole token gain if the attention step's cross-workgroup combine moved into o_proj's prologue?  UPPER BOUND: the token
with the combine simply switched off (tce_w4a16_set_debug_mode(2931): partial states stored plainly, the launch ends; the outputs are then garbage) against the token as
it is, same process, graphs captured per mode, alternating.  Llama-3-8B shapes, 32 layers, position on the device."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tinychatengine_amd import capi
from tinychatengine_amd.decode import SHAPES, DecodeLinears
from tinychatengine_amd.decoder_block import DecoderBlock
dev = torch.device("cuda:0"); L = capi.lib()
shape = SHAPES["llama3-8b"]
heads, hd, ctx_max = 32, 128, 2048
ang = np.random.default_rng(0).uniform(0, 2 * np.pi, (ctx_max, hd // 2))
cos = torch.from_numpy(np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)).to(dev)
sin = torch.from_numpy(np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)).to(dev)
blocks = [DecoderBlock(shape.hidden, heads, shape.ffn, ctx_max, dev, cos, sin, seed=100 + i, kv_heads=8) for i in range(shape.layers)]
for b in blocks:
    b.attention.k_cache.normal_(0, 0.8); b.attention.v_cache.normal_(0, 0.8)
dl = DecodeLinears(shape, device=dev, group_size=128, m=1, layers=1, prepack=True)
hid0 = torch.randn(1, shape.hidden, device=dev).to(torch.float16); hid = hid0.clone()
pos_t = torch.zeros(1, dtype=torch.int32, device=dev)
SETTINGS = [(0, 0, 0), (0, 256, 0), (0, 64, 0), (0, 32, 0), (8, 0, 0), (16, 0, 0), (0, 0, 2), (0, 0, 4), (0, 256, 2), (8, 256, 0), (0, 512, 0)]
for ctx in (512, 2048):
    pos = ctx - 1; pos_t.fill_(pos)
    def token():
        hid.copy_(hid0)
        for b in blocks: b.step(hid, pos, pos_device=pos_t)
        capi.check(capi.w4a16_forward(dl.lm_head.desc(hid, dl.logits), torch.cuda.current_stream().cuda_stream))
    graphs = {}
    for s in SETTINGS:
        try:
            capi.check(L.tce_attention_set_tuning(*s))
            token(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g): token()
            graphs[s] = (g, capi.describe_attention_step(32, ctx, 8))
        except Exception as e:  # a setting the step refuses
            graphs[s] = None
    capi.check(L.tce_attention_set_tuning(0, 0, 0))
    times = {s: [] for s in SETTINGS if graphs[s]}
    for rep in range(3):
        for s in times:
            g = graphs[s][0]
            for _ in range(5): g.replay()
            torch.cuda.synchronize()
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50): g.replay()
            b_.record(); torch.cuda.synchronize()
            times[s].append(a.elapsed_time(b_) / 50)
    for s in times:
        d = graphs[s][1]
        print(json.dumps({"context": ctx, "waves_workgroups_heads": s, "chunks": d.get("chunks"), "workgroups": d.get("workgroups"), "ms": round(min(times[s]), 4), "tokens_per_s": round(1e3 / min(times[s]), 1)}), flush=True)
