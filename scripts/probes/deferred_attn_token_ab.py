#!/usr/bin/env python3
"""Round 5: the whole decode token with the attention combine in o_proj's prologue (DecoderBlock.defer_combine = True) against the in-kernel combine, same process,
one graph per variant and context, alternating; Llama-3-8B shapes (32 / 8 heads) and the 7B-shaped set (32 / 32), position on the device.  Also the pair alone
(attention step + o_proj with residual), 32 layers' worth of distinct weights and caches rotating."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tinychatengine_amd import capi
from tinychatengine_amd.decode import SHAPES, DecodeLinears
from tinychatengine_amd.decoder_block import DecoderBlock
dev = torch.device("cuda:0"); L = capi.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "llama3-8b"
shape = SHAPES[which]
heads, hd, ctx_max = 32, 128, 4096
kv_heads = shape.qkv[1] // 128 if len(shape.qkv) == 3 else heads
ang = np.random.default_rng(0).uniform(0, 2 * np.pi, (ctx_max, hd // 2))
cos = torch.from_numpy(np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)).to(dev)
sin = torch.from_numpy(np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)).to(dev)
blocks = [DecoderBlock(shape.hidden, heads, shape.ffn, ctx_max, dev, cos, sin, seed=100 + i, kv_heads=kv_heads) for i in range(shape.layers)]
for b in blocks:
    b.attention.k_cache.normal_(0, 0.8); b.attention.v_cache.normal_(0, 0.8)
dl = DecodeLinears(shape, device=dev, group_size=128, m=1, layers=1, prepack=True)
hid0 = torch.randn(1, shape.hidden, device=dev).to(torch.float16); hid = hid0.clone()
pos_t = torch.zeros(1, dtype=torch.int32, device=dev)
def timed(g, n=50):
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): g.replay()
    b_.record(); torch.cuda.synchronize()
    return a.elapsed_time(b_) / n
for ctx in (320, 512, 640, 1024, 2048, 4096):
    pos = ctx - 1; pos_t.fill_(pos)
    graphs = {}
    for defer in (False, True):
        for b in blocks: b.defer_combine = defer
        def token():
            hid.copy_(hid0)
            for b in blocks: b.step(hid, pos, pos_device=pos_t)
            capi.check(capi.w4a16_forward(dl.lm_head.desc(hid, dl.logits), torch.cuda.current_stream().cuda_stream))
        def pair():
            for b in blocks:
                st = torch.cuda.current_stream().cuda_stream
                if defer:
                    import ctypes as C
                    b.attention.step(b.qkv_out.view(-1), pos, out=b.attn_out.view(heads, 128), pos_device=pos_t, defer=True)
                    capi.check(L.tce_w4a16_forward_deferred_attention(C.byref(b.o.desc(b.attn_out, hid, flags=capi.TCE_W4_ADD_TO_C)), C.byref(b.attention.deferred), C.c_void_p(pos_t.data_ptr()), int(pos), C.c_void_p(st)))
                else:
                    b.attention.step(b.qkv_out.view(-1), pos, out=b.attn_out.view(heads, 128), pos_device=pos_t)
                    capi.check(capi.w4a16_forward(b.o.desc(b.attn_out, hid, flags=capi.TCE_W4_ADD_TO_C), st))
        for name, fn in (("token", token), ("pair", pair)):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g): fn()
            graphs[(name, defer)] = g
    row = {"model": which, "context": ctx, "slots": int(blocks[0].attention.deferred.slots)}
    for name in ("token", "pair"):
        t = {False: [], True: []}
        for rep in range(3):
            for defer in (False, True): t[defer].append(timed(graphs[(name, defer)]))
        a_, b_ = min(t[False]), min(t[True])
        if name == "token":
            row["token_ms"] = [round(a_, 4), round(b_, 4)]; row["tokens_per_s"] = [round(1e3 / a_, 1), round(1e3 / b_, 1)]; row["gain"] = round(a_ / b_ - 1, 4)
        else:
            row["pair_us_per_layer"] = [round(a_ * 1e3 / shape.layers, 2), round(b_ * 1e3 / shape.layers, 2)]
    print(json.dumps(row), flush=True)
