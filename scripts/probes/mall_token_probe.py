#!/usr/bin/env python3
"""Round 6: is there anything to win by having a decode token's weights in the 256 MiB memory-side cache when their launch starts?

Part A (bound): the 128 block launches of the headline token (no lm_head: 525 MB never fits), as one hipGraph, with the 32 blocks' launches reading
R distinct blocks' weights in rotation: R = 32 is the real token (every byte from HBM), R = 1 / 2 keeps 110 / 220 MB in rotation (memory-side cache hits).
Same launches, same order, same activations' data flow -- only where the weight bytes come from differs.

Part B (cost of a neighbour): the real token (R = 32) while a second stream runs tce_prefetch (a plain streaming read, `wgs` workgroups) over a DISJOINT
buffer: how much does the token slow down, and how many bytes does the neighbour move meanwhile -- i.e. how much HBM time do the launch boundaries leave."""
import ctypes as C, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinychatengine_amd import capi, decode

dev = torch.device("cuda:0")
L = capi.lib()


def rate(fn, n, stream):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        a.record()
        for _ in range(n):
            fn()
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    shape = decode.SHAPES[os.environ.get("SHAPE", "llama3-8b")]
    dl = decode.DecodeLinears(shape, device=dev, prepack=True, dataflow=True)
    s = torch.cuda.Stream()
    st = s.cuda_stream
    per_block = [dl.block_launches(li) for li in range(dl.n_layers)]
    blk_bytes = sum(capi.algorithmic_bytes(1, l.out_features, l.in_features, 128) for l in [*dl.blocks[0]["qkv"], dl.blocks[0]["o"], dl.blocks[0]["gate"], dl.blocks[0]["up"], dl.blocks[0]["down"]])
    out = {"shape": shape.name, "block_bytes": blk_bytes}
    plans = {}
    for R in (32, 1, 2, 4):
        # block li runs with block (li % R)'s WEIGHTS but its own place in the data flow (block 0 reads x, the others the previous down_proj output)
        launches = []
        for li in range(dl.n_layers):
            src = per_block[li % R]
            mine = per_block[li]
            grp = []
            for gs, gm in zip(src, mine):
                row = []
                for d_src, d_me in zip(gs, gm):
                    d = capi.W4A16Desc.from_buffer_copy(d_src)
                    d.A, d.C = d_me.A, d_me.C
                    row.append(d)
                grp.append(row)
            launches += grp
        plans[R] = capi.Plan(launches)
    for R, p in plans.items():
        ms = min(rate(lambda: p.launch(st), 100, s) for _ in range(3))
        out[f"A_token_body_ms_R{R}"] = round(ms, 4)
        print(json.dumps({f"R{R}": ms}), flush=True)
    # Part B
    side = torch.cuda.Stream()
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)  # 1 GiB, disjoint from every weight
    p32 = plans[32]
    N_TOK = 100
    for wgs in (0, 8, 16, 32, 64, 128, 256):
        torch.cuda.synchronize()
        reps = 0
        if wgs:
            # enough passes to outlast the tokens: the side stream is stopped by its own length; count how many passes finished when the tokens end
            ev_done = []
            for i in range(400):
                capi.check(L.tce_prefetch(C.c_void_p(big.data_ptr()), big.numel(), wgs, C.c_void_p(side.cuda_stream)))
                e = torch.cuda.Event(); e.record(side); ev_done.append(e)
            time.sleep(0.002)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            a.record()
            for _ in range(N_TOK):
                p32.launch(st)
            b.record()
        b.synchronize()
        if wgs:
            reps = sum(1 for e in ev_done if e.query())
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / N_TOK
        row = {"B_side_wgs": wgs, "token_body_ms": round(ms, 4), "side_passes_done_at_end(incl. ~2ms head start)": reps,
               "side_GB_per_token_upper": round(reps * big.numel() / 1e9 / N_TOK, 3)}
        print(json.dumps(row), flush=True)
        out.setdefault("B", []).append(row)
    # the side kernel alone: GB/s per workgroup count
    for wgs in (8, 16, 32, 64, 128, 256):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            a.record()
            for i in range(5):
                capi.check(L.tce_prefetch(C.c_void_p(big.data_ptr()), big.numel(), wgs, C.c_void_p(side.cuda_stream)))
            b.record()
        torch.cuda.synchronize()
        out.setdefault("side_alone_GBs", {})[wgs] = round(5 * big.numel() / (a.elapsed_time(b) * 1e6), 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
