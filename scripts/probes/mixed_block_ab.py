#!/usr/bin/env python3
"""Round 6: a rank's linears of one block of the column-sharded model (north_star's one-gather-per-block form: replicated inputs, nothing depends on anything inside
the block) as FOUR launches (q/k/v grouped, o, gate/up grouped, down: what decode.py issued through round 5) against ONE (tce_w4a16_forward_independent).
32 blocks' worth of distinct weights per graph replay (every byte from HBM), no gathers (one GPU): us per block of shard compute at P = 1, 2, 4, 8."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinychatengine_amd import capi, decode

dev = torch.device("cuda:0")


def graph_of(fn, s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    return g


def rate(g, n=60):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    shape_name = os.environ.get("SHAPE", "llama3-8b")
    shape = decode.SHAPES[shape_name]
    s = torch.cuda.Stream()
    rows = []
    for P in (8, 4, 2, 1):
        dl = decode.DecodeLinears(shape, device=dev, prepack=True, rank=0, world=P)
        L = dl.n_layers
        four = lambda: [dl._hip_launch(g) for li in range(L) for g in dl.block_launches(li)]
        one = lambda: [capi.w4a16_forward_independent([d for g in dl.block_launches(li) for d in g], s.cuda_stream) for li in range(L)]
        n_one = capi.w4a16_forward_independent([d for g in dl.block_launches(0) for d in g], s.cuda_stream)
        g4, g1 = graph_of(four, s), graph_of(one, s)
        t4 = min(rate(g4) for _ in range(3)) * 1e3 / L
        t1 = min(rate(g1) for _ in range(3)) * 1e3 / L
        blk_bytes = sum(capi.algorithmic_bytes(1, l.out_features, l.in_features, 128) for l in [*dl.blocks[0]["qkv"], dl.blocks[0]["o"], dl.blocks[0]["gate"], dl.blocks[0]["up"], dl.blocks[0]["down"]])
        # same bits?
        outs = [*dl.out_qkv, dl.out_o, dl.out_gate, dl.out_up, dl.out_down]
        g4.replay(); torch.cuda.synchronize(); want = [o.clone() for o in outs]
        for o in outs:
            o.fill_(float("nan"))
        g1.replay(); torch.cuda.synchronize()
        same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(want, outs))
        sweep = {}
        if os.environ.get("WAVES_SWEEP", "1") == "1":  # the mixed launch's workgroup width (tuning mode 51000 + w; below what the longest K needs: the rule stays)
            need = max(-(-(l.in_features // 128) // 8) for l in [*dl.blocks[0]["qkv"], dl.blocks[0]["o"], dl.blocks[0]["gate"], dl.blocks[0]["up"], dl.blocks[0]["down"]])
            for w in range(need, 17):
                capi.check(capi.lib().tce_w4a16_set_debug_mode(51000 + w))
                gw = graph_of(one, s)
                sweep[w] = round(min(rate(gw) for _ in range(3)) * 1e3 / L, 2)
                del gw
            capi.check(capi.lib().tce_w4a16_set_debug_mode(51000))
        row = {"shape": shape_name, "P": P, "one_launch_us_by_waves_per_workgroup": sweep, "rule": capi.describe_independent([d for g in dl.block_launches(0) for d in g]), "block_shard_MB": round(blk_bytes / 1e6, 2), "four_launches_us_per_block": round(t4, 2), "one_launch_us_per_block": round(t1, 2),
               "launches_reported": n_one, "same_bits": same, "one_launch_frac_of_8TBs": round(blk_bytes / (t1 * 1e-6) / 8e12, 3)}
        print(json.dumps(row), flush=True)
        rows.append(row)
        del dl, g4, g1
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
