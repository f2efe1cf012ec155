#!/usr/bin/env python3
"""Dependent GEMV launches (one stream, in order) with a side branch that touches the NEXT launch's weights while the current one runs:
does streaming out of the memory-side cache shorten the launch?  us per launch for several prefetch sizes (workgroups; 0 = no branch)."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinychatengine_amd import capi, quantize
dev = torch.device("cuda:0"); L = capi.lib()


def mk(segs, K, reps):
    G = 128; zw = quantize.calculate_zeros_width(K, G)
    x = torch.randn(1, K, device=dev).to(torch.float16); sets = []
    for rep in range(reps):
        ds, keep, ranges = [], [], []
        for n in segs:
            qw = torch.randint(-2**31, 2**31 - 1, (n, K // 8), dtype=torch.int32, device=dev)
            sc = (torch.rand((n, zw * 8), device=dev) * 0.01 + 0.001).to(torch.float16)
            zp = torch.full((n, zw), -2004318072, dtype=torch.int32, device=dev)
            out = torch.empty(1, n, dtype=torch.float16, device=dev); keep += [qw, sc, zp, out, x]
            ds.append(capi.W4A16Desc(M=1, N=n, K=K, group_size=G, A=x.data_ptr(), qweight=qw.data_ptr(), scales=sc.data_ptr(), zeros=zp.data_ptr(), C=out.data_ptr(), flags=4))
            ranges.append((qw.data_ptr(), qw.numel() * 4))
        sets.append(((capi.W4A16Desc * len(ds))(*ds), keep, ranges))
    return sets


def timeit(sets, nseg, wgs, frac=1.0, launches=64):
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); side = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(launches):
                if wgs:  # the branch starts where launch i starts and touches launch i+1's weights
                    ev = torch.cuda.Event(); ev.record(s); side.wait_event(ev)
                    for (ptr, nbytes) in sets[(i + 1) % len(sets)][2]:
                        nb = int(nbytes * frac) & ~15
                        capi.check(L.tce_prefetch(C.c_void_p(ptr), nb, wgs, C.c_void_p(side.cuda_stream)))
                capi.check(L.tce_w4a16_forward_group(sets[i % len(sets)][0], nseg, C.c_void_p(s.cuda_stream)))
            if wgs:
                e2 = torch.cuda.Event(); e2.record(side); s.wait_event(e2)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [g.replay() for _ in range(3)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * launches)


def main():
    for segs, K in (([11008, 11008], 4096), ([12288], 4096), ([4096], 4096), ([4096], 11008)):
        reps = max(4, int(1.2e9 / (sum(segs) * K / 2)))
        sets = mk(segs, K, reps)
        row = {"segs": segs, "K": K, "sets": reps}
        for wgs, frac in ((0, 1.0), (64, 1.0), (128, 1.0), (256, 1.0), (512, 1.0), (1024, 1.0), (256, 0.5), (256, 0.25)):
            for _ in range(2):
                us = timeit(sets, len(segs), wgs, frac)
            row[f"wg{wgs}" + (f"_f{frac}" if frac != 1.0 else "")] = round(us, 2)
        print(json.dumps(row), flush=True)
        del sets; torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
