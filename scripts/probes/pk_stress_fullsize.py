#!/usr/bin/env python3
"""Round 5 probe: the dispatcher's own choice at FULL-SIZE prefill shapes -- where two workgroups share a CU (two waves per SIMD, the regime the retired groups-of-32 kernel
failed in) -- launched beside a loaded second stream, every output compared bit for bit with the first (parity of these forms: tests/test_gpu_w4a16_pk.py).
    python scripts/probes/pk_stress_fullsize.py [seconds per shape] [MxNxK ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tinychatengine_amd import capi
from tinychatengine_amd.linear import Linear_half_int4
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]] or [(1024, 11008, 4096), (2048, 11008, 4096), (2048, 4096, 4096), (4096, 4096, 4096), (512, 11008, 4096), (512, 4096, 4096), (1024, 4096, 4096)]
dev = torch.device("cuda:0"); L = capi.lib(); st = torch.cuda.current_stream().cuda_stream
side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big_b = torch.empty_like(big_a)
g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K) in shapes:
    for z8 in (True, False):
        lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack()
        x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        d = lin.desc(x, out)
        d.flags = (d.flags | capi.TCE_W4_ZERO_POINT_IS_8) if z8 else (d.flags & ~capi.TCE_W4_ZERO_POINT_IS_8)
        capi.check(capi.w4a16_forward(d, st)); torch.cuda.synchronize()
        first = out.clone(); rounds = bad = 0; t_end = time.time() + secs
        while time.time() < t_end:
            with torch.cuda.stream(side):
                for _ in range(3): big_b.copy_(big_a, non_blocking=True)
            for _ in range(4):
                out.fill_(float("nan"))
                capi.check(capi.w4a16_forward(d, st))
                torch.cuda.synchronize()
                rounds += 1
                if not torch.equal(out, first): bad += 1
        print(json.dumps({"M": M, "N": N, "K": K, "is": capi.describe_dispatch(d), "launches": rounds, "differing_from_the_first": bad}), flush=True)
        del lin
        torch.cuda.empty_cache()
