#!/usr/bin/env python3
"""Do back-to-back GEMV launches overlap when nothing orders them?  Two capture streams, independent launches.

Upper bound for what a flag-based (in-kernel) dependency instead of stream order could buy: launch ramp and tail of
consecutive linears hidden behind each other."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinychatengine_amd import capi, quantize
dev = torch.device("cuda:0"); L = capi.lib()
def mk(segs, K, reps):
    G = 128; zw = quantize.calculate_zeros_width(K, G)
    x = torch.randn(1, K, device=dev).to(torch.float16); sets = []
    for rep in range(reps):
        ds, keep = [], []
        for n in segs:
            qw = torch.randint(-2**31, 2**31 - 1, (n, K // 8), dtype=torch.int32, device=dev)
            sc = (torch.rand((n, zw * 8), device=dev) * 0.01 + 0.001).to(torch.float16)
            zp = torch.full((n, zw), -2004318072, dtype=torch.int32, device=dev)
            out = torch.empty(1, n, dtype=torch.float16, device=dev); keep += [qw, sc, zp, out, x]
            ds.append(capi.W4A16Desc(M=1, N=n, K=K, group_size=G, A=x.data_ptr(), qweight=qw.data_ptr(), scales=sc.data_ptr(), zeros=zp.data_ptr(), C=out.data_ptr()))
        sets.append(((capi.W4A16Desc * len(ds))(*ds), keep))
    return sets
def timeit(sets, nseg, nstreams, launches=64):
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); side = [torch.cuda.Stream() for _ in range(nstreams - 1)]
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            ev = torch.cuda.Event(); ev.record(s)
            for t in side: t.wait_event(ev)
            streams = [s] + side
            for i in range(launches):
                st = streams[i % nstreams]
                capi.check(L.tce_w4a16_forward_group(sets[i % len(sets)][0], nseg, C.c_void_p(st.cuda_stream)))
            for t in side:
                e2 = torch.cuda.Event(); e2.record(t); s.wait_event(e2)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [g.replay() for _ in range(3)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * launches)
def main():
  for (segs, K) in [([11008, 11008], 4096), ([12288], 4096), ([4096], 4096), ([4096], 11008)]:
      sets = mk(segs, K, reps=24)
      nbytes = sum(segs) * (K // 2 + K // 128 * 2 + quantize.calculate_zeros_width(K, 128) * 4)
      for cfg in [None, (2, 8, 0, 3), (2, 8, 0, 2), (1, 8, 0, 3)]:
          if cfg is None: capi.set_gemv_config()
          else: capi.set_gemv_config(*cfg)
          for ns in (1, 2, 3):
              try:
                  us = timeit(sets, len(segs), ns)
                  print(json.dumps({"segs": segs, "K": K, "cfg": cfg, "streams": ns, "us": round(us, 2), "TBps": round(nbytes / us / 1e6, 2)}), flush=True)
              except Exception as e:
                  print(json.dumps({"segs": segs, "cfg": cfg, "streams": ns, "error": str(e)[:200]}), flush=True)
  capi.set_gemv_config()
if __name__ == '__main__':
  main()
