#!/usr/bin/env python3
"""W8A8 MFMA kernel on the OPT-125M shapes: 1 / 2 / 4 wave quartets per 64x64 tile (debug modes 71 / 72 / 74) and the automatic choice (70)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from tune import dev, time_graph, capi


def main():
    L = capi.lib()
    for (M, N, K) in [(512, 768, 768), (512, 3072, 768), (512, 768, 3072), (108, 768, 768), (108, 3072, 768), (108, 768, 3072), (1, 768, 768), (1, 3072, 768)]:
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
        b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
        bias = torch.randint(-128, 128, (N,), dtype=torch.int8, device=dev)
        out = torch.empty((M, N), dtype=torch.int8, device=dev)
        d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=a.data_ptr(), B=b.data_ptr(), bias=bias.data_ptr(), C=out.data_ptr(), alpha=0.0005,
                          beta=0.02, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8)
        row = {"M": M, "N": N, "K": K}
        for name, mode in (("ks1", 71), ("ks2", 72), ("ks4", 74), ("auto", 70)):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            for _ in range(2):
                us = time_graph(lambda i, sp: capi.check(L.tce_w8a8_matmul(C.byref(d), sp)), 64)
            row[name] = round(us, 2)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
