#!/usr/bin/env python3
"""Is a parity failure of a forced GEMM form reproducible?  Runs test_pk_gemm_matches_oracle's inputs for one shape through a list of modes, N times, and prints worst |err| / tol per (repeat, mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import conftest
from test_gpu_w4a16_pk import _quant, _lin
from conftest import w4a16_close
from tinychatengine_amd import capi
from oracle.oracle import Oracle  # (the checker: this probe is test infrastructure)
M, N, K, G = [int(v) for v in (sys.argv[1:5] or (192, 200, 512, 32))]
modes = [int(v) for v in sys.argv[5:]] or [61, 62, 63, 64, 60]
dev = torch.device("cuda:0"); L = capi.lib(); oracle = Oracle()
import ctypes as C
TK = C.CDLL(os.path.join(os.path.dirname(capi.__file__), "lib", "libtce_testkit.so"))
TK.tce_testkit_poison.argtypes = [C.c_void_p, C.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
POISON = os.environ.get("POISON", "0") == "1"
rng = np.random.default_rng(M + N + K)
qw, sc, zp = _quant(oracle, N, K, G, seed=M * 3 + N + K, random_zeros=False, zero_scale_groups=0)
a = rng.standard_normal((M, K)).astype(np.float16)
ref32 = oracle.w4a16_gemv_q4_6_mt(a, qw, sc, zp, M, N, K, G)
lin = _lin(dev, qw, sc, zp, G).prepack()
x = torch.from_numpy(a).to(dev)
for rep in range(int(os.environ.get('REPS', '3'))):
    row = {}
    for mode in modes:
        capi.check(L.tce_w4a16_set_debug_mode(mode))
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
        d = lin.desc(x, out)
        if POISON:
            assert TK.tce_testkit_poison(sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        ok, worst = w4a16_close(got, ref32)
        bad = np.argwhere(~np.isclose(got.astype(np.float32), ref32, rtol=2e-3, atol=1e-3))
        row[mode] = (round(float(worst), 3), capi.describe_dispatch(d).split()[1:3], len(bad), bad[:2].tolist(), [(float(got[r, c]), float(ref32[r, c])) for r, c in bad[:2]])
        if len(bad) and os.environ.get("EXPLAIN", "0") == "1":  # what is a wrong output made of?  per-group contributions of the element, in float64
            from tinychatengine_amd import quantize as Q
            wd = Q.dequantize_q4_6(torch.from_numpy(qw.view(np.int32)), torch.from_numpy(sc.view(np.float16)), torch.from_numpy(zp.view(np.int32)), G).double().numpy()
            for (r, c) in bad[:3]:
                contrib = (a[r].astype(np.float64) * wd[c].reshape(-1)).reshape(K // G, G).sum(axis=1)
                kb = contrib.reshape(K // 128, 128 // G).sum(axis=1)
                print("   bad", (int(r), int(c)), "got", float(got[r, c]), "ref", float(ref32[r, c]), "diff", float(got[r, c]) - float(ref32[r, c]),
                      "\n      per k-block", np.round(kb, 4).tolist(), "\n      per group", np.round(contrib, 4).tolist(),
                      "\n      scales", np.round(sc.view(np.float16)[c, :K // G].astype(np.float64), 5).tolist(), flush=True)
    L.tce_w4a16_set_debug_mode(60)
    print(rep, row, flush=True)
