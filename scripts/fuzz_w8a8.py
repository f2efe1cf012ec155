#!/usr/bin/env python3
"""Soak test (not part of pytest): seeded random W8A8 problems through tce_w8a8_matmul -- every kernel behind it (MFMA tiles with 1 / 2 wave quartets, the 128-row tiles in their four forced forms, the
wave-per-column kernel for small M, the wave-per-output kernel for per-row operands with long rows, the generic kernel), every epilogue kind, batches, leading
dimensions, `accumulate` -- against the CPU oracle, BIT FOR BIT.  usage: fuzz_w8a8.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from tinychatengine_amd import capi
from oracle.oracle import Oracle

oracle = Oracle()
dev = torch.device("cuda", 0)
L = capi.lib()
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0, bad = time.time(), 0
    kinds = {}
    for case in range(cases):
        M = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 33, 64, 65, 108, 130, 300]))
        N = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 1200)]))
        K = int(rng.choice([16 * rng.integers(1, 12), 16 * rng.integers(12, 80), 16 * rng.integers(80, 300), rng.integers(1, 500), 64 * rng.integers(4, 40)]))
        if M * N * K > 4e8:
            M = 5
        per_row = bool(rng.integers(0, 5) == 0) and M * N * K < 3e7
        batch = 1 if per_row else int(rng.choice([1, 1, 1, 2, 5]))
        pad = lambda n: n + (int(rng.choice([0, 0, 16, 48])) if K % 16 == 0 else 0)
        lda, ldb = pad(K), pad(K)
        out_fp32 = bool(rng.integers(0, 2))
        bias_kind = int(rng.choice([capi.TCE_BIAS_NONE, capi.TCE_BIAS_FP32 if out_fp32 else capi.TCE_BIAS_INT8])) if not per_row else capi.TCE_BIAS_NONE
        accumulate = out_fp32 and bool(rng.integers(0, 3) == 0)
        ldc = N + int(rng.choice([0, 0, 5, 16]))
        qmin = int(rng.choice([-128, 0]))
        alpha, beta = float(rng.choice([0.0005035400390625, 0.003, 1.0e-4])), 0.02130126953125
        A = rng.integers(-128, 128, (batch, M, lda), dtype=np.int8)
        B = rng.integers(-128, 128, (M if per_row else batch, N, ldb), dtype=np.int8)
        b8, bf = rng.integers(-128, 128, N, dtype=np.int8), rng.standard_normal(N).astype(np.float32)
        C0 = rng.standard_normal((batch, M, ldc)).astype(np.float32)
        tA, tB, tb8, tbf = t(A), t(B), t(b8), t(bf)
        out = t(C0.copy()) if out_fp32 else torch.full((batch, M, ldc), 55, dtype=torch.int8, device=dev)
        d = capi.W8A8Desc(M=M, N=N, K=K, batch=batch, A=tA.data_ptr(), B=tB.data_ptr(), bias=(tbf if out_fp32 else tb8).data_ptr() if bias_kind != capi.TCE_BIAS_NONE else None,
                          C=out.data_ptr(), strideA=M * lda, strideB=N * ldb, strideC=M * ldc, alpha=alpha, beta=beta, q_min=qmin, q_max=127, bias_kind=bias_kind,
                          out_kind=capi.TCE_OUT_FP32 if out_fp32 else capi.TCE_OUT_INT8, b_per_row=1 if per_row else 0, accumulate=1 if accumulate else 0, lda=lda if lda != K else 0,
                          ldb=ldb if ldb != K else 0, ldc=ldc if ldc != N else 0)
        mode = int(rng.choice([70, 70, 71, 72, 73]))
        capi.check(L.tce_w4a16_set_debug_mode(mode))
        # the 128-row tiles (K % 64 == 0, K >= 256, a shared B): forced in one of their four forms on a part of the eligible cases
        big = int(rng.choice([75, 75, 76, 77, 176, 177, 78]))
        capi.check(L.tce_w4a16_set_debug_mode(big))
        capi.check(capi.w8a8_matmul(d, None))
        torch.cuda.synchronize()
        L.tce_w4a16_set_debug_mode(70)
        L.tce_w4a16_set_debug_mode(75)
        got = out.cpu().numpy()
        ok = True
        for h in range(batch):
            Ah = np.ascontiguousarray(A[h, :, :K])
            Bh = np.ascontiguousarray(B[:, :, :K]) if per_row else np.ascontiguousarray(B[h, :, :K])
            if out_fp32:
                if bias_kind == capi.TCE_BIAS_FP32:
                    want = oracle.int8_matmul_bias_f32(Ah, Bh, bf, alpha, M, N, K)
                else:
                    want = oracle.int8_matmul_nobias_f32(Ah, Bh, alpha, M, N, K, batch=per_row)
                if accumulate:
                    want = C0[h, :, :N] + want
                ok &= np.array_equal(got[h, :, :N].view(np.uint32), want.view(np.uint32)) and np.array_equal(got[h, :, N:], C0[h, :, N:])
            else:
                if bias_kind == capi.TCE_BIAS_INT8:
                    want = oracle.int8_matmul_bias_i8(Ah, Bh, b8, alpha, beta, qmin, 127, M, N, K)
                else:
                    want = oracle.int8_matmul_nobias_i8(Ah, Bh, alpha, qmin, 127, M, N, K, batch=per_row)
                ok &= np.array_equal(got[h, :, :N], want) and bool((got[h, :, N:] == 55).all())
        key = ("per-row" if per_row else "shared") + (" fp32" if out_fp32 else " int8")
        kinds[key] = kinds.get(key, 0) + 1
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: M={M} N={N} K={K} batch={batch} per_row={per_row} lda={lda} ldb={ldb} ldc={ldc} fp32={out_fp32} bias={bias_kind} acc={accumulate} mode={mode}")
    print(f"fuzz w8a8: {cases} cases {kinds}, {bad} failures, {time.time() - t0:.0f} s (seed {seed})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
