#!/usr/bin/env python3
"""Soak test (not part of pytest): seeded random W8A8 problems through the MatmulOperator mirror -> tce_w8a8_matmul, every kernel family the dispatcher picks
(wave-per-column, 64x64 tiles with 1 / 2 / 4 quartets, the deep-pipeline tile, the 128-row tiles, the generic kernel for odd K) and, every fourth case, a forced
family -- against the CPU oracle, BIT-EXACT (kernels/ref/matmul_ref_int8.cc:11-159).  usage: fuzz_w8a8.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.oracle import Oracle
from tinychatengine_amd import capi
from tinychatengine_amd.matmul import MatmulOperator, matmul_params, matrix


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    orc = Oracle()
    op = MatmulOperator()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    bad, t0 = 0, time.time()
    forced = [dict(quartets_per_tile=1), dict(quartets_per_tile=2), dict(quartets_per_tile=4), dict(deep_pipeline=1), dict(deep_pipeline=2), dict(deep_pipeline=4),
              dict(big_tiles=1), dict(big_tiles=2), dict(big_tiles=3), dict(big_tiles=4), dict(big_tiles=9, deep_pipeline=9)]
    for case in range(cases):
        M = int(rng.choice([1, 2, 5, 8, 9, 16, 33, 64, 65, 108, 130, 200, 512, int(rng.integers(1, 600))]))
        N = int(rng.choice([1, 3, 16, 64, 65, 100, 136, 768, 1000, int(rng.integers(1, 1500))]))
        K = int(rng.choice([16, 48, 64, 80, 192, 768, 1024, 1600, 2048, 3072, 4160, 8192, 16 * int(rng.integers(1, 300)), int(rng.integers(1, 2000))]))
        if M * N * K > 6e8: K = max(16, K // 8)
        A = rng.integers(-128, 128, (M, K), dtype=np.int8)
        B = rng.integers(-128, 128, (N, K), dtype=np.int8)
        alpha, beta = float(np.float32(rng.uniform(1e-4, 2e-3))), float(np.float32(rng.uniform(0.005, 0.05)))
        kind = int(rng.integers(0, 4))  # bias int8 -> int8 | no bias -> int8 | bias fp32 -> fp32 | no bias -> fp32
        tune = forced[int(rng.integers(0, len(forced)))] if case % 4 == 3 else {}
        capi.w8a8_set_tuning(**tune)
        try:
            if kind == 0:
                b8 = rng.integers(-128, 128, N, dtype=np.int8)
                exp = orc.int8_matmul_bias_i8(A, B, b8, alpha, beta, -128, 127, M, N, K)
                out = torch.zeros((M, N), dtype=torch.int8, device=dev)
                p = matmul_params(A=matrix(M, K, t(A)), B=matrix(K, N, t(B)), C=matrix(M, N, out), alpha=alpha, beta=beta)
                p.bias = matrix(1, N, t(b8))
                p.C.qparams.q_min, p.C.qparams.q_max = -128, 127
                op.mat_mul_accelerator_int8_fast_2x2_32unroll(p)
            elif kind == 1:
                qmin = 0 if rng.integers(0, 2) else -128  # (the ReLU clamp of fc1)
                exp = orc.int8_matmul_nobias_i8(A, B, alpha, qmin, 127, M, N, K)
                out = torch.zeros((M, N), dtype=torch.int8, device=dev)
                p = matmul_params(A=matrix(M, K, t(A)), B=matrix(K, N, t(B)), C=matrix(M, N, out), alpha=alpha, beta=beta)
                p.C.qparams.q_min, p.C.qparams.q_max = qmin, 127
                op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias(p)
            elif kind == 2:
                bf = rng.standard_normal(N).astype(np.float32)
                exp = orc.int8_matmul_bias_f32(A, B, bf, alpha, M, N, K)
                out = torch.zeros((M, N), dtype=torch.float32, device=dev)
                p = matmul_params(A=matrix(M, K, t(A)), B=matrix(K, N, t(B)), C=matrix(M, N, out), alpha=alpha, beta=beta)
                p.bias = matrix(1, N, t(bf))
                op.mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32(p)
            else:
                exp = orc.int8_matmul_nobias_f32(A, B, alpha, M, N, K)
                out = torch.zeros((M, N), dtype=torch.float32, device=dev)
                p = matmul_params(A=matrix(M, K, t(A)), B=matrix(K, N, t(B)), C=matrix(M, N, out), alpha=alpha, beta=beta)
                op.mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32(p)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            ok = np.array_equal(got.view(np.uint32) if got.dtype == np.float32 else got, np.asarray(exp).view(np.uint32) if got.dtype == np.float32 else np.asarray(exp))
        except Exception as e:  # noqa: BLE001
            ok = False
            print(f"case {case}: {type(e).__name__}: {e}", flush=True)
            capi.lib().tce_reset_last_error()
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: M={M} N={N} K={K} kind={kind} tuning={tune}", flush=True)
    capi.w8a8_set_tuning()
    print(f"fuzz_w8a8: {cases} cases, {bad} failures, {time.time() - t0:.0f} s (seed {seed})", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
