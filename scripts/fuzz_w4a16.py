#!/usr/bin/env python3
"""Soak test (not part of pytest): seeded random W4A16 problems through tce_w4a16_forward -- every dispatch path incl. the
LDS-DMA GEMM forms (forced tiles / quartets), the small-batch kernel with batch slices, ADD_TO_C, odd leading dimensions --
against the CPU oracle.  usage: fuzz_w4a16.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from tinychatengine_amd import capi
from oracle.oracle import Oracle
oracle = Oracle()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import w4a16_close

dev = torch.device("cuda", 0)
L = capi.lib()
SCRATCH = torch.zeros(int(L.tce_w4a16_gemm_scratch_bytes()), dtype=torch.uint8, device=dev)
FORMS = {}


def make(rng, M, N, K, G, random_zeros):
    zw = oracle.zeros_width(K, G)  # (G = 64 rounds the packed-zero words per row up to an even count)
    qw = rng.integers(0, 2**32, size=(N, K // 8), dtype=np.uint64).astype(np.uint32)
    sc = np.zeros((N, zw * 8), np.float16)
    sc[:, :K // G] = (rng.random((N, K // G), dtype=np.float32) * 0.02 + 0.002).astype(np.float16)
    if random_zeros:
        zp = rng.integers(0, 2**32, size=(N, zw), dtype=np.uint64).astype(np.uint32)
    else:
        zp = np.full((N, zw), 0x88888888, np.uint32)
    a = (rng.standard_normal((M, K), dtype=np.float32)).astype(np.float16)
    return qw, sc, zp, a


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bad = 0
    for case in range(cases):
        G = int(rng.choice([128, 128, 128, 64, 32]))
        K = G * int(rng.integers(1, 48))
        while K % 32 or K % G:
            K += G
        N = int(rng.choice([rng.integers(1, 64), rng.integers(64, 1100), rng.integers(1100, 5000)]))
        M = int(rng.choice([1, 1, 2, 3, 4, 7, 16, 17, 31, 64, 100, 128, 129, 200, 300, 520, 1030]))
        if M * N * K > 3e9:
            M = 64
        rz = bool(rng.integers(0, 2))
        qw, sc, zp, a = make(rng, M, N, K, G, rz)
        ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, G)
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        tq, ts, tz, ta = t(qw.view(np.int32)), t(sc), t(zp.view(np.int32)), t(a)
        configs = [((0, 0), 50)]
        if G == 128 and M > 16:
            mt, nt = [(204, 1), (204, 2), (202, 2), (208, 2), (204, 4), (4, 2), (4, 1)][int(rng.integers(0, 7))]
            configs.append(((mt, nt), 50 + int(rng.integers(0, 3))))
        for cfg, mode in configs:
            capi.set_gemm_config(*cfg)
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            add = bool(rng.integers(0, 3) == 0)
            ldc = N + int(rng.choice([0, 0, 8, 3]))
            c0 = (torch.randn(M, ldc, device=dev) * 0.25).to(torch.float16)
            out = c0.clone()
            flags = (capi.TCE_W4_ADD_TO_C if add else 0) | (capi.TCE_W4_FORCE_GEMM if (cfg != (0, 0)) else 0)
            d = capi.W4A16Desc(M=M, N=N, K=K, group_size=G, A=ta.data_ptr(), qweight=tq.data_ptr(), scales=ts.data_ptr(), zeros=tz.data_ptr(),
                               C=out.data_ptr(), ldc=ldc, flags=flags)
            capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            got = out[:, :N].float().cpu().numpy()
            want = ref32
            if add:  # C = half(C + half(y)): compare y = got - c0 loosely is lossy; rebuild the expectation instead
                want = (c0[:, :N].cpu().numpy().astype(np.float16) + ref32.astype(np.float16)).astype(np.float32)
            ok, worst = w4a16_close(got.astype(np.float16), want)
            tail_ok = ldc == N or torch.equal(out[:, N:], c0[:, N:])
            if not ok and add:  # the half addition adds up to one half ulp of |C + y| on top of the 1e-3 bound
                err = np.abs(got - want)
                tol = 1e-3 * np.maximum(np.abs(ref32), np.sqrt(np.mean(ref32.astype(np.float64) ** 2)) / 64) + np.abs(want) * 2.0 ** -10
                ok = bool((err <= tol).all())
            if not (ok and tail_ok) or np.isnan(got).any():
                bad += 1
                print(f"FAIL case {case}: M={M} N={N} K={K} G={G} rz={rz} cfg={cfg} mode={mode} add={add} ldc={ldc} worst={worst:.3f} tail_ok={tail_ok}", flush=True)
        capi.set_gemm_config(); capi.check(L.tce_w4a16_set_debug_mode(50))
        # round 4: the same problem on the packed copy -- M <= 4 runs the int8-contraction GEMV (groups 128 / 64 / 32 by their row limits), M >= 192 the 128-row GEMM
        need = int(L.tce_w4a16_prepack_bytes(N, K, G)) if K % 128 == 0 else 0
        if need and (M <= 4 or M >= 192):
            buf = torch.empty(need + 256, dtype=torch.uint8, device=dev)
            packed = buf[(-buf.data_ptr()) % 256:][:need]
            capi.check(L.tce_w4a16_prepack(capi.W4A16Desc(M=1, N=N, K=K, group_size=G, qweight=tq.data_ptr(), scales=ts.data_ptr(), zeros=tz.data_ptr()), packed.data_ptr(), None))
            add = bool(rng.integers(0, 3) == 0)
            ldc = N + int(rng.choice([0, 0, 8, 16]))
            c0 = (torch.randn(M, ldc, device=dev) * 0.25).to(torch.float16)
            out = c0.clone()
            flags = (capi.TCE_W4_ADD_TO_C if add else 0) | (0 if rz else capi.TCE_W4_ZERO_POINT_IS_8 * int(rng.integers(0, 2)))
            d = capi.W4A16Desc(M=M, N=N, K=K, group_size=G, A=ta.data_ptr(), qweight=tq.data_ptr(), scales=ts.data_ptr(), zeros=tz.data_ptr(), C=out.data_ptr(), ldc=ldc,
                               flags=flags, prepacked=packed.data_ptr())
            # round 5: beside the dispatcher's choice, two forced forms of the GEMM per case -- the wide ones (128 rows x 64 / 48 columns per wave; they run where the
            # zero-point-8 promise is given, groups of 128) with and without the scratch area that lets the k range be cut across workgroups
            modes = [60]
            if M >= 129:
                modes += [int(v) for v in rng.choice([61, 62, 63, 64, 66, 68, 2669, 2670, 2671, 2672, 2683, 2673, 2674, 2675], size=2, replace=False)]
            for pmode in modes:
                capi.check(L.tce_w4a16_set_debug_mode(pmode))
                out.copy_(c0)
                d.scratch = SCRATCH.data_ptr() if (pmode != 60 or rng.integers(0, 2)) and M > 128 else None
                path = capi.describe_dispatch(d)
                capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                got = out[:, :N].float().cpu().numpy()
                want = (c0[:, :N].cpu().numpy().astype(np.float16) + ref32.astype(np.float16)).astype(np.float32) if add else ref32
                ok, worst = w4a16_close(got.astype(np.float16), want)
                if not ok and add:
                    err = np.abs(got - want)
                    tol = 1e-3 * np.maximum(np.abs(ref32), np.sqrt(np.mean(ref32.astype(np.float64) ** 2)) / 64) + np.abs(want) * 2.0 ** -10
                    ok = bool((err <= tol).all())
                tail_ok = ldc == N or torch.equal(out[:, N:], c0[:, N:])
                FORMS[" ".join(path.split()[:4])] = FORMS.get(" ".join(path.split()[:4]), 0) + 1
                if not (ok and tail_ok) or np.isnan(got).any():
                    bad += 1
                    print(f"FAIL case {case} (packed copy, mode {pmode}, {path}): M={M} N={N} K={K} G={G} rz={rz} add={add} ldc={ldc} flags={flags} worst={worst:.3f} tail_ok={tail_ok}", flush=True)
            capi.check(L.tce_w4a16_set_debug_mode(60))
            assert int(SCRATCH[:4096].to(torch.int32).sum().item()) == 0, "the scratch area's counters must be back at zero"
    print("packed-copy launches by form:", dict(sorted(FORMS.items(), key=lambda kv: -kv[1])), flush=True)
    print(f"fuzz: {cases} cases, {bad} failures, {time.time() - t0:.0f} s (seed {seed})", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
