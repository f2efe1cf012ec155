#!/usr/bin/env python3
"""GPU tuning sweep: every compiled GEMV geometry x the decode shapes, the GEMM tiles at M=512, the W8A8 shapes.

    python scripts/tune.py [--quick] > gpurun_out/tune.jsonl

Method: per (shape, variant) a graph of back-to-back launches rotating over a ring of >= 1.2 GB of
distinct weight sets (defeats the 256 MB Infinity Cache), HIP events around 3 replays, after a warm-up replay.
GB/s = algorithmic bytes (tce_w4a16_algorithmic_bytes) / average launch time (inter-kernel gaps included).
Random packed weights (perf does not depend on the codes), finite scales, zero point 8.
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinychatengine_amd import capi, quantize  # noqa: E402

dev = torch.device("cuda:0")


Z8_FLAG = capi.TCE_W4_ZERO_POINT_IS_8 if os.environ.get("TCE_TUNE_Z8", "1") != "0" else 0


def ring(N, K, G, min_bytes=1.2e9, max_sets=256):
    per = N * K // 2
    n = int(max(4, min(max_sets, -(-min_bytes // per))))
    zw = quantize.calculate_zeros_width(K, G)
    sets = []
    for _ in range(n):
        qw = torch.randint(-2**31, 2**31 - 1, (N, K // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand((N, zw * 8), device=dev) * 0.01 + 0.001).to(torch.float16)
        zp = torch.full((N, zw), -2004318072, dtype=torch.int32, device=dev)
        sets.append((qw, sc, zp))
    return sets


def time_graph(fn_launch, launches, reps=3):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sp = C.c_void_p(s.cuda_stream)
        with torch.cuda.graph(g, stream=s):
            for i in range(launches):
                fn_launch(i, sp)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * launches)  # us per launch


def sweep_gemv(shapes, variants, M=1, G=128, launches=128):
    L = capi.lib()
    for name, segs, K in shapes:
        rings = [ring(n, K, G, min_bytes=1.2e9 / len(segs)) for n in segs]
        nset = min(len(r) for r in rings)
        x = torch.randn(M, K, device=dev).to(torch.float16)
        outs = [torch.empty(M, n, dtype=torch.float16, device=dev) for n in segs]
        arrs = []
        for i in range(nset):
            ds = [capi.W4A16Desc(M=M, N=n, K=K, group_size=G, A=x.data_ptr(), qweight=rings[j][i][0].data_ptr(),
                                 scales=rings[j][i][1].data_ptr(), zeros=rings[j][i][2].data_ptr(), C=outs[j].data_ptr(),
                                 flags=Z8_FLAG)  # the flag the adapter / bench set after tce_w4a16_check_zero_point_8 (TCE_TUNE_Z8=0: without)
                  for j, n in enumerate(segs)]
            arrs.append((capi.W4A16Desc * len(ds))(*ds))
        nbytes = sum(capi.algorithmic_bytes(M, n, K, G) for n in segs)
        best = None
        # "auto" first AND last: the first timing after allocating a ring runs up to 10 % slow; persistent geometries (waves_k = 0) too
        for v in [None] + variants + [(2, 16, 0, 2), (4, 16, 0, 2), (1, 16, 0, 3)] + [None]:
            try:
                capi.set_gemv_config(*(v or (0, 0, 0, 0)))
                rc = L.tce_w4a16_forward_group(arrs[0], len(segs), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                if rc != 0:
                    continue
                us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward_group(arrs[i % nset], len(segs), sp)), launches)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"kind": "gemv", "shape": name, "variant": v, "error": str(e)}), flush=True)
                continue
            rec = {"kind": "gemv", "shape": name, "N": segs, "K": K, "M": M, "variant": v or "auto", "us": round(us, 3),
                   "GBs": round(nbytes / us / 1e3, 1), "frac_8TBs": round(nbytes / us / 1e3 / 8000, 4)}
            print(json.dumps(rec), flush=True)
            if v is not None and (best is None or us < best[1]):
                best = (v, us)
        capi.set_gemv_config()
        if best:
            print(json.dumps({"kind": "gemv_best", "shape": name, "variant": best[0], "us": round(best[1], 3),
                              "GBs": round(nbytes / best[1] / 1e3, 1)}), flush=True)
        del rings, arrs


def experiments():
    """Where does the time go?  The same launch with pieces of the kernel removed (tce_w4a16_set_debug_mode)."""
    L = capi.lib()
    shapes = [("gate+up grouped 2x11008x4096", [11008, 11008], 4096), ("lm_head 128256x4096", [128256], 4096),
              ("o_proj 4096x4096", [4096], 4096), ("down 4096x11008", [4096], 11008), ("qkv 12288x4096", [12288], 4096),
              ("lm_head 32000x4096", [32000], 4096), ("L3 gate+up 2x14336x4096", [14336, 14336], 4096)]
    names = {0: "normal", 1: "stream-only", 3: "x-first"}
    for mode in (0, 3, 1):
        capi.check(L.tce_w4a16_set_debug_mode(mode))
        try:
            for (name, segs, K) in shapes:
                rings = [ring(n, K, 128, min_bytes=1.2e9 / len(segs)) for n in segs]
                nset = min(len(r) for r in rings)
                x = torch.randn(1, K, device=dev).to(torch.float16)
                outs = [torch.empty(1, n, dtype=torch.float16, device=dev) for n in segs]
                arrs = []
                for i in range(nset):
                    ds = [capi.W4A16Desc(M=1, N=n, K=K, group_size=128, A=x.data_ptr(), qweight=rings[j][i][0].data_ptr(),
                                         scales=rings[j][i][1].data_ptr(), zeros=rings[j][i][2].data_ptr(), C=outs[j].data_ptr(),
                                 flags=Z8_FLAG)  # the flag the adapter / bench set after tce_w4a16_check_zero_point_8 (TCE_TUNE_Z8=0: without)
                          for j, n in enumerate(segs)]
                    arrs.append((capi.W4A16Desc * len(ds))(*ds))
                nbytes = sum(capi.algorithmic_bytes(1, n, K, 128) for n in segs)
                for v in [(4, 4, 1, 1), (2, 4, 1, 2), (4, 8, 1, 1), (2, 16, 0, 2), (2, 15, 0, 2), (2, 16, 0, 3), (2, 15, 0, 3), (1, 16, 0, 3), (1, 16, 0, 2), (2, 12, 0, 3), (2, 8, 0, 3), (1, 12, 0, 3)]:
                    if mode != 0 and v[2] == 0:
                        continue
                    capi.set_gemv_config(*v)
                    try:
                        us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward_group(arrs[i % nset], len(segs), sp)), 128)
                    except Exception as e:  # noqa: BLE001
                        print(json.dumps({"kind": "exp", "mode": names[mode], "shape": name, "variant": v, "error": str(e)}), flush=True)
                        continue
                    print(json.dumps({"kind": "exp", "mode": names[mode], "shape": name, "variant": v, "us": round(us, 3),
                                      "GBs": round(nbytes / us / 1e3, 1)}), flush=True)
                capi.set_gemv_config()
                us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward_group(arrs[i % nset], len(segs), sp)), 128)
                print(json.dumps({"kind": "exp", "mode": names[mode], "shape": name, "variant": (0, 0, 0, 0), "us": round(us, 3),
                                  "GBs": round(nbytes / us / 1e3, 1)}), flush=True)
                del rings, arrs
        finally:
            capi.set_gemv_config()
            L.tce_w4a16_set_debug_mode(0)
    # a plain device-to-device read ceiling for reference: torch sum over 1.2 GB of int32 (reads only)
    big = torch.randint(0, 100, (300_000_000,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for fn, label in ((lambda: big.sum(), "torch int32 sum (read 1.2 GB)"), (lambda: big.clone(), "torch clone (read+write 1.2 GB each)")):
        fn(); torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(json.dumps({"kind": "ceiling", "what": label, "ms": round(ms, 4), "GBs_read": round(1.2e9 / ms / 1e6, 1)}), flush=True)


def sweep_gemm(shapes, M=512, G=128):
    L = capi.lib()
    for (N, K) in shapes:
        sets = ring(N, K, G, min_bytes=3e8)
        x = torch.randn(M, K, device=dev).to(torch.float16)
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        ds = [capi.W4A16Desc(M=M, N=N, K=K, group_size=G, A=x.data_ptr(), qweight=s[0].data_ptr(), scales=s[1].data_ptr(),
                             zeros=s[2].data_ptr(), C=out.data_ptr()) for s in sets]
        for v in [None] + capi.gemm_variants():
            capi.set_gemm_config(*(v or (0, 0)))
            try:
                us = time_graph(lambda i, sp: capi.check(L.tce_w4a16_forward(C.byref(ds[i % len(ds)]), sp)), 16)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"kind": "gemm", "N": N, "K": K, "variant": v, "error": str(e)}), flush=True)
                continue
            tf = 2.0 * M * N * K / us / 1e6
            print(json.dumps({"kind": "gemm", "M": M, "N": N, "K": K, "variant": v or "auto", "us": round(us, 2), "TFLOPs": round(tf, 1),
                              "frac_2.5PF": round(tf / 2500, 4)}), flush=True)
        capi.set_gemm_config()


def sweep_w8a8():
    L = capi.lib()
    for (M, N, K) in [(512, 768, 768), (512, 3072, 768), (512, 768, 3072), (108, 3072, 768), (1, 768, 768), (1, 3072, 768)]:
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
        b = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
        bias = torch.randint(-128, 128, (N,), dtype=torch.int8, device=dev)
        out = torch.empty((M, N), dtype=torch.int8, device=dev)
        d = capi.W8A8Desc(M=M, N=N, K=K, batch=1, A=a.data_ptr(), B=b.data_ptr(), bias=bias.data_ptr(), C=out.data_ptr(), alpha=0.0005,
                          beta=0.02, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_INT8)
        us = time_graph(lambda i, sp: capi.check(L.tce_w8a8_matmul(C.byref(d), sp)), 64)
        tops = 2.0 * M * N * K / us / 1e6
        print(json.dumps({"kind": "w8a8", "M": M, "N": N, "K": K, "us": round(us, 2), "TOPs": round(tops, 2)}), flush=True)
    for (M, N) in [(512, 768), (108, 768), (1, 768)]:  # LayerNormQ in front of the int8 linears (bit-exact, one thread per row)
        x = torch.randn(M, N, device=dev); w = torch.randn(N, device=dev); b = torch.randn(N, device=dev)
        out = torch.empty((M, N), dtype=torch.int8, device=dev)
        us = time_graph(lambda i, sp: capi.check(L.tce_layernorm_q(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, sp)), 64)
        print(json.dumps({"kind": "layernorm_q", "M": M, "N": N, "us": round(us, 2)}), flush=True)


def layer_glue():
    """One decoder layer's linears with the element-wise operations around them (attention itself excluded): the
    reference's launch structure -- RMSNorm, q/k/v, o_proj, add, RMSNorm, gate/up, SiLU*mul, down_proj, add = 9 launches --
    on this library's kernels, against the fused prologues/epilogues (4 launches).  One hipGraph of 24 layers each."""
    import ctypes as Ct
    from tinychatengine_amd.linear import Linear_half_int4
    L = capi.lib()
    for (name, h, f, qkv) in [("baseline-named 4096/11008", 4096, 11008, 12288), ("llama3-8b 4096/14336 (GQA)", 4096, 14336, 6144)]:
        reps = 8  # distinct weight sets so that every launch streams from HBM
        zw = lambda k: quantize.calculate_zeros_width(k, 128)
        mk = lambda n, k: Linear_half_int4(torch.randint(-2**31, 2**31 - 1, (n, k // 8), dtype=torch.int32, device=dev),
                                           (torch.rand((n, zw(k) * 8), device=dev) * 0.01).to(torch.float16),
                                           torch.full((n, zw(k)), -2004318072, dtype=torch.int32, device=dev))
        sets = []
        for r in range(reps):
            w = {"qkv": mk(qkv, h), "o": mk(h, h), "gate": mk(f, h), "up": mk(f, h), "down": mk(h, f)}
            w["gu"] = Linear_half_int4.interleave(w["gate"], w["up"])
            sets.append(w)
        gamma = torch.ones(h, device=dev)
        hid = torch.randn(1, h, device=dev).to(torch.float16); xn = torch.empty_like(hid)
        t_qkv = torch.empty(1, qkv, dtype=torch.float16, device=dev); attn = torch.randn(1, h, device=dev).to(torch.float16)
        t_o = torch.empty(1, h, dtype=torch.float16, device=dev); t_g = torch.empty(1, f, dtype=torch.float16, device=dev)
        t_u = torch.empty(1, f, dtype=torch.float16, device=dev); t_d = torch.empty(1, h, dtype=torch.float16, device=dev)

        def one(d):
            return (capi.W4A16Desc * 1)(d)

        def unfused(i, sp):
            w = sets[i % reps]
            capi.check(L.tce_rmsnorm_half(hid.data_ptr(), gamma.data_ptr(), xn.data_ptr(), 1, h, 1e-6, sp))
            capi.check(capi.w4a16_forward(w["qkv"].desc(xn, t_qkv), sp.value))
            capi.check(capi.w4a16_forward(w["o"].desc(attn, t_o), sp.value))
            capi.check(L.tce_add_half(hid.data_ptr(), t_o.data_ptr(), hid.data_ptr(), h, sp))
            capi.check(L.tce_rmsnorm_half(hid.data_ptr(), gamma.data_ptr(), xn.data_ptr(), 1, h, 1e-6, sp))
            capi.check(capi.w4a16_forward_group([w["gate"].desc(xn, t_g), w["up"].desc(xn, t_u)], sp.value))
            capi.check(L.tce_silu_mul_half(t_g.data_ptr(), t_u.data_ptr(), f, sp))
            capi.check(capi.w4a16_forward(w["down"].desc(t_g, t_d), sp.value))
            capi.check(L.tce_add_half(hid.data_ptr(), t_d.data_ptr(), hid.data_ptr(), h, sp))

        def fused(i, sp):
            w = sets[i % reps]
            capi.check(L.tce_w4a16_forward_group_rmsnorm(one(w["qkv"].desc(hid, t_qkv)), 1, gamma.data_ptr(), 1e-6, sp))
            capi.check(capi.w4a16_forward(w["o"].desc(attn, hid, flags=capi.TCE_W4_ADD_TO_C), sp.value))
            capi.check(L.tce_w4a16_forward_group_rmsnorm(one(w["gu"].desc(hid, t_g, flags=capi.TCE_W4_SILU_MUL_PAIRS)), 1, gamma.data_ptr(), 1e-6, sp))
            capi.check(capi.w4a16_forward(w["down"].desc(t_g, hid, flags=capi.TCE_W4_ADD_TO_C), sp.value))

        for nm, fn in (("reference launch structure (9 launches)", unfused), ("fused prologues/epilogues (4 launches)", fused)):
            us = time_graph(fn, 24)
            print(json.dumps({"kind": "layer_glue", "shape": name, "form": nm, "us_per_layer": round(us, 2)}), flush=True)
        del sets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    capi.lib()
    variants = capi.gemv_variants()
    shapes = [
        ("o_proj 4096x4096", [4096], 4096),
        ("qkv fused 12288x4096", [12288], 4096),
        ("gate+up grouped 2x11008x4096", [11008, 11008], 4096),
        ("down 4096x11008", [4096], 11008),
        ("lm_head 32000x4096", [32000], 4096),
        ("L3 qkv grouped 4096+1024+1024", [4096, 1024, 1024], 4096),
        ("L3 gate+up grouped 2x14336x4096", [14336, 14336], 4096),
        ("L3 down 4096x14336", [4096], 14336),
        ("L3 lm_head 128256x4096", [128256], 4096),
        ("tp8 o 512x4096", [512], 4096),
        ("tp8 gate+up 2x1376x4096", [1376, 1376], 4096),
        ("tp8 down 512x11008", [512], 11008),
        ("13B tp8 down 640x13824", [640], 13824),
    ]
    if args.quick:
        shapes = shapes[:5]
    if args.only in ("", "gemv"):
        sweep_gemv(shapes, variants)
        sweep_gemv([("M=2 4096x4096", [4096], 4096), ("M=4 gate+up", [11008, 11008], 4096), ("M=8 4096x4096", [4096], 4096)][:1 if args.quick else 3],
                   [(2, 4, 1, 2), (4, 4, 1, 2), (1, 4, 1, 2)], M=2)
    if args.only == "experiments":
        experiments()
    if args.only in ("", "gemm"):
        sweep_gemm([(4096, 4096), (11008, 4096), (4096, 11008)])
    if args.only in ("", "w8a8"):
        sweep_w8a8()
    if args.only in ("", "glue"):
        layer_glue()


if __name__ == "__main__":
    main()
