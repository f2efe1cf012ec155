#!/usr/bin/env python3
"""Round 5 probe: every form of the packed prefill GEMM under UNEVEN load -- a second stream keeps the memory system busy with large copies while the forms are launched in
rotation (each launch finds other code in the instruction cache), every output compared bit for bit with the form's first result and that against the oracle.
    python scripts/probes/pk_stress.py [seconds per shape]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from test_gpu_w4a16_pk import _quant, _lin
from conftest import w4a16_close
from tinychatengine_amd import capi
from oracle.oracle import Oracle  # (the checker: this probe is test infrastructure)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
# (round 6: the two-quartet forms are offered for every group size again -- mode 62 forces form 2 on groups of 32 directly)
dev = torch.device("cuda:0"); L = capi.lib(); oracle = Oracle()
side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big_b = torch.empty_like(big_a)
small = [torch.randn(1 << 16, device=dev) for _ in range(8)]
CASES = [(192, 200, 512, 32, [61, 62, 63, 64, 60]), (700, 392, 3072, 32, [61, 62, 63, 64, 60]), (260, 300, 1408, 64, [61, 62, 63, 64, 60]),
         (513, 2100, 256, 128, [61, 62, 63, 64, 66, 67, 68, 2670, 2671, 2673, 2674, 2675, 60]), (384, 520, 2048, 128, [61, 62, 63, 64, 66, 67, 672, 68, 2669, 2670, 2671, 2672, 2683, 2673, 2674, 2675, 60]),
         (260, 300, 1408, 128, [2676, 64, 60])]  # (round 6: form 16 -- two quartets per tile and the k range handed off between two workgroups)
if os.environ.get("STRESS_CASES"): CASES = [CASES[int(i)] for i in os.environ["STRESS_CASES"].split(",")]
if os.environ.get("STRESS_MODES"): CASES = [(M, N, K, G, [int(v) for v in os.environ["STRESS_MODES"].split(",")]) for (M, N, K, G, _) in CASES]
for (M, N, K, G, modes) in CASES:
    rng = np.random.default_rng(M + N + K)
    qw, sc, zp = _quant(oracle, N, K, G, seed=M * 3 + N + K, random_zeros=False, zero_scale_groups=0)
    a = rng.standard_normal((M, K)).astype(np.float16)
    ref32 = oracle.w4a16_gemv_q4_6_mt(a, qw, sc, zp, M, N, K, G)
    lin = _lin(dev, qw, sc, zp, G).prepack()
    x = torch.from_numpy(a).to(dev)
    first, bad, runs, first_bad, where = {}, {m: 0 for m in modes}, {m: 0 for m in modes}, {}, []
    outs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(len(modes))]
    t_end = time.time() + secs
    it = 0
    while time.time() < t_end:
        with torch.cuda.stream(side):  # ~1 ms of copies per round, beside the launches below
            for _ in range(3): big_b.copy_(big_a, non_blocking=True)
            for s_ in small: s_.mul_(1.0001)
        for mi, mode in enumerate(modes):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            outs[mi].fill_(float("nan"))
            d = lin.desc(x, outs[mi])
            capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        for mi, mode in enumerate(modes):
            runs[mode] += 1
            if mode not in first:
                first[mode] = outs[mi].clone()
                ok, worst = w4a16_close(first[mode].cpu().numpy(), ref32)
                if not ok: first_bad[mode] = round(float(worst), 2)
            elif not torch.equal(outs[mi], first[mode]):
                bad[mode] += 1
                if len(where) < 6:  # what differs: (row, column) within the 128 x 128 tile, the values
                    ne = (outs[mi] != first[mode]).nonzero()
                    where.append({"mode": mode, "n": int(ne.shape[0]), "tile": [int(ne[0, 0]) // 128, int(ne[0, 1]) // 128], "rows": sorted(set(int(r) % 128 for r in ne[:, 0].tolist()))[:8], "cols": sorted(set(int(c) % 128 for c in ne[:, 1].tolist()))[:20],
                                  "got_vs_first": [(round(float(outs[mi][r, c]), 4), round(float(first[mode][r, c]), 4)) for r, c in ne[:2].tolist()]})
        it += 1
    L.tce_w4a16_set_debug_mode(60)
    print(json.dumps({"M": M, "N": N, "K": K, "G": G, "rounds": it, "first_result_off_the_oracle": first_bad, "results_differing_from_the_first": {m: b for m, b in bad.items() if b}, "where": where}), flush=True)
