#!/usr/bin/env python3
"""Round 6 probe: a decode token's linears as ONE persistent kernel on the int8-contraction body (csrc/w4a16_gemv_i8_token.hip, TCE_PLAN_TAGGED on packed copies) against
the hipGraph of stream-ordered launches -- outputs compared bit for bit, both timed, optionally with the per-stage timeline (wall-clock stamps of every workgroup).
    python scripts/probes/i8_token_probe.py [workload] [--stamps] [--max-units U] [--layers L]"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
if "--stamps" in sys.argv:
    from tinychatengine_amd import lab; lab.use_lab()  # (the token kernel with wall-clock stamps: the diagnostics build)
from tinychatengine_amd import capi
from tinychatengine_amd.decode import SHAPES, DecodeLinears

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="llama3-8b")
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--max-units", type=int, default=0)
ap.add_argument("--layers", type=int, default=None)
ap.add_argument("--reps", type=int, default=200)
args = ap.parse_args()
dev = torch.device("cuda:0"); L = capi.lib()
shape = SHAPES[args.workload]
dl = DecodeLinears(shape, device=dev, group_size=128, m=1, layers=args.layers, dataflow=True, prepack=True)
stream = torch.cuda.current_stream().cuda_stream
plan = dl.make_plan()
n = plan.n_launches
stamps = None
if args.max_units: capi.check(L.tce_w4a16_set_debug_mode(7710 + args.max_units))
if args.stamps:
    stamps = torch.zeros(256 * n * 8, dtype=torch.int64, device=dev)
    capi.check(L.tce_w4a16_set_debug_buffer(C.c_void_p(stamps.data_ptr())))
    capi.check(L.tce_w4a16_set_debug_mode(7702))
tplan = dl.make_plan(tagged=True)
if args.stamps: capi.check(L.tce_w4a16_set_debug_mode(7703))
capi.check(L.tce_w4a16_set_debug_mode(7710))
geo = tplan.geometry() if tplan.kind == 4 else None
outs = [*dl.out_qkv, dl.out_o, dl.out_gate, dl.out_up, dl.out_down, dl.logits]
plan.launch(stream); torch.cuda.synchronize()
want = [o.clone() for o in outs]
bad = None
for rep in range(5):
    for o in outs: o.fill_(float("nan"))
    tplan.launch(stream); tplan.status()
    for name, a, b in zip(("q", "k", "v", "o", "gate", "up", "down", "logits"), want, outs):
        if not torch.equal(a.view(torch.int16), b.view(torch.int16)):
            bad = bad or f"replay {rep}: {name}: {(a.view(torch.int16) != b.view(torch.int16)).sum().item()} of {a.numel()} values differ (nan: {torch.isnan(b.float()).sum().item()})"
def rate(fn, k):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
ms_g = min(rate(lambda: plan.launch(stream), args.reps) for _ in range(3))
ms_t = min(rate(lambda: tplan.launch(stream), args.reps) for _ in range(3))
tplan.status()
res = {"workload": args.workload, "launches": n, "plan_kind": tplan.kind, "geometry": geo, "mismatch": bad, "graph_ms": round(ms_g, 4), "token_ms": round(ms_t, 4),
       "graph_tok_s": round(1e3 / ms_g, 1), "token_tok_s": round(1e3 / ms_t, 1)}
if stamps is not None and tplan.kind == 4:
    ns = geo["rows"]
    tplan.launch(stream); tplan.status()
    st = stamps.cpu().numpy().reshape(256, n, 8)[:, :ns, :].astype(np.float64) / 100.0  # us (100 MHz)
    st -= st[:, 0, 0].min()
    # stamps of wave 0 of every workgroup: 0 stage entered, 1 activations polled, 2 barrier A passed, 3 barrier B passed (planes ready), 5 first unit's weights landed,
    # 6 its arithmetic done, 7 its partial row counted in, 4 all of wave 0's units done
    med = lambda v: round(float(np.median(v)), 2)
    per_kind = {}
    for s_ in range(4, ns - 1):
        prev_done = st[:, s_ - 1, 4].max()
        e, x, a, b, d, wl, ar, cn = (st[:, s_, i] for i in (0, 1, 2, 3, 4, 5, 6, 7))
        k = per_kind.setdefault(s_ % 4, {key: [] for key in ("period", "enter_after_prev_done_max", "x_first_after_prev_done", "x_last_after_prev_done", "A_to_B_med", "B_last_after_prev_done",
                                                            "B_to_weights_med", "weights_to_arith_med", "arith_to_counted_med", "B_to_done_med", "B_to_done_max", "done_spread")})
        k["period"].append(d.max() - prev_done)
        k["enter_after_prev_done_max"].append(e.max() - prev_done)
        k["x_first_after_prev_done"].append(x.min() - prev_done)
        k["x_last_after_prev_done"].append(x.max() - prev_done)
        k["A_to_B_med"].append(np.median(b - a))
        k["B_last_after_prev_done"].append(b.max() - prev_done)
        k["B_to_weights_med"].append(np.median(wl - b))
        k["weights_to_arith_med"].append(np.median(ar - wl))
        k["arith_to_counted_med"].append(np.median(cn - ar))
        k["B_to_done_med"].append(np.median(d - b))
        k["B_to_done_max"].append((d - b).max())
        k["done_spread"].append(d.max() - d.min())
    names = {0: "qkv", 1: "o", 2: "gate+up", 3: "down"}
    res["per_stage_kind_us_median_over_blocks"] = {names[k]: {key: med(v) for key, v in d.items()} for k, d in per_kind.items()}
    res["token_span_us"] = round(float(st[:, ns - 1, 4].max()), 1)
print(json.dumps(res))
