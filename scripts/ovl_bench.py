#!/usr/bin/env python3
"""TCE_PLAN_OVERLAPPED against the stream-ordered hipGraph on one decode token's linears (the bench's data-flow wiring), in ONE process:
verifies every output bit-identical, then times both, for a list of (branches, workgroups per CU per launch) settings.
    python scripts/ovl_bench.py [workload] [layers] [--timeline]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tinychatengine_amd import lab; lab.use_lab()  # (per-launch stamps: the diagnostics build)
from tinychatengine_amd import capi
from tinychatengine_amd.decode import SHAPES, DecodeLinears

L = capi.lib()
dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
workload = args[0] if args else "baseline-named"
layers = int(args[1]) if len(args) > 1 else None
timeline = "--timeline" in sys.argv
dl = DecodeLinears(SHAPES[workload], device=dev, group_size=128, layers=layers, dataflow=True)
st = torch.cuda.current_stream().cuda_stream
outs = [*dl.out_qkv, dl.out_o, dl.out_gate, dl.out_up, dl.out_down, dl.logits]


def rate(plan, n=60):
    for _ in range(10):
        plan.launch(st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        plan.launch(st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


plain = dl.make_plan()
plain.launch(st)
torch.cuda.synchronize()
want = [o.clone() for o in outs]
ms_g = rate(plain)
print(json.dumps({"plan": "hipGraph, stream order", "ms_per_token": round(ms_g, 4), "tokens_per_s": round(1e3 / ms_g, 1)}), flush=True)
for (chains, wpc) in ((1, 3), (1, 4), (1, 2), (2, 3), (2, 2)):
    capi.check(L.tce_w4a16_set_debug_mode(50000 + 100 * chains + wpc))
    try:
        p = dl.make_plan(overlapped=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"plan": f"overlapped chains={chains} wpc={wpc}", "error": str(e)[:200]}), flush=True)
        continue
    if not p.overlapped:
        print(json.dumps({"plan": f"overlapped chains={chains} wpc={wpc}", "error": f"built as kind {p.kind}"}), flush=True)
        continue
    bad = None
    for rep in range(3):
        for o in outs:
            o.fill_(float("nan"))
        p.launch(st)
        try:
            p.status()
        except Exception as e:  # noqa: BLE001
            bad = f"status: {e}"
            break
        if not all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(want, outs)):
            bad = f"outputs differ (replay {rep})"
            break
    if bad:
        print(json.dumps({"plan": f"overlapped chains={chains} wpc={wpc}", "error": bad}), flush=True)
        p.close()
        continue
    ms = rate(p)
    ms2 = rate(plain, 30)
    p.status()
    print(json.dumps({"plan": f"overlapped chains={chains} wpc={wpc}", "ms_per_token": round(ms, 4), "tokens_per_s": round(1e3 / ms, 1), "geometry": p.geometry(),
                      "graph_again_ms": round(ms2, 4), "verified": "bit-identical, 3 replays"}), flush=True)
    p.close()
capi.check(L.tce_w4a16_set_debug_mode(50000))

if timeline:  # per-launch stamps of every workgroup: {entered, x complete, x staged, done} (100 MHz wall clock)
    n = dl.n_layers * 4 + 1
    buf = torch.zeros((n, 4096, 4), dtype=torch.int64, device=dev)
    capi.check(L.tce_w4a16_set_debug_buffer(C.c_void_p(buf.data_ptr())))
    capi.check(L.tce_w4a16_set_debug_mode(2))
    p = dl.make_plan(overlapped=True)
    L.tce_w4a16_set_debug_mode(0)
    for _ in range(5):
        p.launch(st)
    p.status()
    t = buf.cpu().numpy().astype(np.float64) * 0.01  # us
    names = ["qkv", "o", "gate+up", "down"]
    rows = {}
    prev_done = None
    for j in range(n):
        s = t[j]
        live = s[:, 0] > 0
        s = s[live]
        rec = {"workgroups": int(live.sum()), "enter_first": float(s[:, 0].min()), "x_complete_median": float(np.median(s[:, 1])), "x_complete_last": float(s[:, 1].max()),
               "staged_last": float(s[:, 2].max()), "done_first": float(s[:, 3].min()), "done_last": float(s[:, 3].max())}
        if prev_done is not None:
            rec["period"] = rec["done_last"] - prev_done
            rec["entered_before_producer_done"] = prev_done - rec["enter_first"]
            rec["x_after_producer_done"] = rec["x_complete_last"] - prev_done
            rec["stream_after_x"] = rec["done_last"] - rec["x_complete_last"]
        prev_done = rec["done_last"]
        if j >= 4:
            rows.setdefault(names[j % 4] if j < n - 1 else "lm_head", []).append(rec)
    for k, v in rows.items():
        keys = ["workgroups", "period", "entered_before_producer_done", "x_after_producer_done", "stream_after_x"]
        print(json.dumps({"launch": k, **{key: round(float(np.mean([r[key] for r in v if key in r])), 2) for key in keys}}), flush=True)
    p.close()
