#!/usr/bin/env python3
"""Where a token kernel's time goes: per-launch wall-clock stamps of every workgroup (tce_w4a16_set_debug_mode(2) + a debug buffer
while the tagged plan is built).  Prints one JSON line per launch shape (averaged over the blocks, first block skipped) and a total."""
import ctypes as C, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tinychatengine_amd import lab; lab.use_lab()  # (per-launch stamps: the diagnostics build)
from tinychatengine_amd import capi
from tinychatengine_amd.decode import SHAPES, DecodeLinears

L = capi.lib()
dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = [int(v) for v in sys.argv[2:5]] if len(sys.argv) > 4 else None  # rows(+10*bpc), waves, depth of the persistent kernel
dl = DecodeLinears(SHAPES["baseline-named"], device=dev, group_size=128, layers=layers, dataflow=True)
n = layers * 4 + 1
buf = torch.zeros((256 * 2, n, 8), dtype=torch.int64, device=dev)
if cfg:
    capi.set_gemv_config(cfg[0], cfg[1], 0, cfg[2])
capi.check(L.tce_w4a16_set_debug_buffer(C.c_void_p(buf.data_ptr())))
capi.check(L.tce_w4a16_set_debug_mode(2))
plan = dl.make_plan(tagged=True)
L.tce_w4a16_set_debug_mode(0)
assert plan.tagged, plan.kind
geo = plan.geometry()
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    plan.launch(st)
plan.status()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    plan.launch(st)
e1.record()
plan.status()
ms = e0.elapsed_time(e1) / 20
t = buf.cpu().numpy()[: geo["workgroups"]].astype(np.float64) * 0.01  # us; [wg][launch][stamp]
t -= t[:, 0, 0].min()
names = ["qkv 12288x4096", "o 4096x4096", "gate+up 22016x4096", "down 4096x11008"]
end_prev = None
rows = {k: [] for k in names + ["lm_head 32000x4096"]}
for j in range(n):
    s = t[:, j, :]
    rec = {
        "period_us": (s[:, 4].max() - end_prev) if end_prev is not None else None,  # last workgroup done -> last workgroup done
        "enter_skew_us": s[:, 0].max() - s[:, 0].min(),
        "wait_for_activations_us": float(np.mean(s[:, 1] - s[:, 0])), "wait_max_us": float(np.max(s[:, 1] - s[:, 0])),
        "first_x_in_registers_after_last_producer_done_us": (s[:, 1].min() - end_prev) if end_prev is not None else None,
        "stage_image_us": float(np.mean(s[:, 2] - s[:, 1])),
        "stream_and_math_us": float(np.mean(s[:, 3] - s[:, 2])), "stream_and_math_max_us": float(np.max(s[:, 3] - s[:, 2])),
        "wave0_to_workgroup_done_us": float(np.mean(s[:, 4] - s[:, 3])),
        "done_skew_us": s[:, 4].max() - s[:, 4].min(),
    }
    end_prev = s[:, 4].max()
    if j >= 4:
        rows[names[j % 4] if j < n - 1 else "lm_head 32000x4096"].append(rec)
for k, v in rows.items():
    if not v:
        continue
    out = {"launch": k, "n": len(v)}
    for key in v[0]:
        vals = [r[key] for r in v if r[key] is not None]
        out[key] = round(float(np.mean(vals)), 2) if vals else None
    print(json.dumps(out))
print(json.dumps({"ms_per_token": round(ms, 4), "tokens_per_s": round(1e3 / ms, 1), "geometry": geo, "layers": layers,
                  "token_span_from_stamps_us": round(float(t[:, n - 1, 4].max() - t[:, 0, 0].min()), 1)}))
