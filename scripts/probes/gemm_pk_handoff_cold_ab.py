#!/usr/bin/env python3
"""Round 5 probe: the two-run hand-off (mode 695) against the last-arriver forms (694) in bench.py's own setting -- graphs of 16 launches rotating over > 256 MiB of distinct
weights, i.e. weights from HBM -- same process, alternating."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from tinychatengine_amd import capi
dev = torch.device("cuda:0"); L = capi.lib()
res = {}
for rnd in range(2):
    for mode, name in ((694, "last_arriver"), (695, "handoff")):
        capi.check(L.tce_w4a16_set_debug_mode(mode))
        out = bench.other_configs_leg(torch, dev)
        for row in out["w4a16_prefill_gemm_M512"]:
            res.setdefault(f'{row["M"]}x{row["N"]}x{row["K"]}', {}).setdefault(name, []).append((row["prepacked"]["us"], row["prepacked"]["dispatch"].split("group")[0].replace("gemm-pk ", "")))
L.tce_w4a16_set_debug_mode(695)
for k, v in res.items():
    print(json.dumps({"shape": k, **v}), flush=True)
