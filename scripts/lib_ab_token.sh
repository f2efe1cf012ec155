#!/bin/bash
# as lib_ab.sh, on the whole token with attention (161 launches, the fused RMSNorm prologues included): tokens/s at 512 / 2048 keys per build
for v in A B A B; do TCE_LIB_PATH=$PWD/tinychatengine_amd/lib_ab/libtce_hip_$v.so python - <<PY
import json, torch, bench
from tinychatengine_amd.decode import SHAPES, DecodeLinears
dev = torch.device("cuda:0")
shape = SHAPES["baseline-named"]
dl = DecodeLinears(shape, device=dev, group_size=128, m=1, layers=1)
r = bench.whole_token_leg(torch, dev, shape, dl)
print("$v", json.dumps({k: (v.get("tokens_per_s") if isinstance(v, dict) else v) for k, v in r.items() if k != "prefill"})[:300])
PY
done
