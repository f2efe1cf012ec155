#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-x}
timeout 600 python scripts/tune.py --only experiments > gpurun_out/${TAG}_exp.jsonl 2> gpurun_out/${TAG}_exp.err
tail -3 gpurun_out/${TAG}_exp.err
python - <<PY
import json
from collections import defaultdict
t=defaultdict(dict)
for l in open('gpurun_out/${TAG}_exp.jsonl'):
    r=json.loads(l)
    if r['kind']=='exp' and 'us' in r: t[(r['shape'],tuple(r['variant']))][r['mode']]=r['us']
    elif 'error' in r: print(r)
for k,v in t.items(): print(f"{k[0][:30]:30s} {k[1]} " + "  ".join(f"{m}:{u:.2f}" for m,u in v.items()))
PY
