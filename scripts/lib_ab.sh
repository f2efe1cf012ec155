#!/bin/bash
# A/B of two BUILDS of the library on the headline (same box, alternating): put them at tinychatengine_amd/lib_ab/libtce_hip_A.so / _B.so (git-ignored, they travel
# with gpurun) and run `gpurun -- bash scripts/lib_ab.sh`.  capi.py loads TCE_LIB_PATH when it is set.  Prints: build, tokens/s, dominant launch's fraction of 8 TB/s, its us.
for v in A B A B; do TCE_LIB_PATH=$PWD/tinychatengine_amd/lib_ab/libtce_hip_$v.so python bench.py --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['value'], r['frac'], r['avg_launch_us'])"; done
