#!/bin/bash
# round 6, session 2: the groups-of-32 two-quartet kernel -- pads behind the loop's exit (post-loop multiplies), in 8-byte units
export TMPDIR=/tmp
mkdir -p gpurun_out/s2
X2ANY=1 SECS=${SECS:-4} bash scripts/probes/pk_stress_variants.sh asm0 post2 post4 post8 post16 post32 post64 postafter64 postvalu64 post64unpk post64unpk1 post64spread post64hipair hipair unpk unpk1 gapA2 gapB2 gapB8 gapC2 allB4 > gpurun_out/s2/stress.log 2>&1
for v in post64 asm0; do
  echo "== explain $v" >> gpurun_out/s2/explain.log
  TCE_LIB_PATH=$PWD/tinychatengine_amd/lib/abl/libtce_$v.so X2ANY=1 EXPLAIN=1 REPS=2 timeout 120 python scripts/probes/pk_form2_g32_repeat.py 192 200 512 32 62 >> gpurun_out/s2/explain.log 2>&1
done
tail -c 3000 gpurun_out/s2/explain.log
