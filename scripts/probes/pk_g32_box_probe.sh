#!/bin/bash
# Round 5: is the first-execution failure of the groups-of-32 GEMM forms a property of the BOX?  Prints the GPU's unique id, runs the packed-GEMM oracle tests once in a fresh
# process; on a failure: the same tests against the variant builds of scripts/probes/build_pk_variants.sh (same box), then the per-element explanation of a wrong output.
export TMPDIR=/tmp
echo "box: $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -2 | tr '\n' ' ') $(cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -E 'unique_id' | sort -u | tr '\n' ' ' | cut -c1-200)"
run() { timeout 300 python -m pytest tests/test_gpu_w4a16_pk.py -q -m gpu -k "${2:-matches_oracle}" 2>&1 | grep -E "^E +Assertion|passed|failed" | cut -c1-220 | head -${3:-6} | sed "s/^/$1: /"; }
out=$(run current matches_oracle 8)
echo "$out"
if echo "$out" | grep -q failed; then
  echo "== BAD BOX: variants"
  for v in nopin nosched waits; do TCE_LIB_PATH=$PWD/tinychatengine_amd/lib/abl/libtce_$v.so run $v matches_oracle 4; TCE_LIB_PATH=$PWD/tinychatengine_amd/lib/abl/libtce_$v.so run $v-again matches_oracle 4; done
  run current-again matches_oracle 4
  echo "== explain"
  for sh in "192 200 512 32" "700 392 3072 32"; do EXPLAIN=1 REPS=2 timeout 200 python scripts/probes/pk_form2_g32_repeat.py $sh 61 62 63 64 60 2>&1 | grep -v amdgpu | cut -c1-600 | head -30; done
  rocm-smi --showclocks --showpower 2>/dev/null | head -30
fi
