#!/bin/bash
# round 6, session 4: (a) the standalone reproducer, every instruction form; (b) the FIXED sources under the same stress, with the pads that drove the old build to ~100 % failures
export TMPDIR=/tmp
mkdir -p gpurun_out/s4
timeout 600 scripts/probes/pk_opsel_repro > gpurun_out/s4/repro_forms.jsonl 2>&1
X2ANY=1 SECS=${SECS:-8} bash scripts/probes/pk_stress_variants.sh fixed_asm0 fixed_post64 fixed_post256 > gpurun_out/s4/stress_fixed.log 2>&1
cut -c1-330 gpurun_out/s4/repro_forms.jsonl
grep -o '"M": [0-9]*\|"rounds": [0-9]*\|"results_differing_from_the_first": {[^}]*}\|^== .*' gpurun_out/s4/stress_fixed.log | tr '\n' ' '
