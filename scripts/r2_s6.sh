#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
timeout 120 scripts/probes/handoff_probe > $O/handoff_probe.jsonl 2> $O/handoff_probe.err
timeout 600 python -m pytest tests/test_gpu_w4a16_pk.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_pk.log
cat $O/handoff_probe.jsonl; tail -3 $O/handoff_probe.err; tail -8 $O/pytest_pk.log
