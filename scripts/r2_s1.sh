#!/bin/bash
# round-2 session 1: parity suite with the new full-size cases, the VALU/MFMA issue probe, a baseline bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/r2s1_pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/r2s1_pytest.log
timeout 120 scripts/probes/valu_probe > gpurun_out/r2s1_valu_probe.jsonl 2> gpurun_out/r2s1_valu_probe.err
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2s1_bench.json 2> gpurun_out/r2s1_bench.err
tail -5 gpurun_out/r2s1_pytest.log; cat gpurun_out/r2s1_valu_probe.jsonl | cut -c1-400; tail -c 600 gpurun_out/r2s1_bench.err
