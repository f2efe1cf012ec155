#!/bin/bash
# round-2 session 2: the pre-packed GEMM (parity, sweep), the corrected VALU/MFMA issue probe
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_w4a16_pk.py tests/test_gpu_adapter.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/r2s2_pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/r2s2_pytest.log
timeout 600 python scripts/gemm_pk_sweep.py 512 2048 > gpurun_out/r2s2_gemm_pk_sweep.jsonl 2> gpurun_out/r2s2_gemm_pk_sweep.err
timeout 120 scripts/probes/valu_probe > gpurun_out/r2s2_valu_probe.jsonl 2> gpurun_out/r2s2_valu_probe.err
tail -25 gpurun_out/r2s2_pytest.log; cat gpurun_out/r2s2_gemm_pk_sweep.jsonl; tail -3 gpurun_out/r2s2_gemm_pk_sweep.err; cut -c1-330 gpurun_out/r2s2_valu_probe.jsonl
