#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_w4a16_pk.py tests/test_gpu_adapter.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -30 > gpurun_out/r2s3_pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/r2s3_pytest.log
timeout 600 python scripts/gemm_pk_sweep.py 512 2048 > gpurun_out/r2s3_gemm_pk_sweep.jsonl 2> gpurun_out/r2s3_gemm_pk_sweep.err
tail -12 gpurun_out/r2s3_pytest.log; cat gpurun_out/r2s3_gemm_pk_sweep.jsonl; tail -3 gpurun_out/r2s3_gemm_pk_sweep.err
