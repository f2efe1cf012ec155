#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_w8a8.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r2s4_pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/r2s4_pytest.log
timeout 300 python scripts/attention_ops_time.py > gpurun_out/r2s4_attention_time.jsonl 2> gpurun_out/r2s4_attention_time.err
timeout 300 python scripts/opt_decode_layer_time.py > gpurun_out/r2s4_opt_layer.jsonl 2> gpurun_out/r2s4_opt_layer.err
tail -15 gpurun_out/r2s4_pytest.log; cat gpurun_out/r2s4_attention_time.jsonl; tail -3 gpurun_out/r2s4_attention_time.err; cat gpurun_out/r2s4_opt_layer.jsonl; tail -3 gpurun_out/r2s4_opt_layer.err
