#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s13; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_w4a16_pk.py tests/test_gpu_adapter.py tests/test_l2_link.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 > $O/pytest.log
timeout 300 python scripts/gemm_pk_sweep.py 384 512 640 > $O/sweep.jsonl 2> $O/sweep.err
tail -10 $O/pytest.log; cut -c1-600 $O/sweep.jsonl; tail -3 $O/sweep.err
