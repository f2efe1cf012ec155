#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_epilogues.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 > $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 100 > $O/bench.json 2> $O/bench.err
tail -8 $O/pytest.log; cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
