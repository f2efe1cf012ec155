#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chain.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -25 > $O/pytest_chain.log
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
tail -12 $O/pytest_chain.log; cat $O/bench.json | cut -c1-2500; tail -5 $O/bench.err
