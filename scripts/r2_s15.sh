#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_w4a16.py tests/test_gpu_chain.py tests/test_gpu_epilogues.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8 > $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --workload llama3-8b --no-cpu-baseline --no-extras > $O/bench_l3.json 2> $O/bench_l3.err
timeout 600 python scripts/tune.py --only gemv > $O/tune.jsonl 2>> $O/bench.err
tail -5 $O/pytest.log
