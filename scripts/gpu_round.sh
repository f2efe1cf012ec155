#!/bin/bash
# One GPU-box session: parity tests, smoke, tuning sweep, bench.  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r1}
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/${TAG}_box.txt; nproc >> gpurun_out/${TAG}_box.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/${TAG}_box.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/${TAG}_pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/${TAG}_smoke.log
timeout 600 python scripts/tune.py > gpurun_out/${TAG}_tune.jsonl 2> gpurun_out/${TAG}_tune.err
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -5 gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_smoke.log | tail -3; tail -3 gpurun_out/${TAG}_bench.json
