#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r}
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/${TAG}_pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/${TAG}_pytest.log
tail -12 gpurun_out/${TAG}_pytest.log | cut -c1-300
bash scripts/gpu_exp.sh ${TAG}
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -2 gpurun_out/${TAG}_bench.err; cut -c1-300 gpurun_out/${TAG}_bench.json
