#!/bin/bash
# Short GPU-box session: parity tests, smoke, bench, the rocprofv3 passes of the dominant GEMV launch and of the prefill GEMM.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-val}
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/${TAG}_pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --steps 100 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
bash scripts/profile.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1
bash scripts/prof_gemm.sh > gpurun_out/${TAG}_prof_gemm.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log; tail -2 gpurun_out/${TAG}_smoke.log; tail -2 gpurun_out/${TAG}_bench.json | cut -c1-1500; tail -3 gpurun_out/${TAG}_bench.err
