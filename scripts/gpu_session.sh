#!/bin/bash
# One GPU-box session (gpurun -- 'bash scripts/gpu_session.sh TAG'): parity tests, smoke, bench, the GEMV experiment
# sweeps, the GEMM sweep, the rocprofv3 passes of the dominant kernel, the persistent-kernel timeline.  Everything lands
# in gpurun_out/TAG_*; what is to be judged gets copied into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r1b}
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/${TAG}_pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --steps 100 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 600 python scripts/tune.py --only experiments > gpurun_out/${TAG}_exp.jsonl 2> gpurun_out/${TAG}_exp.err
timeout 600 python scripts/tune.py --only gemm > gpurun_out/${TAG}_gemm.jsonl 2>> gpurun_out/${TAG}_exp.err
bash scripts/profile.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1
timeout 200 python scripts/timeline_stream.py > gpurun_out/${TAG}_timeline_stream.jsonl 2>> gpurun_out/${TAG}_exp.err
timeout 200 python scripts/chain_exp.py > gpurun_out/${TAG}_chain.jsonl 2>> gpurun_out/${TAG}_exp.err
tail -4 gpurun_out/${TAG}_pytest.log; tail -2 gpurun_out/${TAG}_bench.json | cut -c1-1500; tail -3 gpurun_out/${TAG}_bench.err
