#!/bin/bash
# Round 6, one box: parity tests, smoke, the bench line, the dominant launch's traffic pass (-> traffic.json, what bench.py's roofline.traffic reads), the counter
# passes of every kernel family (scripts/profile_r5.sh), then the bench line again with the traffic figure in it.  usage: gpu_session_r5.sh [tag] [skip-tests]
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r6}
if [ "$2" != "skip-tests" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/${TAG}_pytest.log
  echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/${TAG}_pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/${TAG}_smoke.log
fi
PROFILE_WORKLOADS="" bash scripts/profile_r4.sh ${TAG} > gpurun_out/${TAG}_profile_dominant.log 2>&1
mkdir -p profiles/r6 && cp gpurun_out/prof_${TAG}_dominant/traffic.json profiles/r6/traffic.json 2>/dev/null
bash scripts/profile_r5.sh ${TAG} > gpurun_out/${TAG}_profile_families.log 2>&1
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -4 gpurun_out/${TAG}_pytest.log; tail -2 gpurun_out/${TAG}_smoke.log; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-1200; tail -3 gpurun_out/${TAG}_bench.err; tail -3 gpurun_out/${TAG}_profile_families.log
