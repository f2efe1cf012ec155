#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chain.py -m gpu -q --tb=short -p no:cacheprovider -k "tagged or True" 2>&1 | tail -15 > $O/pytest_chain.log
for cfg in "" "2 16 3" "4 16 2" "2 16 2" "1 8 3" "12 8 3"; do
  echo "{\"config\": \"$cfg\"}" >> $O/timeline.jsonl
  timeout 300 python scripts/token_timeline.py 32 $cfg >> $O/timeline.jsonl 2>> $O/timeline.err
done
tail -8 $O/pytest_chain.log; cat $O/timeline.jsonl; tail -5 $O/timeline.err
