#!/bin/bash
# round 2, session 5: whole GPU suite, smoke, the bench line (N=1), the N=2 orchestration on one device (peer-write and gloo gathers)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
rm -f gpurun_out/parity_report.jsonl
timeout 900 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 > $O/pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest.log
cp gpurun_out/parity_report.jsonl $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
for G in peer rccl; do
  TCE_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 30 --warmup 5 --backend gloo --gather $G --no-cpu-baseline > $O/bench_n2_$G.json 2> $O/bench_n2_$G.err
done
tail -6 $O/pytest.log; cat $O/smoke.log | tail -2; cat $O/bench_n1.json; tail -3 $O/bench_n1.err; for G in peer rccl; do cat $O/bench_n2_$G.json; tail -3 $O/bench_n2_$G.err; done
