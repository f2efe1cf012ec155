#!/bin/bash
# round 2, session 9: the whole GPU suite, smoke, bench lines (N=1; N=2 orchestration on one device), rocprofv3 evidence
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s9; mkdir -p $O
rm -f gpurun_out/parity_report.jsonl
timeout 900 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 > $O/pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest.log
cp gpurun_out/parity_report.jsonl $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --workload llama3-8b --no-cpu-baseline --no-extras > $O/bench_n1_llama3.json 2> $O/bench_n1_llama3.err
timeout 600 python bench.py --workload llama2-13b --no-cpu-baseline --no-extras > $O/bench_n1_13b.json 2> $O/bench_n1_13b.err
TCE_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 30 --warmup 5 --backend gloo --gather peer --no-cpu-baseline > $O/bench_n2_peer.json 2> $O/bench_n2_peer.err
TCE_ALGO_BYTES=46558208 bash scripts/profile.sh r2 > $O/profile.log 2>&1
bash scripts/prof_gemm_pk.sh > $O/prof_gemm_pk.log 2>&1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tail -6 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-700 $O/bench_n1.json; tail -3 $O/bench_n1.err; cut -c1-400 $O/bench_n1_llama3.json; cut -c1-400 $O/bench_n1_13b.json; cut -c1-600 $O/bench_n2_peer.json; tail -3 $O/bench_n2_peer.err; tail -30 $O/profile.log; tail -40 $O/prof_gemm_pk.log
