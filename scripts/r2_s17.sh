#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s17; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_comm.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -6 > $O/pytest.log
for G in peer rccl; do
TCE_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 30 --warmup 5 --backend gloo --gather $G --no-cpu-baseline > $O/bench_n2_$G.json 2> $O/bench_n2_$G.err
done
TCE_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29518 \
    bench.py --gpus 4 --steps 20 --warmup 5 --backend gloo --gather peer --gathers-per-block 4 --no-cpu-baseline > $O/bench_n4_peer4.json 2> $O/bench_n4_peer4.err
timeout 300 python bench.py --force-dist --steps 50 --no-cpu-baseline --no-extras > $O/bench_force_dist.json 2> $O/bench_force_dist.err
tail -3 $O/pytest.log; for f in bench_n2_peer bench_n2_rccl bench_n4_peer4 bench_force_dist; do cut -c1-420 $O/$f.json; grep -v "socket.cpp\|amdgpu.ids\|Gloo" $O/$f.err | tail -3; done
