"""CPU, world_size 2 over gloo: the multi-GPU path (column-sharded linears + all-gather per block).

The kernels cannot run here, so the launch callback of DecodeLinears.run_token_distributed is replaced by a CPU
stand-in that evaluates each descriptor with the oracle (test infrastructure) straight from the descriptor's pointers.
What is under test is everything around the kernel: the row-range sharding of weights/scales/zeros, the descriptor
contents each rank builds, and the gather order/placement -- the gathered result must equal the unsharded computation
bit for bit."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _np(ptr, shape, dtype):
    n = int(np.prod(shape))
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def _oracle_launch(orc):
    """A stand-in for tce_w4a16_forward[_group]: run every descriptor of the group through the oracle."""
    from tinychatengine_amd import quantize
    def launch(group):
        for d in group:
            zw = quantize.calculate_zeros_width(d.K, d.group_size)
            a = _np(d.A, (d.M, d.K), np.float16)
            qw = _np(d.qweight, (d.N, d.K // 8), np.uint32)
            sc = _np(d.scales, (d.N, zw * 8), np.float16)
            zp = _np(d.zeros, (d.N, zw), np.uint32)
            _, c16 = orc.w4a16_gemv_q4_6(a, qw, sc, zp, d.M, d.N, d.K, d.group_size)
            _np(d.C, (d.M, d.N), np.float16)[...] = c16
    return launch


def _worker(rank, world, port, gathers, q, m=1):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import Oracle
        from tinychatengine_amd.decode import SHAPES, DecodeLinears
        orc = Oracle()
        dl = DecodeLinears(SHAPES["tiny"], device="cpu", rank=rank, world=world, m=m)
        dl.run_token_distributed(gathers_per_block=gathers, launch=_oracle_launch(orc))
        out = {"logits": dl.g_logits.clone(), "down": dl.g_down.clone()}
        if gathers == 4:
            out.update(o=dl.g_o.clone(), gate=dl.g_gate.clone(), up=dl.g_up.clone(), k=dl.g_qkv[1].clone())
        gathered = [None] * world
        dist.all_gather_object(gathered, {k: v.numpy().view(np.uint16).tobytes() for k, v in out.items()})
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gathers,m", [(1, 1), (4, 1), (4, 64)])
def test_column_sharded_token_equals_unsharded(gathers, m):
    """m = 64: a sharded PROMPT chunk -- the ranks' [M][N/P] blocks are gathered rank-major and the rows laid side by side (the sharded path used to
    assert M == 1)."""
    from oracle.oracle import Oracle
    from tinychatengine_amd.decode import SHAPES, DecodeLinears
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, gathers, q, m)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank holds the same gathered vectors
    assert gathered[0] == gathered[1]
    # ... and they equal the single-rank computation
    orc = Oracle()
    ref = DecodeLinears(SHAPES["tiny"], device="cpu", rank=0, world=1, m=m)
    launch = _oracle_launch(orc)
    for li in range(ref.n_layers):
        ref.run_block(li, launch=launch)
    ref.run_lm_head(launch=launch)
    exp = {"logits": ref.logits, "down": ref.out_down}
    if gathers == 4:
        exp.update(o=ref.out_o, gate=ref.out_gate, up=ref.out_up, k=ref.out_qkv[1])
    for k, v in exp.items():
        assert gathered[0][k] == v.numpy().view(np.uint16).tobytes(), k


def test_shards_are_contiguous_row_ranges():
    from tinychatengine_amd.linear import Linear_half_int4
    w = torch.randn(64, 256) * 0.02
    full = Linear_half_int4.from_float(w, 128)
    parts = [full.shard(r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p.weight for p in parts]), full.weight)
    assert torch.equal(torch.cat([p.scale for p in parts]), full.scale)
    assert torch.equal(torch.cat([p.zero_point for p in parts]), full.zero_point)
    assert all(p.out_features == 16 and p.in_features == 256 for p in parts)
    with pytest.raises(ValueError):
        full.shard(0, 3)


def test_decode_linears_shards_agree_across_world_sizes():
    from tinychatengine_amd.decode import SHAPES, DecodeLinears
    one = DecodeLinears(SHAPES["tiny"], device="cpu", world=1)
    two = [DecodeLinears(SHAPES["tiny"], device="cpu", rank=r, world=2) for r in range(2)]
    for name in ("o", "gate", "up", "down"):
        assert torch.equal(torch.cat([t.blocks[1][name].weight for t in two]), one.blocks[1][name].weight)
    assert torch.equal(torch.cat([t.lm_head.scale for t in two]), one.lm_head.scale)
    assert two[0].token_bytes() < one.token_bytes() < 2 * two[0].token_bytes()  # weights are halved, activations replicated
    assert len(one.token_launches()) == one.n_layers * 4 + 1 and len(one.token_launches(grouped=False)) == one.n_layers * 7 + 1
