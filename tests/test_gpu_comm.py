"""Multi-GPU path behind the C ABI: tce_w4a16_shard + the peer-write all-gather tce_allgather_f16 (csrc/comm.hip).
The GPU box has ONE device, so the ranks of these tests share it -- as several communicators in one process (streams run
concurrently) and as two PROCESSES that map each other's windows through hipIpcMemHandles, which is the production topology
minus the xGMI hop.  What is checked: the gathered result of column-sharded linears is BIT-IDENTICAL to the unsharded linear,
over many exchanges per slot (epochs, buffer parity), several slots, and graph replay."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_descriptor_arithmetic_needs_no_gpu():
    """tce_w4a16_shard against the row-range arithmetic of Linear_half_int4.shard (CPU: pointers are only computed)."""
    from tinychatengine_amd import capi
    N, K, G = 11008, 4096, 128
    zw = (K // G + 7) // 8
    base_w, base_s, base_z = 0x10000000, 0x20000000, 0x30000000
    full = capi.W4A16Desc(M=1, N=N, K=K, group_size=G, A=0x1000, qweight=base_w, scales=base_s, zeros=base_z, C=0x2000, flags=capi.TCE_W4_ZERO_POINT_IS_8)
    for world in (1, 2, 4, 8):
        for rank in range(world):
            sh = capi.W4A16Desc()
            capi.check(capi.lib().tce_w4a16_shard(C.byref(full), rank, world, C.byref(sh)))
            n = N // world
            assert (sh.N, sh.K, sh.M, sh.flags) == (n, K, 1, full.flags)
            assert sh.qweight == base_w + rank * n * (K // 2) and sh.scales == base_s + rank * n * zw * 8 * 2 and sh.zeros == base_z + rank * n * zw * 4
    sh = capi.W4A16Desc()
    assert capi.lib().tce_w4a16_shard(C.byref(full), 0, 3, C.byref(sh)) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert capi.lib().tce_w4a16_shard(C.byref(full), 8, 8, C.byref(sh)) == capi.TCE_ERR_BAD_ARG


def test_comm_argument_checks_need_no_gpu():
    """What the multi-GPU entry points refuse before they touch the device (CPU: no HIP call is reached)."""
    from tinychatengine_amd import capi
    L = capi.lib()
    h = C.c_void_p()
    assert L.tce_comm_create(2, 2, 4096, 4, C.byref(h)) == capi.TCE_ERR_BAD_ARG      # rank >= world
    assert L.tce_comm_create(0, 9, 4096, 4, C.byref(h)) == capi.TCE_ERR_BAD_ARG      # more than TCE_COMM_MAX_RANKS
    assert L.tce_comm_create(0, 2, 0, 4, C.byref(h)) == capi.TCE_ERR_BAD_ARG         # empty vectors
    assert L.tce_comm_create(0, 2, 4096, 0, C.byref(h)) == capi.TCE_ERR_BAD_ARG      # no slots
    assert L.tce_comm_create(0, 2, 4096, 4, None) == capi.TCE_ERR_BAD_ARG
    for fn in (L.tce_comm_status, L.tce_comm_reset, L.tce_comm_device):
        assert fn(None) == capi.TCE_ERR_BAD_ARG
    assert L.tce_comm_set_timeout_ms(None, 100) == capi.TCE_ERR_BAD_ARG
    assert L.tce_comm_export(None, None) == capi.TCE_ERR_BAD_ARG
    assert L.tce_comm_connect(None, None) == capi.TCE_ERR_BAD_ARG
    assert L.tce_comm_connect_local(None, None) == capi.TCE_ERR_BAD_ARG
    assert L.tce_allgather_f16(None, 0, None, None, 16, None) == capi.TCE_ERR_BAD_ARG


gpu = pytest.mark.gpu


@gpu
def test_comm_timeout_reset_and_device():
    """A lone rank of a 2-rank group: its exchange gives up after the (shortened) bound and flags the communicator; tce_comm_reset re-arms it."""
    from tinychatengine_amd import capi
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    a, b = capi.Comm(0, 2, 1024, slots=2), capi.Comm(1, 2, 1024, slots=2)
    capi.Comm.connect_local([a, b])
    assert a.device == 0 and b.device == 0
    a.set_timeout_ms(50)
    with pytest.raises(capi.TceError):
        a.set_timeout_ms(0)
    src = torch.ones(512, dtype=torch.float16, device=dev)
    dst = torch.zeros(1024, dtype=torch.float16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert a.status() == 0
    a.allgather(0, src.data_ptr(), dst.data_ptr(), 1024, st)   # rank 1 never sends
    torch.cuda.synchronize()
    assert a.status() == 1
    a.reset()
    assert a.status() == 0
    # both ranks now: rank 1 catches up on slot 0 (its first exchange there), then a complete exchange works again
    s1 = torch.cuda.Stream()
    src1 = torch.full((512,), 2.0, dtype=torch.float16, device=dev)
    dst1 = torch.zeros(1024, dtype=torch.float16, device=dev)
    b.allgather(0, src1.data_ptr(), dst1.data_ptr(), 1024, s1.cuda_stream)   # epoch 1 of slot 0 on rank 1: rank 0's flag of epoch 1 is already there
    torch.cuda.synchronize()
    a.allgather(0, src.data_ptr(), dst.data_ptr(), 1024, st)
    b.allgather(0, src1.data_ptr(), dst1.data_ptr(), 1024, s1.cuda_stream)
    torch.cuda.synchronize()
    assert a.status() == 0 and b.status() == 0
    want = torch.cat([torch.ones(512), torch.full((512,), 2.0)]).to(torch.float16).to(dev)
    assert torch.equal(dst, want) and torch.equal(dst1, want)
    a.close(); b.close()


def _sharded_forward(lin, x, rank, world, stream):
    """Rank's slice of lin(x) through the C ABI: tce_w4a16_shard + tce_w4a16_forward on `stream`."""
    from tinychatengine_amd import capi
    full = lin.desc(x, torch.empty(1, lin.out_features, dtype=torch.float16, device=x.device))
    sh = capi.W4A16Desc()
    capi.check(capi.lib().tce_w4a16_shard(C.byref(full), rank, world, C.byref(sh)))
    out = torch.full((1, lin.out_features // world), float("nan"), dtype=torch.float16, device=x.device)
    sh.C = out.data_ptr()
    capi.check(capi.w4a16_forward(sh, stream.cuda_stream))
    return out


@gpu
@pytest.mark.parametrize("world", [2])  # more ranks than that in ONE process end up behind each other on a shared hardware queue
def test_sharded_linears_gathered_by_peer_writes_equal_the_unsharded_linear(world):  # (the wait then times out, by design); see the IPC test
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    shapes = [(4096, 4096), (11008, 4096), (5120, 5120)]
    lins = [Linear_half_int4.from_float(torch.empty(n, k, device=dev).normal_(0, 0.02, generator=g), 128) for n, k in shapes]
    comms = [capi.Comm(r, world, 16384, slots=3) for r in range(world)]
    capi.Comm.connect_local(comms)
    # the two in-process ranks wait for each other INSIDE their kernels: their streams must sit on different hardware queues.  Streams of one priority share a pool of
    # queues by creation order -- which streams the tests that ran earlier in the process took decides whether two of them collide (seen: this test failing with a
    # timed-out gather when another test file ran first) -- streams of different priorities never share a queue
    streams = [torch.cuda.Stream(priority=0 if r % 2 == 0 else -1) for r in range(world)]
    try:
        capi.set_gemv_config(2, 4, 1, 2)  # one geometry for shards and whole: bit-identical rows (tests/test_gpu_w4a16.py)
        for round_ in range(5):  # epochs 1..5 on every slot: both buffer parities, flags from earlier rounds still in the windows
            for si, lin in enumerate(lins):
                x = torch.empty(1, lin.in_features, device=dev).normal_(0, 1, generator=g).to(torch.float16)
                want = lin.forward(x)
                torch.cuda.synchronize()
                fulls = [torch.full((1, lin.out_features), float("nan"), dtype=torch.float16, device=dev) for _ in range(world)]
                keep = []
                for r in range(world):  # every rank: shard GEMV, then the exchange, on its own stream
                    with torch.cuda.stream(streams[r]):
                        part = _sharded_forward(lin, x, r, world, streams[r])
                        keep.append(part)
                        comms[r].allgather(si, part.data_ptr(), fulls[r].data_ptr(), lin.out_features, streams[r].cuda_stream)
                torch.cuda.synchronize()
                for r in range(world):
                    assert comms[r].status() == 0
                    assert torch.equal(fulls[r], want), f"round {round_} linear {si} rank {r}"
    finally:
        capi.set_gemv_config()
        for c in comms:
            c.close()


def _ipc_rank(rank, world, conn, seed, use_graph=False):
    """Child process of the IPC test: its own HIP context on device 0, handles exchanged through the pipe."""
    sys.path.insert(0, REPO)
    import torch as T
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    dev = T.device("cuda:0")
    g = T.Generator(device=dev).manual_seed(seed)  # same seed in both processes: identical weights and inputs
    lin = Linear_half_int4.from_float(T.empty(4096, 4096, device=dev).normal_(0, 0.02, generator=g), 128)
    comm = capi.Comm(rank, world, 4096, slots=2)
    conn.send(comm.export())
    handles = conn.recv()
    comm.connect(handles)
    capi.set_gemv_config(2, 4, 1, 2)
    ok = True
    st = T.cuda.current_stream()
    if use_graph:  # [shard GEMV, exchange] captured once, replayed with changing inputs: the epochs live on the device
        x = T.zeros(1, 4096, dtype=T.float16, device=dev)
        part = T.zeros(1, 4096 // world, dtype=T.float16, device=dev)
        got = T.zeros(1, 4096, dtype=T.float16, device=dev)
        full = lin.desc(x, got)
        sh = capi.W4A16Desc()
        capi.check(capi.lib().tce_w4a16_shard(C.byref(full), rank, world, C.byref(sh)))
        sh.C = part.data_ptr()
        gr = T.cuda.CUDAGraph()
        cs = T.cuda.Stream()
        with T.cuda.stream(cs):
            with T.cuda.graph(gr, stream=cs, capture_error_mode="thread_local"):
                capi.check(capi.w4a16_forward(sh, cs.cuda_stream))
                comm.allgather(0, part.data_ptr(), got.data_ptr(), 4096, cs.cuda_stream)
        for it in range(8):
            x.copy_(T.empty(1, 4096, device=dev).normal_(0, 1, generator=g).to(T.float16))
            want = lin.forward(x)
            T.cuda.synchronize()
            gr.replay()
            T.cuda.synchronize()
            ok = ok and comm.status() == 0 and bool(T.equal(got, want))
    for it in range(0 if use_graph else 8):
        x = T.empty(1, 4096, device=dev).normal_(0, 1, generator=g).to(T.float16)
        want = lin.forward(x)
        full = lin.desc(x, want)
        sh = capi.W4A16Desc()
        capi.check(capi.lib().tce_w4a16_shard(C.byref(full), rank, world, C.byref(sh)))
        part = T.empty(1, 4096 // world, dtype=T.float16, device=dev)
        sh.C = part.data_ptr()
        capi.check(capi.w4a16_forward(sh, st.cuda_stream))
        got = T.full((1, 4096), float("nan"), dtype=T.float16, device=dev)
        comm.allgather(it % 2, part.data_ptr(), got.data_ptr(), 4096, st.cuda_stream)
        T.cuda.synchronize()
        ok = ok and comm.status() == 0 and bool(T.equal(got, want))
    conn.send(ok)
    conn.recv()  # keep the window mapped until the peer is done too
    comm.close()


@gpu
@pytest.mark.parametrize("world,use_graph", [(2, False), (4, False), (2, True)])
def test_processes_exchange_through_ipc_windows(world, use_graph):
    """The production topology on the one-GPU box: one process per rank, each maps the others' windows with hipIpcOpenMemHandle.
    use_graph: every rank captures its [shard GEMV, exchange] pair into a hipGraph and replays it."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(world)]
    procs = [ctx.Process(target=_ipc_rank, args=(r, world, pipes[r][1], 77, use_graph)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        handles = [pipes[r][0].recv() for r in range(world)] if all(pipes[r][0].poll(300) for r in range(world)) else None
        assert handles is not None, "a rank did not come up"
        for r in range(world):
            pipes[r][0].send(handles)
        results = []
        for r in range(world):
            assert pipes[r][0].poll(300), f"rank {r} did not finish"
            results.append(pipes[r][0].recv())
        for r in range(world):
            pipes[r][0].send(True)
        assert all(results), results
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()


@gpu
def test_rccl_side_of_the_communicator_and_the_size_dispatch():
    """Round 4: RCCL behind the C ABI.  One GPU = one RCCL rank (RCCL refuses two ranks on a device), so this exercises the plumbing -- librccl opened on first use,
    the opaque id blob, tce_comm_rccl_init, the dispatch of tce_allgather_f16 by size -- with world = 1, where an all-gather is a copy: a vector far larger than
    the window and its 64 KiB-per-rank regime must take the RCCL path and arrive intact; a small one stays on the peer-write kernel."""
    from tinychatengine_amd import capi
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    comm = capi.Comm(0, 1, 1024, slots=2)
    capi.Comm.connect_local([comm])
    st = torch.cuda.current_stream().cuda_stream
    big = torch.randn(1 << 20, device=dev).to(torch.float16)
    out = torch.zeros_like(big)
    with pytest.raises(capi.TceError):  # no RCCL side yet: beyond the window -> refused, not truncated
        comm.allgather(0, big.data_ptr(), out.data_ptr(), big.numel(), st)
    comm.rccl_init(capi.Comm.rccl_unique_id())
    comm.allgather(0, big.data_ptr(), out.data_ptr(), big.numel(), st)
    torch.cuda.synchronize()
    assert torch.equal(out, big)
    small = torch.randn(512, device=dev).to(torch.float16)
    o2 = torch.zeros_like(small)
    comm.allgather(1, small.data_ptr(), o2.data_ptr(), 512, st)
    torch.cuda.synchronize()
    assert torch.equal(o2, small) and comm.status() == 0
    # rows: M = 300 x 4096 halves = 2.4 MB -> RCCL regime + the interleave kernel (world 1: rows unchanged), into a wider destination
    src = torch.randn(300, 4096, device=dev).to(torch.float16)
    dst = torch.zeros(300, 4096 + 64, dtype=torch.float16, device=dev)
    ws = torch.empty(int(capi.lib().tce_allgather_rows_workspace_bytes(300, 4096)), dtype=torch.uint8, device=dev)
    comm.allgather_rows(0, src.data_ptr(), dst.data_ptr(), 300, 4096, ws.data_ptr(), st, ldd=4096 + 64)
    torch.cuda.synchronize()
    assert torch.equal(dst[:, :4096], src) and torch.count_nonzero(dst[:, 4096:]).item() == 0
    comm.close()


@gpu
def test_allgather_rows_two_ranks_peer_regime():
    """The column-sharded outputs of M > 1 rows (a sharded prompt chunk): two ranks in one process, each with its [M][N/2] block; every rank ends with [M][N],
    rank 0's columns left of rank 1's, over several exchanges on one slot."""
    from tinychatengine_amd import capi
    dev = torch.device("cuda:0")
    world, M, N = 2, 6, 1024
    comms = [capi.Comm(r, world, M * N, slots=2) for r in range(world)]
    capi.Comm.connect_local(comms)
    streams = [torch.cuda.Stream(priority=0 if r % 2 == 0 else -1) for r in range(world)]  # (different priorities: never one hardware queue, see above)
    ws = [torch.empty(M * N, dtype=torch.float16, device=dev) for _ in range(world)]
    try:
        for it in range(4):
            parts = [torch.randn(M, N // world, device=dev).to(torch.float16) for _ in range(world)]
            fulls = [torch.zeros(M, N, dtype=torch.float16, device=dev) for _ in range(world)]
            torch.cuda.synchronize()
            for r in range(world):
                comms[r].allgather_rows(1, parts[r].data_ptr(), fulls[r].data_ptr(), M, N, ws[r].data_ptr(), streams[r].cuda_stream)
            torch.cuda.synchronize()
            want = torch.cat(parts, dim=1)
            for r in range(world):
                assert comms[r].status() == 0
                assert torch.equal(fulls[r], want), (it, r)
    finally:
        for c in comms:
            c.close()


@gpu
def test_allgather_rows_beyond_64k_slices_without_rccl_uses_the_window():
    """M = 64 rows of a 4096-wide tensor are 262144 halves -- 256 KiB slices for two ranks, beyond the peer-write kernel's preferred regime.  With no RCCL
    communicator the exchange used to be refused (TCE_ERR_UNSUPPORTED_KIND); it fits the window, so the peer-write kernel carries it (slower than the links, correct)."""
    from tinychatengine_amd import capi
    dev = torch.device("cuda:0")
    world, M, N = 2, 64, 4096
    comms = [capi.Comm(r, world, M * N, slots=2) for r in range(world)]
    capi.Comm.connect_local(comms)
    streams = [torch.cuda.Stream(priority=0 if r % 2 == 0 else -1) for r in range(world)]  # (different priorities: never one hardware queue, see above)
    ws = [torch.empty(M * N, dtype=torch.float16, device=dev) for _ in range(world)]
    try:
        for it in range(3):
            parts = [torch.randn(M, N // world, device=dev).to(torch.float16) for _ in range(world)]
            fulls = [torch.zeros(M, N, dtype=torch.float16, device=dev) for _ in range(world)]
            torch.cuda.synchronize()
            for r in range(world):
                comms[r].allgather_rows(0, parts[r].data_ptr(), fulls[r].data_ptr(), M, N, ws[r].data_ptr(), streams[r].cuda_stream)
            torch.cuda.synchronize()
            want = torch.cat(parts, dim=1)
            for r in range(world):
                assert comms[r].status() == 0
                assert torch.equal(fulls[r], want), (it, r)
        # and a window that is too small still refuses, by name
        small = [capi.Comm(r, world, 4096, slots=1) for r in range(world)]
        capi.Comm.connect_local(small)
        with pytest.raises(capi.TceError):
            small[0].allgather_rows(0, parts[0].data_ptr(), fulls[0].data_ptr(), M, N, ws[0].data_ptr(), streams[0].cuda_stream)
        capi.lib().tce_reset_last_error()
        for c in small:
            c.close()
    finally:
        for c in comms:
            c.close()


@gpu
def test_one_process_two_devices_through_connect_local():
    """the first multi-GPU lease must not be the first test of the cross-device path.  One process, one communicator per DEVICE
    (tce_comm_connect_local maps the peers' windows without IPC handles), a column-sharded linear per device, the gather across the link.  Skips on a one-GPU box --
    every gpurun box so far."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    if int(capi.lib().tce_device_count()) < 2 or torch.cuda.device_count() < 2:
        pytest.skip("needs two devices in one process (tce_device_count() < 2)")
    world = 2
    devs = [torch.device(f"cuda:{r}") for r in range(world)]
    g = torch.Generator(device=devs[0]).manual_seed(15)
    w = torch.empty(4096, 4096, device=devs[0]).normal_(0, 0.02, generator=g)
    x0 = torch.empty(1, 4096, device=devs[0]).normal_(0, 1, generator=g).to(torch.float16)
    lin0 = Linear_half_int4.from_float(w, 128)
    want = lin0.forward(x0)
    torch.cuda.synchronize()
    comms = []
    for r in range(world):
        with torch.cuda.device(devs[r]):
            comms.append(capi.Comm(r, world, 16384, slots=2))
    capi.Comm.connect_local(comms)
    try:
        fulls, keep = [], []
        for r in range(world):
            with torch.cuda.device(devs[r]):
                assert comms[r].device() == r
                lin = Linear_half_int4(lin0.weight.to(devs[r]), lin0.scale.to(devs[r]), lin0.zero_point.to(devs[r]), 128)
                x = x0.to(devs[r])
                st = torch.cuda.current_stream(devs[r])
                part = _sharded_forward(lin, x, r, world, st)
                full = torch.full((1, 4096), float("nan"), dtype=torch.float16, device=devs[r])
                comms[r].allgather(0, part.data_ptr(), full.data_ptr(), 4096, st.cuda_stream)
                fulls.append(full)
                keep.append((lin, x, part))
        for r in range(world):
            torch.cuda.synchronize(devs[r])
        for r in range(world):
            assert comms[r].status() == 0
            assert torch.equal(fulls[r].cpu(), want.cpu()), r
    finally:
        for c in comms:
            c.close()


# ---- round 6: the exchange INSIDE the block's one launch (tce_w4a16_forward_independent_gather) ----

def _block_shards(T, capi, dev, g, world, hidden=1024, ffn=3584):
    """The linears of a small sharded block: (full linears, per rank [shards]) -- q, o, gate of K = hidden, down of K = ffn (the gathered one, last)."""
    from tinychatengine_amd.linear import Linear_half_int4
    fulls = [Linear_half_int4.from_float(T.empty(n, k, device=dev).normal_(0, 0.02, generator=g), 128).prepack()
             for n, k in ((hidden, hidden), (hidden, hidden), (ffn, hidden), (hidden, ffn))]
    return fulls, [[f.shard(r, world).prepack() for f in fulls] for r in range(world)]


@gpu
def test_exchange_inside_the_block_launch_two_ranks_in_one_process():
    """Two ranks in one process (streams of different priorities): per round every rank issues ONE call -- its four shards, the last one's slice exchanged inside the
    launch -- and ends with the complete down output, bit-identical to the unsharded linear, the other shards bit-identical to the rows of theirs.  Rounds alternate with
    the plain gather kernel on the SAME slot (one protocol), cover both buffer parities, and a residual-add epilogue on the gathered linear."""
    from tinychatengine_amd import capi
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    world = 2
    g = torch.Generator(device=dev).manual_seed(11)
    fulls, shards = _block_shards(torch, capi, dev, g, world)
    comms = [capi.Comm(r, world, 4096, slots=2) for r in range(world)]
    capi.Comm.connect_local(comms)
    streams = [torch.cuda.Stream(priority=0 if r % 2 == 0 else -1) for r in range(world)]
    try:
        for round_ in range(6):
            xs = [torch.empty(1, f.in_features, device=dev).normal_(0, 1, generator=g).to(torch.float16) for f in fulls]
            add = round_ % 3 == 2  # the gathered linear adds to its slice buffer (o_proj / down_proj + residual): the exchanged values are the sums
            resid = torch.empty(1, fulls[3].out_features, device=dev).normal_(0, 1, generator=g).to(torch.float16)
            wants = [f.forward(x) for f, x in zip(fulls, xs)]
            if add:
                wants[3] = fulls[3].forward_add(xs[3], resid.clone())
            torch.cuda.synchronize()
            outs = [[torch.full((1, s.out_features), float("nan"), dtype=torch.float16, device=dev) for s in shards[r]] for r in range(world)]
            full_down = [torch.full((1, fulls[3].out_features), float("nan"), dtype=torch.float16, device=dev) for _ in range(world)]
            n_launches = []
            for r in range(world):
                n_loc = shards[r][3].out_features
                if add:
                    outs[r][3].copy_(resid[:, r * n_loc:(r + 1) * n_loc])
                with torch.cuda.stream(streams[r]):
                    descs = [s.desc(x, o, flags=(capi.TCE_W4_ADD_TO_C if add and i == 3 else 0)) for i, (s, x, o) in enumerate(zip(shards[r], xs, outs[r]))]
                    if round_ % 3 == 1:  # the two-call form on the same slot
                        capi.w4a16_forward_independent(descs, streams[r].cuda_stream)
                        comms[r].allgather(0, outs[r][3].data_ptr(), full_down[r].data_ptr(), fulls[3].out_features, streams[r].cuda_stream)
                    else:
                        n_launches.append(comms[r].forward_independent_gather(descs, 3, 0, full_down[r].data_ptr(), streams[r].cuda_stream))
            torch.cuda.synchronize()
            assert all(n == 1 for n in n_launches), n_launches
            for r in range(world):
                assert comms[r].status() == 0
                assert torch.equal(full_down[r].view(torch.int16), wants[3].view(torch.int16)), f"round {round_} rank {r}: gathered vector"
                for i in range(4):
                    n_loc = shards[r][i].out_features
                    assert torch.equal(outs[r][i].view(torch.int16), wants[i][:, r * n_loc:(r + 1) * n_loc].view(torch.int16)), f"round {round_} rank {r} linear {i}"
    finally:
        for c in comms:
            c.close()


@gpu
def test_exchange_inside_the_launch_single_rank_and_fallbacks():
    """world = 1: the exchange is a copy through the window.  A vector beyond 16384 halves, or a gathered linear the one-launch form does not take, goes through
    tce_w4a16_forward_independent + the gather kernel: same result, the launch count says so.  Bad arguments are refused by name."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    st = torch.cuda.current_stream().cuda_stream
    comm = capi.Comm(0, 1, 32768, slots=2)
    capi.Comm.connect_local([comm])
    try:
        for n, k, prepack, expect in ((256, 1024, True, 1), (20480, 1024, True, 2), (256, 1024, False, 2)):
            lin = Linear_half_int4.from_float(torch.empty(n, k, device=dev).normal_(0, 0.02, generator=g), 128)
            if prepack:
                lin.prepack()
            x = torch.empty(1, k, device=dev).normal_(0, 1, generator=g).to(torch.float16)
            want = lin.forward(x)
            for it in range(3):
                part = torch.full((1, n), float("nan"), dtype=torch.float16, device=dev)
                full = torch.full((1, n), float("nan"), dtype=torch.float16, device=dev)
                assert comm.forward_independent_gather([lin.desc(x, part)], 0, 1, full.data_ptr(), st) == expect
                torch.cuda.synchronize()
                assert comm.status() == 0 and torch.equal(part, want) and torch.equal(full, want), (n, k, prepack, it)
        with pytest.raises(capi.TceError):
            comm.forward_independent_gather([lin.desc(x, part)], 1, 0, full.data_ptr(), st)  # gathered: not an index of the call
        with pytest.raises(capi.TceError):
            comm.forward_independent_gather([lin.desc(x, part)], 0, 5, full.data_ptr(), st)  # slot out of range
        capi.lib().tce_reset_last_error()
    finally:
        comm.close()


def _ipc_rank_fused(rank, world, conn, seed, use_graph):
    """Child process: a rank of the sharded block, the block output exchanged inside the block's one launch."""
    sys.path.insert(0, REPO)
    import torch as T
    from tinychatengine_amd import capi
    dev = T.device("cuda:0")
    g = T.Generator(device=dev).manual_seed(seed)  # same seed everywhere: identical weights and inputs
    fulls, shards = _block_shards(T, capi, dev, g, world)
    mine = shards[rank]
    comm = capi.Comm(rank, world, 4096, slots=2)
    conn.send(comm.export())
    comm.connect(conn.recv())
    ok = True
    xs = [T.zeros(1, f.in_features, dtype=T.float16, device=dev) for f in fulls]
    outs = [T.zeros(1, s.out_features, dtype=T.float16, device=dev) for s in mine]
    got = T.zeros(1, fulls[3].out_features, dtype=T.float16, device=dev)
    descs = [s.desc(x, o) for s, x, o in zip(mine, xs, outs)]
    cs = T.cuda.Stream()
    gr = None
    if use_graph:
        gr = T.cuda.CUDAGraph()
        with T.cuda.stream(cs):
            with T.cuda.graph(gr, stream=cs, capture_error_mode="thread_local"):
                comm.forward_independent_gather(descs, 3, 1, got.data_ptr(), cs.cuda_stream)
    for it in range(8):
        for x in xs:
            x.copy_(T.empty(1, x.shape[1], device=dev).normal_(0, 1, generator=g).to(T.float16))
        want = fulls[3].forward(xs[3])
        got.fill_(float("nan"))
        T.cuda.synchronize()
        if gr is not None:
            gr.replay()
        else:
            ok = ok and comm.forward_independent_gather(descs, 3, 1, got.data_ptr(), cs.cuda_stream) == 1
        T.cuda.synchronize()
        ok = ok and comm.status() == 0 and bool(T.equal(got.view(T.int16), want.view(T.int16)))
    conn.send(ok)
    conn.recv()
    comm.close()


@gpu
@pytest.mark.parametrize("world,use_graph", [(2, False), (4, False), (2, True)])
def test_processes_exchange_inside_the_block_launch(world, use_graph):
    """One process per rank on the one GPU (IPC windows), the block's one launch carrying the exchange: every rank's complete vector bit-identical to the unsharded linear,
    eagerly and as a replayed hipGraph (the epoch and the arrival counter live on the device)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(world)]
    procs = [ctx.Process(target=_ipc_rank_fused, args=(r, world, pipes[r][1], 91, use_graph)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        handles = [pipes[r][0].recv() for r in range(world)] if all(pipes[r][0].poll(300) for r in range(world)) else None
        assert handles is not None, "a rank did not come up"
        for r in range(world):
            pipes[r][0].send(handles)
        results = []
        for r in range(world):
            assert pipes[r][0].poll(300), f"rank {r} did not finish"
            results.append(pipes[r][0].recv())
        for r in range(world):
            pipes[r][0].send(True)
        assert all(results), results
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
