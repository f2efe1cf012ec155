"""GPU tests of the token kernel on the int8-contraction body (csrc/w4a16_gemv_i8_token.hip, round 6): TCE_PLAN_TAGGED over linears that carry a packed copy.

What must hold: the plan's outputs are the stream-ordered plan's, bit for bit (same arithmetic, another distribution of the work), on every replay, with the input changed
between replays (a stage that read its activations early would see the previous replay's values), across the 16-bit tag's wrap, with grouped launches, slices, both fused
epilogues, ragged k-chunks and ragged last tiles; the chain's first link is held to the oracle; lists the kernel does not take in full are taken as far as they go.
"""
import numpy as np
import pytest

from conftest import w4a16_close

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU (they must not silently pass without it)"
    from tinychatengine_amd import capi
    capi.lib()
    capi.set_gemv_config()
    return torch.device("cuda:0")


def _mk(dev, g, n, k, G=128):
    from tinychatengine_amd.linear import Linear_half_int4
    return Linear_half_int4.from_float(torch.empty(n, k, device=dev).normal_(0, 1.0 / np.sqrt(k), generator=g), G).prepack()


def _chain(dev, dims, seed):
    """dims[0] = K of the first linear; every further entry N or (N, K_next): the linear's width and how much of its output the next linear reads (a prefix slice)"""
    g = torch.Generator(device=dev).manual_seed(seed)
    widths = [d if isinstance(d, int) else d[0] for d in dims]
    reads = [d if isinstance(d, int) else d[1] for d in dims]
    bufs = [torch.zeros((1, d), dtype=torch.float16, device=dev) for d in widths]
    lins = [_mk(dev, g, widths[i + 1], reads[i]) for i in range(len(dims) - 1)]
    launches = [[l.desc(bufs[i][:, :reads[i]], bufs[i + 1])] for i, l in enumerate(lins)]
    return launches, bufs, lins


@pytest.mark.parametrize("dims", [[4096, 11008, 4096, 1024, 4096, 256, 2048], [4096, 14336, 4096, 6144, 5120, 13824, 5120], [1024, (264, 256), (1416, 1408), (40, 0)]])
def test_int8_token_plan_keeps_the_data_dependences(dev, oracle, dims):
    """A chain of dependent linears (K = 11008 / 13824: ragged last k-chunks; 264 / 1416 / 40 rows: ragged last tiles, read as 256- / 1408-wide prefixes; 14 and 11 chunks per tile) as one kernel."""
    from tinychatengine_amd import capi
    launches, bufs, lins = _chain(dev, dims, seed=11)
    plain = capi.Plan(launches)
    tok = capi.Plan(launches, tagged=True)
    assert tok.kind == 4 and tok.tagged and not plain.chained, tok.kind
    assert tok.geometry()["rows"] == len(launches)  # every launch inside the kernel
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(3)
    for it in range(40):
        x0 = torch.from_numpy(rng.standard_normal((1, bufs[0].shape[1])).astype(np.float16)).to(dev)
        results = []
        for plan in (plain, tok):
            for b in bufs[1:]:
                b.fill_(float("nan"))
            bufs[0].copy_(x0)
            plan.launch(s)
            plan.status()
            results.append([b.cpu().numpy().copy() for b in bufs[1:]])
        for li, (a, b) in enumerate(zip(*results)):
            assert not np.isnan(b.astype(np.float32)).any(), f"replay {it}: launch {li} left outputs unwritten"
            assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"replay {it}: launch {li} differs from the stream-ordered plan ({(a.view(np.uint16) != b.view(np.uint16)).sum()} values)"
        if it == 0:  # the chain's first link against the oracle
            l0 = lins[0]
            ref32, _ = oracle.w4a16_gemv_q4_6(x0.cpu().numpy(), l0.weight.cpu().numpy().view(np.uint32), l0.scale.cpu().numpy(), l0.zero_point.cpu().numpy().view(np.uint32), 1, bufs[1].shape[1], bufs[0].shape[1], 128)
            ok, worst = w4a16_close(results[1][0], ref32)
            assert ok, f"worst |err|/tol = {worst:.3f}"
    # back-to-back replays without a host sync in between
    x0 = torch.from_numpy(rng.standard_normal((1, bufs[0].shape[1])).astype(np.float16)).to(dev)
    bufs[0].copy_(x0)
    plain.launch(s)
    torch.cuda.synchronize()
    want = bufs[-1].cpu().numpy().copy()
    for _ in range(200):
        tok.launch(s)
    tok.status()
    assert np.array_equal(bufs[-1].cpu().numpy().view(np.uint16), want.view(np.uint16))
    plain.close()
    tok.close()


def test_int8_token_plan_decoder_block_dataflow(dev, oracle):
    """Two decoder blocks wired the way the linears feed each other: grouped launches (q / k / v; the interleaved gate + up), a consumer that reads a SLICE of a producer's
    output, the SiLU-mul pair epilogue, the residual-add epilogue, an input from outside the plan, buffers reused from block to block, and 70 000 back-to-back replays (the
    16-bit tag wraps at 65 535).  Bit-identical to the stream-ordered plan throughout."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    h, f = 1024, 2816
    g = torch.Generator(device=dev).manual_seed(5)
    e = lambda n: torch.zeros((1, n), dtype=torch.float16, device=dev)
    x, qkv, o_out, act = e(h), e(3 * h), e(h), e(f)
    launches, keep = [], []
    mkf = lambda n, k: Linear_half_int4.from_float(torch.empty(n, k, device=dev).normal_(0, 1.0 / np.sqrt(k), generator=g), 128)
    for blk in range(2):
        q = _mk(dev, g, h, h); k = _mk(dev, g, h, h); v = _mk(dev, g, h, h); o = _mk(dev, g, h, h); down = _mk(dev, g, h, f)
        gu = Linear_half_int4.interleave(mkf(f, h), mkf(f, h)).prepack()
        keep += [q, k, v, o, down, gu]
        launches.append([q.desc(x, qkv[:, :h]), k.desc(x, qkv[:, h:2 * h]), v.desc(x, qkv[:, 2 * h:])])
        launches.append([o.desc(qkv[:, :h], o_out)])
        launches.append([gu.desc(o_out, act, flags=capi.TCE_W4_SILU_MUL_PAIRS)])
        launches.append([down.desc(act, x, flags=capi.TCE_W4_ADD_TO_C)])
    plain = capi.Plan(launches)
    tok = capi.Plan(launches, tagged=True)
    assert tok.kind == 4 and not plain.chained
    s = torch.cuda.current_stream().cuda_stream
    bufs = [qkv, o_out, act, x]
    for it in range(25):
        x0 = torch.empty((1, h), device=dev).normal_(0, 1, generator=g).to(torch.float16)
        results = []
        for plan in (plain, tok):
            for b in bufs:
                b.fill_(float("nan"))
            x.copy_(x0)
            plan.launch(s)
            plan.status()
            results.append([b.cpu().numpy().copy() for b in bufs])
        for name, a, b in zip(("qkv", "o", "act", "x"), *results):
            assert not np.isnan(b.astype(np.float32)).any(), f"replay {it}: {name} has unwritten / poisoned values"
            assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"replay {it}: {name} differs from the stream-ordered plan"
    x0 = torch.empty((1, h), device=dev).normal_(0, 1, generator=g).to(torch.float16)
    for r in range(70):
        for _ in range(999):
            tok.launch(s)
        x.copy_(x0)
        tok.launch(s)
    tok.status()
    got = x.cpu().numpy().copy()
    x.copy_(x0)
    plain.launch(s)
    torch.cuda.synchronize()
    assert np.array_equal(got.view(np.uint16), x.cpu().numpy().view(np.uint16))
    plain.close()
    tok.close()


def test_int8_token_plan_takes_a_prefix(dev, oracle):
    """A list whose tail the kernel does not take (a linear with general zero points; a linear without a packed copy): the prefix runs inside the kernel, the rest follows it
    as ordinary launches of the same graph -- the outputs stay the stream-ordered plan's."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    g = torch.Generator(device=dev).manual_seed(9)
    dims = [2048, 4096, 2048, 1024]
    launches, bufs, lins = _chain(dev, dims, seed=21)
    tail = Linear_half_int4.from_float(torch.empty(512, 1024, device=dev).normal_(0, 1.0 / 32, generator=g), 128)  # no packed copy
    out = torch.zeros((1, 512), dtype=torch.float16, device=dev)
    launches.append([tail.desc(bufs[-1], out)])
    plain = capi.Plan(launches)
    tok = capi.Plan(launches, tagged=True)
    assert tok.kind == 4 and tok.geometry()["rows"] == 3
    s = torch.cuda.current_stream().cuda_stream
    for it in range(10):
        x0 = torch.empty((1, dims[0]), device=dev).normal_(0, 1, generator=g).to(torch.float16)
        res = []
        for plan in (plain, tok):
            out.fill_(float("nan"))
            bufs[0].copy_(x0)
            plan.launch(s)
            plan.status()
            res.append(out.cpu().numpy().copy())
        assert np.array_equal(res[0].view(np.uint16), res[1].view(np.uint16)), f"replay {it}"
    plain.close()
    tok.close()


def test_int8_token_plan_is_not_taken_where_it_does_not_apply(dev, oracle):
    """No packed copies: round 2's token kernel (kind 2) as before; forced off (tce_w4a16_set_debug_mode(7701)): likewise; a hazard protected by position only: stream-ordered."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    g = torch.Generator(device=dev).manual_seed(13)
    dims = [1024, 2048, 1024]
    bufs = [torch.zeros((1, d), dtype=torch.float16, device=dev) for d in dims]
    lins = [Linear_half_int4.from_float(torch.empty(dims[i + 1], dims[i], device=dev).normal_(0, 1.0 / 32, generator=g), 128) for i in range(2)]
    p = capi.Plan([[l.desc(bufs[i], bufs[i + 1])] for i, l in enumerate(lins)], tagged=True)
    assert p.kind == 2
    p.close()
    for l in lins:
        l.prepack()
    launches = [[l.desc(bufs[i], bufs[i + 1])] for i, l in enumerate(lins)]
    L = capi.lib()
    try:
        capi.check(L.tce_w4a16_set_debug_mode(7701))
        p = capi.Plan(launches, tagged=True)
        assert p.kind == 2
        p.close()
    finally:
        capi.check(L.tce_w4a16_set_debug_mode(7700))
    p = capi.Plan(launches, tagged=True)
    assert p.kind == 4
    p.close()
    # launch 1 overwrites launch 0's INPUT without consuming launch 0's output: only stream order protects launch 0's read
    other = torch.zeros((1, 1024), dtype=torch.float16, device=dev)
    l2 = _mk(dev, g, 1024, 1024)
    hazard = [[lins[0].desc(bufs[0], bufs[1])], [l2.desc(other, bufs[0])]]
    p = capi.Plan(hazard, tagged=True)
    assert p.kind == 0
    p.close()
