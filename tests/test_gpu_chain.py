"""GPU tests of the persistent GEMV kernel (w4a16_gemv_stream.hip) and of token plans (TCE_PLAN_TAGGED: one token
kernel walks the launch list, consumers poll tagged output words instead of waiting at kernel boundaries).

Parity: the persistent kernel against the oracle on the same inputs (every geometry it compiles), and -- because a
row's arithmetic does not depend on which wave computes it -- bit-identical to the workgroup-per-row-block kernel.
Ordering: a chained plan whose launch i+1 consumes launch i's output must give exactly what the stream-ordered plan
gives, on every replay, with the input changed between replays (a launch that read its activations early would see
the previous replay's values).
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


def _quant(oracle, N, K, G, seed, random_zeros):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    qw, sc, zp, _, _ = oracle.quantize_q4_6(w, G)
    if random_zeros:
        nib = rng.integers(0, 16, (N, zp.shape[1] * 8), dtype=np.uint32)
        zp = (nib.reshape(N, -1, 8) << (np.arange(8, dtype=np.uint32) * 4)).sum(axis=2).astype(np.uint32)
    return qw, sc, zp


def _dev(dev, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in arrs]


def _group(dev, oracle, Ns, K, G, seed, random_zeros=False, z8_flag=False):
    from tinychatengine_amd import capi
    rng = np.random.default_rng(seed + 99)
    a = rng.standard_normal((1, K)).astype(np.float16)
    (ta,) = _dev(dev, a)
    keep, descs, refs, outs = [ta], [], [], []
    for i, N in enumerate(Ns):
        qw, sc, zp = _quant(oracle, N, K, G, seed + i, random_zeros)
        ref32, _ = oracle.w4a16_gemv_q4_6(a, qw, sc, zp, 1, N, K, G)
        tq, ts, tz = _dev(dev, qw.view(np.int32), sc.view(np.float16), zp.view(np.int32))
        out = torch.full((1, N), float("nan"), dtype=torch.float16, device=dev)
        keep += [tq, ts, tz]
        outs.append(out)
        refs.append(ref32)
        descs.append(capi.W4A16Desc(M=1, N=N, K=K, group_size=G, A=ta.data_ptr(), qweight=tq.data_ptr(), scales=ts.data_ptr(),
                                    zeros=tz.data_ptr(), C=out.data_ptr(), flags=capi.TCE_W4_ZERO_POINT_IS_8 if z8_flag else 0))
    return descs, outs, refs, keep


def _launch(descs):
    from tinychatengine_amd import capi
    arr = (capi.W4A16Desc * len(descs))(*descs)
    import ctypes as C
    capi.check(capi.lib().tce_w4a16_forward_group(arr, len(descs), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()


STREAM_CASES = [
    # Ns, K, G, random zero points
    ([256], 4096, 128, False), ([100, 37], 4096, 128, True), ([512, 512], 1408, 128, False), ([64, 64, 200], 2048, 64, True),
    ([4096], 4096, 128, False), ([333], 11008, 128, True), ([48, 16, 16, 16], 1024, 32, True), ([2050], 14336, 128, False),
    ([1], 32, 32, True), ([11008, 11008], 4096, 128, False),
]
GEOMETRIES = [(4, 8, 2), (2, 8, 2), (2, 8, 3), (1, 8, 3), (1, 5, 2), (24, 8, 2), (22, 7, 3), (4, 16, 2)]  # (10*bpc + rows, waves, depth)


@pytest.mark.parametrize("Ns,K,G,rz", STREAM_CASES)
def test_persistent_gemv_matches_oracle_and_row_block_kernel(dev, oracle, Ns, K, G, rz):
    from tinychatengine_amd import capi
    descs, outs, refs, keep = _group(dev, oracle, Ns, K, G, seed=sum(Ns) + K, random_zeros=rz)
    # the row-block kernel without a K split sums in the same order as the persistent one: bit-identical outputs
    capi.set_gemv_config(2, 4, 1, 1)
    _launch(descs)
    base = [o.cpu().numpy().copy() for o in outs]
    try:
        for geo in GEOMETRIES[: 8 if sum(Ns) < 8192 else 3]:
            for o in outs:
                o.fill_(float("nan"))
            capi.set_gemv_config(geo[0], geo[1], 0, geo[2])
            _launch(descs)
            for o, ref32, b, N in zip(outs, refs, base, Ns):
                got = o.cpu().numpy()
                assert not np.isnan(got.astype(np.float32)).any(), f"{Ns} K={K} geo={geo}: unwritten outputs"
                ok, worst = w4a16_close(got, ref32)
                assert ok, f"{Ns} K={K} g{G} geo={geo}: worst |err|/tol = {worst:.3f}"
                assert np.array_equal(got.view(np.uint16), b.view(np.uint16)), f"{Ns} K={K} geo={geo}: differs from the row-block kernel"
    finally:
        capi.set_gemv_config()


def test_persistent_gemv_zero_point_8_flag(dev, oracle):
    from tinychatengine_amd import capi
    descs, outs, refs, keep = _group(dev, oracle, [700, 300], 4096, 128, seed=5, z8_flag=True)
    try:
        capi.set_gemv_config(2, 8, 0, 3)
        _launch(descs)
        for o, ref32 in zip(outs, refs):
            ok, worst = w4a16_close(o.cpu().numpy(), ref32)
            assert ok, f"worst |err|/tol = {worst:.3f}"
    finally:
        capi.set_gemv_config()


def _mlp_chain(dev, oracle, dims, G, seed):
    """Linears dims[0]->dims[1]->...: launch i reads launch i-1's output buffer.  Returns (launch lists, x0, last out, weights)."""
    from tinychatengine_amd import capi
    bufs = [torch.zeros((1, d), dtype=torch.float16, device=dev) for d in dims]
    launches, keep, wts = [], [], []
    for i in range(len(dims) - 1):
        K, N = dims[i], dims[i + 1]
        rng = np.random.default_rng(seed + i)
        w = (rng.standard_normal((N, K)) * (1.0 / np.sqrt(K))).astype(np.float32)  # keeps the activations O(1) down the chain
        qw, sc, zp, _, _ = oracle.quantize_q4_6(w, G)
        tq, ts, tz = _dev(dev, qw.view(np.int32), sc.view(np.float16), zp.view(np.int32))
        keep += [tq, ts, tz]
        wts.append((qw, sc, zp, N, K))
        launches.append([capi.W4A16Desc(M=1, N=N, K=K, group_size=G, A=bufs[i].data_ptr(), qweight=tq.data_ptr(), scales=ts.data_ptr(),
                                        zeros=tz.data_ptr(), C=bufs[i + 1].data_ptr(), flags=capi.TCE_W4_ZERO_POINT_IS_8)])
    return launches, bufs, keep, wts


@pytest.mark.parametrize("tagged", [False, True, "overlapped", "overlapped-ring-depth-2"])
def test_chained_plan_keeps_the_data_dependences(dev, oracle, tagged):
    """Both spellings of the flag (TCE_PLAN_CHAINED is a synonym of TCE_PLAN_TAGGED since round 2), and TCE_PLAN_OVERLAPPED (round 3:
    one kernel per launch on alternating graph branches, ordered by the same tagged words)."""
    from tinychatengine_amd import capi
    dims = [4096, 11008, 4096, 1024, 4096, 256, 2048]
    launches, bufs, keep, wts = _mlp_chain(dev, oracle, dims, 128, seed=11)
    capi.set_gemv_config(2, 8, 0, 2)  # the stream-ordered plan on the persistent kernel too: outputs must be bit-identical
    plain = capi.Plan(launches)
    capi.set_gemv_config()
    if isinstance(tagged, str):
        capi.check(capi.lib().tce_w4a16_set_debug_mode(50002 if tagged.endswith("depth-2") else 50000))
        chained = capi.Plan(launches, overlapped=True)
        capi.check(capi.lib().tce_w4a16_set_debug_mode(50000))
        assert chained.overlapped and not plain.chained
    else:
        chained = capi.Plan(launches, chained=not tagged, tagged=tagged)
        assert chained.tagged and not plain.chained
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(3)
    for it in range(40):
        x0 = torch.from_numpy(rng.standard_normal((1, dims[0])).astype(np.float16)).to(dev)
        results = []
        for plan in (plain, chained):
            for b in bufs[1:]:
                b.fill_(float("nan"))
            bufs[0].copy_(x0)
            plan.launch(s)
            plan.status()
            results.append([b.cpu().numpy().copy() for b in bufs[1:]])
        for li, (a, b) in enumerate(zip(*results)):
            assert not np.isnan(b.astype(np.float32)).any(), f"replay {it}: launch {li} left outputs unwritten"
            assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"replay {it}: launch {li} differs from the stream-ordered plan"
        if it == 0:  # and the chain's first link against the oracle
            qw, sc, zp, N, K = wts[0]
            ref32, _ = oracle.w4a16_gemv_q4_6(x0.cpu().numpy(), qw, sc, zp, 1, N, K, 128)
            ok, worst = w4a16_close(results[1][0], ref32)
            assert ok, f"worst |err|/tol = {worst:.3f}"
    # back-to-back replays without a host sync in between
    x0 = torch.from_numpy(rng.standard_normal((1, dims[0])).astype(np.float16)).to(dev)
    bufs[0].copy_(x0)
    plain.launch(s)
    torch.cuda.synchronize()
    want = bufs[-1].cpu().numpy().copy()
    for _ in range(50):
        chained.launch(s)
    chained.status()
    assert np.array_equal(bufs[-1].cpu().numpy().view(np.uint16), want.view(np.uint16))
    plain.close()
    chained.close()


@pytest.mark.parametrize("kind", ["tagged", "overlapped"])
def test_tagged_plan_decoder_block_dataflow(dev, oracle, kind):
    """A tagged (token kernel) / overlapped (one kernel per launch, alternating branches) plan over the launch shapes of two decoder blocks, wired the way the linears feed each other (the attention between
    qkv and o is not part of the path: o reads the q slice): grouped launches (q / k / v as three linears; gate + up), a consumer
    that reads a SLICE of a producer's output, the SiLU-mul pair epilogue, the residual-add epilogue, an input that comes from
    outside the plan, buffers reused from block to block (the latest writer is the producer), and 70 000 back-to-back replays
    (the 16-bit token tag wraps at 65 535).  Bit-identical to the stream-ordered plan throughout."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    h, f, G = 1024, 2816, 128
    g = torch.Generator(device=dev).manual_seed(5)
    mk = lambda n, k: Linear_half_int4.from_float(torch.empty(n, k, device=dev).normal_(0, 1.0 / np.sqrt(k), generator=g), G)
    e = lambda n: torch.zeros((1, n), dtype=torch.float16, device=dev)
    x, qkv, o_out, act = e(h), e(3 * h), e(h), e(f)
    launches, keep = [], []
    for blk in range(2):
        q, k, v, o, down = mk(h, h), mk(h, h), mk(h, h), mk(h, h), mk(h, f)
        gu = Linear_half_int4.interleave(mk(f, h), mk(f, h))
        keep += [q, k, v, o, down, gu]
        launches.append([q.desc(x, qkv[:, :h]), k.desc(x, qkv[:, h:2 * h]), v.desc(x, qkv[:, 2 * h:])])  # block 0: x from outside the plan
        launches.append([o.desc(qkv[:, :h], o_out)])                                                   # a slice of launch 0's first output
        launches.append([gu.desc(o_out, act, flags=capi.TCE_W4_SILU_MUL_PAIRS)])
        launches.append([down.desc(act, x, flags=capi.TCE_W4_ADD_TO_C)])                               # x += down(act): next block's input
    capi.set_gemv_config(2, 8, 0, 2)
    plain = capi.Plan(launches)
    capi.set_gemv_config()
    tagged = capi.Plan(launches, tagged=kind == "tagged", overlapped=kind == "overlapped")
    assert (tagged.tagged if kind == "tagged" else tagged.overlapped) and not plain.chained
    s = torch.cuda.current_stream().cuda_stream
    bufs = [qkv, o_out, act, x]
    for it in range(25):
        x0 = torch.empty((1, h), device=dev).normal_(0, 1, generator=g).to(torch.float16)
        results = []
        for plan in (plain, tagged):
            for b in bufs:
                b.fill_(float("nan"))
            x.copy_(x0)
            plan.launch(s)
            plan.status()
            results.append([b.cpu().numpy().copy() for b in bufs])
        for name, a, b in zip(("qkv", "o", "act", "x"), *results):
            assert not np.isnan(b.astype(np.float32)).any(), f"replay {it}: {name} has unwritten / poisoned values"
            assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"replay {it}: {name} differs from the stream-ordered plan"
    # tag wrap-around: x is an input AND an output, so 70 000 replays are a 140 000-block-deep recurrence; restart it now and then
    x0 = torch.empty((1, h), device=dev).normal_(0, 1, generator=g).to(torch.float16)
    for r in range(70):
        for _ in range(999):
            tagged.launch(s)
        x.copy_(x0)
        tagged.launch(s)
    tagged.status()
    got = x.cpu().numpy().copy()
    x.copy_(x0)
    plain.launch(s)
    torch.cuda.synchronize()
    assert np.array_equal(got.view(np.uint16), x.cpu().numpy().view(np.uint16))
    plain.close()
    tagged.close()


def test_tagged_plan_with_the_fused_rmsnorm_prologue(dev, oracle):
    """A consumer whose activations go through the fused RMSNorm prologue (two passes over x: both from the polled copy in LDS)."""
    from tinychatengine_amd import capi
    launches, bufs, keep, wts = _mlp_chain(dev, oracle, [1024, 512, 11008, 256], 128, seed=4)
    gammas = [(1 + 0.1 * torch.randn(d, device=dev)).float() for d in (512, 11008)]
    for l, g in zip(launches[1:], gammas):
        l[0].rmsnorm_gamma = g.data_ptr()
        l[0].rmsnorm_eps = 1e-6
    capi.set_gemv_config(2, 8, 0, 2)
    plain = capi.Plan(launches)
    capi.set_gemv_config()
    plan = capi.Plan(launches, tagged=True)
    assert plan.tagged and not plain.chained
    s = torch.cuda.current_stream().cuda_stream
    for it in range(10):
        x0 = torch.randn((1, 1024), device=dev).to(torch.float16)
        res = []
        for p in (plain, plan):
            for b in bufs[1:]:
                b.fill_(float("nan"))
            bufs[0].copy_(x0)
            p.launch(s)
            p.status()
            res.append([b.cpu().numpy().copy() for b in bufs[1:]])
        for a, b in zip(*res):
            assert not np.isnan(b.astype(np.float32)).any()
            assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), f"replay {it}"
    plain.close()
    plan.close()


def test_tagged_plan_is_not_taken_when_a_hazard_is_protected_by_position_only(dev, oracle):
    """Launch 1 overwrites the vector launch 0 reads from outside the plan, and nothing makes launch 1 wait for launch 0 (its own
    input is external too): stream order protects that, polls would not -- the plan must come out stream-ordered.  Same for an
    output written twice by launches that do not feed each other."""
    from tinychatengine_amd import capi
    descs, outs, refs, keep = _group(dev, oracle, [256, 256], 256, 128, seed=9)
    a, b = descs
    other = torch.randn((1, 256), device=dev).to(torch.float16)
    b.A = other.data_ptr()
    b.C = a.A                                   # WAR: launch 1 writes what launch 0 reads
    plan = capi.Plan([[a], [b]], tagged=True)
    assert not plan.chained
    plan.close()
    b.C = a.C                                   # WAW: both write the same output, neither feeds the other
    plan = capi.Plan([[a], [b]], tagged=True)
    assert not plan.chained
    plan.close()
    b.A, b.C = a.C, outs[1].data_ptr()          # and the plain chain is taken
    plan = capi.Plan([[a], [b]], tagged=True)
    assert plan.tagged
    plan.close()


def test_chained_plan_falls_back_when_a_launch_is_not_a_decode_gemv(dev, oracle):
    from tinychatengine_amd import capi
    descs, outs, refs, keep = _group(dev, oracle, [64], 1024, 128, seed=2)
    a2 = torch.randn((2, 1024), device=dev).to(torch.float16)
    out2 = torch.empty((2, 64), dtype=torch.float16, device=dev)
    d2 = capi.W4A16Desc(M=2, N=64, K=1024, group_size=128, A=a2.data_ptr(), qweight=descs[0].qweight, scales=descs[0].scales,
                        zeros=descs[0].zeros, C=out2.data_ptr())
    plan = capi.Plan([[descs[0]], [d2]], chained=True)
    assert not plan.chained
    plan.launch(torch.cuda.current_stream().cuda_stream)
    plan.status()
    ok, worst = w4a16_close(outs[0].cpu().numpy(), refs[0])
    assert ok
    plan.close()


def test_tuned_plan_matches_the_untuned_plan(dev_chain=None):
    """TCE_PLAN_TUNED: a stream-ordered plan whose launch geometries were timed at creation computes the untuned plan's bits (only geometries that keep every row's summation order are candidates), leaves the caller's output buffers untouched during the timing, and replays."""
    import torch
    from tinychatengine_amd import capi
    from tinychatengine_amd.decode import SHAPES, DecodeLinears
    capi.lib()
    dev = torch.device("cuda:0")
    dl = DecodeLinears(SHAPES["tiny"], device=dev)
    st = torch.cuda.current_stream().cuda_stream
    plan = dl.make_plan()
    plan.launch(st)
    torch.cuda.synchronize()
    outs = [*dl.out_qkv, dl.out_o, dl.out_gate, dl.out_up, dl.out_down, dl.logits]
    want = [o.clone() for o in outs]
    for o in outs:
        o.fill_(3.0)
    tuned = dl.make_plan(tuned=True)
    torch.cuda.synchronize()
    assert all(bool((o == 3.0).all()) for o in outs), "plan creation wrote into the caller's buffers"
    for _ in range(2):
        for o in outs:
            o.fill_(float("nan"))
        tuned.launch(st)
        torch.cuda.synchronize()
        for a, b in zip(outs, want):
            assert torch.isfinite(a.float()).all()
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
