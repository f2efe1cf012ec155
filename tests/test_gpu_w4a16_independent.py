"""GPU parity tests of tce_w4a16_forward_independent (round 6; SURVEY section 8e (i): the column-sharded block -- a rank's q / k / v / o / gate / up / down
shards read replicated inputs and do not depend on each other): up to eight decode linears with their OWN activations, K and N as ONE launch
(csrc/w4a16_gemv_i8.hip, the kernel with ARGS = I8MixArgs: wave groups of each linear's own K-split width inside workgroups sized for the longest K).

Held here: every output against the oracle; every output BIT-IDENTICAL to the same descriptor through tce_w4a16_forward (a row's bits depend on K and the
group size only -- not on what else is in the launch); the launch count the call reports; ragged N / ragged K / K not a multiple of 1024; linears whose K differs
by more than the workgroup can pack (14 waves: three 4-wave tiles, one 11-wave tile); general zero points; the epilogue flags per linear; the fallbacks
(no packed copy, M > 1, groups of 64: one launch per linear, same results); the Llama-3-8B block at 8 / 4 / 2 ranks at full size.
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
    capi.set_gemm_config()
    capi.set_gemv_i8()
    return torch.device("cuda:0")


def _lin(oracle, dev, N, K, G, seed, random_zeros=False, prepack=True):
    from tinychatengine_amd.linear import Linear_half_int4
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    qw, sc, zp, _, _ = oracle.quantize_q4_6(w, G)
    if random_zeros:
        nib = rng.integers(0, 16, (N, zp.shape[1] * 8), dtype=np.uint32)
        zp = (nib.reshape(N, -1, 8) << (np.arange(8, dtype=np.uint32) * 4)).sum(axis=2).astype(np.uint32)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    lin = Linear_half_int4(t(qw.view(np.int32)), t(sc.view(np.float16)), t(zp.view(np.int32)), G)
    if prepack:
        lin.prepack()
    return lin, (qw, sc, zp)


def _run_both(capi, lins, xs, flags=None, m=1):
    """(outputs of ONE tce_w4a16_forward_independent call, outputs of tce_w4a16_forward per linear, launches reported)"""
    st = torch.cuda.current_stream().cuda_stream
    flags = flags or [0] * len(lins)
    width = lambda l, f: l.out_features // 2 if f & capi.TCE_W4_SILU_MUL_PAIRS else l.out_features
    nan = lambda l, f: torch.full((m, width(l, f)), float("nan"), dtype=torch.float16, device=xs[0].device)
    outs_a = [nan(l, f) for l, f in zip(lins, flags)]
    outs_b = [nan(l, f) for l, f in zip(lins, flags)]
    n = capi.w4a16_forward_independent([l.desc(x, o, flags=f) for l, x, o, f in zip(lins, xs, outs_a, flags)], st)
    for l, x, o, f in zip(lins, xs, outs_b, flags):
        capi.check(capi.w4a16_forward(l.desc(x, o, flags=f), st))
    torch.cuda.synchronize()
    return outs_a, outs_b, n


def _assert_same_bits(outs_a, outs_b, what):
    for i, (a, b) in enumerate(zip(outs_a, outs_b)):
        assert not torch.isnan(a.float()).any(), f"{what}: linear {i} has unwritten outputs"
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"{what}: linear {i} differs from tce_w4a16_forward"


# (N, K) per linear: ragged N, K in whole and part 1024-k waves, very different K in one launch
CASES = {
    "two, same K": [(256, 4096), (100, 4096)],
    "block of a sharded llama (small)": [(96, 1024), (32, 1024), (32, 1024), (96, 1024), (224, 1024), (224, 1024), (96, 3584)],
    "K 4096 beside 14336": [(72, 4096), (40, 14336), (264, 4096)],
    "K 1408 / 11008 / 128 / 16384": [(100, 1408), (48, 11008), (16, 128), (24, 16384)],
    "eight linears": [(16 * (i + 1), 1024 * (1 + i % 3)) for i in range(8)],
    "one tile each": [(16, 2048), (16, 4096), (8, 1024)],
}


@pytest.mark.parametrize("name", list(CASES))
def test_independent_linears_one_launch_same_bits(dev, oracle, name):
    from tinychatengine_amd import capi
    shapes = CASES[name]
    lins, xs, refs = [], [], []
    for i, (N, K) in enumerate(shapes):
        lin, (qw, sc, zp) = _lin(oracle, dev, N, K, 128, seed=1000 + 17 * i + N + K)
        a = np.random.default_rng(7 * i + K).standard_normal((1, K)).astype(np.float16)
        refs.append(oracle.w4a16_gemv_q4_6(a, qw, sc, zp, 1, N, K, 128)[0])
        lins.append(lin)
        xs.append(torch.from_numpy(a).to(dev))
    outs_a, outs_b, n = _run_both(capi, lins, xs)
    assert n == 1, f"{name}: {n} launches"
    _assert_same_bits(outs_a, outs_b, name)
    for i, (o, r) in enumerate(zip(outs_a, refs)):
        ok, worst = w4a16_close(o.cpu().numpy(), r)
        assert ok, f"{name}: linear {i} worst |err|/tol = {worst:.3f}"


def test_independent_linears_zero_points_and_epilogues(dev, oracle):
    """General zero points in ONE of the linears (the launch then reads every linear's zero points: same values), SiLU-mul pairs on one, residual add on another."""
    from tinychatengine_amd import capi
    shapes = [(128, 4096), (64, 2048), (96, 14336), (40, 1024)]
    lins, xs = [], []
    for i, (N, K) in enumerate(shapes):
        lin, _ = _lin(oracle, dev, N, K, 128, seed=50 + i, random_zeros=(i == 1))
        lins.append(lin)
        xs.append(torch.from_numpy(np.random.default_rng(i).standard_normal((1, K)).astype(np.float16)).to(dev))
    flags = [capi.TCE_W4_SILU_MUL_PAIRS, 0, capi.TCE_W4_ADD_TO_C, 0]
    st = torch.cuda.current_stream().cuda_stream
    width = lambda l, f: l.out_features // 2 if f & capi.TCE_W4_SILU_MUL_PAIRS else l.out_features
    base = [torch.randn((1, width(l, f)), device=dev).to(torch.float16) for l, f in zip(lins, flags)]
    outs_a, outs_b = [b.clone() for b in base], [b.clone() for b in base]
    n = capi.w4a16_forward_independent([l.desc(x, o, flags=f) for l, x, o, f in zip(lins, xs, outs_a, flags)], st)
    for l, x, o, f in zip(lins, xs, outs_b, flags):
        capi.check(capi.w4a16_forward(l.desc(x, o, flags=f), st))
    torch.cuda.synchronize()
    assert n == 1
    _assert_same_bits(outs_a, outs_b, "zero points + epilogues")
    assert not torch.equal(outs_a[2], base[2])  # the residual add happened


def test_independent_linears_fallbacks_give_the_same_results(dev, oracle):
    """What the one-launch form does not take is issued linear by linear: no packed copy, two rows, groups of 64, a K beyond 16384."""
    from tinychatengine_amd import capi
    st = torch.cuda.current_stream().cuda_stream
    for what, kw, K, G, m in (("no packed copy", dict(prepack=False), 1024, 128, 1), ("two rows", {}, 1024, 128, 2), ("groups of 64", {}, 1024, 64, 1), ("K 28672", {}, 28672, 128, 1)):
        lins, xs = [], []
        for i in range(3):
            Ki = K if i == 0 else 1024
            lin, _ = _lin(oracle, dev, 48 + 16 * i, Ki, G, seed=300 + i, **kw)
            lins.append(lin)
            xs.append(torch.from_numpy(np.random.default_rng(i).standard_normal((m, Ki)).astype(np.float16)).to(dev))
        outs_a, outs_b, n = _run_both(capi, lins, xs, m=m)
        assert n == 3, f"{what}: {n} launches"
        _assert_same_bits(outs_a, outs_b, what)
    with pytest.raises(capi.TceError):
        capi.w4a16_forward_independent([], st)


@pytest.mark.parametrize("P", [8, 4, 2])
def test_independent_linears_llama3_block_shards_full_size(dev, P):
    """A rank's seven shards of a Llama-3-8B block at P ranks (q 4096, k / v 1024, o 4096, gate / up 14336 of K = 4096; down 4096 of K = 14336; rows / P): one launch,
    bit-identical to the seven tce_w4a16_forward calls (whose own parity at these widths is tests/test_gpu_w4a16.py's and test_sharding's)."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    g = torch.Generator(device=dev).manual_seed(P)
    lins, xs = [], []
    for n, k in ((4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096), (4096, 14336)):
        w = torch.empty((n // P, k), dtype=torch.float32, device=dev).normal_(0.0, 0.02, generator=g)
        lins.append(Linear_half_int4.from_float(w, 128).prepack())
        xs.append(torch.empty((1, k), dtype=torch.float32, device=dev).normal_(0.0, 1.0, generator=g).to(torch.float16))
    outs_a, outs_b, n = _run_both(capi, lins, xs)
    assert n == 1
    _assert_same_bits(outs_a, outs_b, f"llama3-8b block shards, P = {P}")


def test_independent_linears_in_a_graph_replay(dev, oracle):
    """Captured into a hipGraph and replayed with changing activations: the launch holds no state."""
    from tinychatengine_amd import capi
    lins, xs = [], []
    for i, (N, K) in enumerate([(64, 4096), (48, 14336), (32, 1024)]):
        lin, _ = _lin(oracle, dev, N, K, 128, seed=700 + i)
        lins.append(lin)
        xs.append(torch.randn((1, K), device=dev).to(torch.float16))
    outs = [torch.zeros((1, l.out_features), dtype=torch.float16, device=dev) for l in lins]
    s = torch.cuda.Stream()
    descs = [l.desc(x, o) for l, x, o in zip(lins, xs, outs)]
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        capi.w4a16_forward_independent(descs, s.cuda_stream)  # (warm: code object load outside the capture)
        torch.cuda.synchronize()
        with torch.cuda.graph(gph, stream=s):
            capi.w4a16_forward_independent(descs, s.cuda_stream)
    for rep in range(3):
        for x in xs:
            x.copy_(torch.randn_like(x.float()).to(torch.float16))
        gph.replay()
        torch.cuda.synchronize()
        want = [torch.full_like(o, float("nan")) for o in outs]
        for l, x, w in zip(lins, xs, want):
            capi.check(capi.w4a16_forward(l.desc(x, w), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        _assert_same_bits(outs, want, f"replay {rep}")


def test_plan_of_independent_groups_equals_the_stream_ordered_plan(dev):
    """TCE_PLAN_INDEPENDENT: a rank's token with ONE launch per block (tiny model, two ranks' worth of rows) against the four-launches-per-block plan -- every output
    of the token bit for bit; the flag does not combine with the other plan flags."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.decode import SHAPES, DecodeLinears
    dl = DecodeLinears(SHAPES["tiny"], device=dev, rank=1, world=2, prepack=True)
    st = torch.cuda.current_stream().cuda_stream
    outs = [*dl.out_qkv, dl.out_o, dl.out_gate, dl.out_up, dl.out_down, dl.logits]
    four, one = dl.make_plan(), dl.make_plan(one_launch_per_block=True)
    assert four.n_launches == dl.n_layers * 4 + 1 and one.n_launches == dl.n_layers + 1
    four.launch(st)
    torch.cuda.synchronize()
    want = [o.clone() for o in outs]
    for rep in range(2):
        for o in outs:
            o.fill_(float("nan"))
        one.launch(st)
        torch.cuda.synchronize()
        _assert_same_bits(outs, want, f"plan, replay {rep}")
    descs = (capi.W4A16Desc * 1)(dl.token_launches()[0][0])
    groups = (capi.C.c_int32 * 1)(1)
    h = capi.C.c_void_p()
    rc = capi.lib().tce_plan_create_ex(descs, groups, 1, capi.TCE_PLAN_INDEPENDENT | capi.TCE_PLAN_TUNED, capi.C.byref(h))
    assert rc == capi.TCE_ERR_BAD_ARG
