"""GPU tests of the fused decode epilogues: gate_proj + up_proj + SiLuMul_half as one launch
(TCE_W4_SILU_MUL_PAIRS) and o_proj / down_proj + add_half as one launch (TCE_W4_ADD_TO_C).

The element-wise arithmetic is binary16 as in the reference kernels (Int4llamaDecoderLayer.cu:12-30) and restated in
the oracle (orc_silu_mul_half / orc_add_half).  The fused launch must equal the oracle ops applied to the UNFUSED
kernel's own fp16 outputs: bit for bit for the add, and within one half ulp-step for SiLU*mul (the device exponential
may differ from the C library's in the last float bit; see the oracle's comment)."""
import numpy as np
import pytest

from conftest import w4a16_close

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from tinychatengine_amd import capi
    capi.lib()
    capi.set_gemv_config()
    return torch.device("cuda:0")


def _ulps(a, b):
    """Distance in binary16 steps between two half arrays (same sign assumed near zero handled by the bit trick)."""
    ia = a.view(np.int16).astype(np.int32)
    ib = b.view(np.int16).astype(np.int32)
    ia = np.where(ia < 0, -32768 - ia, ia)
    ib = np.where(ib < 0, -32768 - ib, ib)
    return np.abs(ia - ib)


CASES = [(1, 11008, 4096), (1, 344, 4096), (1, 72, 1408), (2, 520, 2048), (4, 256, 4096), (1, 14336, 4096), (12, 96, 1024)]


@pytest.mark.parametrize("M,H,K", CASES)
def test_gate_up_silu_mul_fused(dev, oracle, M, H, K):
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    g = torch.Generator(device=dev).manual_seed(H + K + M)
    gate = Linear_half_int4.from_float(torch.empty(H, K, device=dev).normal_(0, 0.05, generator=g))
    up = Linear_half_int4.from_float(torch.empty(H, K, device=dev).normal_(0, 0.05, generator=g))
    gu = Linear_half_int4.interleave(gate, up)
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    configs = [None] if M > 1 or H > 2000 else [None, (2, 4, 1, 1), (4, 4, 1, 1), (2, 2, 2, 1), (2, 8, 0, 2), (4, 16, 0, 2)]
    try:
        for cfg in configs:
            capi.set_gemv_config(*(cfg or (0, 0, 0, 0)))
            y = torch.empty(M, 2 * H, dtype=torch.float16, device=dev)
            capi.check(capi.w4a16_forward(gu.desc(x, y), torch.cuda.current_stream().cuda_stream))  # same kernel as the fused call
            fused = gu.forward_silu_mul(x)
            torch.cuda.synchronize()
            yn = y.cpu().numpy()
            want = oracle.silu_mul_half(yn[:, 0::2], yn[:, 1::2])
            got = fused.cpu().numpy()
            assert got.shape == (M, H)
            d = _ulps(got, want)
            assert d.max() <= 1, f"cfg {cfg}: {int((d > 1).sum())} outputs differ by more than one half step (max {int(d.max())})"
            assert (d > 0).mean() < 2e-3, f"cfg {cfg}: {float((d > 0).mean()):.4f} of the outputs differ from the oracle"
    finally:
        capi.set_gemv_config()
    # and the unfused outputs themselves are the two projections (row 2n = gate n, row 2n+1 = up n)
    if M == 1:
        a = x.cpu().numpy()
        ref_g, _ = oracle.w4a16_gemv_q4_6(a, gate.weight.cpu().numpy().view(np.uint32), gate.scale.cpu().numpy(),
                                          gate.zero_point.cpu().numpy().view(np.uint32), M, H, K, 128)
        ok, worst = w4a16_close(yn[:, 0::2], ref_g)
        assert ok, f"interleaved gate rows: worst |err|/tol = {worst:.3f}"


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (1, 4096, 11008), (3, 264, 2048), (1, 100, 1408), (64, 256, 1024), (100, 200, 512)])
def test_projection_plus_residual_fused(dev, oracle, M, N, K):
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    g = torch.Generator(device=dev).manual_seed(N + K + M)
    lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.05, generator=g))
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    res = torch.empty(M, N, device=dev).normal_(0, 2, generator=g).to(torch.float16)
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    capi.check(capi.w4a16_forward(lin.desc(x, y), torch.cuda.current_stream().cuda_stream))
    inout = res.clone()
    lin.forward_add(x, inout)
    torch.cuda.synchronize()
    want = oracle.add_half(res.cpu().numpy(), y.cpu().numpy())
    assert np.array_equal(inout.cpu().numpy().view(np.uint16), want.view(np.uint16))


def test_pair_epilogue_rejects_bad_descriptors(dev, oracle):
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    lin = Linear_half_int4.from_float(torch.randn(48, 256, device=dev) * 0.05)
    odd = Linear_half_int4(lin.weight[:47].contiguous(), lin.scale[:47].contiguous(), lin.zero_point[:47].contiguous())
    x = torch.randn(1, 256, device=dev).to(torch.float16)
    out = torch.empty(1, 24, dtype=torch.float16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    assert capi.w4a16_forward(odd.desc(x, out, flags=capi.TCE_W4_SILU_MUL_PAIRS), s) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert capi.w4a16_forward(lin.desc(x, out, flags=capi.TCE_W4_SILU_MUL_PAIRS | capi.TCE_W4_ADD_TO_C), s) == capi.TCE_ERR_BAD_ARG


@pytest.mark.parametrize("n", [8, 4096, 11008, 11008 * 3 + 8, 4099 * 8])
def test_standalone_glue_kernels(dev, oracle, n):
    """tce_add_half bit-exact, tce_silu_mul_half within one half step of the oracle (same caveat as the fused form)."""
    import ctypes as C
    from tinychatengine_amd import capi
    g = torch.Generator(device=dev).manual_seed(n)
    a = (torch.empty(n, device=dev).normal_(0, 3, generator=g)).to(torch.float16)
    b = (torch.empty(n, device=dev).normal_(0, 3, generator=g)).to(torch.float16)
    c = torch.empty_like(a)
    L, s = capi.lib(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(L.tce_add_half(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, s))
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy().view(np.uint16), oracle.add_half(a.cpu().numpy(), b.cpu().numpy()).view(np.uint16))
    want = oracle.silu_mul_half(a.cpu().numpy(), b.cpu().numpy())
    capi.check(L.tce_silu_mul_half(a.data_ptr(), b.data_ptr(), n, s))
    torch.cuda.synchronize()
    d = _ulps(a.cpu().numpy(), want)
    assert d.max() <= 1 and (d > 0).mean() < 2e-3


@pytest.mark.parametrize("m,n", [(1, 4096), (3, 11008), (1, 520), (2, 14336), (5, 64)])
def test_rmsnorm_kernel_matches_oracle(dev, oracle, m, n):
    from tinychatengine_amd.linear import rmsnorm_half
    g = torch.Generator(device=dev).manual_seed(m * 31 + n)
    x = torch.empty(m, n, device=dev).normal_(0, 2.5, generator=g).to(torch.float16)
    gamma = (1 + 0.2 * torch.empty(n, device=dev).normal_(0, 1, generator=g)).float()
    got = rmsnorm_half(x, gamma, 1e-6).cpu().numpy()
    want = oracle.rmsnorm_half(x.cpu().numpy(), gamma.cpu().numpy(), 1e-6)
    d = _ulps(got, want)
    # the sum of squares is associated differently from the reference's block reduction: the common factor rs may differ in
    # its last bit, which moves at most a few per cent of the outputs by one binary16 step
    assert d.max() <= 1 and (d > 0).mean() < 0.08, f"max {int(d.max())} steps, {float((d > 0).mean()):.4f} differ"


@pytest.mark.parametrize("Ns,K", [([4096, 4096, 4096], 4096), ([11008, 11008], 4096), ([72], 1408), ([520, 264], 11008), ([4096], 4096)])
def test_rmsnorm_prologue_fused(dev, oracle, Ns, K):
    """input_layernorm + q/k/v (or post_attention_layernorm + gate/up) as one launch: against the oracle's RMSNorm followed
    by the oracle's GEMV, and against this library's own two-launch form."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4, forward_group, forward_group_rmsnorm, rmsnorm_half
    g = torch.Generator(device=dev).manual_seed(sum(Ns) + K)
    lins = [Linear_half_int4.from_float(torch.empty(n, K, device=dev).normal_(0, 0.05, generator=g)) for n in Ns]
    x = torch.empty(1, K, device=dev).normal_(0, 3, generator=g).to(torch.float16)
    gamma = (1 + 0.2 * torch.empty(K, device=dev).normal_(0, 1, generator=g)).float()
    xn_ref = oracle.rmsnorm_half(x.cpu().numpy(), gamma.cpu().numpy(), 1e-6)
    # automatic, row-block geometries, persistent geometries (waves_k = 0)
    cfgs = [None, (4, 4, 1, 1), (2, 16, 0, 2)] if sum(Ns) > 9000 else [None, (2, 4, 1, 2), (4, 4, 1, 1), (1, 2, 2, 1), (2, 2, 2, 2), (1, 2, 4, 1), (2, 8, 0, 2), (1, 16, 0, 3), (4, 16, 0, 2)]
    try:
        for cfg in cfgs:
            capi.set_gemv_config(*(cfg or (0, 0, 0, 0)))
            fused = [torch.full((1, n), float("nan"), dtype=torch.float16, device=dev) for n in Ns]
            forward_group_rmsnorm(lins, x, fused, gamma, 1e-6)
            two = [torch.empty(1, n, dtype=torch.float16, device=dev) for n in Ns]
            forward_group(lins, rmsnorm_half(x, gamma, 1e-6), two)
            torch.cuda.synchronize()
            xn_gpu = rmsnorm_half(x, gamma, 1e-6).cpu().numpy()
            d = _ulps(xn_gpu, xn_ref)
            assert d.max() <= 1 and (d > 0).mean() < 0.08
            for l, f, t, n in zip(lins, fused, two, Ns):
                fo = f.cpu().numpy()
                assert not np.isnan(fo.astype(np.float32)).any()
                # the fused launch stages exactly the vector the stand-alone kernel writes (same wave-level sum): identical bits
                assert np.array_equal(fo.view(np.uint16), t.cpu().numpy().view(np.uint16)), f"cfg {cfg} N={n}: fused != two launches"
                # and the GEMV on that vector is the oracle's GEMV on it
                ref32, _ = oracle.w4a16_gemv_q4_6(xn_gpu, l.weight.cpu().numpy().view(np.uint32), l.scale.cpu().numpy(),
                                                  l.zero_point.cpu().numpy().view(np.uint32), 1, n, K, 128)
                ok, worst = w4a16_close(fo, ref32)
                assert ok, f"cfg {cfg} N={n}: worst |err|/tol = {worst:.3f}"
    finally:
        capi.set_gemv_config()
    # M > 1 is not a decode shape
    x2 = torch.randn(2, K, device=dev).to(torch.float16)
    with pytest.raises(capi.TceError):
        forward_group_rmsnorm(lins[:1], x2, [torch.empty(2, Ns[0], dtype=torch.float16, device=dev)], gamma, 1e-6)


def test_decoder_layer_linears_fused_in_plans(dev, oracle):
    """A decoder layer's linears with their glue (attention replaced by a fixed vector): the reference's 9-launch
    structure on the stand-alone kernels against 4 fused launches, issued directly, as a stream-ordered plan and as a
    chained (token-kernel) plan.  Same device arithmetic everywhere, so the hidden state must come out bit-identical."""
    import ctypes as C
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4, forward_group, rmsnorm_half
    h, f, eps = 1024, 2816, 1e-5
    g = torch.Generator(device=dev).manual_seed(77)
    mk = lambda n, k: Linear_half_int4.from_float(torch.empty(n, k, device=dev).normal_(0, 1.0 / np.sqrt(k), generator=g))
    qkv, o, gate, up, down = mk(3 * h, h), mk(h, h), mk(f, h), mk(f, h), mk(h, f)
    gu = Linear_half_int4.interleave(gate, up)
    gam1 = (1 + 0.1 * torch.empty(h, device=dev).normal_(0, 1, generator=g)).float()
    gam2 = (1 + 0.1 * torch.empty(h, device=dev).normal_(0, 1, generator=g)).float()
    hid0 = torch.empty(1, h, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    attn = torch.empty(1, h, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    L, s = capi.lib(), torch.cuda.current_stream().cuda_stream
    sp = C.c_void_p(s)

    # --- reference launch structure ---
    hid = hid0.clone()
    xn = rmsnorm_half(hid, gam1, eps)
    t_qkv = qkv(xn)
    t_o = o(attn)
    capi.check(L.tce_add_half(hid.data_ptr(), t_o.data_ptr(), hid.data_ptr(), h, sp))
    xn = rmsnorm_half(hid, gam2, eps)
    t_g, t_u = torch.empty(1, f, dtype=torch.float16, device=dev), torch.empty(1, f, dtype=torch.float16, device=dev)
    forward_group([gate, up], xn, [t_g, t_u])
    capi.check(L.tce_silu_mul_half(t_g.data_ptr(), t_u.data_ptr(), f, sp))
    t_d = down(t_g)
    capi.check(L.tce_add_half(hid.data_ptr(), t_d.data_ptr(), hid.data_ptr(), h, sp))
    torch.cuda.synchronize()
    want_hid, want_qkv = hid.cpu().numpy().copy(), t_qkv.cpu().numpy().copy()

    # --- fused: 4 launches ---
    hid_f = torch.empty_like(hid0)
    q_f = torch.empty(1, 3 * h, dtype=torch.float16, device=dev)
    act = torch.empty(1, f, dtype=torch.float16, device=dev)
    launches = [[qkv.desc(hid_f, q_f, gamma=gam1, eps=eps)],
                [o.desc(attn, hid_f, flags=capi.TCE_W4_ADD_TO_C)],
                [gu.desc(hid_f, act, flags=capi.TCE_W4_SILU_MUL_PAIRS, gamma=gam2, eps=eps)],
                [down.desc(act, hid_f, flags=capi.TCE_W4_ADD_TO_C)]]

    def check(what):
        torch.cuda.synchronize()
        assert np.array_equal(q_f.cpu().numpy().view(np.uint16), want_qkv.view(np.uint16)), f"{what}: q/k/v"
        assert np.array_equal(hid_f.cpu().numpy().view(np.uint16), want_hid.view(np.uint16)), f"{what}: hidden state"

    hid_f.copy_(hid0)
    for d in launches:
        capi.check(capi.w4a16_forward(d[0], s))
    check("direct")
    for tagged in (False, True):
        # the token-kernel form is NOT taken for this list: o_proj (fixed `attn`, nothing to wait for) overwrites the hidden state
        # that the q/k/v launch reads from outside the plan -- stream order protects that by position, tagged polls would not
        plan = capi.Plan(launches, tagged=tagged)
        assert not plan.chained
        for _ in range(3):
            hid_f.copy_(hid0); q_f.zero_(); act.zero_()
            plan.launch(s)
            plan.status()
            check(f"plan tagged={tagged}")
        plan.close()

    # --- the same four fused launches with o_proj fed by the q slice of the q/k/v output (in the model the attention sits between
    #     them): now every hazard is downstream in the data flow, and the token kernel takes the list -- fused RMSNorm prologues,
    #     SiLU-mul pairs and residual adds included.  Bit-identical to the launches issued one by one.
    launches[1] = [o.desc(q_f[:, :h], hid_f, flags=capi.TCE_W4_ADD_TO_C)]
    hid_f.copy_(hid0)
    for d in launches:
        capi.check(capi.w4a16_forward(d[0], s))
    torch.cuda.synchronize()
    want_hid, want_qkv = hid_f.cpu().numpy().copy(), q_f.cpu().numpy().copy()
    plan = capi.Plan(launches, tagged=True)
    assert plan.tagged
    for it in range(5):
        hid_f.copy_(hid0); q_f.fill_(float("nan")); act.fill_(float("nan"))
        plan.launch(s)
        plan.status()
        check(f"token kernel, replay {it}")
    plan.close()


@pytest.mark.parametrize("M,H,K", [(512, 2752, 1024), (200, 136, 512), (300, 1032, 2048), (130, 264, 512)])
def test_gate_up_silu_mul_on_the_prefill_gemm(dev, oracle, M, H, K):
    """TCE_W4_SILU_MUL_PAIRS on a batch with pre-packed weights: the pair epilogue of the 128-row GEMM (w4a16_gemm_pk.hip) against the
    gate GEMM and the up GEMM on the same kernel followed by tce_silu_mul_half (the prefill form of Int4llamaDecoderLayer.cu:96-102); column tails (H = 136:
    272 interleaved columns, not a multiple of 128 or 16) and the k range cut across workgroups (few tiles) included.  Below the GEMM's row threshold (M = 130)
    the flag runs on the GEMV kernel as before: there the tolerance is the linears' (another accumulation order), one binary16 step on the products."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    g = torch.Generator(device=dev).manual_seed(M + H + K)
    rnd = lambda n, k: torch.empty(n, k, device=dev).normal_(0.0, k ** -0.5, generator=g)
    gate, up = Linear_half_int4.from_float(rnd(H, K)), Linear_half_int4.from_float(rnd(H, K))
    both = Linear_half_int4.interleave(gate, up)
    for l in (gate, up, both):
        l.prepack()
    x = torch.randn(M, K, device=dev, generator=g).to(torch.float16)
    st = torch.cuda.current_stream().cuda_stream
    a, b = torch.empty((M, H), dtype=torch.float16, device=dev), torch.empty((M, H), dtype=torch.float16, device=dev)
    capi.check(capi.w4a16_forward(gate.desc(x, a), st))
    capi.check(capi.w4a16_forward(up.desc(x, b), st))
    capi.check(capi.lib().tce_silu_mul_half(a.data_ptr(), b.data_ptr(), a.numel(), st))
    fused = torch.full((M, H), 7.0, dtype=torch.float16, device=dev)
    d = both.desc(x, fused, flags=capi.TCE_W4_SILU_MUL_PAIRS)
    capi.check(capi.w4a16_forward(d, st))
    torch.cuda.synchronize()
    import ctypes as C
    buf = C.create_string_buffer(256)
    capi.lib().tce_w4a16_describe_dispatch(C.byref(d), buf, 256)
    want, got = a.cpu().numpy(), fused.cpu().numpy()
    assert np.isfinite(got).all()
    if M >= 192:
        assert buf.value.decode().startswith("gemm-pk"), buf.value
        # the fused launch (N = 2 H columns) and the separate ones (N = H) may run different forms of the kernel (tile width, k range cut across workgroups): the
        # same products summed in another fp32 order, so a gate or up value can round to the neighbouring half -- few elements, one step on the factors
        differ = got != want
        assert differ.mean() < 5e-3, f"{differ.sum()} elements differ"
        g64, w64 = got.astype(np.float64), want.astype(np.float64)
        assert np.all(np.abs(g64 - w64) <= 2.0 ** -8 * np.abs(w64) + 2.0 ** -14), "more than rounding-order differences"
    else:
        assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 2e-2 * np.abs(want.astype(np.float64)).max()
