"""GPU parity of the pre-packed 128-row prefill GEMM (csrc/w4a16_gemm_pk.hip) and of its load-time re-layout
(tce_w4a16_prepack -> q4_mfma, csrc/w4a16_mfma_layout.hpp).  Oracle and tolerance as for every W4A16 path
(oracle.w4a16_gemv_q4_6 = the reference's arithmetic, kernels/cuda/gemv_cuda.cu:181-193; conftest.w4a16_close)."""
import numpy as np
import pytest

from conftest import record_parity, w4a16_close, w4a16_report

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from tinychatengine_amd import capi
    capi.lib()
    capi.set_gemm_config()
    return torch.device("cuda:0")


def _quant(oracle, N, K, G, seed, random_zeros, zero_scale_groups=0):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    qw, sc, zp, _, _ = oracle.quantize_q4_6(w, G)
    if random_zeros:
        nib = rng.integers(0, 16, (N, zp.shape[1] * 8), dtype=np.uint32)
        zp = (nib.reshape(N, -1, 8) << (np.arange(8, dtype=np.uint32) * 4)).sum(axis=2).astype(np.uint32)
    sc = sc.copy()
    for _ in range(zero_scale_groups):  # groups whose scale is 0 (an all-zero weight group): they must contribute exactly nothing
        sc[rng.integers(0, N), rng.integers(0, K // G)] = 0.0
    if zero_scale_groups:
        sc[0, 0] = 0.0  # a leading one
        sc[1, :] = 0.0  # and a whole row
    return qw, sc, zp


def _lin(dev, qw, sc, zp, G):
    from tinychatengine_amd.linear import Linear_half_int4
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    return Linear_half_int4(t(qw.view(np.int32)), t(sc.view(np.float16)), t(zp.view(np.int32)), G)


def test_prepack_layout_matches_its_specification(dev, oracle):
    """words / consts of the packed buffer against a numpy restatement of w4a16_mfma_layout.hpp."""
    from tinychatengine_amd import capi
    for (N, K, G) in [(40, 256, 128), (136, 384, 64), (16, 128, 32)]:
        qw, sc, zp = _quant(oracle, N, K, G, seed=N + K, random_zeros=True, zero_scale_groups=3)
        lin = _lin(dev, qw, sc, zp, G).prepack()
        torch.cuda.synchronize()
        buf = lin.packed.cpu().numpy()
        nt, nkb, ng = (N + 15) // 16, K // 128, K // G
        words = buf[: nt * nkb * 64 * 4 * 4].view(np.uint32).reshape(nt, nkb, 64, 4)
        c_off = (nt * nkb * 1024 + 255) // 256 * 256
        consts = buf[c_off: c_off + nt * ng * 16 * 8].view(np.uint32).reshape(nt, ng, 16, 2)
        codes = oracle.unpack_q4_6(qw, N, K).astype(np.int64)
        zn = ((zp[:, :, None] >> (np.arange(8, dtype=np.uint32) * 4)) & 0xF).reshape(N, -1)[:, :ng].astype(np.int64)
        s32 = sc[:, :ng].astype(np.float32)
        for n in range(nt * 16):
            for g in range(ng):
                e_expected = None
                if n < N:
                    nz = [s for s in s32[n, : g + 1] if s != 0]
                    first = [s for s in s32[n] if s != 0]
                    e_expected = nz[-1] if nz else (first[0] if first else np.float32(1.0))
                    z = zn[n, g]
                else:
                    e_expected, z = np.float32(1.0), 8
                got = consts[n // 16, g, n % 16]
                assert got[0] == np.float32(e_expected).view(np.uint32), (N, K, G, n, g)
                assert got[1] == (((0xD400 | (int(z) << 4)) << 16) | (0xE400 | int(z))), (n, g)
        for n in range(nt * 16):
            for k in range(0, K, 8):
                kb, kk = k // 128, k % 128
                s, q = kk // 32, (kk // 8) % 4
                w = int(words[n // 16, kb, q * 16 + n % 16, s])
                got = [(w >> (4 * ((e >> 1) + 4 * (e & 1)))) & 0xF for e in range(8)]
                if n < N:
                    g = k // G
                    exp = [int(zn[n, g])] * 8 if s32[n, g] == 0 else [int(c) for c in codes[n, k:k + 8]]
                else:
                    exp = [8] * 8
                assert got == exp, (N, K, G, n, k, got, exp)


PK_SHAPES = [
    # M, N, K, G
    (256, 256, 512, 128), (200, 136, 384, 128), (513, 2100, 256, 128), (128, 128, 128, 128), (300, 264, 1024, 64), (192, 200, 512, 32),
    (384, 520, 1408 + 128 * 5, 128),
    (260, 300, 1408, 64), (700, 392, 3072, 32),  # uneven runs when the k range is cut across workgroups (11 k-blocks in 2; 24 in 3 / 4), groups of 64 / 32
    (260, 300, 1408, 128), (200, 136, 1152, 128), (130, 520, 1024, 128),  # round 6, form 16 (mode 2676): 11 / 9 / 8 k-blocks handed off between two workgroups of two quartets each
]


@pytest.mark.parametrize("M,N,K,G", PK_SHAPES)
def test_pk_gemm_matches_oracle(dev, oracle, M, N, K, G):
    """Both forms (one / two wave quartets per tile) and the automatic choice; random zero points, zero-scale groups, row tails
    (M % 128 != 0), column tails (N % 128, N % 16 != 0), a single k-block, an odd number of k-blocks (quartet 1 runs one step less)."""
    from tinychatengine_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(M + N + K)
    for rz, zs in ((False, 0), (True, 4), (False, 3)):  # (the wide forms -- modes 2670 .. 2672 -- run on the linears whose zero points are all 8; the others keep the narrow forms under these modes)
        qw, sc, zp = _quant(oracle, N, K, G, seed=M * 3 + N + K, random_zeros=rz, zero_scale_groups=zs)
        a = rng.standard_normal((M, K)).astype(np.float16)
        ref32 = oracle.w4a16_gemv_q4_6_mt(a, qw, sc, zp, M, N, K, G)
        lin = _lin(dev, qw, sc, zp, G).prepack()
        x = torch.from_numpy(a).to(dev)
        try:
            for mode in (61, 62, 63, 64, 66, 67, 672, 68, 2669, 2670, 2671, 2672, 2683, 2673, 2674, 2675, 2676, 60):  # 2676 (round 6): form 16 -- two quartets alternating a run's k-blocks, the k range handed off between two workgroups (8+ k-blocks, groups of 128; otherwise form 2);  64: the k range cut across workgroups (needs the scratch area desc() attaches); 66 / 67 / 672 / 68 / 2669: the 256-row wave tiles (round 5; groups of 128 -- other group sizes run form 1 under these modes): whole tiles / k range cut / two quartets alternating a tile's k-blocks (even counts; odd: one quartet) / two quartets side by side on 256 x 256
                capi.check(L.tce_w4a16_set_debug_mode(mode))
                out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
                d = lin.desc(x, out)
                if mode != 60:  # (60 = automatic: the cost models of the two GEMMs decide, test_pk_dispatch_rules)
                    assert capi.describe_dispatch(d).startswith("gemm-pk"), capi.describe_dispatch(d)
                if mode in (2670, 2671, 2672, 2683, 2675) and not rz and G == 128 and M > 128:  # round 5: 128 rows x 64 columns per wave
                    assert "wave=128x64" in capi.describe_dispatch(d), capi.describe_dispatch(d)
                if mode == 2676 and G == 128 and K >= 1024:
                    assert "quartets=2 ksplit=2" in capi.describe_dispatch(d), capi.describe_dispatch(d)
                if mode in (2673, 2674) and not rz and G == 128 and M > 128:  # the same on 128 x 192 tiles
                    assert "wave=128x48" in capi.describe_dispatch(d), capi.describe_dispatch(d)
                capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                got = out.cpu().numpy()
                assert not np.isnan(got.astype(np.float32)).any(), f"mode {mode}: unwritten outputs"
                ok, worst = w4a16_close(got, ref32)
                assert ok, f"pk mode {mode} {M}x{N}x{K} g{G} rz={rz}: worst |err|/tol = {worst:.3f}"
        finally:
            L.tce_w4a16_set_debug_mode(60)


def test_pk_gemm_add_to_c_and_strides(dev, oracle):
    """TCE_W4_ADD_TO_C, a padded lda and a wider ldc: bit for bit against the plain pre-packed launch."""
    from tinychatengine_amd import capi
    M, N, K, G = 260, 264, 512, 128
    qw, sc, zp = _quant(oracle, N, K, G, seed=7, random_zeros=True)
    lin = _lin(dev, qw, sc, zp, G).prepack()
    rng = np.random.default_rng(3)
    a = rng.standard_normal((M, K)).astype(np.float16)
    x = torch.from_numpy(a).to(dev)
    plain = torch.zeros(M, N, dtype=torch.float16, device=dev)
    capi.check(capi.w4a16_forward(lin.desc(x, plain), torch.cuda.current_stream().cuda_stream))
    xp = torch.full((M, K + 24), float("nan"), dtype=torch.float16, device=dev)
    xp[:, :K] = x
    c0 = (torch.randn(M, N + 8, device=dev) * 0.5).to(torch.float16)
    acc = c0.clone()
    d = lin.desc(x, acc, ldc=N + 8, flags=capi.TCE_W4_ADD_TO_C)
    d.A, d.lda = xp.data_ptr(), K + 24
    capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(acc[:, :N], c0[:, :N] + plain) and torch.equal(acc[:, N:], c0[:, N:])
    ref32 = oracle.w4a16_gemv_q4_6_mt(a, qw, sc, zp, M, N, K, G)
    assert w4a16_close(plain.cpu().numpy(), ref32)[0]


def test_pk_gemm_with_the_tail_tiles_cut(dev, oracle):
    """More than 256 tiles: the first 256 run whole, the k-blocks of the others as short runs beside them (partials through the
    scratch area, fixed order).  260 tiles here; every row of two row blocks against the oracle, all rows against the one-quartet form."""
    from tinychatengine_amd import capi
    M, N, K, G = 640, 6656, 1024, 128
    qw, sc, zp = _quant(oracle, N, K, G, seed=77, random_zeros=True, zero_scale_groups=3)
    lin = _lin(dev, qw, sc, zp, G).prepack()
    rng = np.random.default_rng(5)
    a = rng.standard_normal((M, K)).astype(np.float16)
    x = torch.from_numpy(a).to(dev)
    out = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
    d = lin.desc(x, out)
    st = torch.cuda.current_stream().cuda_stream
    try:
        capi.check(capi.lib().tce_w4a16_set_debug_mode(64))  # "cut the k range" by name (the automatic rule decides by its cost model)
        assert "of-the-tiles-past-256" in capi.describe_dispatch(d), capi.describe_dispatch(d)
        for rep in range(3):
            out.fill_(float("nan"))
            capi.check(capi.w4a16_forward(d, st))
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            assert not np.isnan(got.astype(np.float32)).any()
            if rep:
                assert np.array_equal(got.view(np.uint16), first.view(np.uint16)), "the sum must not depend on which workgroup arrives last"
            first = got
    finally:
        capi.lib().tce_w4a16_set_debug_mode(60)
    rows = list(range(0, 128)) + list(range(512, 640))
    ref32 = oracle.w4a16_gemv_q4_6_mt(a[rows], qw, sc, zp, len(rows), N, K, G)
    ok, worst = w4a16_close(got[rows], ref32)
    assert ok, worst
    try:  # the whole-tile form on the same inputs: the cut tiles differ from it by fp32 reassociation only
        capi.check(capi.lib().tce_w4a16_set_debug_mode(61))
        out2 = torch.empty_like(out)
        capi.check(capi.w4a16_forward(lin.desc(x, out2), st))
        torch.cuda.synchronize()
    finally:
        capi.lib().tce_w4a16_set_debug_mode(60)
    whole = out2.cpu().numpy().astype(np.float32)
    assert np.abs(got.astype(np.float32) - whole).max() <= 2e-3 * np.abs(whole).max()
    from tinychatengine_amd.linear import gemm_scratch
    assert int(gemm_scratch(dev)[:4096].to(torch.int32).sum().item()) == 0


def test_pk_gemm_k_split_exchange_under_repetition(dev, oracle):
    """The partial tiles of a k range cut across workgroups travel through the scratch area (write-through stores, a counter, coherent
    loads): 150 back-to-back launches on ONE scratch area with the activations changing between launches and no host synchronisation in
    between -- a partial read before it landed, or a stale one from the launch before, shows up against the whole-tile form."""
    from tinychatengine_amd import capi
    M, N, K, G = 256, 1024, 2048, 128  # 16 tiles, cut in 4 (the rule's choice when forced): 64 workgroups, 48 partial tiles per launch
    qw, sc, zp = _quant(oracle, N, K, G, seed=123, random_zeros=True)
    lin = _lin(dev, qw, sc, zp, G).prepack()
    g = torch.Generator(device=dev).manual_seed(9)
    xs = [torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16) for _ in range(150)]
    outs = [torch.empty(M, N, dtype=torch.float16, device=dev) for _ in range(150)]
    st = torch.cuda.current_stream().cuda_stream
    try:
        capi.check(capi.lib().tce_w4a16_set_debug_mode(64))
        assert "ksplit=" in capi.describe_dispatch(lin.desc(xs[0], outs[0]))
        for x, o in zip(xs, outs):
            capi.check(capi.w4a16_forward(lin.desc(x, o), st))
        torch.cuda.synchronize()
        capi.check(capi.lib().tce_w4a16_set_debug_mode(61))
        whole = torch.empty(M, N, dtype=torch.float16, device=dev)
        for i, (x, o) in enumerate(zip(xs, outs)):
            capi.check(capi.w4a16_forward(lin.desc(x, whole), st))
            torch.cuda.synchronize()
            w = whole.float()
            assert (o.float() - w).abs().max().item() <= 2e-3 * w.abs().max().item(), f"launch {i}"
        # the residual-add epilogue behind the exchange: C = hadd(C, y) with y the summed tile
        res = torch.empty(M, N, device=dev).normal_(0, 2, generator=g).to(torch.float16)
        got = res.clone()
        capi.check(capi.lib().tce_w4a16_set_debug_mode(64))
        capi.check(capi.w4a16_forward(lin.desc(xs[0], got, flags=capi.TCE_W4_ADD_TO_C), st))
        torch.cuda.synchronize()
        want = (res.float() + outs[0].float()).to(torch.float16)  # hadd of two halves = one rounding of the exact sum
        assert torch.equal(got, want)
    finally:
        capi.lib().tce_w4a16_set_debug_mode(60)


def test_pk_dispatch_rules(dev, oracle):
    """No packed copy -> the other kernels; M <= 128 -> the other kernels; in between the two GEMMs' cost models decide (the 64-row
    tiles keep M = 512 at N = 4096, where 128-row tiles are too few to fill 256 CUs); K % 128 != 0 -> no packed form at all."""
    from tinychatengine_amd import capi
    qw, sc, zp = _quant(oracle, 256, 512, 128, seed=1, random_zeros=False)
    lin = _lin(dev, qw, sc, zp, 128)
    x = torch.zeros(512, 512, dtype=torch.float16, device=dev)
    out = torch.zeros(512, 256, dtype=torch.float16, device=dev)
    assert capi.describe_dispatch(lin.desc(x, out)).startswith("gemm-dma")
    lin.prepack()
    try:
        capi.check(capi.lib().tce_w4a16_set_debug_mode(61))
        assert capi.describe_dispatch(lin.desc(x, out)).startswith("gemm-pk tile=128x")
    finally:
        capi.lib().tce_w4a16_set_debug_mode(60)
    assert not capi.describe_dispatch(lin.desc(x[:64], out[:64])).startswith("gemm-pk")
    # describe_dispatch only reads the descriptor: the full-size decisions without full-size tensors
    def decision(M, N, K):
        d = lin.desc(x, out)
        d.M, d.N, d.K, d.lda, d.ldc = M, N, K, K, N
        return capi.describe_dispatch(d).split()[0]
    assert decision(2048, 4096, 4096) == decision(512, 11008, 4096) == decision(4096, 4096, 11008) == "gemm-pk"
    assert decision(128, 11008, 4096) != "gemm-pk"
    # few tiles (M = 512 at N = 4096: 128 tiles for 256 CUs): with a scratch area the k range is cut across workgroups, without one the
    # 64-row kernel keeps the launch
    d = lin.desc(x, out)
    d.M, d.N, d.K, d.lda, d.ldc = 512, 4096, 4096, 4096, 4096
    assert d.scratch and "ksplit=2" in capi.describe_dispatch(d), capi.describe_dispatch(d)
    d.K = d.lda = 11008
    assert "quartets=2 ksplit=2" in capi.describe_dispatch(d), capi.describe_dispatch(d)  # (round 6, form 16: two runs of two quartets each, 50.9-53.8 us against four one-quartet runs' 53.1-54.1)
    d.K = d.lda = 14336
    assert "ksplit=4" in capi.describe_dispatch(d), capi.describe_dispatch(d)  # (round 4: the scratch area holds 512 units; four runs per tile win on the long k range -- also against round 5's two handed-off runs once the weights come from HBM: 53.5 against 54.55 us)
    d.scratch = None
    assert capi.describe_dispatch(d).startswith("gemm-dma")
    assert capi.describe_dispatch(lin.desc(x[:1], out[:1])).startswith("gemv")
    assert int(capi.lib().tce_w4a16_prepack_bytes(256, 1440, 32)) == 0


@pytest.mark.parametrize("N,K", [(4096, 4096), (11008, 4096), (4096, 11008)])
def test_full_size_prefill_on_the_packed_kernel(dev, oracle, N, K):
    """BASELINE configs[2] on the pre-packed kernel: M = 512, 128 of the rows against the oracle (threaded over weight rows),
    the duplicate-row property on all of them, and the floor-reliance report of the round-1 verdict."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    G, M = 128, 512
    g = torch.Generator(device=dev).manual_seed(177 + N)
    lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), G).prepack()
    xh = torch.empty(M // 2, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    x = torch.cat([xh, xh], dim=0).contiguous()
    y = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
    what = capi.describe_dispatch(lin.desc(x, y))
    # N = 4096: 128 tiles, every tile's k range cut in two (K = 4096) or four (K = 11008: round 4); N = 11008: 344 tiles, the 88 past the first 256 cut in two (one run per CU at most)
    # N = 11008 (round 5): 128 x 192 tiles -- 232 of them for the 256 CUs --, two quartets alternating a tile's k-blocks (until then: 344 tiles of 128 x 128, the 88 past the first 256 cut in two)
    assert what.startswith("gemm-pk") and (("quartets=2 ksplit=2 " in what) if N == 4096 else "tile=128x192 wave=128x48 quartets=2" in what), what
    for rep in range(3):  # (the scratch counters must be back to zero after every call)
        y.fill_(float("nan"))
        lin.forward(x, y)
        torch.cuda.synchronize()
        if rep:
            assert torch.equal(y, y_first)
        y_first = y.clone()
    from tinychatengine_amd.linear import gemm_scratch
    assert int(gemm_scratch(dev)[:4096].to(torch.int32).sum().item()) == 0
    assert torch.equal(y[: M // 2], y[M // 2:]), "duplicate rows must produce identical outputs"
    rows = [r for r in range(0, 512) if (r % 4) == ((r // 64) % 4)]
    ref32 = oracle.w4a16_gemv_q4_6_mt(x[rows].cpu().numpy(), lin.weight.cpu().numpy().view(np.uint32), lin.scale.cpu().numpy(),
                                      lin.zero_point.cpu().numpy().view(np.uint32), len(rows), N, K, G)
    got = y[rows].cpu().numpy()
    ok, worst = w4a16_close(got, ref32)
    assert ok, worst
    rep = w4a16_report(got, ref32)
    record_parity(f"prefill M=512 {N}x{K} pre-packed kernel (128 rows)", rep)
    assert rep["frac_over_plain"] < 0.01 and rep["worst_plain"] <= 1.0, rep


@pytest.mark.parametrize("M,N,K", [(2048, 4096, 4096), (512, 11008, 4096), (1024, 4096, 11008)])
def test_wide_form_against_the_narrow_form_at_full_size(dev, oracle, M, N, K):
    """Round 5, the wide form (128 rows x 64 columns per wave; modes 2670 / 2671 / 2672): one quartet per 128 x 256 tile walks the k-blocks in the order the 128 x 128
    form does, with the same rescale sequence -- every output BIT-identical to form 1's (which the full-size tests above hold to the oracle); two quartets alternating
    the k-blocks and the k range cut across workgroups add partial sums in a fixed order: close to it, identical from call to call; 64 rows against the oracle."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4, gemm_scratch
    L = capi.lib()
    g = torch.Generator(device=dev).manual_seed(91 + N + M)
    lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack()
    assert lin.zeros_are_8
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    outs = {}
    try:
        for mode in (61, 2670, 2671, 2672, 2673, 2674, 2675):
            capi.check(L.tce_w4a16_set_debug_mode(mode))
            ys = []
            for rep in range(2):
                y = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
                d = lin.desc(x, y)
                if mode != 61:
                    assert ("wave=128x48" if mode in (2673, 2674) else "wave=128x64") in capi.describe_dispatch(d), capi.describe_dispatch(d)
                capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                ys.append(y)
            assert torch.equal(ys[0], ys[1]), f"mode {mode}: two calls differ"
            assert not torch.isnan(ys[0].float()).any(), f"mode {mode}: unwritten outputs"
            outs[mode] = ys[0]
    finally:
        L.tce_w4a16_set_debug_mode(60)
    assert int(gemm_scratch(dev)[:4096].to(torch.int32).sum().item()) == 0
    assert torch.equal(outs[2670], outs[61]), "one quartet per wide tile: the same sums in the same order as the 128 x 128 form"
    assert torch.equal(outs[2673], outs[61]), "one quartet per 128 x 192 tile: likewise"
    assert torch.equal(outs[2675], outs[61]), "128 x 512 tiles, two quartets side by side: likewise"
    rows = list(range(0, M, M // 64))
    ref32 = oracle.w4a16_gemv_q4_6_mt(x[rows].cpu().numpy(), lin.weight.cpu().numpy().view(np.uint32), lin.scale.cpu().numpy(),
                                      lin.zero_point.cpu().numpy().view(np.uint32), len(rows), N, K, 128)
    for mode in (2670, 2671, 2672, 2673, 2674, 2675):
        ok, worst = w4a16_close(outs[mode][rows].cpu().numpy(), ref32)
        assert ok, f"mode {mode}: worst |err|/tol = {worst:.3f}"


def test_two_run_k_cut_as_a_directed_hand_off(dev, oracle):
    """Round 5: a k range cut in TWO runs is a hand-off -- run 0 (two k-blocks shorter) writes its tile through and raises the tile's counter, run 1 adds it to its own in run
    order -- instead of both meeting at the counter.  Same arithmetic as the last-arriver form with the cut moved by a k-block: against the oracle, repeatable bit for
    bit, the counters back at zero, and the A/B switch (mode 694) still gives the old form."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4, gemm_scratch
    L = capi.lib()
    for (M, N, K) in ((512, 4096, 4096), (300, 1000, 1408), (512, 2048, 1024)):
        g = torch.Generator(device=dev).manual_seed(7 + N)
        lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack()
        x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
        rows = list(range(0, M, max(1, M // 48)))
        ref32 = oracle.w4a16_gemv_q4_6_mt(x[rows].cpu().numpy(), lin.weight.cpu().numpy().view(np.uint32), lin.scale.cpu().numpy(), lin.zero_point.cpu().numpy().view(np.uint32), len(rows), N, K, 128)
        try:
            capi.check(L.tce_w4a16_set_debug_mode(642))  # every tile's k range cut in two
            outs = {}
            for sw in (695, 694):
                capi.check(L.tce_w4a16_set_debug_mode(sw))
                ys = []
                for rep in range(3):
                    y = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
                    d = lin.desc(x, y)
                    assert "ksplit=2" in capi.describe_dispatch(d), capi.describe_dispatch(d)
                    capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
                    torch.cuda.synchronize()
                    ys.append(y)
                assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]), (sw, M, N, K)
                assert int(gemm_scratch(dev)[:4096].to(torch.int32).sum().item()) == 0
                ok, worst = w4a16_close(ys[0][rows].cpu().numpy(), ref32)
                assert ok, f"switch {sw} {M}x{N}x{K}: worst |err|/tol = {worst:.3f}"
                outs[sw] = ys[0]
        finally:
            L.tce_w4a16_set_debug_mode(695)
            L.tce_w4a16_set_debug_mode(60)


@pytest.mark.parametrize("mode,what", [(65, "two runs, directed hand-off"), (694, "two runs through the last arriver")])
def test_a_failed_exchange_is_loud_and_sticky(dev, oracle, mode, what):
    """Round 6: a k-cut exchange that gives up must not return a plausible number.  The state such a fault leaves behind -- the tile's counter word
    poisoned -- is planted by hand (the wait itself has never been seen to run out): every launch that meets it stores NaN for that tile, counts a fault
    (tce_w4a16_gemm_scratch_faults), and keeps doing so until the host zeroes the area's first 4096 bytes; the other tiles and, after the clearing, all tiles are the
    uncut kernel's."""
    import ctypes as C
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4, gemm_scratch
    L = capi.lib()
    M, N, K = 512, 4096, 4096
    g = torch.Generator(device=dev).manual_seed(77)
    lin = Linear_half_int4.from_float(torch.empty(N, K, device=dev).normal_(0, 0.02, generator=g), 128).prepack()
    x = torch.empty(M, K, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    scratch = gemm_scratch(dev)
    st = torch.cuda.current_stream().cuda_stream

    def faults():
        n = C.c_uint32(0)
        capi.check(L.tce_w4a16_gemm_scratch_faults(C.c_void_p(scratch.data_ptr()), C.c_void_p(st), C.byref(n)))
        return n.value

    def run():
        y = torch.full((M, N), 7.0, dtype=torch.float16, device=dev)
        capi.check(capi.w4a16_forward(lin.desc(x, y), st))
        torch.cuda.synchronize()
        return y
    try:
        if mode == 694:
            capi.check(L.tce_w4a16_set_debug_mode(694))
        assert "ksplit=2" in capi.describe_dispatch(lin.desc(x, torch.empty((M, N), dtype=torch.float16, device=dev)))
        assert faults() == 0
        good = run()
        assert not torch.isnan(good.float()).any() and faults() == 0
        words = scratch[:4096].view(torch.int32)
        words[3] = -2147483648  # 0x80000000: the poison a run 1 leaves when its wait runs out
        torch.cuda.synchronize()
        for rep in range(2):
            bad = run()
            nan = torch.isnan(bad.float())
            assert nan.any(), f"{what}: a poisoned counter must not yield a finite tile"
            rows, cols = nan.any(dim=1).nonzero().flatten(), nan.any(dim=0).nonzero().flatten()
            assert rows.numel() == 128 and cols.numel() == 128 and bool(nan[rows[0]:rows[-1] + 1, cols[0]:cols[-1] + 1].all()), "exactly one 128 x 128 tile"
            assert torch.equal(bad[~nan], good[~nan]), "the other tiles are untouched"
            assert faults() >= rep + 1
        words.zero_()  # the host's recovery
        torch.cuda.synchronize()
        assert torch.equal(run(), good) and faults() == 0
    finally:
        scratch[:4096].zero_()
        L.tce_w4a16_set_debug_mode(695)
        torch.cuda.synchronize()
