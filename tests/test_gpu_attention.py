"""GPU parity of the attention operators around the int4 linears against the oracle's restatement of the reference's CUDA
arithmetic (PARITY UNPINNED: those kernels exist only as CUDA; the binary16 fma primitive itself is pinned to exact rational arithmetic in
tests/test_oracle.py).  BMM_F16T: bit-exact.  softmax: exact except where the device exponential and libm differ in the last float bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from tinychatengine_amd import capi
    capi.lib()
    return torch.device("cuda:0")


# (batch = heads, M = query rows, N = keys, K = head_dim): decode q k^T, decode p v on the transposed V (K = keys), prefill-like, odd sizes
BMM_SHAPES = [(32, 1, 777, 128), (32, 1, 128, 777), (4, 9, 9, 128), (3, 5, 70, 96), (2, 3, 11, 13), (1, 1, 1, 1), (2, 17, 300, 64)]


@pytest.mark.parametrize("batch,M,N,K", BMM_SHAPES)
def test_bmm_f16t_bit_exact(dev, oracle, batch, M, N, K):
    from tinychatengine_amd.attention_ops import BMM_F16T
    rng = np.random.default_rng(batch * 1000 + M + N + K)
    A = (rng.standard_normal((batch, M, K)) * 0.7).astype(np.float16)
    B = (rng.standard_normal((batch, N, K)) * 0.7).astype(np.float16)
    A[0, 0, : min(K, 3)] = np.float16(6e-5)  # subnormal products on the way
    for alpha in (0.08838834764831845, 1.0):  # 1/sqrt(128) as alpha_half.bin stores it; the p v product's 1.0
        want = oracle.bmm_f16t(A, B, np.float16(alpha))
        c = torch.full((batch, M, N), float("nan"), dtype=torch.float16, device=dev)
        BMM_F16T(alpha).forward(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), c)
        torch.cuda.synchronize()
        got = c.cpu().numpy()
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), f"alpha {alpha}: {(got.view(np.uint16) != want.view(np.uint16)).sum()} of {got.size} differ"


def test_bmm_f16t_unaligned_rows_take_the_scalar_path(dev, oracle):
    """K % 8 != 0 or a base that is not 16-byte aligned: same bits through the element-wise loop."""
    from tinychatengine_amd import capi
    import ctypes as C
    rng = np.random.default_rng(5)
    batch, M, N, K = 2, 3, 40, 72
    A = (rng.standard_normal((batch, M, K))).astype(np.float16)
    B = (rng.standard_normal((batch, N, K))).astype(np.float16)
    want = oracle.bmm_f16t(A, B, np.float16(0.5))
    bufa = torch.zeros(A.size + 1, dtype=torch.float16, device=dev)
    bufa[1:] = torch.from_numpy(A.reshape(-1)).to(dev)  # offset by 2 bytes
    tb = torch.from_numpy(B).to(dev)
    c = torch.empty((batch, M, N), dtype=torch.float16, device=dev)
    alpha_bits = int(np.array([0.5], np.float16).view(np.uint16)[0])
    capi.check(capi.lib().tce_bmm_f16t(C.c_void_p(bufa.data_ptr() + 2), C.c_void_p(tb.data_ptr()), C.c_void_p(c.data_ptr()), batch, M, N, K, alpha_bits, None))
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy().view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("rows,n", [(32, 777), (32, 1), (5, 64), (3, 2049), (64, 130)])
def test_softmax_half_matches_oracle(dev, oracle, rows, n):
    from tinychatengine_amd.attention_ops import softmax
    rng = np.random.default_rng(rows + n)
    x = (rng.standard_normal((rows, n)) * 4).astype(np.float16)
    x[0, :] = x[0, 0]            # a constant row
    if n > 2:
        x[-1, 1] = np.float16(-60000.0)  # an entry that underflows to 0
    want = oracle.softmax_half(x)
    got = softmax(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert not np.isnan(got.astype(np.float32)).any()
    diff = got.view(np.uint16).astype(np.int32) - want.view(np.uint16).astype(np.int32)
    # expf on the device vs libm: a last-bit difference in float moves the binary16 exponential only when it sits on a rounding boundary; the
    # sum then moves by at most an ulp and with it every quotient of that row
    assert np.abs(diff).max() <= 2, f"worst difference {np.abs(diff).max()} binary16 steps"
    assert (diff != 0).mean() <= 0.35, f"{(diff != 0).mean():.3f} of the elements differ"
    assert np.abs(got.astype(np.float64).sum(axis=1) - 1.0).max() < 0.05


@pytest.mark.parametrize("heads,keys,hd,masked", [(32, 777, 128, False), (32, 2048, 128, True), (4, 1, 128, False), (3, 70, 96, True), (2, 33, 20, False)])
def test_fused_decode_attention_is_the_composition(dev, oracle, heads, keys, hd, masked):
    """tce_attention_decode_f16 == tce_bmm_f16t -> (+ mask, inf scrub) -> tce_softmax_half -> tce_bmm_f16t bit for bit (same operations, same
    order, one launch), and its two products equal the oracle's BMM on the same inputs."""
    from tinychatengine_amd.attention_ops import BMM_F16T, attention_decode, softmax
    rng = np.random.default_rng(heads * 7 + keys + hd)
    q = (rng.standard_normal((heads, hd)) * 1.5).astype(np.float16)
    K = (rng.standard_normal((heads, keys, hd)) * 1.5).astype(np.float16)
    V = (rng.standard_normal((heads, keys, hd))).astype(np.float16)
    Vt = np.ascontiguousarray(V.transpose(0, 2, 1))
    alpha = float(np.float16(1.0 / np.sqrt(hd)))
    mask = None
    if masked:
        mask = np.zeros(keys, np.float16)
        mask[keys // 3] = np.float16(-65504.0)  # a masked key (what prepare_decoder_attention_mask writes)
    tq, tK, tVt = (torch.from_numpy(x).to(dev) for x in (q, K, Vt))
    tm = torch.from_numpy(mask).to(dev) if masked else None
    fused = attention_decode(tq, tK, tVt, torch.empty((heads, hd), dtype=torch.float16, device=dev), alpha, tm)
    # the same thing as separate launches
    s = BMM_F16T(alpha).forward(tq.view(heads, 1, hd), tK, torch.empty((heads, 1, keys), dtype=torch.float16, device=dev))
    assert np.array_equal(s.cpu().numpy().view(np.uint16), oracle.bmm_f16t(q.reshape(heads, 1, hd), K, np.float16(alpha)).view(np.uint16))
    if masked:
        s = s + tm.view(1, 1, keys)  # __hadd
    s = torch.where(torch.isfinite(s), s, torch.full_like(s, -65504.0))  # check_inf_half
    p = softmax(s.contiguous())
    o = BMM_F16T(1.0).forward(p, tVt, torch.empty((heads, 1, hd), dtype=torch.float16, device=dev))
    torch.cuda.synchronize()
    assert torch.equal(fused.view(-1), o.view(-1)), f"{(fused.view(-1) != o.view(-1)).sum().item()} of {fused.numel()} outputs differ"
    want = oracle.bmm_f16t(p.cpu().numpy(), Vt, np.float16(1.0))  # the oracle's product on the device's probabilities
    assert np.array_equal(fused.cpu().numpy().reshape(-1).view(np.uint16), want.reshape(-1).view(np.uint16))
    if masked:
        assert float(p.view(heads, keys)[:, keys // 3].abs().max()) == 0.0


@pytest.mark.parametrize("heads,ln,hd,start", [(32, 1, 128, 77), (32, 9, 128, 0), (4, 3, 64, 5), (2, 2, 510, 1)])
def test_rope_bit_exact(dev, oracle, heads, ln, hd, start):
    from tinychatengine_amd.attention_ops import rotary_pos_emb
    rng = np.random.default_rng(heads + ln + hd + start)
    q = rng.standard_normal((heads, ln, hd)).astype(np.float16)
    k = rng.standard_normal((heads, ln, hd)).astype(np.float16)
    pos = np.arange(start + ln + 3)[:, None] * (10000.0 ** (-np.arange(0, hd, 2) / hd))[None, :]
    cos = np.concatenate([np.cos(pos), np.cos(pos)], axis=1).astype(np.float16)  # the tables the reference precomputes
    sin = np.concatenate([np.sin(pos), np.sin(pos)], axis=1).astype(np.float16)
    wq, wk = oracle.rope_half(q, k, cos, sin, start)
    tq, tk = torch.from_numpy(q).to(dev), torch.from_numpy(k).to(dev)
    rotary_pos_emb(tq, tk, torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev), start)
    torch.cuda.synchronize()
    assert np.array_equal(tq.cpu().numpy().view(np.uint16), wq.view(np.uint16)) and np.array_equal(tk.cpu().numpy().view(np.uint16), wk.view(np.uint16))


# ---------------------------------------------------------------------------------------------------------------------------------
# tce_attention_decode_step_f16: the whole decode step of the attention block as one launch, fp32 arithmetic (csrc/attention_fast.hip)
# ---------------------------------------------------------------------------------------------------------------------------------
def _rope_tables(positions, hd, seed):
    rng = np.random.default_rng(seed)  # any table: the kernel applies what it is given (the reference loads cos.bin / sin.bin)
    ang = rng.uniform(0, 2 * np.pi, (positions, hd))
    return np.cos(ang).astype(np.float16), np.sin(ang).astype(np.float16)


def _attention_reference_f64(q_rot, K, V, alpha, mask):
    """softmax(alpha * q K^T + mask) V per head in float64 on binary16 inputs; scores outside binary16 range -> -65504 (check_inf_half)."""
    s = alpha * np.einsum("hd,hkd->hk", q_rot.astype(np.float64), K.astype(np.float64))
    if mask is not None:
        s = s + mask.astype(np.float64)[None, :]
    s = np.where(np.abs(s) <= 65504.0, s, -65504.0)
    s = s - s.max(axis=1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(axis=1, keepdims=True)
    return np.einsum("hk,hkd->hd", p, V.astype(np.float64))


@pytest.mark.parametrize("heads,max_keys,steps,start,cut", [(32, 64, 40, 0, 0), (8, 2048, 3, 2045, 0), (8, 2048, 3, 2045, 1024), (32, 2048, 2, 1000, 0),
                                                             (4, 300, 5, 250, 0), (32, 700, 3, 600, 0), (4, 300, 3, 100, 64)])
def test_attention_decode_step_against_float64_and_the_binary16_chain_kernel(dev, oracle, heads, max_keys, steps, start, cut):
    """cut = 0: the fitted rule picks the chunks (one per head up to 320 keys, four up to 640, eight beyond); otherwise the key range is cut
    for that many workgroups (8 heads, 1024 workgroups: 32 chunks, the combine's one-by-one tail past 16 chunks)."""
    from tinychatengine_amd.attention_ops import DecodeAttention, attention_decode
    from tinychatengine_amd import capi
    capi.check(capi.lib().tce_w4a16_set_debug_mode(3000 + cut))
    try:
        _attention_step_case(dev, oracle, heads, max_keys, steps, start)
    finally:
        capi.check(capi.lib().tce_w4a16_set_debug_mode(3000))


@pytest.mark.parametrize("waves", [8, 16])
def test_attention_decode_step_wider_workgroups(dev, oracle, waves):
    """The 8- and 16-wave forms of the step kernel (kept for the sweep that ruled them out) compute the same thing."""
    from tinychatengine_amd import capi
    capi.check(capi.lib().tce_w4a16_set_debug_mode(2900 + waves))
    try:
        _attention_step_case(dev, oracle, 32, 700, 2, 600)
        _attention_step_case(dev, oracle, 4, 300, 2, 100)
    finally:
        capi.check(capi.lib().tce_w4a16_set_debug_mode(2900))


def _attention_step_case(dev, oracle, heads, max_keys, steps, start):
    from tinychatengine_amd.attention_ops import DecodeAttention, attention_decode
    hd = 128
    rng = np.random.default_rng(heads + max_keys + start)
    cos, sin = _rope_tables(max_keys, hd, 5)
    att = DecodeAttention(heads, hd, max_keys, dev, torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev))
    # a past of `start` keys already in the caches (rotated keys, as the reference would have appended them)
    Kc = np.zeros((heads, max_keys, hd), np.float16)
    Vc = np.zeros((heads, max_keys, hd), np.float16)
    if start:
        Kc[:, :start] = (rng.standard_normal((heads, start, hd)) * 0.8).astype(np.float16)
        Vc[:, :start] = (rng.standard_normal((heads, start, hd)) * 0.8).astype(np.float16)
        att.k_cache.copy_(torch.from_numpy(Kc))
        att.v_cache.copy_(torch.from_numpy(Vc))
    alpha = float(np.float16(1.0 / np.sqrt(hd)))
    for t in range(steps):
        pos = start + t
        qkv = (rng.standard_normal((3, heads, hd)) * 0.9).astype(np.float16)
        mask = None
        if t % 2 == 1:  # an additive mask with a few keys switched off, as a padding mask would
            mask = np.zeros(pos + 1, np.float16)
            mask[rng.integers(0, pos + 1, size=max(1, (pos + 1) // 7))] = np.float16(-65504.0)
            mask[pos] = 0
        out = att.step(torch.from_numpy(qkv.reshape(-1)).to(dev), pos, mask=None if mask is None else torch.from_numpy(mask).to(dev))
        torch.cuda.synchronize()
        # the reference's RoPE (binary16 arithmetic) on q and the new key: the oracle's restatement of RotaryPosEmb_cuda_forward
        q_rot, k_rot = oracle.rope_half(qkv[0][:, None, :], qkv[1][:, None, :], cos, sin, pos)
        Kc[:, pos] = k_rot[:, 0]
        Vc[:, pos] = qkv[2]
        assert np.array_equal(att.k_cache[:, pos].cpu().numpy().view(np.uint16), Kc[:, pos].view(np.uint16)), "appended key differs from the reference's rotated key"
        assert np.array_equal(att.v_cache[:, pos].cpu().numpy().view(np.uint16), Vc[:, pos].view(np.uint16))
        ref = _attention_reference_f64(q_rot[:, 0], Kc[:, : pos + 1], Vc[:, : pos + 1], alpha, mask)
        got = out.cpu().numpy().astype(np.float64)
        tol = 2e-3 * np.abs(ref).max(axis=1, keepdims=True) + 2.0 ** -11 * np.abs(ref)
        assert np.all(np.abs(got - ref) <= tol), f"step {t}: worst |err|/tol = {(np.abs(got - ref) / tol).max():.3f}"
        if t == steps - 1:  # and against the bit-exact (binary16-chain) form of the same block, on the same rotated inputs
            K_t = torch.from_numpy(np.ascontiguousarray(Kc[:, : pos + 1])).to(dev)
            Vt_t = torch.from_numpy(np.ascontiguousarray(Vc[:, : pos + 1].transpose(0, 2, 1))).to(dev)
            exact = torch.empty((heads, hd), dtype=torch.float16, device=dev)
            attention_decode(torch.from_numpy(np.ascontiguousarray(q_rot[:, 0])).to(dev), K_t, Vt_t, exact, alpha,
                             mask=None if mask is None else torch.from_numpy(mask).to(dev))
            torch.cuda.synchronize()
            e = exact.cpu().numpy().astype(np.float64)
            # the chain kernel accumulates in binary16 (the reference's arithmetic): its own distance from float64 is the larger one
            # (binary16 running sums over `keys` terms: ~1 % of the largest output per 1000 keys)
            tol2 = (2e-2 + 2e-5 * (pos + 1)) * np.abs(ref).max(axis=1, keepdims=True)
            assert np.all(np.abs(got - e) <= tol2), f"fp32 form vs binary16-chain form: {(np.abs(got - e) / tol2).max():.3f}"


def test_attention_decode_step_chunk_merge_under_repetition(dev, oracle):
    """The chunk merge crosses XCDs (the workgroups of a head sit on different L2s): partial results are stored write-through, a counter
    elects the last workgroup, which reads them back with coherent loads.  200 back-to-back steps on ONE workspace, the values of every
    past key and of q changing between steps, no host synchronisation inside a batch of 25: a partial that was read before it landed, or a
    stale one from the previous step, would show up as a wrong head (the float64 reference is recomputed per step)."""
    from tinychatengine_amd.attention_ops import DecodeAttention
    heads, hd, max_keys = 32, 128, 2048
    rng = np.random.default_rng(99)
    att = DecodeAttention(heads, hd, max_keys, dev, None, None)  # no rotation: the reference below is then plain numpy
    alpha = float(np.float16(1.0 / np.sqrt(hd)))
    pos = max_keys - 1
    for batch in range(8):
        cases = []
        for i in range(25):
            K = (rng.standard_normal((heads, max_keys, hd)) * 0.8).astype(np.float16)
            V = (rng.standard_normal((heads, max_keys, hd)) * 0.8).astype(np.float16)
            qkv = (rng.standard_normal((3, heads, hd)) * 0.9).astype(np.float16)
            cases.append((K, V, qkv))
        outs = []
        for (K, V, qkv) in cases:  # enqueue everything, synchronise once
            att.k_cache.copy_(torch.from_numpy(K).to(dev), non_blocking=True)
            att.v_cache.copy_(torch.from_numpy(V).to(dev), non_blocking=True)
            outs.append(att.step(torch.from_numpy(qkv.reshape(-1)).to(dev), pos).clone())
        torch.cuda.synchronize()
        for i, ((K, V, qkv), out) in enumerate(zip(cases, outs)):
            K = K.copy(); V = V.copy()
            K[:, pos] = qkv[1]
            V[:, pos] = qkv[2]
            ref = _attention_reference_f64(qkv[0], K, V, alpha, None)
            got = out.cpu().numpy().astype(np.float64)
            tol = 2e-3 * np.abs(ref).max(axis=1, keepdims=True) + 2.0 ** -11 * np.abs(ref)
            assert np.all(np.abs(got - ref) <= tol), f"batch {batch} step {i}: worst |err|/tol = {(np.abs(got - ref) / tol).max():.3f}"


def test_attention_decode_step_random_contexts(dev, oracle):
    """Forty random (heads, capacity, position, mask) cases, one step each on a cache filled up to the position: every chunk rule, key runs
    that are no multiple of anything, the token's own row at the end of a chunk, at its start and alone (position 0)."""
    from tinychatengine_amd.attention_ops import DecodeAttention
    hd = 128
    rng = np.random.default_rng(2024)
    cases = [(32, 321, 320), (32, 322, 321), (32, 1025, 1024), (32, 1026, 1025), (5, 1, 0), (1, 17, 16), (40, 4100, 4099)]
    while len(cases) < 40:
        heads = int(rng.choice([1, 3, 8, 32, 40]))
        max_keys = int(rng.integers(1, 3000))
        cases.append((heads, max_keys, int(rng.integers(0, max_keys))))
    for (heads, max_keys, pos) in cases:
        cos, sin = _rope_tables(max_keys, hd, 7)
        att = DecodeAttention(heads, hd, max_keys, dev, torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev))
        Kc = np.zeros((heads, max_keys, hd), np.float16)
        Vc = np.zeros((heads, max_keys, hd), np.float16)
        Kc[:, :pos] = (rng.standard_normal((heads, pos, hd)) * 0.8).astype(np.float16)
        Vc[:, :pos] = (rng.standard_normal((heads, pos, hd)) * 0.8).astype(np.float16)
        # rows past the position hold garbage the kernel must not look at
        Kc[:, pos:] = np.float16(300.0)
        Vc[:, pos:] = np.float16(-300.0)
        att.k_cache.copy_(torch.from_numpy(Kc))
        att.v_cache.copy_(torch.from_numpy(Vc))
        qkv = (rng.standard_normal((3, heads, hd)) * 0.9).astype(np.float16)
        mask = None
        if rng.integers(0, 2):
            mask = np.zeros(pos + 1, np.float16)
            mask[rng.integers(0, pos + 1, size=max(1, (pos + 1) // 5))] = np.float16(-65504.0)
            mask[pos] = 0
        alpha = float(np.float16(1.0 / np.sqrt(hd)))
        out = att.step(torch.from_numpy(qkv.reshape(-1)).to(dev), pos, mask=None if mask is None else torch.from_numpy(mask).to(dev))
        torch.cuda.synchronize()
        q_rot, k_rot = oracle.rope_half(qkv[0][:, None, :], qkv[1][:, None, :], cos, sin, pos)
        Kc[:, pos] = k_rot[:, 0]
        Vc[:, pos] = qkv[2]
        assert np.array_equal(att.k_cache[:, pos].cpu().numpy().view(np.uint16), Kc[:, pos].view(np.uint16)), (heads, max_keys, pos)
        if pos + 1 < max_keys:  # nothing behind the position was written
            assert float(att.k_cache[:, pos + 1:].float().min()) == 300.0 and float(att.v_cache[:, pos + 1:].float().max()) == -300.0, (heads, max_keys, pos)
        ref = _attention_reference_f64(q_rot[:, 0], Kc[:, : pos + 1], Vc[:, : pos + 1], alpha, mask)
        got = out.cpu().numpy().astype(np.float64)
        tol = 2e-3 * np.abs(ref).max(axis=1, keepdims=True) + 2.0 ** -11 * np.abs(ref)
        assert np.all(np.abs(got - ref) <= tol), f"heads {heads} capacity {max_keys} position {pos} mask {mask is not None}: worst |err|/tol = {(np.abs(got - ref) / tol).max():.3f}"


@pytest.mark.parametrize("heads,kv_heads,max_keys,start,steps", [(32, 8, 2048, 2040, 3), (32, 8, 700, 600, 2), (8, 2, 300, 0, 6), (16, 8, 1100, 1000, 2),
                                                                 (32, 8, 257, 255, 2), (4, 1, 4200, 4190, 2)])
@pytest.mark.parametrize("fuse", [0, 2, 4])
def test_attention_decode_step_grouped_queries(dev, oracle, heads, kv_heads, max_keys, start, steps, fuse):
    """Grouped-query attention (Llama-3-8B: 32 query heads over 8 key / value heads, model.h:83): query head i reads key / value head
    i // (heads // kv_heads) -- the reference's `repeat` (non_cuda/Int4llamaAttention.cc:166-185).  Checked against float64 on the repeated
    caches, against the multi-head kernel run on physically repeated caches (same arithmetic per query head: bit-identical where both
    use one chunk, tolerance-level where their chunkings differ), and the appended rows against the reference's RoPE."""
    from tinychatengine_amd.attention_ops import DecodeAttention
    from tinychatengine_amd import capi
    hd, rep = 128, heads // kv_heads
    if fuse and rep % fuse:
        pytest.skip("the fused form needs heads / kv_heads to be a multiple of the heads per workgroup")
    capi.check(capi.lib().tce_w4a16_set_debug_mode(2920 + fuse))  # 0: the rule (one query head per workgroup); 2 / 4: fused
    try:
        _gqa_case(dev, oracle, heads, kv_heads, max_keys, start, steps)
    finally:
        capi.check(capi.lib().tce_w4a16_set_debug_mode(2920))


def _gqa_case(dev, oracle, heads, kv_heads, max_keys, start, steps):
    from tinychatengine_amd.attention_ops import DecodeAttention
    hd, rep = 128, heads // kv_heads
    rng = np.random.default_rng(heads * 7 + max_keys)
    cos, sin = _rope_tables(max_keys, hd, 5)
    tc, ts = torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev)
    gqa = DecodeAttention(heads, hd, max_keys, dev, tc, ts, kv_heads=kv_heads)
    mha = DecodeAttention(heads, hd, max_keys, dev, tc, ts)
    Kc = np.zeros((kv_heads, max_keys, hd), np.float16)
    Vc = np.zeros((kv_heads, max_keys, hd), np.float16)
    Kc[:, :start] = (rng.standard_normal((kv_heads, start, hd)) * 0.8).astype(np.float16)
    Vc[:, :start] = (rng.standard_normal((kv_heads, start, hd)) * 0.8).astype(np.float16)
    # behind the position: bit patterns of inf / nan (an uninitialised cache), which must never reach the result
    Kc[:, start:] = np.array([0x7C00, 0x7E00, 0xFC00, 0x7FFF], np.uint16).view(np.float16)[rng.integers(0, 4, (kv_heads, max_keys - start, hd))]
    Vc[:, start:] = np.array([0x7C00, 0x7E00, 0xFC00, 0x7FFF], np.uint16).view(np.float16)[rng.integers(0, 4, (kv_heads, max_keys - start, hd))]
    gqa.k_cache.copy_(torch.from_numpy(Kc)); gqa.v_cache.copy_(torch.from_numpy(Vc))
    mha.k_cache.copy_(torch.from_numpy(np.repeat(Kc, rep, axis=0))); mha.v_cache.copy_(torch.from_numpy(np.repeat(Vc, rep, axis=0)))
    alpha = float(np.float16(1.0 / np.sqrt(hd)))
    for t in range(steps):
        pos = start + t
        q = (rng.standard_normal((heads, hd)) * 0.9).astype(np.float16)
        kv = (rng.standard_normal((2, kv_heads, hd)) * 0.9).astype(np.float16)
        mask = None
        if t % 2 == 1:
            mask = np.zeros(pos + 1, np.float16)
            mask[rng.integers(0, pos + 1, size=max(1, (pos + 1) // 7))] = np.float16(-65504.0)
            mask[pos] = 0
        tm = None if mask is None else torch.from_numpy(mask).to(dev)
        row = np.concatenate([q.reshape(-1), kv[0].reshape(-1), kv[1].reshape(-1)])
        out = gqa.step(torch.from_numpy(row).to(dev), pos, mask=tm)
        row_mha = np.concatenate([q.reshape(-1), np.repeat(kv[0], rep, axis=0).reshape(-1), np.repeat(kv[1], rep, axis=0).reshape(-1)])
        out_mha = mha.step(torch.from_numpy(row_mha).to(dev), pos, mask=tm)
        torch.cuda.synchronize()
        q_rot, _ = oracle.rope_half(q[:, None, :], q[:, None, :], cos, sin, pos)
        _, k_rot = oracle.rope_half(kv[0][:, None, :], kv[0][:, None, :], cos, sin, pos)
        Kc[:, pos] = k_rot[:, 0]
        Vc[:, pos] = kv[1]
        assert np.array_equal(gqa.k_cache[:, pos].cpu().numpy().view(np.uint16), Kc[:, pos].view(np.uint16)), "appended key differs from the reference's rotated key"
        assert np.array_equal(gqa.v_cache[:, pos].cpu().numpy().view(np.uint16), Vc[:, pos].view(np.uint16))
        if pos + 1 < max_keys:  # nothing behind the position was written
            assert np.array_equal(gqa.k_cache[:, pos + 1:].cpu().numpy().view(np.uint16), Kc[:, pos + 1:].view(np.uint16))
        ref = _attention_reference_f64(q_rot[:, 0], np.repeat(Kc[:, : pos + 1], rep, axis=0), np.repeat(Vc[:, : pos + 1], rep, axis=0), alpha, mask)
        got = out.cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), f"step {t}: inf / nan bits behind the position leaked into the output"
        tol = 2e-3 * np.abs(ref).max(axis=1, keepdims=True) + 2.0 ** -11 * np.abs(ref)
        assert np.all(np.abs(got - ref) <= tol), f"step {t}: worst |err|/tol = {(np.abs(got - ref) / tol).max():.3f}"
        g2 = out_mha.cpu().numpy().astype(np.float64)
        assert np.all(np.abs(got - g2) <= 2 * tol), f"step {t}: grouped vs multi-head kernel on repeated caches: {(np.abs(got - g2) / tol).max():.3f}"


def test_attention_decode_step_ignores_inf_nan_bits_behind_the_position(dev, oracle):
    """The multi-head entry point on a cache whose rows at and behind `pos` hold inf / nan bit patterns (a zero weight times
    an infinite value row is NaN; the kernel now zeroes such rows instead of weighting them)."""
    from tinychatengine_amd.attention_ops import DecodeAttention
    heads, hd, max_keys = 8, 128, 400
    rng = np.random.default_rng(3)
    for pos in (0, 5, 17, 100, 333):
        att = DecodeAttention(heads, hd, max_keys, dev, None, None)
        K = (rng.standard_normal((heads, max_keys, hd)) * 0.8).astype(np.float16)
        V = (rng.standard_normal((heads, max_keys, hd)) * 0.8).astype(np.float16)
        K[:, pos:] = np.array([0x7C00], np.uint16).view(np.float16)[0]
        V[:, pos:] = np.array([0xFE00], np.uint16).view(np.float16)[0]
        att.k_cache.copy_(torch.from_numpy(K)); att.v_cache.copy_(torch.from_numpy(V))
        qkv = (rng.standard_normal((3, heads, hd)) * 0.9).astype(np.float16)
        out = att.step(torch.from_numpy(qkv.reshape(-1)).to(dev), pos)
        torch.cuda.synchronize()
        K[:, pos] = qkv[1]; V[:, pos] = qkv[2]
        ref = _attention_reference_f64(qkv[0], K[:, : pos + 1], V[:, : pos + 1], float(np.float16(1.0 / np.sqrt(hd))), None)
        got = out.cpu().numpy().astype(np.float64)
        assert np.isfinite(got).all(), f"pos {pos}"
        tol = 2e-3 * np.abs(ref).max(axis=1, keepdims=True) + 2.0 ** -11 * np.abs(ref)
        assert np.all(np.abs(got - ref) <= tol), f"pos {pos}"


@pytest.mark.parametrize("heads,kv_heads,bound", [(32, 8, 700), (8, 8, 300), (4, 1, 2100)])
def test_attention_decode_step_with_the_position_on_the_device(dev, oracle, heads, kv_heads, bound):
    """tce_attention_decode_step_pos_f16: ONE captured launch (cut for the bound) replayed for growing contexts -- a one-element device tensor holds
    the position -- must equal the by-value entry point at every position: the appended rows bit for bit, the outputs within the merge's
    reassociation (the two forms may cut a short context into different chunks)."""
    from tinychatengine_amd.attention_ops import DecodeAttention
    hd = 128
    rng = np.random.default_rng(bound)
    cos, sin = _rope_tables(bound + 1, hd, 5)
    tc, ts = torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev)
    a_dev = DecodeAttention(heads, hd, bound + 1, dev, tc, ts, kv_heads=kv_heads)
    a_val = DecodeAttention(heads, hd, bound + 1, dev, tc, ts, kv_heads=kv_heads)
    K = (rng.standard_normal((kv_heads, bound + 1, hd)) * 0.8).astype(np.float16)
    V = (rng.standard_normal((kv_heads, bound + 1, hd)) * 0.8).astype(np.float16)
    for a_ in (a_dev, a_val):
        a_.k_cache.copy_(torch.from_numpy(K)); a_.v_cache.copy_(torch.from_numpy(V))
    qkv = torch.from_numpy((rng.standard_normal((heads + 2 * kv_heads) * hd) * 0.9).astype(np.float16)).to(dev)
    pos_t = torch.zeros(1, dtype=torch.int32, device=dev)
    out_dev = torch.empty((heads, hd), dtype=torch.float16, device=dev)
    a_dev.step(qkv, bound, out=out_dev, pos_device=pos_t)  # warm-up outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        a_dev.step(qkv, bound, out=out_dev, pos_device=pos_t)
    for pos in (0, 1, 15, 16, 63, 64, 255, 256, 257, bound // 2, bound - 1, bound):
        if pos > bound:
            continue
        for a_ in (a_dev, a_val):  # both start from the same past (a step appends its row at `pos`)
            a_.k_cache.copy_(torch.from_numpy(K)); a_.v_cache.copy_(torch.from_numpy(V))
        pos_t.fill_(pos)
        g.replay()
        want = a_val.step(qkv, pos)
        torch.cuda.synchronize()
        assert torch.equal(a_dev.k_cache, a_val.k_cache) and torch.equal(a_dev.v_cache, a_val.v_cache), f"pos {pos}: caches differ"
        got, ref = out_dev.float().cpu().numpy(), want.float().cpu().numpy()
        tol = 2e-3 * np.abs(ref).max(axis=1, keepdims=True) + 2.0 ** -10 * np.abs(ref)
        assert np.isfinite(got).all() and np.all(np.abs(got - ref) <= tol), f"pos {pos}: worst |err|/tol = {(np.abs(got - ref) / tol).max():.3f}"


def test_attention_decode_step_argument_checks(dev):
    from tinychatengine_amd import capi
    L = capi.lib()
    assert int(L.tce_attention_decode_workspace_bytes(32, 2048, 64)) == 0  # head_dim 128 only
    z = torch.zeros(4096, dtype=torch.float16, device=dev)
    ws = torch.zeros(int(L.tce_attention_decode_workspace_bytes(2, 64, 128)), dtype=torch.uint8, device=dev)
    p = z.data_ptr()
    assert L.tce_attention_decode_step_f16(p, p, p, None, None, None, p, ws.data_ptr(), 2, 128, 64, 64, 0x3C00, None) == capi.TCE_ERR_BAD_ARG  # pos == max_keys
    assert L.tce_attention_decode_step_f16(p, p, p, p, None, None, p, ws.data_ptr(), 2, 128, 64, 0, 0x3C00, None) == capi.TCE_ERR_BAD_ARG  # cos without sin
    assert L.tce_attention_decode_step_gqa_f16(p, p, p, None, None, None, p, ws.data_ptr(), 6, 4, 128, 64, 0, 0x3C00, None) == capi.TCE_ERR_BAD_ARG  # 6 query heads over 4


@pytest.mark.parametrize("heads,kv_heads,max_keys,pos,m,causal,masked", [(8, 8, 256, 0, 100, True, False), (32, 8, 700, 37, 130, True, False), (4, 1, 200, 0, 64, False, True),
                                                                         (4, 4, 300, 50, 7, True, True), (2, 2, 1100, 0, 1024, True, False), (8, 2, 400, 128, 65, False, False),
                                                                         (2, 1, 130, 1, 129, True, False)])
def test_attention_prefill_against_float64_and_the_decode_step(dev, oracle, heads, kv_heads, max_keys, pos, m, causal, masked):
    """tce_attention_prefill_f16 (m new rows on top of `pos` cached keys): the appended keys / values bit for bit against the oracle's RotaryPosEmb
    (RotaryPosEmb.cu:4-34) -- and therefore against what m decode steps append --, the outputs against a float64 evaluation of
    softmax(alpha q K^T + mask) V per row and head (2e-3 * max|out| + one binary16 ulp: the decode step's tolerance), and the LAST row against the decode
    step run on the same caches.  Grouped queries, a context in front of the chunk, m that is not a multiple of the 64-row block or the 64-key tile, an
    explicit additive mask (the reference's attention_mask) with and without the causal cut, inf / nan bit patterns behind pos + m."""
    from tinychatengine_amd.attention_ops import DecodeAttention
    hd, rep = 128, heads // kv_heads
    rng = np.random.default_rng(heads * 7 + m + pos)
    cos, sin = _rope_tables(max_keys, hd, 9)
    tc, ts = torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev)
    att = DecodeAttention(heads, hd, max_keys, dev, tc, ts, kv_heads=kv_heads)
    Kc = np.full((kv_heads, max_keys, hd), np.array([0x7C00], np.uint16).view(np.float16)[0])  # +inf bits where nothing was appended yet
    Vc = np.full((kv_heads, max_keys, hd), np.array([0xFE00], np.uint16).view(np.float16)[0])  # nan bits
    Kc[:, :pos] = (rng.standard_normal((kv_heads, pos, hd)) * 0.8).astype(np.float16)
    Vc[:, :pos] = (rng.standard_normal((kv_heads, pos, hd)) * 0.8).astype(np.float16)
    att.k_cache.copy_(torch.from_numpy(Kc)); att.v_cache.copy_(torch.from_numpy(Vc))
    width = (heads + 2 * kv_heads) * hd
    qkv = (rng.standard_normal((m, width)) * 0.9).astype(np.float16)
    tgz = pos + m
    mask = None
    if masked:  # the reference's attention_mask: causal as the lowest half, plus a few keys switched off for every row (padding)
        mask = np.zeros((m, tgz), np.float16)
        for r in range(m):
            mask[r, pos + r + 1:] = np.float16(-65504.0)
        off = rng.integers(0, max(1, pos), size=3) if pos else []
        for c in off:
            mask[:, c] = np.float16(-65504.0)
    t_qkv = torch.from_numpy(qkv).to(dev)
    out = att.prefill(t_qkv, pos, mask=None if mask is None else torch.from_numpy(mask).to(dev), causal=causal)
    torch.cuda.synchronize()
    q = qkv[:, : heads * hd].reshape(m, heads, hd).transpose(1, 0, 2)
    k = qkv[:, heads * hd: (heads + kv_heads) * hd].reshape(m, kv_heads, hd).transpose(1, 0, 2)
    v = qkv[:, (heads + kv_heads) * hd:].reshape(m, kv_heads, hd).transpose(1, 0, 2)
    q_rot, _ = oracle.rope_half(np.ascontiguousarray(q), np.ascontiguousarray(q), cos, sin, pos)
    _, k_rot = oracle.rope_half(np.ascontiguousarray(k), np.ascontiguousarray(k), cos, sin, pos)
    Kc[:, pos:tgz] = k_rot
    Vc[:, pos:tgz] = v
    assert np.array_equal(att.k_cache.cpu().numpy().view(np.uint16), Kc.view(np.uint16)), "appended keys differ from the reference's rotated keys (or rows outside pos .. pos + m were touched)"
    assert np.array_equal(att.v_cache.cpu().numpy().view(np.uint16), Vc.view(np.uint16))
    alpha = float(np.float16(1.0 / np.sqrt(hd)))
    got = out.cpu().numpy().astype(np.float64).reshape(m, heads, hd)
    assert np.isfinite(got).all()
    Kr, Vr = np.repeat(Kc[:, :tgz], rep, axis=0).astype(np.float64), np.repeat(Vc[:, :tgz], rep, axis=0).astype(np.float64)
    worst = 0.0
    for r in range(m):
        s = alpha * np.einsum("hd,hkd->hk", q_rot[:, r].astype(np.float64), Kr)
        if mask is not None:
            s = s + mask[r].astype(np.float64)[None, :]
        if causal:
            s[:, pos + r + 1:] = -np.inf
        s = s - s.max(axis=1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(axis=1, keepdims=True)
        ref = np.einsum("hk,hkd->hd", p, Vr)
        tol = 2e-3 * np.abs(ref).max(axis=1, keepdims=True) + 2.0 ** -11 * np.abs(ref)
        err = np.abs(got[r] - ref) / tol
        worst = max(worst, err.max())
        assert err.max() <= 1.0, f"row {r}: worst |err|/tol = {err.max():.3f}"
    if causal and not masked:  # the decode step for the last row, on caches that hold everything in front of it
        dec = DecodeAttention(heads, hd, max_keys, dev, tc, ts, kv_heads=kv_heads)
        dec.k_cache.copy_(att.k_cache); dec.v_cache.copy_(att.v_cache)
        one = dec.step(t_qkv[m - 1].contiguous(), tgz - 1)
        torch.cuda.synchronize()
        assert torch.equal(dec.k_cache.view(torch.int16), att.k_cache.view(torch.int16)) and torch.equal(dec.v_cache.view(torch.int16), att.v_cache.view(torch.int16)), \
            "a decode step appends other bits than the prefill did"
        a, b = one.float().cpu().numpy().reshape(heads, hd), got[m - 1]
        tol = 4e-3 * np.abs(b).max(axis=1, keepdims=True) + 2.0 ** -10 * np.abs(b)
        assert np.all(np.abs(a - b) <= tol), f"decode step vs prefill, last row: {(np.abs(a - b) / tol).max():.3f}"


@pytest.mark.parametrize("waves", [4, 8, 14, 18])
def test_attention_prefill_both_block_sizes(dev, oracle, waves):
    """Blocks of 64, 128 and 256 query rows (4 / 8 waves per workgroup with one 16-row tile per wave; 14 / 18: with two; the launch picks by the number of
    blocks) compute the same thing."""
    from tinychatengine_amd import capi
    capi.check(capi.lib().tce_w4a16_set_debug_mode(2950 + waves))
    try:
        test_attention_prefill_against_float64_and_the_decode_step(dev, oracle, 8, 2, 400, 37, 200, True, False)
        test_attention_prefill_against_float64_and_the_decode_step(dev, oracle, 4, 4, 300, 50, 130, True, True)
    finally:
        capi.check(capi.lib().tce_w4a16_set_debug_mode(2950))


def test_attention_prefill_argument_checks(dev):
    from tinychatengine_amd import capi
    L = capi.lib()
    assert int(L.tce_attention_prefill_workspace_bytes(32, 100, 64)) == 0 and int(L.tce_attention_prefill_workspace_bytes(32, 100, 128)) == 32 * 100 * 128 * 2
    z = torch.zeros(1 << 16, dtype=torch.float16, device=dev)
    p = z.data_ptr()
    args = lambda **kw: [kw.get("qkv", p), 0, p, p, None, None, None, 0, 1, p, 0, kw.get("ws", p), kw.get("heads", 4), kw.get("kv", 2), kw.get("hd", 128), 64, kw.get("pos", 0), kw.get("m", 8), 0x2DA8, None]
    assert L.tce_attention_prefill_f16(*args(kv=3)) == capi.TCE_ERR_BAD_ARG
    assert L.tce_attention_prefill_f16(*args(hd=64)) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert L.tce_attention_prefill_f16(*args(pos=60, m=8)) == capi.TCE_ERR_BAD_ARG          # pos + m > max_keys
    assert L.tce_attention_prefill_f16(*args(ws=None)) == capi.TCE_ERR_BAD_ARG
    assert L.tce_attention_prefill_f16(*args(qkv=p + 2)) == capi.TCE_ERR_UNSUPPORTED_SHAPE
