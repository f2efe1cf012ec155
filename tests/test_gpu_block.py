"""A whole decoder block on this library's calls (tinychatengine_amd/decoder_block.py: 5 launches per layer -- fused RMSNorm + q/k/v, the
one-launch attention step, o_proj + residual, fused RMSNorm + gate/up + SiLU*mul, down_proj + residual) against a float64 evaluation
of Int4llamaDecoderLayer::forward's mathematics (Int4llamaDecoderLayer.cu:73-115, Int4llamaAttention.cu:116-229) on the dequantized
weights, token after token with a growing KV cache.  This checks the COMPOSITION -- layouts handed from one call to the next (q | k | v
rows, head-major; interleaved gate / up rows; the cache the attention step appends to; the residual stream updated in place) -- each
call's own arithmetic is held to the oracle elsewhere.  Tolerance: binary16 intermediates, 2 % of the largest activation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _rmsnorm(x, gamma, eps):
    return x / np.sqrt(np.mean(x * x) + eps) * gamma


def _rope(v, cos, sin):  # v [heads][128]; RotaryPosEmb_cuda_forward: v * cos + rotate_half(v) * sin
    half = v.shape[-1] // 2
    rot = np.concatenate([-v[:, half:], v[:, :half]], axis=1)
    return v * cos[None, :] + rot * sin[None, :]


@pytest.mark.parametrize("hidden,heads,kv_heads,ffn,layers,tokens", [(256, 2, 2, 512, 2, 6), (512, 4, 4, 1408, 1, 3), (512, 4, 1, 1408, 2, 5), (1024, 8, 4, 512, 1, 3)])
def test_decoder_blocks_against_float64(hidden, heads, kv_heads, ffn, layers, tokens):
    """kv_heads < heads: grouped-query attention (Llama-3-8B's form, llm/include/model.h:83): the fused projection's row is q | k | v with
    kv_heads key and value heads, query head i attends over key / value head i // (heads // kv_heads)."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.decoder_block import DecoderBlock, dequantize
    assert torch.cuda.is_available()
    capi.lib()
    dev = torch.device("cuda:0")
    max_keys, hd = 64, 128
    rng = np.random.default_rng(hidden + ffn)
    ang = rng.uniform(0, 2 * np.pi, (max_keys, hd // 2))
    cos = np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)  # the reference's tables repeat the half (RotaryPosEmb.cc)
    sin = np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)
    blocks = [DecoderBlock(hidden, heads, ffn, max_keys, dev, torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev), seed=10 + i, kv_heads=kv_heads) for i in range(layers)]
    rep = heads // kv_heads
    W = [{k: dequantize(getattr(b, k)) for k in ("qkv", "o", "gate", "up", "down")} for b in blocks]
    G = [(b.gamma1.cpu().numpy().astype(np.float64), b.gamma2.cpu().numpy().astype(np.float64)) for b in blocks]
    Kc = [np.zeros((kv_heads, 0, hd)) for _ in blocks]
    Vc = [np.zeros((kv_heads, 0, hd)) for _ in blocks]
    alpha = float(np.float16(1.0 / np.sqrt(hd)))
    for pos in range(tokens):
        x0 = rng.standard_normal(hidden).astype(np.float16)
        h_gpu = torch.from_numpy(x0.reshape(1, -1).copy()).to(dev)
        for b in blocks:
            b.step(h_gpu, pos)
        torch.cuda.synchronize()
        h = x0.astype(np.float64)
        c, s = cos[pos].astype(np.float64), sin[pos].astype(np.float64)
        for li in range(layers):
            qkv = W[li]["qkv"] @ _rmsnorm(h, G[li][0], 1e-6)
            kvw = kv_heads * hd
            q = qkv[:hidden].reshape(heads, hd)
            k = qkv[hidden:hidden + kvw].reshape(kv_heads, hd)
            v = qkv[hidden + kvw:].reshape(kv_heads, hd)
            q, k = _rope(q, c, s), _rope(k, c, s)
            Kc[li] = np.concatenate([Kc[li], k[:, None, :]], axis=1)
            Vc[li] = np.concatenate([Vc[li], v[:, None, :]], axis=1)
            sc = alpha * np.einsum("hd,hkd->hk", q, np.repeat(Kc[li], rep, axis=0))  # the reference's `repeat` (Int4llamaAttention.cc:166-185)
            p = np.exp(sc - sc.max(axis=1, keepdims=True))
            p /= p.sum(axis=1, keepdims=True)
            attn = np.einsum("hk,hkd->hd", p, np.repeat(Vc[li], rep, axis=0)).reshape(-1)
            h = h + W[li]["o"] @ attn
            hn = _rmsnorm(h, G[li][1], 1e-6)
            gate, up = W[li]["gate"] @ hn, W[li]["up"] @ hn
            h = h + W[li]["down"] @ (gate / (1.0 + np.exp(-gate)) * up)
        got = h_gpu.cpu().numpy().astype(np.float64).reshape(-1)
        tol = 2e-2 * np.abs(h).max()
        assert np.abs(got - h).max() <= tol, f"token {pos}: max |err| {np.abs(got - h).max():.4f} vs tol {tol:.4f}"


@pytest.mark.parametrize("hidden,heads,kv_heads,ffn,layers,prompt,chunk,prepacked", [(512, 4, 1, 1408, 2, 37, 0, False), (1024, 8, 4, 512, 1, 200, 0, True), (512, 4, 4, 1408, 2, 70, 33, True)])
def test_prefill_blocks_against_token_by_token_decode(hidden, heads, kv_heads, ffn, layers, prompt, chunk, prepacked):
    """DecoderBlock.prefill (m rows at once: RMSNorm, GEMMs, the prefill attention, 10 launches per layer) against the SAME blocks stepped token by token
    through the decode path (5 launches per layer, held to float64 above): the residual streams of all rows within binary16-intermediate tolerance, the caches
    within the linears' tolerance -- the two paths differ in accumulation order (GEMM / GEMV, fp32 attention both), not in mathematics.  `chunk`: the prompt
    enters in two pieces (the second on top of the first's cache); `prepacked`: the 128-row GEMM on the q4_mfma copies."""
    from tinychatengine_amd import capi
    from tinychatengine_amd.decoder_block import DecoderBlock
    capi.lib()
    dev = torch.device("cuda:0")
    max_keys, hd = 256, 128
    rng = np.random.default_rng(hidden + prompt)
    ang = rng.uniform(0, 2 * np.pi, (max_keys, hd // 2))
    cos = torch.from_numpy(np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)).to(dev)
    sin = torch.from_numpy(np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)).to(dev)
    mk = lambda: [DecoderBlock(hidden, heads, ffn, max_keys, dev, cos, sin, seed=40 + i, kv_heads=kv_heads) for i in range(layers)]
    a_blocks, b_blocks = mk(), mk()
    if prepacked:
        for b in a_blocks:
            b.prepare_prefill()
    x = torch.from_numpy(rng.standard_normal((prompt, hidden)).astype(np.float16)).to(dev)
    # token by token
    want = torch.empty_like(x)
    for t in range(prompt):
        h = x[t:t + 1].clone()
        for b in b_blocks:
            b.step(h, t)
        want[t] = h[0]
    # all rows at once (or in two pieces)
    got = x.clone()
    pieces = [(0, prompt)] if not chunk else [(0, chunk), (chunk, prompt)]
    for lo, hi in pieces:
        rows = got[lo:hi].contiguous()
        for b in a_blocks:
            b.prefill(rows, lo)
        got[lo:hi] = rows
    torch.cuda.synchronize()
    w, g = want.float().cpu().numpy(), got.float().cpu().numpy()
    assert np.isfinite(g).all()
    tol = 2e-2 * np.abs(w).max()
    assert np.abs(g - w).max() <= tol, f"residual stream: max |err| {np.abs(g - w).max():.4f} vs tol {tol:.4f}"
    for a, b in zip(a_blocks, b_blocks):
        for ca, cb in ((a.attention.k_cache, b.attention.k_cache), (a.attention.v_cache, b.attention.v_cache)):
            ka, kb = ca[:, :prompt].float().cpu().numpy(), cb[:, :prompt].float().cpu().numpy()
            assert np.abs(ka - kb).max() <= 2e-2 * np.abs(kb).max(), "caches of the two paths"
