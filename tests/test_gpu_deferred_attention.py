"""Round 5: the attention step without its cross-workgroup combine + o_proj combining the partial states in its prologue
(tce_attention_decode_step_deferred_f16 + tce_w4a16_forward_deferred_attention) against the pair it replaces
(tce_attention_decode_step_pos_f16 + tce_w4a16_forward): the SAME bits out of o_proj (the combine is the same operations in the same order, the row is rounded to
binary16 the same way), at contexts with one, four and eight chunk slots, grouped-query and one key / value head per query head, the position by value and on the device
(one captured graph replayed over positions that change the number of live slots), with the residual epilogue."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tinychatengine_amd import capi
    capi.lib()
    return torch.device("cuda:0")


def _setup(dev, heads, kv_heads, max_keys, seed):
    from tinychatengine_amd.attention_ops import DecodeAttention
    from tinychatengine_amd.linear import Linear_half_int4
    g = torch.Generator(device=dev).manual_seed(seed)
    ang = np.random.default_rng(seed).uniform(0, 2 * np.pi, (max_keys, 64))
    cos = torch.from_numpy(np.concatenate([np.cos(ang), np.cos(ang)], axis=1).astype(np.float16)).to(dev)
    sin = torch.from_numpy(np.concatenate([np.sin(ang), np.sin(ang)], axis=1).astype(np.float16)).to(dev)
    atts = [DecodeAttention(heads, 128, max_keys, dev, cos, sin, kv_heads=kv_heads) for _ in range(2)]
    k0 = torch.empty(kv_heads, max_keys, 128, device=dev).normal_(0, 0.8, generator=g).to(torch.float16)
    v0 = torch.empty(kv_heads, max_keys, 128, device=dev).normal_(0, 0.8, generator=g).to(torch.float16)
    for a in atts:
        a.k_cache.copy_(k0)
        a.v_cache.copy_(v0)
    hidden = heads * 128
    o = Linear_half_int4.from_float(torch.empty(hidden, hidden, device=dev).normal_(0, hidden ** -0.5, generator=g), 128).prepack()
    qkv = torch.empty((heads + 2 * kv_heads) * 128, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    res = torch.empty(1, hidden, device=dev).normal_(0, 1, generator=g).to(torch.float16)
    return atts, o, qkv, res


def _plain(att, o, qkv, res, pos, pos_t=None):
    from tinychatengine_amd import capi
    x = torch.full((1, o.in_features), float("nan"), dtype=torch.float16, device=qkv.device)
    y = res.clone()
    att.step(qkv, pos, out=x.view(att.heads, 128), pos_device=pos_t)
    capi.check(capi.w4a16_forward(o.desc(x, y, flags=capi.TCE_W4_ADD_TO_C), torch.cuda.current_stream().cuda_stream))
    return x, y


def _deferred(att, o, qkv, res, pos, pos_t=None):
    from tinychatengine_amd import capi
    x = torch.full((1, o.in_features), float("nan"), dtype=torch.float16, device=qkv.device)
    y = res.clone()
    att.step(qkv, pos, out=x.view(att.heads, 128), pos_device=pos_t, defer=True)
    capi.check(capi.lib().tce_w4a16_forward_deferred_attention(C.byref(o.desc(x, y, flags=capi.TCE_W4_ADD_TO_C)), C.byref(att.deferred),
                                                               C.c_void_p(pos_t.data_ptr() if pos_t is not None else 0), int(pos), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return x, y, att.deferred.slots


@pytest.mark.parametrize("heads,kv_heads", [(32, 8), (32, 32), (8, 8)])
def test_deferred_combine_gives_the_same_bits_as_the_in_kernel_combine(dev, heads, kv_heads):
    atts, o, qkv, res = _setup(dev, heads, kv_heads, 2048, seed=3 + heads + kv_heads)
    seen = set()
    for pos in (5, 200, 319, 320, 511, 640, 700, 1023, 1024, 2047):
        x0, y0 = _plain(atts[0], o, qkv, res, pos)
        x1, y1, slots = _deferred(atts[1], o, qkv, res, pos)
        torch.cuda.synchronize()
        seen.add(slots)
        assert torch.isfinite(y0.float()).all()
        assert torch.equal(y0, y1), f"{heads}/{kv_heads} heads, position {pos} ({slots} slots): o_proj + residual differs"
        assert torch.equal(atts[0].k_cache, atts[1].k_cache) and torch.equal(atts[0].v_cache, atts[1].v_cache)  # the appended rows
        if slots == 1:
            assert torch.equal(x0, x1)  # nothing deferred: the step wrote the row itself
    assert {1, 4, 8} <= seen, seen


def test_deferred_combine_with_the_position_on_the_device_in_one_graph(dev):
    """One captured pair of launches, replayed at positions that leave one, some and all of the bound's eight slots live."""
    atts, o, qkv, res = _setup(dev, 32, 8, 2048, seed=11)
    pos_t = torch.zeros(1, dtype=torch.int32, device=dev)
    bound = 2047
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        pos_t.fill_(100)
        _deferred(atts[1], o, qkv, res, bound, pos_t)  # warm (appends the token's row at position 100: the other cache gets the same)
        torch.cuda.synchronize()
        atts[0].k_cache.copy_(atts[1].k_cache)
        atts[0].v_cache.copy_(atts[1].v_cache)
        g = torch.cuda.CUDAGraph()
        out = {}
        with torch.cuda.graph(g, stream=s):
            x1, y1, slots = _deferred(atts[1], o, qkv, res, bound, pos_t)
            out["y"] = y1
        assert slots == 8
        for pos in (3, 255, 256, 300, 1000, 1791, 1792, 2047):
            pos_t.fill_(pos)
            g.replay()
            torch.cuda.synchronize()
            got = out["y"].clone()
            x0, y0 = _plain(atts[0], o, qkv, res, bound, pos_t)
            torch.cuda.synchronize()
            assert torch.equal(y0, got), f"position {pos} under the bound {bound}"


def test_deferred_pair_refuses_what_it_cannot_run(dev):
    from tinychatengine_amd import capi
    from tinychatengine_amd.linear import Linear_half_int4
    atts, o, qkv, res = _setup(dev, 32, 8, 2048, seed=5)
    x = torch.zeros(1, 4096, dtype=torch.float16, device=dev)
    y = torch.zeros(1, 4096, dtype=torch.float16, device=dev)
    atts[0].step(qkv, 1000, out=x.view(32, 128), defer=True)
    info = atts[0].deferred
    assert info.slots == 8 and info.stride == 132 and info.heads == 32
    L = capi.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev).manual_seed(1)
    unpacked = Linear_half_int4.from_float(torch.empty(4096, 4096, device=dev).normal_(0, 0.02, generator=g), 128)
    assert L.tce_w4a16_forward_deferred_attention(C.byref(unpacked.desc(x, y)), C.byref(info), None, 1000, st) == capi.TCE_ERR_UNSUPPORTED_SHAPE  # no packed copy
    two = torch.zeros(2, 4096, dtype=torch.float16, device=dev)
    assert L.tce_w4a16_forward_deferred_attention(C.byref(o.desc(two, torch.zeros(2, 4096, dtype=torch.float16, device=dev))), C.byref(info), None, 1000, st) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert L.tce_w4a16_forward_deferred_attention(C.byref(o.desc(x, y)), None, None, 1000, st) == capi.TCE_ERR_BAD_ARG
    L.tce_reset_last_error()
    torch.cuda.synchronize()
