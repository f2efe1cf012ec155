"""Host logic of tce_w4a16_forward: which kernel family / GEMM tile a shape is sent to (tce_w4a16_describe_dispatch: no launch, no HIP call,
so this runs without a GPU).  The expectations are the measured cross-over points of DESIGN.md section 5 / profiles/r1/m_sweep_17_384.jsonl."""
import pytest

from tinychatengine_amd import capi


def _desc(M, N, K, G=128, flags=0):
    # pointers are never dereferenced by the query; 16 satisfies the alignment checks
    return capi.W4A16Desc(M=M, N=N, K=K, group_size=G, A=16, qweight=16, scales=16, zeros=16, C=16, flags=flags)


@pytest.mark.parametrize("shape,expect", [
    ((1, 4096, 4096), "gemv passes=1 kernel=row-block"),      # decode
    ((1, 128256, 4096), "gemv passes=1 kernel=row-block"),    # the Llama-3 lm_head (round 1 handed it to the persistent kernel; re-tuned in round 2)
    ((1, 32000, 4096), "gemv passes=1 kernel=row-block"),
    ((2, 11008, 4096), "gemv passes=1 kernel=row-block"),                      # M = 2 stays on the GEMV kernels
    ((3, 4096, 4096), "small-batch slices=1"),
    ((16, 22016, 4096), "small-batch slices=1"),
    ((32, 4096, 4096), "small-batch slices=2"),                # N too small to fill the chip with GEMM tiles
    ((64, 4096, 11008), "small-batch slices=4"),
    ((32, 22016, 4096), "gemm-dma tile=32x128 quartets=2 group=128"),   # enough column tiles: the GEMM takes over
    ((128, 4096, 4096), "gemm-dma tile=64x64 quartets=2 group=128"),
    ((512, 4096, 4096), "gemm-dma tile=64x128 quartets=2 group=128"),  # one workgroup per CU: two quartets
    ((512, 11008, 4096), "gemm-dma tile=64x128 quartets=1 group=128"), # 688 workgroups on 512 slots
    ((4096, 4096, 4096), "gemm-dma tile=64x128 quartets=1 group=128"),
    ((512, 4096, 4096, 64), "gemm-dma tile=64x128 quartets=2 group=64"),
    ((9, 4096, 4096, 64), "gemv passes=3 kernel=row-block"),                   # other group sizes below M = 17: GEMV kernel, 4 rows per pass
    ((12, 4096, 4096, 128, capi.TCE_W4_FORCE_GEMV), "gemv passes=3 kernel=row-block"),
])
def test_dispatch_of_documented_regimes(shape, expect):
    assert capi.describe_dispatch(_desc(*shape)) == expect


def test_forced_paths_and_errors():
    assert capi.describe_dispatch(_desc(1, 4096, 4096, 128, capi.TCE_W4_FORCE_GEMM)).startswith("gemm-dma")
    assert capi.describe_dispatch(_desc(64, 4096, 4096, 128, capi.TCE_W4_FORCE_GEMV)) == "gemv passes=16 kernel=row-block"
    with pytest.raises(capi.TceError):
        capi.describe_dispatch(_desc(4, 4096, 4160))  # K not a multiple of the group size
    try:
        capi.set_gemm_config(4, 2)
        assert capi.describe_dispatch(_desc(512, 4096, 4096)) == "gemm tile=64x128"
        capi.set_gemm_config(204, 1)
        assert capi.describe_dispatch(_desc(512, 4096, 4096)).startswith("gemm-dma tile=64x64")
    finally:
        capi.set_gemm_config()


def test_cost_model_prefers_fewer_rounds():
    """688 workgroups of 64x128 on 512 slots are two rounds; the dispatcher must not pick a form that needs three."""
    for M, N in ((512, 11008), (192, 11008), (256, 22016), (2048, 4096)):
        s = capi.describe_dispatch(_desc(M, N, 4096))
        assert s.startswith("gemm-dma"), s
        r, c = (int(v) for v in s.split("tile=")[1].split()[0].split("x"))
        q = int(s.split("quartets=")[1].split()[0])
        wgs = -(-M // r) * -(-N // c)
        slots = 256 * (1 if q == 2 or r >= 128 else 2)
        assert -(-wgs // slots) <= 4 or wgs / slots > 3, (s, wgs, slots)


def test_attention_step_cut_rule():
    """The fitted rule of the fast attention step: one chunk per head and no combine up to 320 keys, four chunks up to 1024 (up to 640 with four query heads per key / value head: round 4),
    eight chunks of at most 512 keys beyond; every chunk a multiple of 16 keys (four waves x four keys per step), all keys covered,
    never more chunks than the workspace was sized for (64-key chunks)."""
    from tinychatengine_amd import capi
    for heads in (1, 8, 32, 40):
        for keys in list(range(1, 700)) + [1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8192, 16384, 100000]:
            d = capi.describe_attention_step(heads, keys)
            chunk, chunks = d["keys-per-chunk"], d["chunks"]
            assert chunk % 16 == 0 and chunk <= 1024 and d["waves"] == 4, (heads, keys, d)
            assert chunks * chunk >= keys > (chunks - 1) * chunk, (heads, keys, d)
            assert chunks <= (keys + 63) // 64 or chunks == 1, (heads, keys, d)
            assert d["workgroups"] == heads * chunks and d["combine"] == ("yes" if chunks > 1 else "no")
            if keys <= 320:
                assert chunks == 1
            elif keys <= 640 or (keys <= 1024):  # (multi-head launches -- describe_attention_step(heads, keys) -- keep four chunks up to 1024 keys; grouped queries: below)
                assert chunks in (3, 4), (keys, d)  # (ceil(keys / 4) rounded up to 16 keys can leave the fourth chunk empty: 321..336 keys)
            elif keys <= 4096:
                assert chunks in (6, 7, 8), (keys, d)  # (641..672 keys: ceil(keys / 8) rounded up to 16 keys leaves fewer chunks)
            else:
                assert chunk == 512
    for keys, want in ((512, (4,)), (640, (4,)), (641, (6, 7, 8)), (1024, (8,)), (2048, (8,))):  # grouped queries, 32 over 8
        assert capi.describe_attention_step(32, keys, 8)["chunks"] in want, keys


def _pk(M, N, K, G=128, z8=True, scratch=True):
    # a descriptor that carries a packed copy (and the scratch area): the query reads no memory
    return capi.W4A16Desc(M=M, N=N, K=K, group_size=G, A=16, qweight=16, scales=16, zeros=16, C=16, prepacked=4096, scratch=8192 if scratch else None,
                          flags=capi.TCE_W4_ZERO_POINT_IS_8 if z8 else 0)


def test_packed_prefill_dispatch_rules_of_round_5():
    """The cost model's choices the GPU tests also see (tests/test_gpu_w4a16_pk.py), held here without a GPU: which form of the packed prefill GEMM a shape is sent to."""
    d = capi.describe_dispatch
    # few tiles: the k range of every 128 x 128 tile cut across workgroups -- two runs (a directed hand-off) at K = 4096, four through the last arriver at K = 11008
    # (round 6: with at most 128 tiles the two runs are walked by TWO quartets each -- form 16, two waves per SIMD on every CU: 512 x 4096 x 4096 27.3 -> 26.1 us,
    #  x 11008 53.6 -> 50.9-53.8; very long k ranges and launches of 64 tiles keep four runs of one quartet; other group sizes keep the one-quartet runs)
    assert "tile=128x128 quartets=2 ksplit=2 " in d(_pk(512, 4096, 4096)), d(_pk(512, 4096, 4096))
    assert "tile=128x128 quartets=2 ksplit=2 " in d(_pk(512, 4096, 11008)), d(_pk(512, 4096, 11008))
    assert "tile=128x128 quartets=2 ksplit=2 " in d(_pk(384, 4096, 4096)), d(_pk(384, 4096, 4096))
    assert "tile=128x128 quartets=1 ksplit=4 " in d(_pk(512, 4096, 14336)), d(_pk(512, 4096, 14336))
    assert "tile=128x128 quartets=1 ksplit=4 " in d(_pk(512, 2048, 8192)), d(_pk(512, 2048, 8192))
    assert "tile=128x128 quartets=1 ksplit=2 " in d(_pk(512, 4096, 4096, G=64)), d(_pk(512, 4096, 4096, G=64))
    assert "ksplit" not in d(_pk(1024, 4096, 4096))  # 256 tiles: whole tiles
    assert "ksplit" not in d(_pk(512, 4096, 4096, scratch=False))  # no scratch area, no cut
    # the wide forms (128 rows x 64 / 48 columns per wave): linears whose zero points are all 8, groups of 128
    assert "tile=128x192 wave=128x48 quartets=2" in d(_pk(512, 11008, 4096))
    assert "tile=128x256 wave=128x64 quartets=2" in d(_pk(2048, 4096, 4096))
    assert "tile=128x256 wave=128x64 quartets=1" in d(_pk(2048, 11008, 4096))
    assert "tile=128x512 wave=128x64" in d(_pk(4096, 4096, 4096))
    for shape in ((512, 11008, 4096), (2048, 4096, 4096), (4096, 4096, 4096)):
        assert "wave=" not in d(_pk(*shape, z8=False)), d(_pk(*shape, z8=False))  # real zero points: the narrow forms
        assert d(_pk(*shape, z8=False)).startswith("gemm-pk")
    # two quartets per workgroup on the narrow body: every group size again (round 5 kept them to groups of 128 for the groups-of-32 instantiation's lost accumulator
    # lanes; round 6 found the instruction form behind them, removed it and made the build refuse it: isa_lint.py RULE 1, profiles/r6/pk_lost_lanes_rule.md)
    for (M, N, K, q) in ((192, 200, 512, 2), (700, 392, 3072, 1), (1024, 4096, 4096, 2), (256, 2048, 1024, 2), (384, 1024, 4096, 1)):
        for G in (64, 32):
            s = d(_pk(M, N, K, G=G))
            assert s.startswith("gemm-pk") and f"quartets={q}" in s and f"group={G}" in s, s
    assert "quartets=2" in d(_pk(1024, 4096, 4096, z8=False))


def test_independent_linears_dispatch():
    """tce_w4a16_forward_independent (round 6): one launch for decode rows on packed copies whatever their K; the workgroup is as wide as the longest K needs and a
    shorter K packs floor(waves / own waves) tiles into it.  No GPU needed (tce_w4a16_describe_independent)."""
    def d(N, K, M=1, G=128, packed=256, flags=0):
        x = _desc(M, N, K, G, flags)
        x.prepacked = packed
        return x
    # a rank's shards of a Llama-3-8B block at 8 ranks: q, k, v, o, gate, up (K = 4096: 4 waves a tile, 3 tiles per 14-wave workgroup) and down (K = 14336: 14 waves)
    shards = [d(512, 4096), d(128, 4096), d(128, 4096), d(512, 4096), d(1792, 4096), d(1792, 4096), d(512, 14336)]
    tiles = [32, 8, 8, 32, 112, 112]
    assert capi.describe_independent(shards) == f"gemv-i8-mixed waves=14 workgroups={sum(-(-t // 3) for t in tiles) + 32}"
    assert capi.describe_independent([d(256, 4096), d(100, 4096)]) == "gemv-i8-mixed waves=4 workgroups=23"
    assert capi.describe_independent([d(16, 128), d(16, 1408)]) == "gemv-i8-mixed waves=2 workgroups=2"   # 1408 = 11 units: two waves; the 128-wide linear packs two tiles per workgroup
    # what the one-launch form does not take goes linear by linear
    assert capi.describe_independent([d(256, 4096)]) == "one-by-one launches=1"
    assert capi.describe_independent([d(256, 4096), d(64, 4096, packed=0)]) == "one-by-one launches=2"
    assert capi.describe_independent([d(256, 4096), d(64, 4096, M=2)]) == "one-by-one launches=2"
    assert capi.describe_independent([d(256, 4096), d(64, 4096, G=64)]) == "one-by-one launches=2"
    assert capi.describe_independent([d(256, 4096), d(64, 28672)]) == "one-by-one launches=2"
    assert capi.describe_independent([d(256, 4096), d(64, 4096, flags=capi.TCE_W4_FORCE_GEMM)]) == "one-by-one launches=2"
    with pytest.raises(capi.TceError):
        capi.describe_independent([d(16, 1024)] * 9)

    # the width rule, restated workgroup by workgroup (the library computes a CU's load in closed form per linear and keeps the last answer per thread): the most loaded of
    # 256 CUs in index order, widths within 5 % of the best compared by their workgroup count, the narrowest on ties
    def rule(shapes):
        wk = [-(-(K // 128) // 8) for _, K in shapes]
        tiles = [-(-N // 16) for N, _ in shapes]
        need = max(wk)
        def blocks(w):
            return sum(-(-t // (w // k)) for t, k in zip(tiles, wk))
        def worst(w):
            load, b = [0] * 256, 0
            for t, k in zip(tiles, wk):
                nsub = w // k
                for t0 in range(0, t, nsub):
                    load[b % 256] += min(nsub, t - t0) * k
                    b += 1
            return max(load)
        best = min(worst(w) for w in range(need, 17))
        ok = [w for w in range(need, 17) if worst(w) * 20 <= best * 21]
        w = min(ok, key=lambda w: (blocks(w), w))
        return f"gemv-i8-mixed waves={w} workgroups={blocks(w)}"
    import random
    rng = random.Random(6)
    for _ in range(200):
        shapes = [(rng.choice((16, 48, 100, 128, 512, 1792, 3584, 5120, 14336, 40000)), 128 * rng.choice((1, 8, 11, 32, 40, 86, 108, 112, 128)))
                  for _ in range(rng.randint(2, 8))]
        assert capi.describe_independent([d(N, K) for N, K in shapes]) == rule(shapes), shapes
        assert capi.describe_independent([d(N, K) for N, K in shapes]) == rule(shapes)  # (the remembered answer)


def test_w8a8_dispatch_rules():
    """tce_w8a8_describe_dispatch (ABI 0.1.13): the form tce_w8a8_matmul would run, without a GPU -- BASELINE config 4's OPT-125M launches and the rules round 6 fitted
    (the whole tile in every wave for chains of >= 12 k-steps: the smallest of 32 x 48 / 32 x 64 / 64 x 64 whose workgroups are at most one per CU, 32 x 48 also at two)."""
    def d(M, N, K, batch=1, **kw):
        return capi.W8A8Desc(M=M, N=N, K=K, batch=batch, alpha=1.0, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE, out_kind=capi.TCE_OUT_INT8, **kw)
    f = capi.describe_w8a8_dispatch
    assert f(d(1, 768, 768)) == "w8a8 wave-per-column rows=1" and f(d(1, 768, 3072)) == "w8a8 wave-per-column rows=1"
    assert f(d(512, 768, 3072)) == "w8a8 k-slice tile=32x48 waves=4 workgroups=256"
    assert f(d(512, 768, 768)) == "w8a8 k-slice tile=32x48 waves=4 workgroups=256"
    assert f(d(108, 768, 3072)) == "w8a8 k-slice tile=32x48 waves=4 workgroups=64"
    assert f(d(108, 768, 3072), with_scratch=True) == "w8a8 k-slice tile=32x48 waves=4 workgroups=64"  # (the cut across workgroups no longer takes it)
    assert f(d(108, 3072, 768)) == "w8a8 k-slice tile=32x48 waves=4 workgroups=256"
    assert f(d(1024, 768, 768)) == "w8a8 k-slice tile=32x48 waves=4 workgroups=512"   # two per CU
    assert f(d(512, 1024, 4096)) == "w8a8 k-slice tile=32x64 waves=4 workgroups=256"
    assert f(d(512, 2048, 8192)) == "w8a8 k-slice tile=64x64 waves=4 workgroups=256"
    assert f(d(512, 3072, 768)) == "w8a8 tile=64x64 quartets=1"                         # 384 tiles of 64 x 64: the quartet kernel
    assert f(d(2048, 768, 3072)) == "w8a8 tile=64x64 deep-pipeline quartets=1"
    assert f(d(512, 4096, 4096)) == "w8a8 k-slice tile=64x64 waves=4 workgroups=512"    # OPT-6.7B widths at 512 rows: 512 tiles x 64 k-steps (until round 6 the deep-pipeline tile)
    assert f(d(512, 4096, 2048)) == "w8a8 tile=128x64 quartets=2" and f(d(1024, 4096, 8192)) == "w8a8 tile=128x128 quartets=2"  # (the 128-row tiles keep what round 3 / 4 gave them)
    assert f(d(2048, 4096, 4096)) == "w8a8 tile=128x128 quartets=1" and f(d(512, 16384, 4096)) == "w8a8 tile=128x128 quartets=1"  # the 128-row tiles
    # 128 x 64 tiles from 384 of them on (round 6: 1024 x 3072 x 768 12.1 -> 9.7 us); one quartet under 32 k-steps, two from there on while they are fewer than 512
    assert f(d(1024, 3072, 768)) == "w8a8 tile=128x64 quartets=1" and f(d(640, 5120, 1280)) == "w8a8 tile=128x64 quartets=1"
    assert f(d(768, 4096, 4096)) == "w8a8 tile=128x64 quartets=2" and f(d(1024, 3072, 8192)) == "w8a8 tile=128x64 quartets=2"
    assert f(d(512, 8192, 2048)) == "w8a8 tile=128x64 quartets=1"                       # 512 tiles: one quartet, as before
    assert f(d(512, 512, 64, batch=12)) == "w8a8 tile=64x64 quartets=1"                # the attention BMMs: one k-step
    assert f(d(512, 64, 512, batch=12)).startswith("w8a8 tile=64x64 quartets=")        # eight k-steps: under the k-slice rule's twelve
    assert f(d(40, 33, 50)) == "w8a8 generic (one output per thread)"                   # K % 16 != 0
    L = capi.lib()
    try:  # a forced form of one family switches the other families' rules off, never the other way round
        capi.check(L.tce_w4a16_set_debug_mode(19001))
        assert f(d(512, 768, 3072)) == "w8a8 tile=32x64 quartets=2"                    # (round 6's first step, reached only like this now)
        capi.check(L.tce_w4a16_set_debug_mode(19904))
        assert f(d(512, 768, 3072)) == "w8a8 k-slice tile=64x64 waves=4 workgroups=96"
        capi.check(L.tce_w4a16_set_debug_mode(19000))
        capi.check(L.tce_w4a16_set_debug_mode(72))
        assert f(d(512, 768, 3072)) == "w8a8 tile=32x64 quartets=2"
        capi.check(L.tce_w4a16_set_debug_mode(70))
        capi.check(L.tce_w4a16_set_debug_mode(184))
        assert f(d(108, 768, 3072), with_scratch=True) == "w8a8 tile=64x64 quartets=2 kcut=4"
    finally:
        for m in (19000, 70, 180, 190, 170, 75):
            L.tce_w4a16_set_debug_mode(m)
    with pytest.raises(capi.TceError):
        f(d(0, 16, 64))
