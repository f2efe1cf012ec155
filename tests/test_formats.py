"""CPU: the weight formats and on-disk files of the path: the product-side quantizer
(tinychatengine_amd/quantize.py) against the oracle and the golden vectors produced by the reference's own
quantize_row_q4_6 / q4_5, the .bin round trip Linear_half_int4 loads, and the q/k/v merge."""
import numpy as np
import pytest
import torch

from tinychatengine_amd import quantize as Q


@pytest.mark.parametrize("N,K,G", [(32, 1408, 128), (16, 512, 64), (8, 256, 32), (24, 11008, 128)])
def test_product_quantizer_matches_oracle(oracle, N, K, G):
    g = torch.Generator().manual_seed(N + K)
    w = torch.randn(N, K, generator=g) * 0.02
    w[0, :G] = 0  # d == 0 branch
    qw, sc, zp = Q.quantize_q4_6(w, G)
    oqw, osc, ozp, codes, d = oracle.quantize_q4_6(w.numpy(), G)
    assert np.array_equal(qw.numpy().view(np.uint32), oqw)
    assert np.array_equal(sc.numpy().view(np.uint16), osc.view(np.uint16))
    assert np.array_equal(zp.numpy().view(np.uint32), ozp)
    assert sc.shape[1] == Q.calculate_zeros_width(K, G) * 8
    if N % 8 == 0:
        q5, s5, z5 = Q.quantize_q4_5(w, G)
        oq5, os5, oz5 = oracle.pack_q4_5(codes, d.reshape(-1), N, K, G)
        assert np.array_equal(q5.numpy().view(np.uint32), oq5) and np.array_equal(s5.numpy().view(np.uint16), os5.view(np.uint16))
        assert np.array_equal(z5.numpy().view(np.uint32), oz5)
    assert torch.equal(Q.unpack_q4_6(qw), torch.from_numpy(codes))
    deq = Q.dequantize_q4_6(qw, sc, zp, G).reshape(N, K)
    assert torch.allclose(deq, w, atol=float(w.abs().max()) / 7.0)  # 4-bit grid: |err| <= d/2 <= max/16


def test_product_quantizer_matches_reference_python_golden(golden):
    M, N, K, G = (int(v) for v in golden["w4_dims"])
    qw, sc, zp = Q.quantize_q4_6(torch.from_numpy(golden["w4_w"]), G)
    assert np.array_equal(qw.numpy().view(np.uint32), golden["w4_qweight"])
    assert np.array_equal(sc.numpy().view(np.uint16), golden["w4_scales"].view(np.uint16))
    assert np.array_equal(zp.numpy().view(np.uint32), golden["w4_zeros"])
    M, N, K, G = (int(v) for v in golden["awq_dims"])
    q5, s5, z5 = Q.quantize_q4_5(torch.from_numpy(golden["awq_w"]), G)
    assert np.array_equal(q5.numpy().view(np.uint32), golden["awq_qweight"])
    assert np.array_equal(s5.numpy().view(np.uint16), golden["awq_scales"].view(np.uint16))


def test_zeros_width_table():
    # quantize_methods.py:9-21 / utils.cu:162-178; SURVEY App. C padded rows
    assert Q.calculate_zeros_width(4096, 128) == 4 and Q.calculate_zeros_width(11008, 128) == 11
    assert Q.calculate_zeros_width(13824, 128) == 14 and Q.calculate_zeros_width(4096, 64) == 8
    assert Q.calculate_zeros_width(4096, 32) == 16 and Q.calculate_zeros_width(1408, 128) == 2
    with pytest.raises(NotImplementedError):
        Q.calculate_zeros_width(4096, 16)


def test_on_disk_round_trip_and_qkv_merge(tmp_path):
    """weight_int4.bin / scaling_factor_int4.bin / zero_point_int4.bin (model_quantizer.py:57-64, linear.h:206-209)."""
    from tinychatengine_amd.linear import Linear_half_int4
    g = torch.Generator().manual_seed(3)
    parts = []
    for i, n in enumerate((64, 16, 16)):
        w = torch.randn(n, 1408, generator=g) * 0.02
        t = Q.quantize_q4_6(w, 128)
        d = tmp_path / f"p{i}"
        Q.save_linear_q4_6(str(d), *t)
        assert (d / "weight_int4.bin").stat().st_size == n * 1408 // 2
        assert (d / "scaling_factor_int4.bin").stat().st_size == n * 16 * 2  # 11 groups padded to 16
        assert (d / "zero_point_int4.bin").stat().st_size == n * 2 * 4
        back = Q.load_linear_q4_6(str(d), n, 1408, 128)
        assert all(torch.equal(a, b) for a, b in zip(t, back))
        parts.append(back)
    qkv = Q.merge_qkv_q4_6(*parts)  # llm/tools/llama_qkv_merger.py:27-48: row-wise concatenation
    lin = Linear_half_int4(*qkv, group_size=128)
    assert lin.out_features == 96 and lin.in_features == 1408 and lin.zeros_are_8
    assert torch.equal(lin.weight[64:80], parts[1][0])
    sh = lin.shard(1, 2)
    assert torch.equal(sh.weight, lin.weight[48:96]) and sh.out_features == 48
