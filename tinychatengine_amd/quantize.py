"""The int4 weight formats consumed by the hot path, restated from the reference's offline quantizer
(llm/tools/quantize_methods.py) as torch ops so that synthetic weights can be produced directly in HBM.

q4_6  ("CUDA GEMV", what Linear_half_int4 loads)   quantize_row_q4_6, quantize_methods.py:370-442
q4_5  ("CUDA GEMM" / AWQ order)                     quantize_row_q4_5, quantize_methods.py:299-368
On-disk file names / dtypes: llm/tools/model_quantizer.py:35-67.
"""
from __future__ import annotations

import os

import numpy as np
import torch

QK4_6 = 128  # llm/tools/quantize_constants.py:8


def make_divisible(c: int, divisor: int) -> int:
    return (c + divisor - 1) // divisor


def calculate_zeros_width(in_features: int, group_size: int = 128, pack_num: int = 8) -> int:
    """quantize_methods.py:9-21 == llm/src/nn_modules/cuda/utils.cu:162-178."""
    if group_size >= 128:
        mult = 1
    elif group_size == 64:
        mult = 2
    elif group_size == 32:
        mult = 4
    else:
        raise NotImplementedError(group_size)
    w = make_divisible(in_features // group_size, pack_num)
    return make_divisible(w, mult) * mult


def group_codes(w: torch.Tensor, group_size: int):
    """Per group of `group_size` along the input dim: d = x[argmax|x|] / -8, code = trunc(clip(x/d + 8.5, 0, 15)).

    Returns (codes uint8 [N][K], d float32 [N][K/G]).  float32 arithmetic, one op at a time (no FMA), exactly like the
    numpy reference (quantize_methods.py:392-411)."""
    assert w.dtype == torch.float32 and w.dim() == 2
    n, k = w.shape
    assert k % group_size == 0
    x = w.reshape(-1, group_size)
    idx = torch.argmax(x.abs(), dim=1, keepdim=True)  # first maximal element, like np.argmax
    d = torch.gather(x, 1, idx).squeeze(1) / -8.0
    inv = torch.where(d == 0, torch.zeros_like(d), 1.0 / d)
    t = x * inv[:, None]
    t = t + 8.5
    codes = t.clamp_(0, 15).to(torch.int32).to(torch.uint8)  # astype(int32): truncation
    return codes.reshape(n, k), d.reshape(n, k // group_size)


def pack_q4_6(codes: torch.Tensor, d: torch.Tensor, group_size: int):
    """codes uint8 [N][K], d f32 [N][K/G] -> (qweight int32 [N][K/8], scales fp16 [N][zw*8], zeros int32 [N][zw])."""
    n, k = codes.shape
    c = codes.reshape(n, k // 8, 8).to(torch.int64)
    shifts = torch.arange(0, 32, 4, device=codes.device, dtype=torch.int64)
    words = (c << shifts).sum(dim=2)  # nibble i of word j = code[n][8j+i]
    qweight = (words & 0xFFFFFFFF).to(torch.int64)
    qweight = torch.where(qweight >= 2**31, qweight - 2**32, qweight).to(torch.int32)
    zw = calculate_zeros_width(k, group_size)
    scales = torch.zeros((n, zw * 8), dtype=torch.float16, device=codes.device)
    scales[:, : k // group_size] = d.to(torch.float16)
    zeros = torch.full((n, zw), -2004318072, dtype=torch.int32, device=codes.device)  # 0x88888888
    return qweight.contiguous(), scales.contiguous(), zeros.contiguous()


def quantize_q4_6(w: torch.Tensor, group_size: int = QK4_6):
    codes, d = group_codes(w, group_size)
    return pack_q4_6(codes, d, group_size)


def pack_q4_5(codes: torch.Tensor, d: torch.Tensor, group_size: int):
    """AWQ GEMM layout: qweight int32 [K][N/8] with nibble order 0 2 4 6 1 3 5 7 along n, scales fp16 [K/G][N],
    zeros int32 [K/G][N/8] = 0x88888888 (quantize_methods.py:341-366)."""
    n, k = codes.shape
    assert n % 8 == 0
    order = torch.tensor([0, 2, 4, 6, 1, 3, 5, 7], device=codes.device)
    ct = codes.t().reshape(k, n // 8, 8).to(torch.int64)[:, :, order]
    shifts = torch.arange(0, 32, 4, device=codes.device, dtype=torch.int64)
    words = (ct << shifts).sum(dim=2)
    qweight = torch.where(words >= 2**31, words - 2**32, words).to(torch.int32)
    scales = d.t().contiguous().to(torch.float16)
    zeros = torch.full((k // group_size, n // 8), -2004318072, dtype=torch.int32, device=codes.device)
    return qweight.contiguous(), scales, zeros


def quantize_q4_5(w: torch.Tensor, group_size: int = 128):
    codes, d = group_codes(w, group_size)
    return pack_q4_5(codes, d, group_size)


def unpack_q4_6(qweight: torch.Tensor) -> torch.Tensor:
    """int32 [N][K/8] -> codes uint8 [N][K]."""
    n, kw = qweight.shape
    shifts = torch.arange(0, 32, 4, device=qweight.device, dtype=torch.int64)
    c = (qweight.to(torch.int64)[:, :, None] >> shifts) & 0xF
    return c.reshape(n, kw * 8).to(torch.uint8)


def dequantize_q4_6(qweight, scales, zeros, group_size: int) -> torch.Tensor:
    """fp32 [N][K] = s * (q - z): a convenience for building test inputs (NOT used by any compute path)."""
    codes = unpack_q4_6(qweight).to(torch.float32)
    n, k = codes.shape
    ng = k // group_size
    shifts = torch.arange(0, 32, 4, device=qweight.device, dtype=torch.int64)
    z = ((zeros.to(torch.int64)[:, :, None] >> shifts) & 0xF).reshape(n, -1)[:, :ng].to(torch.float32)
    s = scales[:, :ng].to(torch.float32)
    return (codes.reshape(n, ng, group_size) - z[:, :, None]) * s[:, :, None]


# ---- on-disk format (llm/tools/model_quantizer.py:35-67, llm/include/ops/linear.h:206-209) ----
def save_linear_q4_6(dirname: str, qweight, scales, zeros) -> None:
    os.makedirs(dirname, exist_ok=True)
    qweight.cpu().numpy().astype(np.int32).tofile(os.path.join(dirname, "weight_int4.bin"))
    scales.cpu().numpy().astype(np.float16).tofile(os.path.join(dirname, "scaling_factor_int4.bin"))
    zeros.cpu().numpy().astype(np.int32).tofile(os.path.join(dirname, "zero_point_int4.bin"))
    np.zeros(scales.numel(), np.float16).tofile(os.path.join(dirname, "offset_int4.bin"))  # unused on the GPU path


def load_linear_q4_6(dirname: str, out_features: int, in_features: int, group_size: int = QK4_6, device="cpu"):
    zw = calculate_zeros_width(in_features, group_size)
    qweight = np.fromfile(os.path.join(dirname, "weight_int4.bin"), dtype=np.int32).reshape(out_features, in_features // 8)
    scales = np.fromfile(os.path.join(dirname, "scaling_factor_int4.bin"), dtype=np.float16).reshape(out_features, zw * 8)
    zeros = np.fromfile(os.path.join(dirname, "zero_point_int4.bin"), dtype=np.int32).reshape(out_features, zw)
    return (torch.from_numpy(qweight).to(device), torch.from_numpy(scales).to(device), torch.from_numpy(zeros).to(device))


def merge_qkv_q4_6(q, k, v):
    """llm/tools/llama_qkv_merger.py:27-48: the fused qkv linear is the row-wise concatenation of the three."""
    return tuple(torch.cat([a, b, c], dim=0).contiguous() for a, b, c in zip(q, k, v))
