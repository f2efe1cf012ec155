"""Python mirror of the reference operator interface for the hot path: ``matmul_params`` + ``matmul::MatmulOperator``
(reference kernels/matmul.h:52-92, 110-153).  Same method names, same argument meaning, same error behaviour
(the reference asserts / prints "Unsupported group size" and exits -- here: ValueError / TceError), so the parity tests
read like llm/tests/cuda/test_ops.cu.  Every method enqueues HIP work through the C ABI and returns immediately;
data lives in torch tensors on the GPU (torch is the allocator and stream owner, nothing more).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch

from . import capi


def _ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("tinychatengine_amd operates on device tensors only (no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


@dataclass
class quantization_params:  # kernels/matmul.h:52-57
    scale: float = 1.0
    per_channel: bool = False
    zero_point: int = 0
    q_min: int = -128
    q_max: int = 127


@dataclass
class matrix:  # kernels/matmul.h:59-71 -- one tensor instead of seven typed pointers
    row: int = 0
    column: int = 0
    data: torch.Tensor | None = None
    qparams: quantization_params = field(default_factory=quantization_params)


@dataclass
class matmul_params:  # kernels/matmul.h:78-92
    A: matrix = field(default_factory=matrix)
    B: matrix = field(default_factory=matrix)
    C: matrix = field(default_factory=matrix)
    bias: matrix = field(default_factory=matrix)
    alpha: float = 1.0
    beta: float = 0.0
    half_scales: torch.Tensor | None = None      # fp16 [N][zw*8]      (q4_6)  or [K/G][N] (q4_5 via fp16_scales)
    fp16_scales: torch.Tensor | None = None
    int32_zero_point: torch.Tensor | None = None  # int32 [N][zw]
    block_size: int = 128                         # QK under QM_CUDA (llm/include/common.h:18)


class MatmulOperator:
    """matmul::MatmulOperator -- only the methods that take a quantized B (the hot path)."""

    # ---- W4A16 ----
    def gemv_forward_cuda(self, params: matmul_params) -> None:
        """kernels/cuda/gemv_cuda.cu:213-260.  IC = A.column, OC = C.column, M = C.row; B.row/B.column ignored."""
        A, B, Cm = params.A, params.B, params.C
        if params.block_size not in (32, 64, 128):
            raise capi.TceError(capi.TCE_ERR_UNSUPPORTED_GROUP, f"Unsupported group size: {params.block_size}")
        d = capi.W4A16Desc(M=Cm.row, N=Cm.column, K=A.column, group_size=params.block_size,
                           A=_ptr(A.data), qweight=_ptr(B.data), scales=_ptr(params.half_scales),
                           zeros=_ptr(params.int32_zero_point), C=_ptr(Cm.data))
        capi.check(capi.w4a16_forward(d, _stream()))

    def naive_mat_mul_fp16_int4(self, params: matmul_params) -> None:
        """kernels/cuda/matmul_int4.cu:8-48 (AWQ q4_5 layout, binary16 arithmetic); B.row = K, B.column = N/8."""
        A, B, Cm = params.A, params.B, params.C
        capi.check(capi.lib().tce_w4a16_awq_fp16acc(Cm.row, Cm.column, B.row, params.block_size, _ptr(A.data), _ptr(B.data),
                                                    _ptr(params.fp16_scales), _ptr(Cm.data), _stream()))

    def gemm_forward_cuda(self, params: matmul_params, split_k_iters: int = 1, workspace: torch.Tensor | None = None,
                          repack: bool = True) -> torch.Tensor:
        """Declared in kernels/matmul.h:142, undefined in the reference.  q4_5 layout in, fp16 [M][N] out, fp32
        accumulation (split_k_iters is accepted for signature compatibility; this implementation does not split K)."""
        A, B, Cm = params.A, params.B, params.C
        K, N, G = B.row, Cm.column, params.block_size
        need = int(capi.lib().tce_w4a16_awq_workspace_bytes(N, K, G))
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.uint8, device=A.data.device)
        if workspace.numel() * workspace.element_size() < need:
            raise ValueError("workspace too small")
        capi.check(capi.lib().tce_w4a16_gemm_awq(Cm.row, N, K, G, _ptr(A.data), _ptr(B.data), _ptr(params.half_scales),
                                                 _ptr(Cm.data), _ptr(workspace), int(repack), _stream()))
        return workspace

    # ---- W8A8: the eight methods of kernels/ref/matmul_ref_int8.cc:161-192 ----
    def _int8(self, p: matmul_params, bias_kind: int, out_kind: int, b_per_row: bool) -> None:
        A, B, Cm = p.A, p.B, p.C
        if A.column != B.row or Cm.row != A.row or Cm.column != B.column:  # the reference's asserts (:19-21)
            raise ValueError("shape mismatch: need A.column == B.row, C.row == A.row, C.column == B.column")
        d = capi.W8A8Desc(M=A.row, N=B.column, K=A.column, batch=1, A=_ptr(A.data), B=_ptr(B.data),
                          bias=_ptr(p.bias.data) if bias_kind != capi.TCE_BIAS_NONE else None, C=_ptr(Cm.data),
                          alpha=p.alpha, beta=p.beta, q_min=Cm.qparams.q_min, q_max=Cm.qparams.q_max,
                          bias_kind=bias_kind, out_kind=out_kind, b_per_row=int(b_per_row))
        capi.check(capi.w8a8_matmul(d, _stream()))

    def mat_mul_accelerator_int8_fast_2x2_32unroll(self, p):
        self._int8(p, capi.TCE_BIAS_INT8, capi.TCE_OUT_INT8, False)

    def mat_mul_accelerator_int8_fast_32unroll_over_column(self, p):
        self._int8(p, capi.TCE_BIAS_INT8, capi.TCE_OUT_INT8, False)

    def mat_mul_accelerator_int8_fast_2x2_32unroll_nobias(self, p):
        self._int8(p, capi.TCE_BIAS_NONE, capi.TCE_OUT_INT8, False)

    def mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_batch(self, p):
        self._int8(p, capi.TCE_BIAS_NONE, capi.TCE_OUT_INT8, True)

    def mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32(self, p):
        self._int8(p, capi.TCE_BIAS_FP32, capi.TCE_OUT_FP32, False)

    def mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32_over_column(self, p):
        self._int8(p, capi.TCE_BIAS_FP32, capi.TCE_OUT_FP32, False)

    def mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32(self, p):
        self._int8(p, capi.TCE_BIAS_NONE, capi.TCE_OUT_FP32, False)

    def mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32_batch(self, p):
        self._int8(p, capi.TCE_BIAS_NONE, capi.TCE_OUT_FP32, True)
