"""The L2 callers of the hot path -- the ops that own quantized weights, fill a descriptor and call one operator
method (reference llm/include/ops/linear.h:186-221, llm/src/ops/cuda/linear.cu:5-40, llm/src/ops/W8A8*.cc,
llm/src/ops/BMM_S8T_*.cc) -- and their multi-GPU form: every linear sharded column-wise (output channels) across
ranks, outputs joined by an RCCL all-gather over xGMI (no reference counterpart).
"""
from __future__ import annotations

import torch

from . import capi, quantize
from .matmul import MatmulOperator, matmul_params, matrix, _ptr, _stream


_gemm_scratch: dict = {}


def gemm_scratch(device) -> torch.Tensor:
    """tce_w4a16_desc.scratch for `device`: one zeroed area shared by every linear (their calls are ordered by the stream the harness
    uses; a host that runs prefill GEMMs on several streams at once gives each stream its own)."""
    key = str(device)
    if key not in _gemm_scratch:
        _gemm_scratch[key] = torch.zeros(int(capi.lib().tce_w4a16_gemm_scratch_bytes()), dtype=torch.uint8, device=device)
    return _gemm_scratch[key]


class Linear_half_int4:
    """W4A16 linear on the q4_6 layout.  Mirrors Linear_half_int4 (linear.h:186-221 / linear.cu:5-40)."""

    def __init__(self, qweight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, group_size: int = quantize.QK4_6):
        self.weight, self.scale, self.zero_point = qweight, scales, zeros  # names as in linear.h:215-218
        self.out_features, self.in_features = qweight.shape[0], qweight.shape[1] * 8
        self.group_size = group_size
        zw = quantize.calculate_zeros_width(self.in_features, group_size)
        if tuple(scales.shape) != (self.out_features, zw * 8) or tuple(zeros.shape) != (self.out_features, zw):
            raise ValueError("scales/zeros do not have the padded q4_6 shapes (quantize_methods.py:431-440)")
        # the reference quantizer writes zero point 8 everywhere; verified once here so the GEMV can skip the zeros stream
        self.zeros_are_8 = bool((zeros == -2004318072).all().item())
        self.packed = None  # q4_mfma copy for the prefill GEMM (prepack())
        self._op = MatmulOperator()

    def prepack(self) -> "Linear_half_int4":
        """Builds the q4_mfma copy of the weights once (tce_w4a16_prepack; load-time work like the reference's offline
        re-layouts, llm/tools/model_quantizer.py): descriptors built afterwards carry it, and tce_w4a16_forward runs the
        128-row MFMA kernel on it for large batches.  K % 128 != 0: no packed form, nothing changes."""
        if self.packed is None:
            need = int(capi.lib().tce_w4a16_prepack_bytes(self.out_features, self.in_features, self.group_size))
            if need:
                buf = torch.empty(need + 256, dtype=torch.uint8, device=self.weight.device)
                off = (-buf.data_ptr()) % 256
                self._packed_storage = buf
                self.packed = buf[off:off + need]
                d = capi.W4A16Desc(M=1, N=self.out_features, K=self.in_features, group_size=self.group_size, qweight=_ptr(self.weight),
                                   scales=_ptr(self.scale), zeros=_ptr(self.zero_point))
                capi.check(capi.lib().tce_w4a16_prepack(d, self.packed.data_ptr(), _stream()))
        return self

    @classmethod
    def from_float(cls, w: torch.Tensor, group_size: int = quantize.QK4_6):
        return cls(*quantize.quantize_q4_6(w, group_size), group_size=group_size)

    @classmethod
    def load(cls, dirname: str, out_features: int, in_features: int, group_size: int = quantize.QK4_6, device="cuda"):
        return cls(*quantize.load_linear_q4_6(dirname, out_features, in_features, group_size, device), group_size=group_size)

    def desc(self, x: torch.Tensor, out: torch.Tensor, ldc: int = 0, flags: int = 0, gamma: torch.Tensor | None = None, eps: float = 0.0,
             allow_host: bool = False) -> capi.W4A16Desc:
        """The C-ABI descriptor of self(x) -> out.  Everything the kernels assume about the buffers is checked here (device
        memory, dtype, contiguity, sizes): a sliced / transposed x, an fp16 gamma or a short output would otherwise be a silent
        out-of-bounds device read.  allow_host=True skips only the is_cuda check (CPU tests of the sharding logic build
        descriptors for host tensors and never hand them to the library)."""
        m = x.numel() // self.in_features
        pairs = bool(flags & capi.TCE_W4_SILU_MUL_PAIRS)
        n_out = self.out_features // 2 if pairs else self.out_features
        for name, t, dt in (("x", x, torch.float16), ("out", out, torch.float16), ("gamma", gamma, torch.float32)):
            if t is None:
                continue
            if t.dtype != dt:
                raise ValueError(f"{name} must be {dt}, got {t.dtype}")
            if not t.is_contiguous():
                raise ValueError(f"{name} must be contiguous")
            if not allow_host and not t.is_cuda:
                raise ValueError(f"{name} must be a device tensor (no CPU fallback)")
        if x.numel() != m * self.in_features or m < 1:
            raise ValueError("x is not [M][in_features]")
        if out.numel() < (m - 1) * (ldc or n_out) + n_out:
            raise ValueError("out is too small for [M][N] with this ldc")
        if gamma is not None and gamma.numel() != self.in_features:
            raise ValueError("gamma must be fp32 [in_features]")
        return capi.W4A16Desc(M=m, N=self.out_features, K=self.in_features, group_size=self.group_size, A=x.data_ptr(),
                              qweight=self.weight.data_ptr(), scales=self.scale.data_ptr(), zeros=self.zero_point.data_ptr(),
                              C=out.data_ptr(), ldc=ldc, flags=flags | (capi.TCE_W4_ZERO_POINT_IS_8 if self.zeros_are_8 else 0),
                              rmsnorm_gamma=gamma.data_ptr() if gamma is not None else None, rmsnorm_eps=float(eps),
                              prepacked=self.packed.data_ptr() if self.packed is not None else None,
                              scratch=gemm_scratch(self.weight.device).data_ptr() if self.packed is not None and m > 128 else None)

    @classmethod
    def interleave(cls, gate: "Linear_half_int4", up: "Linear_half_int4") -> "Linear_half_int4":
        """One linear whose row 2n is gate's row n and row 2n+1 up's row n: the load-time layout of the fused
        gate_proj + up_proj + SiLuMul launch (TCE_W4_SILU_MUL_PAIRS).  A pure row permutation of the three q4_6 arrays,
        like the reference's offline qkv merge (llm/tools/llama_qkv_merger.py:27-48)."""
        assert gate.weight.shape == up.weight.shape and gate.group_size == up.group_size
        il = lambda a, b: torch.stack((a, b), dim=1).reshape(2 * a.shape[0], *a.shape[1:]).contiguous()
        return cls(il(gate.weight, up.weight), il(gate.scale, up.scale), il(gate.zero_point, up.zero_point), gate.group_size)

    def forward_silu_mul(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """silu(gate(x)) * up(x) for an interleaved gate/up linear, fp16 arithmetic of SiLuMul_half
        (Int4llamaDecoderLayer.cu:20-30, 96-102); one launch, output [..., out_features / 2]."""
        if out is None:
            out = torch.empty((*x.shape[:-1], self.out_features // 2), dtype=torch.float16, device=x.device)
        capi.check(capi.w4a16_forward(self.desc(x, out, flags=capi.TCE_W4_SILU_MUL_PAIRS), _stream()))
        return out

    def forward_add(self, x: torch.Tensor, residual_inout: torch.Tensor) -> torch.Tensor:
        """residual_inout = hadd(residual_inout, self(x)): o_proj / down_proj with the add_half behind it
        (Int4llamaDecoderLayer.cu:12-18, 86-88, 107-108) as one launch."""
        capi.check(capi.w4a16_forward(self.desc(x, residual_inout, flags=capi.TCE_W4_ADD_TO_C), _stream()))
        return residual_inout

    def forward(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        m = x.numel() // self.in_features
        assert self.out_features > 8 and self.out_features % 16 == 0  # linear.cu:16-17
        if out is None:
            out = torch.empty((*x.shape[:-1], self.out_features), dtype=torch.float16, device=x.device)
        if self.packed is not None:  # the descriptor carries the q4_mfma copy (what the C++ adapter does from its tensor cache)
            capi.check(capi.w4a16_forward(self.desc(x, out), _stream()))
            return out
        p = matmul_params(A=matrix(m, self.in_features, x), B=matrix(self.in_features // 8, self.out_features, self.weight),
                          C=matrix(m, self.out_features, out), half_scales=self.scale,
                          int32_zero_point=self.zero_point, block_size=self.group_size)
        self._op.gemv_forward_cuda(p)
        return out

    __call__ = forward

    def shard(self, rank: int, world: int) -> "Linear_half_int4":
        """Rows [rank*N/P, (rank+1)*N/P): a contiguous byte range of weights, scales and zeros in q4_6."""
        n = self.out_features
        # forward() keeps the reference wrapper's N % 16 == 0 (linear.cu:16-17): a shard must satisfy it too to be runnable
        if n % world or (n // world) % 16:
            raise ValueError(f"N={n} does not shard {world}-way into multiples of 16 rows (Linear_half_int4::forward's own requirement)")
        lo, hi = rank * (n // world), (rank + 1) * (n // world)
        return Linear_half_int4(self.weight[lo:hi].contiguous(), self.scale[lo:hi].contiguous(),
                                self.zero_point[lo:hi].contiguous(), self.group_size)

    def weight_bytes(self) -> int:
        return capi.algorithmic_bytes(1, self.out_features, self.in_features, self.group_size)


def forward_group(linears: list[Linear_half_int4], x: torch.Tensor, outs: list[torch.Tensor]) -> None:
    """Several linears reading the same activation (q/k/v, gate/up) as ONE launch (tce_w4a16_forward_group)."""
    capi.check(capi.w4a16_forward_group([l.desc(x, o) for l, o in zip(linears, outs)], _stream()))


def rmsnorm_half(x: torch.Tensor, gamma: torch.Tensor, eps: float, out: torch.Tensor | None = None) -> torch.Tensor:
    """LlamaRMSNorm_cuda::forward (generalT5LayerNorm, llm/src/ops/cuda/LlamaRMSNorm.cu:68-115): x fp16 [..., n], gamma fp32 [n]."""
    if out is None:
        out = torch.empty_like(x)
    n = x.shape[-1]
    if x.dtype != torch.float16 or out.dtype != torch.float16 or gamma.dtype != torch.float32 or gamma.numel() != n:
        raise ValueError("rmsnorm_half: x / out fp16 [..., n], gamma fp32 [n]")
    capi.check(capi.lib().tce_rmsnorm_half(_ptr(x), _ptr(gamma), _ptr(out), x.numel() // n, n, float(eps), _stream()))
    return out


def forward_group_rmsnorm(linears: list[Linear_half_int4], x: torch.Tensor, outs: list[torch.Tensor], gamma: torch.Tensor, eps: float) -> None:
    """RMSNorm(x) * gamma fed to several linears (q/k/v after input_layernorm, gate/up after post_attention_layernorm,
    Int4llamaDecoderLayer.cu:78, 92-99) as ONE launch: the normalisation happens while x is staged (decode, M = 1)."""
    import ctypes as C
    descs = [l.desc(x, o) for l, o in zip(linears, outs)]
    arr = (capi.W4A16Desc * len(descs))(*descs)
    if gamma.dtype != torch.float32 or gamma.numel() != x.shape[-1]:
        raise ValueError("gamma must be fp32 [in_features]")
    capi.check(capi.lib().tce_w4a16_forward_group_rmsnorm(arr, len(descs), _ptr(gamma), float(eps), C.c_void_p(_stream() or 0)))


class W8A8B8O8Linear:
    """int8 -> int8 linear, alpha/beta epilogue (llm/src/ops/W8A8B8O8Linear.cc:15-78); relu=True is W8A8B8O8LinearReLU
    (q_min = 0, W8A8B8O8LinearReLU.cc:32)."""

    def __init__(self, weight: torch.Tensor, bias_int8: torch.Tensor, alpha: float, beta: float, relu: bool = False):
        self.weight, self.bias, self.alpha, self.beta = weight, bias_int8, float(alpha), float(beta)
        self.q_min = 0 if relu else -128
        self._op = MatmulOperator()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, k = self.weight.shape
        m = x.numel() // k
        out = torch.empty((*x.shape[:-1], n), dtype=torch.int8, device=x.device)
        p = matmul_params(A=matrix(m, k, x), B=matrix(k, n, self.weight), C=matrix(m, n, out), bias=matrix(1, n, self.bias),
                          alpha=self.alpha, beta=self.beta)
        p.C.qparams.q_min, p.C.qparams.q_max = self.q_min, 127
        if m == 1:  # W8A8B8O8Linear.cc:61-75
            self._op.mat_mul_accelerator_int8_fast_32unroll_over_column(p)
        else:
            self._op.mat_mul_accelerator_int8_fast_2x2_32unroll(p)
        return out

    __call__ = forward


def layernorm_q_linears(x: torch.Tensor, ln_weight: torch.Tensor, ln_bias: torch.Tensor, linears: list, ln_out: torch.Tensor | None = None) -> list[torch.Tensor]:
    """LayerNormQ::forward (llm/src/ops/LayerNormQ.cc:12-52) and the W8A8 linears that read its output -- q_proj / k_proj / v_proj
    (Int8OPTAttention.cc:186-201) or fc1 -- as ONE launch for decode (x fp32 [m][k], m <= 8).  `linears`: W8A8B8O8Linear /
    W8A8BFP32OFP32Linear objects.  Bit-exact against the separate launches."""
    import ctypes as C
    k = x.shape[-1]
    m = x.numel() // k
    descs, outs = [], []
    for lin in linears:
        n = lin.weight.shape[0]
        fp32 = isinstance(lin, W8A8BFP32OFP32Linear)
        out = torch.empty((m, n), dtype=torch.float32 if fp32 else torch.int8, device=x.device)
        outs.append(out)
        descs.append(capi.W8A8Desc(M=m, N=n, K=k, batch=1, A=None, B=_ptr(lin.weight), bias=_ptr(lin.bias), C=_ptr(out), strideA=0, strideB=0, strideC=0,
                                   alpha=lin.alpha, beta=0.0 if fp32 else lin.beta, q_min=-128 if fp32 else lin.q_min, q_max=127,
                                   bias_kind=capi.TCE_BIAS_FP32 if fp32 else capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_FP32 if fp32 else capi.TCE_OUT_INT8,
                                   b_per_row=0))
    arr = (capi.W8A8Desc * len(descs))(*descs)
    if x.dtype != torch.float32 or ln_weight.dtype != torch.float32 or ln_bias.dtype != torch.float32:
        raise ValueError("layernorm_q_linears: x, weight, bias are fp32")
    capi.check(capi.lib().tce_layernorm_q_w8a8_group(_ptr(x), _ptr(ln_weight), _ptr(ln_bias), m, k, arr, len(descs), _ptr(ln_out), C.c_void_p(_stream() or 0)))
    return outs


class W8A8BFP32OFP32Linear:
    """int8 -> fp32 linear with fp32 bias (llm/src/ops/W8A8BFP32OFP32Linear.cc:12-73)."""

    def __init__(self, weight: torch.Tensor, bias_fp32: torch.Tensor, alpha: float):
        self.weight, self.bias, self.alpha = weight, bias_fp32, float(alpha)
        self._op = MatmulOperator()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, k = self.weight.shape
        m = x.numel() // k
        out = torch.empty((*x.shape[:-1], n), dtype=torch.float32, device=x.device)
        p = matmul_params(A=matrix(m, k, x), B=matrix(k, n, self.weight), C=matrix(m, n, out), bias=matrix(1, n, self.bias),
                          alpha=self.alpha)
        self._op.mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32(p)
        return out

    __call__ = forward


def bmm_s8t_s8n(a: torch.Tensor, b: torch.Tensor, alpha: float, out_fp32: bool) -> torch.Tensor:
    """BMM_S8T_S8N_F32T / BMM_S8T_S8N_S8T (llm/src/ops/BMM_S8T_S8N_F32T.cc:12-62, BMM_S8T_S8N_S8T.cc:12-63):
    a int8 [b][m][k], b int8 [b][n][k] -> [b][m][n]; the reference loops heads on the host, here one launch."""
    bs, m, k = a.shape
    n = b.shape[1]
    out = torch.empty((bs, m, n), dtype=torch.float32 if out_fp32 else torch.int8, device=a.device)
    d = capi.W8A8Desc(M=m, N=n, K=k, batch=bs, A=_ptr(a), B=_ptr(b), bias=None, C=_ptr(out), strideA=m * k, strideB=n * k,
                      strideC=m * n, alpha=float(alpha), beta=0.0, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE,
                      out_kind=capi.TCE_OUT_FP32 if out_fp32 else capi.TCE_OUT_INT8, b_per_row=0)
    capi.check(capi.w8a8_matmul(d, _stream()))
    return out
