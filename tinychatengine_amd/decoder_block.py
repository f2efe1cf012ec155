"""One decode token through whole decoder blocks -- the caller side of the hot path and of its "next" rows, on this
library's calls only.

The reference's Int4llamaDecoderLayer::forward (llm/src/nn_modules/cuda/Int4llamaDecoderLayer.cu:73-115) with
Int4llamaAttention::forward inside it (Int4llamaAttention.cu:116-229) issues, per layer and token, ~20 kernel launches and 4 memcpys
per head: two RMSNorms, qkv_proj, shape_qkv, RoPE, the KV copies, qk_bmm, batch_Add, check_inf, softmax, transpose, pv_bmm, unshape,
o_proj, add, gate_proj, up_proj, SiLuMul, down_proj, add.  Here a layer is FIVE launches:

    1  input_layernorm + q/k/v projection        tce_w4a16_forward, descriptor with rmsnorm_gamma            
    2  RoPE + KV append + attention              tce_attention_decode_step_f16                                
    3  o_proj + residual add                     tce_w4a16_forward, TCE_W4_ADD_TO_C
    4  post_attention_layernorm + gate/up + SiLU*mul   tce_w4a16_forward, rmsnorm_gamma + TCE_W4_SILU_MUL_PAIRS (gate / up rows interleaved)
    5  down_proj + residual add                  tce_w4a16_forward, TCE_W4_ADD_TO_C

Used by tests/test_gpu_block.py (against a float64 evaluation of the same layer on the dequantized weights) and by bench.py's
`other_configs.decode_with_attention` leg.  Synthetic weights; nothing here loads a checkpoint.
"""
from __future__ import annotations

import numpy as np
import ctypes as C

import torch

from . import capi
from .attention_ops import DecodeAttention
from .linear import Linear_half_int4, _stream


class DecoderBlock:
    def __init__(self, hidden: int, heads: int, ffn: int, max_keys: int, device, cos: torch.Tensor, sin: torch.Tensor, seed: int = 0,
                 group_size: int = 128, eps: float = 1e-6, kv_heads: int | None = None, prepack: bool = True, defer_combine: bool | None = None):
        """kv_heads < heads: grouped-query attention (Llama-3-8B: 32 / 8, llm/include/model.h:83) -- the fused projection is
        (heads + 2 * kv_heads) * 128 rows wide and the caches hold kv_heads heads."""
        assert hidden % heads == 0 and hidden // heads == 128, "the attention step is built for head_dim 128 (Llama)"
        self.hidden, self.heads, self.ffn, self.eps = hidden, heads, ffn, eps
        self.kv_heads = heads if kv_heads is None else kv_heads
        # the attention combine in o_proj's prologue (tce_attention_decode_step_deferred_f16 + tce_w4a16_forward_deferred_attention): bit-identical, and MEASURED at
        # +0.5 .. +0.8 % of the whole token (-0.4 % at 2048 keys; profiles/r5/deferred_attn_token_ab.jsonl) -- under the 1 % the round's stop rule asks for: opt-in
        self.defer_combine = bool(defer_combine) and prepack and group_size == 128 and hidden % 1024 == 0 and hidden <= 4096
        g = torch.Generator(device=device).manual_seed(seed)
        rnd = lambda n, k: torch.empty(n, k, device=device).normal_(0.0, k ** -0.5, generator=g)
        self.qkv = Linear_half_int4.from_float(rnd((heads + 2 * self.kv_heads) * 128, hidden), group_size)   # rows: q | k | v, head-major (llama_qkv_merger.py:27-48)
        self.o = Linear_half_int4.from_float(rnd(hidden, hidden), group_size)
        self.gate = Linear_half_int4.from_float(rnd(ffn, hidden), group_size)
        self.up = Linear_half_int4.from_float(rnd(ffn, hidden), group_size)
        self.gate_up = Linear_half_int4.interleave(self.gate, self.up)                  # row 2n = gate n, row 2n + 1 = up n
        self.down = Linear_half_int4.from_float(rnd(hidden, ffn), group_size)
        self.gamma1 = (1.0 + 0.1 * torch.empty(hidden, device=device).normal_(0, 1, generator=g)).float()
        self.gamma2 = (1.0 + 0.1 * torch.empty(hidden, device=device).normal_(0, 1, generator=g)).float()
        if prepack:  # load-time re-layout: the four launches of a decode step then run on the packed copies (csrc/w4a16_gemv_i8.hip, fused prologue / epilogues included)
            self.prepare_prefill()
        self.attention = DecodeAttention(heads, 128, max_keys, device, cos, sin, kv_heads=self.kv_heads)
        e = lambda n: torch.empty((1, n), dtype=torch.float16, device=device)
        self.qkv_out, self.attn_out, self.act = e((heads + 2 * self.kv_heads) * 128), e(hidden), e(ffn)
        self.xn2 = e(hidden)  # post_attention_layernorm of the residual row (step_chained)

    def step(self, hidden_state: torch.Tensor, pos: int, pos_device: torch.Tensor | None = None) -> None:
        """hidden_state fp16 [1][hidden], updated in place (it is the residual stream).  pos_device: the position on the device (`pos` then
        bounds it): the five launches can be captured once and replayed token after token."""
        st = _stream()
        capi.check(capi.w4a16_forward(self.qkv.desc(hidden_state, self.qkv_out, gamma=self.gamma1, eps=self.eps), st))
        if self.defer_combine:  # round 5: the attention step ends at its per-chunk partial states, o_proj's prologue combines them (same row, bit for bit; one launch boundary instead of a cross-workgroup exchange)
            self.attention.step(self.qkv_out.view(-1), pos, out=self.attn_out.view(self.heads, 128), pos_device=pos_device, defer=True)
            capi.check(capi.lib().tce_w4a16_forward_deferred_attention(C.byref(self.o.desc(self.attn_out, hidden_state, flags=capi.TCE_W4_ADD_TO_C)), C.byref(self.attention.deferred),
                                                                       C.c_void_p(pos_device.data_ptr() if pos_device is not None else 0), int(pos), C.c_void_p(st)))
        else:
            self.attention.step(self.qkv_out.view(-1), pos, out=self.attn_out.view(self.heads, 128), pos_device=pos_device)
            capi.check(capi.w4a16_forward(self.o.desc(self.attn_out, hidden_state, flags=capi.TCE_W4_ADD_TO_C), st))
        capi.check(capi.w4a16_forward(self.gate_up.desc(hidden_state, self.act, flags=capi.TCE_W4_SILU_MUL_PAIRS, gamma=self.gamma2, eps=self.eps), st))
        capi.check(capi.w4a16_forward(self.down.desc(self.act, hidden_state, flags=capi.TCE_W4_ADD_TO_C), st))

    def step_chained(self, hidden_state: torch.Tensor, xn_in: torch.Tensor, pos: int, next_gamma: torch.Tensor, xn_out: torch.Tensor, workspace: torch.Tensor,
                     pos_device: torch.Tensor | None = None) -> None:
        """The same five launches with the norms on the PRODUCER side (round 4, tce_w4a16_forward_residual_rmsnorm): `xn_in` is input_layernorm(hidden_state), already
        formed by the launch that produced hidden_state (the previous layer's down_proj; tce_rmsnorm_half for the first layer); o_proj's residual epilogue forms
        post_attention_layernorm, down_proj's forms the NEXT layer's input_layernorm (`next_gamma`) into `xn_out`.  q/k/v and gate/up are plain launches.
        Same bits as step(): every piece has the arithmetic of the separate kernels."""
        st = _stream()
        capi.check(capi.w4a16_forward(self.qkv.desc(xn_in, self.qkv_out), st))
        self.attention.step(self.qkv_out.view(-1), pos, out=self.attn_out.view(self.heads, 128), pos_device=pos_device)
        capi.check(capi.w4a16_forward_residual_rmsnorm(self.o.desc(self.attn_out, hidden_state, flags=capi.TCE_W4_ADD_TO_C), self.gamma2.data_ptr(), self.eps,
                                                        self.xn2.data_ptr(), workspace.data_ptr(), st))
        capi.check(capi.w4a16_forward(self.gate_up.desc(self.xn2, self.act, flags=capi.TCE_W4_SILU_MUL_PAIRS), st))
        capi.check(capi.w4a16_forward_residual_rmsnorm(self.down.desc(self.act, hidden_state, flags=capi.TCE_W4_ADD_TO_C), next_gamma.data_ptr(), self.eps,
                                                        xn_out.data_ptr(), workspace.data_ptr(), st))

    LAUNCHES = 5

    # ---- m > 1 rows: a prompt, or a chunk of one (the same reference code path with sqlen > 1) ----
    def prepare_prefill(self) -> "DecoderBlock":
        """Load-time work for the prefill path: the q4_mfma copies the 128-row GEMM reads (tce_w4a16_prepack)."""
        for l in (self.qkv, self.o, self.gate_up, self.down):
            l.prepack()
        return self

    def prefill(self, hidden_rows: torch.Tensor, pos: int) -> None:
        """hidden_rows fp16 [m][hidden], updated in place; the m rows take positions pos .. pos + m - 1.  EIGHT launches where the reference issues its ~20 + the
        per-head copies: RMSNorm, q/k/v GEMM, [rotation + KV append], attention, o_proj GEMM + residual, RMSNorm, gate/up GEMM with the SiLU*mul pair epilogue,
        down_proj GEMM + residual (the GEMMs have no fused norm prologue: that is a decode form)."""
        from .linear import rmsnorm_half
        m = hidden_rows.shape[0]
        assert hidden_rows.dtype == torch.float16 and hidden_rows.is_contiguous() and hidden_rows.shape[1] == self.hidden
        st = _stream()
        buf = getattr(self, "_pf", None)
        if buf is None or buf[0].shape[0] < m:
            e = lambda n: torch.empty((m, n), dtype=torch.float16, device=hidden_rows.device)
            buf = self._pf = (e(self.hidden), e((self.heads + 2 * self.kv_heads) * 128), e(self.hidden), e(self.ffn), e(self.ffn))
        xn, qkv, attn, g, u = (t[:m] for t in buf)
        rmsnorm_half(hidden_rows, self.gamma1, self.eps, out=xn)
        capi.check(capi.w4a16_forward(self.qkv.desc(xn, qkv), st))
        self.attention.prefill(qkv, pos, out=attn, causal=True)
        capi.check(capi.w4a16_forward(self.o.desc(attn, hidden_rows, flags=capi.TCE_W4_ADD_TO_C), st))
        rmsnorm_half(hidden_rows, self.gamma2, self.eps, out=xn)
        if m > 128 and self.gate_up.packed is not None:  # gate + up + SiLU*mul as ONE GEMM launch on the interleaved rows (pair epilogue of the 128-row GEMM)
            capi.check(capi.w4a16_forward(self.gate_up.desc(xn, g, flags=capi.TCE_W4_SILU_MUL_PAIRS), st))
        else:
            capi.check(capi.w4a16_forward(self.gate.desc(xn, g), st))
            capi.check(capi.w4a16_forward(self.up.desc(xn, u), st))
            capi.check(capi.lib().tce_silu_mul_half(g.data_ptr(), u.data_ptr(), g.numel(), st))
        capi.check(capi.w4a16_forward(self.down.desc(g, hidden_rows, flags=capi.TCE_W4_ADD_TO_C), st))

    PREFILL_LAUNCHES = 8  # with pre-packed weights and m > 128 (otherwise 10: gate, up and SiLU*mul as three launches)

    def linear_bytes(self) -> int:
        return sum(capi.algorithmic_bytes(1, l.out_features, l.in_features, l.group_size) for l in (self.qkv, self.o, self.gate_up, self.down))


def dequantize(lin: Linear_half_int4) -> np.ndarray:
    """fp64 [N][K] = scale * (code - zero point) of a q4_6 linear (for the float64 reference of the tests)."""
    qw = lin.weight.cpu().numpy().view(np.uint32)
    n, k8 = qw.shape
    codes = ((qw[:, :, None] >> (np.arange(8, dtype=np.uint32) * 4)) & 0xF).reshape(n, k8 * 8).astype(np.float64)
    g = lin.group_size
    ng = k8 * 8 // g
    sc = lin.scale.cpu().numpy().astype(np.float64)[:, :ng]
    zp = lin.zero_point.cpu().numpy().view(np.uint32)
    z = ((zp[:, :, None] >> (np.arange(8, dtype=np.uint32) * 4)) & 0xF).reshape(n, -1)[:, :ng].astype(np.float64)
    return (codes.reshape(n, ng, g) - z[:, :, None]).reshape(n, -1) * np.repeat(sc, g, axis=1)
