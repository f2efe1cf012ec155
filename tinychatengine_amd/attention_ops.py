"""Python mirror of the two fp16 operators between the q/k/v and the o linears of the reference's Llama attention
(`BMM_F16T`, `softmax`: llm/src/ops/cuda/BMM_F16T.cu, softmax.cu), over the C ABI (`tce_bmm_f16t`, `tce_softmax_half`).
torch is only the owner of the device buffers."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class BMM_F16T:
    """`BMM_F16T(alpha)`; `forward(a, weight, c)`: a [batch][M][K], weight [batch][N][K] -> c [batch][M][N] =
    hmul(alpha, sum over k by binary16 fma) -- BMM_F16T::forward (BMM_F16T.cu:52-76)."""

    def __init__(self, alpha: float = 1.0):
        self.alpha_bits = int(np.array([alpha], np.float16).view(np.uint16)[0])

    def forward(self, a: torch.Tensor, weight: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
        assert a.dtype == weight.dtype == c.dtype == torch.float16 and a.is_contiguous() and weight.is_contiguous() and c.is_contiguous()
        batch, M, K = a.shape
        assert weight.shape[0] == batch and weight.shape[2] == K and tuple(c.shape) == (batch, M, weight.shape[1])  # BMM_F16T.cu:57-60
        capi.check(capi.lib().tce_bmm_f16t(C.c_void_p(a.data_ptr()), C.c_void_p(weight.data_ptr()), C.c_void_p(c.data_ptr()), batch, M,
                                           weight.shape[1], K, self.alpha_bits, C.c_void_p(_stream())))
        return c


def softmax(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """softmax over the last dimension with softmax_cuda's arithmetic (softmax.cu:4-40)."""
    assert x.dtype == torch.float16 and x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    n = x.shape[-1]
    capi.check(capi.lib().tce_softmax_half(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel() // n, n, C.c_void_p(_stream())))
    return out


def attention_decode(q: torch.Tensor, K: torch.Tensor, Vt: torch.Tensor, out: torch.Tensor, alpha: float, mask: torch.Tensor | None = None) -> torch.Tensor:
    """One decode step of all heads as one launch: q [heads][hd], K [heads][keys][hd], Vt [heads][hd][keys], mask [keys] or None ->
    out [heads][hd]; the reference's qk_bmm -> batch_Add -> check_inf_half -> softmax -> pv_bmm (Int4llamaAttention.cu:184-211)."""
    heads, hd = q.shape
    keys = K.shape[1]
    assert tuple(K.shape) == (heads, keys, hd) and tuple(Vt.shape) == (heads, hd, keys) and tuple(out.shape) == (heads, hd)
    assert all(x.dtype == torch.float16 and x.is_contiguous() for x in (q, K, Vt, out))
    alpha_bits = int(np.array([alpha], np.float16).view(np.uint16)[0])
    capi.check(capi.lib().tce_attention_decode_f16(C.c_void_p(q.data_ptr()), C.c_void_p(K.data_ptr()), C.c_void_p(Vt.data_ptr()),
                                                   C.c_void_p(mask.data_ptr() if mask is not None else 0), C.c_void_p(out.data_ptr()), heads, keys, hd,
                                                   alpha_bits, C.c_void_p(_stream())))
    return out


def rotary_pos_emb(q: torch.Tensor | None, k: torch.Tensor | None, cos: torch.Tensor, sin: torch.Tensor, start_idx: int) -> None:
    """RotaryPosEmb_cuda_forward in place: q, k [heads][len][hd] (either may be None), cos / sin [positions][hd] (RotaryPosEmb.cu:4-34)."""
    ref = q if q is not None else k
    heads, ln, hd = ref.shape
    assert cos.dtype == sin.dtype == torch.float16 and cos.shape[-1] == hd and cos.shape[0] >= start_idx + ln
    capi.check(capi.lib().tce_rope_half(C.c_void_p(q.data_ptr() if q is not None else 0), C.c_void_p(k.data_ptr() if k is not None else 0),
                                        C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), heads, ln, hd, start_idx, C.c_void_p(_stream())))


class DecodeAttention:
    """One decode step of the Llama attention block between the fused q/k/v linear and o_proj as ONE launch
    (tce_attention_decode_step_f16; Int4llamaAttention.cu:130-217 without its copies and transposes): fixed-capacity caches
    [heads][max_keys][hd], RoPE on q and the new key with the reference's binary16 arithmetic, fp32 online softmax over key chunks."""

    def __init__(self, heads: int, head_dim: int, max_keys: int, device, cos: torch.Tensor | None = None, sin: torch.Tensor | None = None,
                 kv_heads: int | None = None):
        """kv_heads < heads: grouped-query attention (tce_attention_decode_step_gqa_f16; query head i reads key / value head
        i // (heads // kv_heads), Int4llamaAttention.cc:166-185); the caches are [kv_heads][max_keys][hd] and the projection's row is
        [heads + 2 * kv_heads][hd]."""
        self.heads, self.hd, self.max_keys = heads, head_dim, max_keys
        self.kv_heads = heads if kv_heads is None else kv_heads
        self.k_cache = torch.zeros((self.kv_heads, max_keys, head_dim), dtype=torch.float16, device=device)
        self.v_cache = torch.zeros((self.kv_heads, max_keys, head_dim), dtype=torch.float16, device=device)
        need = int(capi.lib().tce_attention_decode_workspace_bytes(heads, max_keys, head_dim))
        if need == 0:
            raise ValueError("unsupported attention shape (head_dim must be 128)")
        self.workspace = torch.zeros(need, dtype=torch.uint8, device=device)  # zeroed once: the arrival counters
        self.cos, self.sin = cos, sin
        self.alpha_bits = int(np.array([1.0 / np.sqrt(head_dim)], np.float16).view(np.uint16)[0])

    def step(self, qkv: torch.Tensor, pos: int, out: torch.Tensor | None = None, mask: torch.Tensor | None = None, pos_device: torch.Tensor | None = None,
             defer: bool = False) -> torch.Tensor:
        """pos_device (int32 device tensor of one element): the position is read on the device and `pos` only bounds it
        (tce_attention_decode_step_pos_f16): the launch can be captured once and replayed for growing contexts.
        defer = True (tce_attention_decode_step_deferred_f16): with several chunks per head the launch stops at its partial states -- `out` is then NOT the
        attention output until a linear issued through tce_w4a16_forward_deferred_attention with `self.deferred` has consumed it (DecoderBlock.step does)."""
        assert qkv.dtype == torch.float16 and qkv.is_contiguous() and qkv.numel() == (self.heads + 2 * self.kv_heads) * self.hd and qkv.is_cuda
        if out is None:
            out = torch.empty((self.heads, self.hd), dtype=torch.float16, device=qkv.device)
        p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
        if pos_device is not None:
            assert pos_device.dtype == torch.int32 and pos_device.is_cuda and pos_device.numel() == 1
        if defer:  # round 5: the launch ends at its partial states; the linear that reads `out` combines them in its prologue (step_deferred's docstring)
            info = capi.AttnDeferred()
            capi.check(capi.lib().tce_attention_decode_step_deferred_f16(p(qkv), p(self.k_cache), p(self.v_cache), p(self.cos), p(self.sin), p(mask), p(out), p(self.workspace),
                                                                         self.heads, self.kv_heads, self.hd, self.max_keys, p(pos_device), int(pos), self.alpha_bits, C.byref(info),
                                                                         C.c_void_p(_stream())))
            self.deferred = info
            return out
        capi.check(capi.lib().tce_attention_decode_step_pos_f16(p(qkv), p(self.k_cache), p(self.v_cache), p(self.cos), p(self.sin), p(mask), p(out), p(self.workspace),
                                                                self.heads, self.kv_heads, self.hd, self.max_keys, p(pos_device), int(pos), self.alpha_bits, C.c_void_p(_stream())))
        return out

    def prefill(self, qkv: torch.Tensor, pos: int, out: torch.Tensor | None = None, mask: torch.Tensor | None = None, causal: bool = True) -> torch.Tensor:
        """m > 1 new rows on top of `pos` cached keys (tce_attention_prefill_f16; Int4llamaAttention.cu:116-229 with sqlen > 1): qkv [m][(heads + 2 kv_heads) * hd]
        as the fused projection writes it, mask fp16 [m][pos + m] additive or None, causal: row r sees keys 0 .. pos + r.  Appends rows pos .. pos + m - 1 to the
        caches and returns out [m][heads * hd] (o_proj's input rows)."""
        assert qkv.dtype == torch.float16 and qkv.is_contiguous() and qkv.dim() == 2 and qkv.shape[1] == (self.heads + 2 * self.kv_heads) * self.hd and qkv.is_cuda
        m = qkv.shape[0]
        if out is None:
            out = torch.empty((m, self.heads * self.hd), dtype=torch.float16, device=qkv.device)
        need = int(capi.lib().tce_attention_prefill_workspace_bytes(self.heads, m, self.hd))
        ws = getattr(self, "_prefill_ws", None)
        if ws is None or ws.numel() < need:
            ws = self._prefill_ws = torch.empty(need, dtype=torch.uint8, device=qkv.device)
        if mask is not None:
            assert mask.dtype == torch.float16 and mask.is_contiguous() and tuple(mask.shape) == (m, pos + m)
        p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
        capi.check(capi.lib().tce_attention_prefill_f16(p(qkv), 0, p(self.k_cache), p(self.v_cache), p(self.cos), p(self.sin), p(mask), 0, 1 if causal else 0, p(out), 0, p(ws),
                                                        self.heads, self.kv_heads, self.hd, self.max_keys, int(pos), m, self.alpha_bits, C.c_void_p(_stream())))
        return out
