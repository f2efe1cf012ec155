"""One SmoothQuant W8A8 OPT decoder layer on this library's calls -- the caller side of the int8 half of the hot path and of its "next"
row (LayerNormQ -> W8A8 fusion and the int8 BMMs of the attention).

The reference's Int8OPTDecoderLayer::forward (llm/src/nn_modules/Int8OPTDecoderLayer.cc:24-59) with Int8OPTAttention::forward inside it
(llm/src/nn_modules/Int8OPTAttention.cc:183-284) issues per layer: LayerNormQ, q / k / v projections, three `shape` transposes, the copy of
the whole past into the other cache buffer, the qk BMM (a host loop over heads), batch_Add, softmax, the int8 conversion, a transpose of
the whole value cache, the pv BMM (host loop over heads), `unshape`, out_proj, add, LayerNormQ, fc1, fc2, add.  Here a layer is

    decode (m <= 8 new rows)                                   prefill (m > 8)
    (decode, the default: steps 2-5 are ONE launch, tce_opt_attention_decode -- 5 launches per layer; `fused_attention = False` issues them separately)
    1  LayerNormQ + q, k, v        tce_layernorm_q_w8a8_group   tce_layernorm_q + 3 x tce_w8a8_matmul
    2  KV append (k as rows, v as columns)  tce_opt_kv_append   same
    3  qk BMM, all heads           tce_w8a8_matmul (the head's 64 columns of q as they lie: lda = embed; keys from the cache)
    4  + mask, softmax, -> int8    tce_opt_softmax_q
    5  pv BMM, all heads           tce_w8a8_matmul (V^T from the cache; each head writes its 64 columns of the [m][embed] row: no unshape)
    6  out_proj + residual add     tce_w8a8_matmul (fp32 out, accumulate)
    7  LayerNormQ + fc1 (ReLU)     tce_layernorm_q_w8a8_group   tce_layernorm_q + tce_w8a8_matmul
    8  fc2 + residual add          tce_w8a8_matmul (fp32 out, accumulate)

5 launches (decode; 8 with the attention's steps issued separately) / 12 (prefill); every buffer is allocated once, so a step is a fixed launch sequence (capturable in a hipGraph for a
fixed position).  Synthetic parameters; nothing here loads a checkpoint.  The arithmetic of every launch is the reference's
(tests/test_gpu_w8a8.py holds the layer to the oracle's composition of the same steps).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi
from .linear import _stream


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Int8OPTDecoderLayer:
    def __init__(self, embed: int, heads: int, ffn: int, max_keys: int, max_rows: int, device, seed: int = 0):
        assert embed % heads == 0
        self.embed, self.heads, self.hd, self.ffn, self.max_keys, self.max_rows = embed, heads, embed // heads, ffn, max_keys, max_rows
        g = torch.Generator(device=device).manual_seed(seed)
        i8 = lambda *s: torch.randint(-127, 128, s, dtype=torch.int8, device=device, generator=g)
        f32 = lambda *s: torch.empty(s, device=device).normal_(0, 1, generator=g)
        # parameters (shapes of llm/src/ops/W8A8B8O8Linear.cc:7-12, W8A8BFP32OFP32Linear.cc:6-10, LayerNormQ.cc; scales chosen so that the int8
        # outputs use their range without saturating everywhere)
        self.ln1_w, self.ln1_b = (1 + 0.1 * f32(embed)) * 20, f32(embed) * 3
        self.ln2_w, self.ln2_b = (1 + 0.1 * f32(embed)) * 20, f32(embed) * 3
        self.Wq, self.Wk, self.Wv = i8(embed, embed), i8(embed, embed), i8(embed, embed)
        self.bq, self.bk, self.bv = i8(embed), i8(embed), i8(embed)
        self.a_qkv, self.b_qkv = 2.0e-4, 0.05
        self.Wo, self.bo, self.a_o = i8(embed, embed), f32(embed) * 0.1, 1.0e-4
        self.W1, self.b1, self.a_1, self.b_1 = i8(ffn, embed), i8(ffn), 2.0e-4, 0.05
        self.W2, self.b2, self.a_2 = i8(embed, ffn), f32(embed) * 0.1, 5.0e-5
        self.a_qk, self.a_pv = 2.0e-3, 1.0 / 127.0
        self.fused_attention = True  # decode steps: KV append + both BMMs + softmax as one launch (False: the four separate launches)
        # state and scratch, allocated once
        z8 = lambda *s: torch.zeros(s, dtype=torch.int8, device=device)
        self.k_cache = z8(heads, max_keys, self.hd)
        self.vt_cache = z8(heads, self.hd, max_keys)
        self.ln_out, self.q, self.k, self.v = z8(max_rows, embed), z8(max_rows, embed), z8(max_rows, embed), z8(max_rows, embed)
        self.attn, self.fc1 = z8(max_rows, embed), z8(max_rows, ffn)
        self.ldp = (max_keys + 15) // 16 * 16
        self.scores = torch.zeros((heads, max_rows, max_keys), dtype=torch.float32, device=device)
        self.probs = z8(heads, max_rows, self.ldp)

    # ---- descriptors ----
    def _lin(self, m, A, W, bias, out, alpha, beta=0.0, fp32=False, q_min=-128, accumulate=False):
        n, k = W.shape
        return capi.W8A8Desc(M=m, N=n, K=k, batch=1, A=_p(A), B=_p(W), bias=_p(bias), C=_p(out), alpha=alpha, beta=beta, q_min=q_min, q_max=127,
                             bias_kind=capi.TCE_BIAS_FP32 if fp32 else capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_FP32 if fp32 else capi.TCE_OUT_INT8,
                             accumulate=1 if accumulate else 0)

    def step(self, hidden: torch.Tensor, pos: int, mask: torch.Tensor) -> None:
        """hidden fp32 [m][embed], updated in place (the residual stream); the m new rows take positions pos .. pos + m - 1; mask fp32 [m][pos + m]
        additive (Int8OPTDecoderLayer_input.attention_mask)."""
        L, st = capi.lib(), C.c_void_p(_stream() or 0)
        m, E, H, hd = hidden.shape[0], self.embed, self.heads, self.hd
        tgz = pos + m
        assert hidden.dtype == torch.float32 and hidden.is_contiguous() and m <= self.max_rows and tgz <= self.max_keys and tuple(mask.shape) == (m, tgz)
        fused = m <= 8
        # 1. LayerNormQ + q, k, v
        qkv = [self._lin(m, self.ln_out, W, b, o, self.a_qkv, self.b_qkv) for W, b, o in ((self.Wq, self.bq, self.q), (self.Wk, self.bk, self.k), (self.Wv, self.bv, self.v))]
        if fused:
            capi.check(L.tce_layernorm_q_w8a8_group(_p(hidden), _p(self.ln1_w), _p(self.ln1_b), m, E, (capi.W8A8Desc * 3)(*qkv), 3, None, st))
        else:
            capi.check(L.tce_layernorm_q(_p(hidden), _p(self.ln1_w), _p(self.ln1_b), _p(self.ln_out), m, E, st))
            for d in qkv:
                capi.check(L.tce_w8a8_matmul(C.byref(d), st))
        if fused and self.fused_attention:
            # 2-5 as ONE launch for a decode step (tce_opt_attention_decode: the append, qk, + mask / softmax / int8, pv -- bit-identical to the four launches below)
            capi.check(L.tce_opt_attention_decode(_p(self.q), _p(self.k), _p(self.v), _p(self.k_cache), _p(self.vt_cache), _p(mask), _p(self.attn), H, hd, m, pos, self.max_keys, E,
                                                  self.a_qk, self.a_pv, st))
            return self._rest(hidden, m, fused, L, st)
        # 2. KV append
        capi.check(L.tce_opt_kv_append(_p(self.k), _p(self.v), _p(self.k_cache), _p(self.vt_cache), H, hd, m, pos, self.max_keys, st))
        # 3. qk BMM: head h contracts its 64 columns of q with its keys; scores [heads][m][tgz] fp32 (BMM_S8T_S8N_F32T.cc:12-62)
        if m == 1:  # the reference's decode form: row h of A (= head h's query) has its own B_h (the *_batch member, :45-52)
            qk = capi.W8A8Desc(M=H, N=tgz, K=hd, batch=1, A=_p(self.q), B=_p(self.k_cache), C=_p(self.scores), alpha=self.a_qk, q_min=-128, q_max=127,
                               bias_kind=capi.TCE_BIAS_NONE, out_kind=capi.TCE_OUT_FP32, b_per_row=1, strideB=self.max_keys * hd)
        else:
            qk = capi.W8A8Desc(M=m, N=tgz, K=hd, batch=H, A=_p(self.q), B=_p(self.k_cache), C=_p(self.scores), strideA=hd, strideB=self.max_keys * hd, strideC=m * tgz,
                               lda=E, alpha=self.a_qk, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE, out_kind=capi.TCE_OUT_FP32)
        capi.check(L.tce_w8a8_matmul(C.byref(qk), st))
        # 4. + mask, softmax, int8 probabilities [heads][m][ldp]
        ldp = (tgz + 15) // 16 * 16
        capi.check(L.tce_opt_softmax_q(_p(self.scores), _p(mask), _p(self.probs), H, m, tgz, ldp, st))
        # 5. pv BMM: probabilities x V^T -> the head's 64 columns of the [m][embed] row (BMM_S8T_S8N_S8T.cc:12-63; the reference's unshape is the ldc)
        if m == 1:
            pv = capi.W8A8Desc(M=H, N=hd, K=tgz, batch=1, A=_p(self.probs), B=_p(self.vt_cache), C=_p(self.attn), alpha=self.a_pv, q_min=-128, q_max=127,
                               bias_kind=capi.TCE_BIAS_NONE, out_kind=capi.TCE_OUT_INT8, b_per_row=1, strideB=hd * self.max_keys, lda=ldp, ldb=self.max_keys)
        else:
            pv = capi.W8A8Desc(M=m, N=hd, K=tgz, batch=H, A=_p(self.probs), B=_p(self.vt_cache), C=_p(self.attn), strideA=m * ldp, strideB=hd * self.max_keys, strideC=hd,
                               lda=ldp, ldb=self.max_keys, ldc=E, alpha=self.a_pv, q_min=-128, q_max=127, bias_kind=capi.TCE_BIAS_NONE, out_kind=capi.TCE_OUT_INT8)
        capi.check(L.tce_w8a8_matmul(C.byref(pv), st))
        self._rest(hidden, m, fused, L, st)

    def _rest(self, hidden, m, fused, L, st):
        E = self.embed
        # 6. out_proj + residual add (W8A8BFP32OFP32Linear, then `add`: Int8OPTDecoderLayer.cc:39)
        d = self._lin(m, self.attn, self.Wo, self.bo, hidden, self.a_o, fp32=True, accumulate=True)
        capi.check(L.tce_w8a8_matmul(C.byref(d), st))
        # 7. final_layer_norm + fc1 (ReLU: q_min = 0, W8A8B8O8LinearReLU.cc:32)
        f1 = self._lin(m, self.ln_out, self.W1, self.b1, self.fc1, self.a_1, self.b_1, q_min=0)
        if fused:
            capi.check(L.tce_layernorm_q_w8a8_group(_p(hidden), _p(self.ln2_w), _p(self.ln2_b), m, E, (capi.W8A8Desc * 1)(f1), 1, None, st))
        else:
            capi.check(L.tce_layernorm_q(_p(hidden), _p(self.ln2_w), _p(self.ln2_b), _p(self.ln_out), m, E, st))
            capi.check(L.tce_w8a8_matmul(C.byref(f1), st))
        # 8. fc2 + residual add (:51-54)
        d = self._lin(m, self.fc1, self.W2, self.b2, hidden, self.a_2, fp32=True, accumulate=True)
        capi.check(L.tce_w8a8_matmul(C.byref(d), st))

    def launches(self, m: int) -> int:
        return (5 if self.fused_attention else 8) if m <= 8 else 12

    def int8_ops(self, m: int, tgz: int) -> int:
        """Multiply-accumulates x 2 of one step (the linears and both BMMs)."""
        E, F = self.embed, self.ffn
        return 2 * m * (4 * E * E + 2 * E * F + 2 * tgz * E)
