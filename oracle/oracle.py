"""ctypes front-end of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product package ``tinychatengine_amd``
never does (tests/test_boundary.py greps for that).

Two libraries are wrapped with identical call signatures:

* ``orc``  -> ``oracle/libtce_oracle.so``      our plain-C restatement (oracle/tce_oracle.c)
* ``ref``  -> ``oracle/_ref/libtce_ref.so``    the reference's own sources compiled from
  /root/reference by ``oracle/Makefile`` (present here and -- as a prebuilt file -- on the GPU box)
* ``ref_avx`` -> ``oracle/_ref/libtce_ref_avx.so``  the reference's AVX2 W4A8 fast path (timed CPU baseline only)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libtce_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libtce_ref.so")
REF_AVX_SO = os.path.join(_HERE, "_ref", "libtce_ref_avx.so")
REF_X86NAIVE_SO = os.path.join(_HERE, "_ref", "libtce_ref_x86naive.so")
REF_ARMNAIVE_SO = os.path.join(_HERE, "_ref", "libtce_ref_armnaive.so")
REF_METALNAIVE_SO = os.path.join(_HERE, "_ref", "libtce_ref_metalnaive.so")


def build(with_ref: bool | None = None) -> None:
    """Compile the oracle (always) and the reference build (when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if with_ref is None:
        with_ref = os.path.isdir("/root/reference/kernels")
    if with_ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
        # the reference's own level-2 callers linked against this repository's adapter (oracle/_ref/l2_harness; tests/test_l2_link.py);
        # needs libtce_hip.so, i.e. tinychatengine_amd.build first -- skipped quietly when that has not run yet
        lib = os.path.join(_HERE, "..", "tinychatengine_amd", "lib", "libtce_hip.so")
        if os.path.exists(lib) and os.path.isdir("/root/reference/llm/src/ops"):
            subprocess.check_call(["make", "-s", "-C", _HERE, "l2link"])
        # the reference's CUDA glue kernels run on the CPU through oracle/cuda_emul (oracle/_ref/glue_harness; tests/test_oracle_glue.py)
        if os.path.isdir("/root/reference/llm/src/ops/cuda"):
            subprocess.check_call(["make", "-s", "-C", _HERE, "glue"])


def _p(a: np.ndarray | None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle wants contiguous arrays"
    return a.ctypes.data_as(C.c_void_p)


class _Lib:
    """Common call surface of libtce_oracle.so (prefix orc_) and libtce_ref.so (prefix ref_)."""

    def __init__(self, path: str, prefix: str):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)

    def _f(self, name, restype=None):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        return fn

    # ---- W4 ----
    def naive_mat_mul_int4(self, A, B, scales, zero_point, M, N, K, G):
        A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.uint8)
        scales = np.ascontiguousarray(scales, np.float32)
        out = np.empty((M, N), np.float32)
        self._f("naive_mat_mul_int4")(C.c_int(M), C.c_int(N), C.c_int(K), C.c_int(G), _p(A), _p(B), _p(scales),
                                      C.c_float(zero_point), _p(out))
        return out

    def naive_mat_mul_int4_with_offset(self, A, B, scales, offset, zero_point, M, N, K, G):
        A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.uint8)
        scales = np.ascontiguousarray(scales, np.float32); offset = np.ascontiguousarray(offset, np.float32)
        out = np.empty((M, N), np.float32)
        self._f("naive_mat_mul_int4_with_offset")(C.c_int(M), C.c_int(N), C.c_int(K), C.c_int(G), _p(A), _p(B),
                                                  _p(scales), _p(offset), C.c_float(zero_point), _p(out))
        return out

    def ref_int4_fast(self, A, B, scales, offset, M, N, K, G=32, b_row=None):
        b_row = K // 2 if b_row is None else b_row  # what Linear_FP_int4::forward_fast passes (linear.cc:138-139)
        A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.uint8)
        scales = np.ascontiguousarray(scales, np.float32); offset = np.ascontiguousarray(offset, np.float32)
        out = np.empty((M, N), np.float32)
        rc = self._f("ref_int4_fast", C.c_int)(C.c_int(M), C.c_int(N), C.c_int(K), C.c_int(G), C.c_int(b_row), _p(A), _p(B),
                                               _p(scales), _p(offset), _p(out))
        if rc != 0:
            raise ValueError("block size must be 32")
        return out

    def naive_mat_mul_fp16_int4(self, A_f16, qweight_q4_5, scales_f16, M, N, K, G):
        A = np.ascontiguousarray(A_f16).view(np.uint16); q = np.ascontiguousarray(qweight_q4_5).view(np.uint32)
        s = np.ascontiguousarray(scales_f16).view(np.uint16)
        out = np.empty((M, N), np.uint16)
        self._f("naive_mat_mul_fp16_int4")(C.c_int(M), C.c_int(N), C.c_int(K), C.c_int(G), _p(A), _p(q), _p(s), _p(out))
        return out.view(np.float16)

    # ---- W8A8 ----
    def int8_matmul_bias_i8(self, A, B, bias, alpha, beta, qmin, qmax, M, N, K, over_column=False):
        A = np.ascontiguousarray(A, np.int8); B = np.ascontiguousarray(B, np.int8); bias = np.ascontiguousarray(bias, np.int8)
        out = np.empty((M, N), np.int8)
        name = "int8_matmul_bias_i8_over_column" if (over_column and self.prefix == "ref_") else "int8_matmul_bias_i8"
        self._f(name)(C.c_int(M), C.c_int(N), C.c_int(K), _p(A), _p(B), _p(bias), C.c_float(alpha), C.c_float(beta),
                      C.c_int(qmin), C.c_int(qmax), _p(out))
        return out

    def int8_matmul_nobias_i8(self, A, B, alpha, qmin, qmax, M, N, K, batch=False):
        A = np.ascontiguousarray(A, np.int8); B = np.ascontiguousarray(B, np.int8)
        out = np.empty((M, N), np.int8)
        name = "int8_matmul_nobias_batch_i8" if batch else "int8_matmul_nobias_i8"
        self._f(name)(C.c_int(M), C.c_int(N), C.c_int(K), _p(A), _p(B), C.c_float(alpha), C.c_int(qmin), C.c_int(qmax),
                      _p(out))
        return out

    def int8_matmul_bias_f32(self, A, B, bias, alpha, M, N, K):
        A = np.ascontiguousarray(A, np.int8); B = np.ascontiguousarray(B, np.int8); bias = np.ascontiguousarray(bias, np.float32)
        out = np.empty((M, N), np.float32)
        self._f("int8_matmul_bias_f32")(C.c_int(M), C.c_int(N), C.c_int(K), _p(A), _p(B), _p(bias), C.c_float(alpha), _p(out))
        return out

    def int8_matmul_nobias_f32(self, A, B, alpha, M, N, K, batch=False):
        A = np.ascontiguousarray(A, np.int8); B = np.ascontiguousarray(B, np.int8)
        out = np.empty((M, N), np.float32)
        name = "int8_matmul_nobias_batch_f32" if batch else "int8_matmul_nobias_f32"
        self._f(name)(C.c_int(M), C.c_int(N), C.c_int(K), _p(A), _p(B), C.c_float(alpha), _p(out))
        return out

    def naive_mat_mul_int8(self, A, B_kn, A_zp, C_zp, A_sc, B_sc, C_sc, qmin, qmax, M, N, K):
        A = np.ascontiguousarray(A, np.int8); B = np.ascontiguousarray(B_kn, np.int8)
        out = np.empty((M, N), np.int8)
        self._f("naive_mat_mul_int8")(C.c_int(M), C.c_int(N), C.c_int(K), _p(A), _p(B), C.c_int32(A_zp), C.c_int32(C_zp),
                                      C.c_float(A_sc), C.c_float(B_sc), C.c_float(C_sc), C.c_int(qmin), C.c_int(qmax), _p(out))
        return out


class Oracle(_Lib):
    """libtce_oracle.so: the restatement plus the weight-format helpers that only it has."""

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(with_ref=False)
        super().__init__(path, "orc_")
        self.lib.orc_zeros_width.restype = C.c_int
        self.lib.orc_f32_to_f16.restype = C.c_uint16
        self.lib.orc_f32_to_f16.argtypes = [C.c_float]
        self.lib.orc_f64_to_f16.restype = C.c_uint16
        self.lib.orc_f64_to_f16.argtypes = [C.c_double]
        self.lib.orc_f16_to_f32.restype = C.c_float
        self.lib.orc_f16_to_f32.argtypes = [C.c_uint16]

    def zeros_width(self, K, G):
        return int(self.lib.orc_zeros_width(C.c_int(K), C.c_int(G)))

    def group_quantize(self, w: np.ndarray, G: int):
        w = np.ascontiguousarray(w, np.float32)
        codes = np.empty(w.size, np.uint8)
        d = np.empty(w.size // G, np.float32)
        self.lib.orc_group_quantize(_p(w), C.c_int64(w.size), C.c_int(G), _p(codes), _p(d))
        return codes.reshape(w.shape), d

    def pack_q4_6(self, codes, d, N, K, G):
        zw = self.zeros_width(K, G)
        qw = np.empty((N, K // 8), np.uint32); sc = np.empty((N, zw * 8), np.uint16); zp = np.empty((N, zw), np.uint32)
        self.lib.orc_pack_q4_6(_p(np.ascontiguousarray(codes, np.uint8)), _p(np.ascontiguousarray(d, np.float32)), C.c_int(N),
                               C.c_int(K), C.c_int(G), _p(qw), _p(sc), _p(zp))
        return qw, sc.view(np.float16), zp

    def pack_q4_5(self, codes, d, N, K, G):
        qw = np.empty((K, N // 8), np.uint32); sc = np.empty((K // G, N), np.uint16); zp = np.empty((K // G, N // 8), np.uint32)
        self.lib.orc_pack_q4_5(_p(np.ascontiguousarray(codes, np.uint8)), _p(np.ascontiguousarray(d, np.float32)), C.c_int(N),
                               C.c_int(K), C.c_int(G), _p(qw), _p(sc), _p(zp))
        return qw, sc.view(np.float16), zp

    def pack_sequential(self, codes, N, K):
        out = np.empty((N, K // 2), np.uint8)
        self.lib.orc_pack_sequential(_p(np.ascontiguousarray(codes, np.uint8)), C.c_int(N), C.c_int(K), _p(out))
        return out

    def unpack_q4_6(self, qweight, N, K):
        out = np.empty((N, K), np.uint8)
        self.lib.orc_unpack_q4_6(_p(np.ascontiguousarray(qweight).view(np.uint32)), C.c_int(N), C.c_int(K), _p(out))
        return out

    def quantize_q4_6(self, w: np.ndarray, G: int = 128):
        """fp32 [N][K] -> (qweight u32 [N][K/8], scales f16 [N][zw*8], zeros u32 [N][zw], codes u8 [N][K], d f32 [N][K/G])."""
        N, K = w.shape
        codes, d = self.group_quantize(w, G)
        qw, sc, zp = self.pack_q4_6(codes, d, N, K, G)
        return qw, sc, zp, codes, d.reshape(N, K // G)

    def w4a16_gemv_q4_6(self, A_f16, qweight, scales_f16, zeros, M, N, K, G):
        A = np.ascontiguousarray(A_f16).view(np.uint16); q = np.ascontiguousarray(qweight).view(np.uint32)
        s = np.ascontiguousarray(scales_f16).view(np.uint16); z = np.ascontiguousarray(zeros).view(np.uint32)
        c32 = np.empty((M, N), np.float32); c16 = np.empty((M, N), np.uint16)
        self.lib.orc_w4a16_gemv_q4_6(C.c_int(M), C.c_int(N), C.c_int(K), C.c_int(G), _p(A), _p(q), _p(s), _p(z), _p(c32), _p(c16))
        return c32, c16.view(np.float16)

    def w4a16_gemv_q4_6_mt(self, A_f16, qweight, scales_f16, zeros, M, N, K, G, threads: int | None = None):
        """The same function over row ranges of the weight matrix on several host threads (an output column depends on
        its own weight row only, so the result is bit-identical to the single call; ctypes releases the GIL)."""
        from concurrent.futures import ThreadPoolExecutor
        threads = threads or min(64, os.cpu_count() or 1)
        A = np.ascontiguousarray(A_f16).view(np.uint16); q = np.ascontiguousarray(qweight).view(np.uint32).reshape(N, K // 8)
        s = np.ascontiguousarray(scales_f16).view(np.uint16).reshape(N, -1); z = np.ascontiguousarray(zeros).view(np.uint32).reshape(N, -1)
        step = max(16, -(-N // (threads * 4)))
        c32 = np.empty((M, N), np.float32)

        def part(n0):
            n1 = min(N, n0 + step)
            o = np.empty((M, n1 - n0), np.float32)
            self.lib.orc_w4a16_gemv_q4_6(C.c_int(M), C.c_int(n1 - n0), C.c_int(K), C.c_int(G), _p(A), _p(q[n0:n1]), _p(s[n0:n1]), _p(z[n0:n1]), _p(o), None)
            c32[:, n0:n1] = o

        with ThreadPoolExecutor(max_workers=threads) as pool:
            list(pool.map(part, range(0, N, step)))
        return c32

    def add_half(self, a_f16, b_f16):
        a = np.ascontiguousarray(a_f16, np.float16).view(np.uint16); b = np.ascontiguousarray(b_f16, np.float16).view(np.uint16)
        out = np.empty(a.shape, np.uint16)
        self.lib.orc_add_half(_p(a), _p(b), _p(out), C.c_int64(a.size))
        return out.view(np.float16)

    def silu_mul_half(self, gate_f16, up_f16):
        a = np.ascontiguousarray(gate_f16, np.float16).view(np.uint16); b = np.ascontiguousarray(up_f16, np.float16).view(np.uint16)
        out = np.empty(a.shape, np.uint16)
        self.lib.orc_silu_mul_half(_p(a), _p(b), _p(out), C.c_int64(a.size))
        return out.view(np.float16)

    def layernorm_q(self, x_f32, w_f32, b_f32):
        x = np.ascontiguousarray(x_f32, np.float32); w = np.ascontiguousarray(w_f32, np.float32); b = np.ascontiguousarray(b_f32, np.float32)
        m, n = x.reshape(-1, x.shape[-1]).shape
        out = np.empty(x.shape, np.int8)
        self.lib.orc_layernorm_q(_p(x), _p(w), _p(b), _p(out), C.c_int(m), C.c_int(n))
        return out

    def opt_softmax_q(self, scores_f32, mask_f32):
        """batch_Add + softmax + int8 probabilities of the OPT attention: scores [heads][sq][tgz], mask [sq][tgz] -> int8 [heads][sq][tgz]."""
        s = np.ascontiguousarray(scores_f32, np.float32); m = np.ascontiguousarray(mask_f32, np.float32)
        heads, sq, tgz = s.shape
        assert m.shape == (sq, tgz)
        out = np.empty(s.shape, np.int8)
        self.lib.orc_opt_softmax_q(_p(s), _p(m), _p(out), C.c_int(heads), C.c_int(sq), C.c_int(tgz))
        return out

    def rmsnorm_half(self, x_f16, gamma_f32, eps):
        x = np.ascontiguousarray(x_f16, np.float16); g = np.ascontiguousarray(gamma_f32, np.float32)
        m, n = x.reshape(-1, x.shape[-1]).shape
        out = np.empty(x.shape, np.uint16)
        self.lib.orc_rmsnorm_half(_p(x.view(np.uint16)), _p(g), _p(out), C.c_int(m), C.c_int(n), C.c_float(eps))
        return out.view(np.float16)

    # ---- attention ops (CUDA-only in the reference; pinned against its kernel sources run through oracle/cuda_emul: tests/test_oracle_glue.py) ----
    def hfma(self, a_bits: int, b_bits: int, c_bits: int) -> int:
        self.lib.orc_hfma.restype = C.c_uint16
        self.lib.orc_hfma.argtypes = [C.c_uint16] * 3
        return int(self.lib.orc_hfma(a_bits, b_bits, c_bits))

    def bmm_f16t(self, A_f16, B_f16, alpha_f16) -> np.ndarray:
        """A [batch][M][K], B [batch][N][K] -> C [batch][M][N] = hmul(alpha, sequential hfma over k) (BMM_F16T.cu:28-45)."""
        A = np.ascontiguousarray(A_f16, np.float16); B = np.ascontiguousarray(B_f16, np.float16)
        batch, M, K = A.shape
        N = B.shape[1]
        out = np.empty((batch, M, N), np.uint16)
        alpha = int(np.array([alpha_f16], np.float16).view(np.uint16)[0])
        self.lib.orc_bmm_f16t.argtypes = [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_uint16]
        self.lib.orc_bmm_f16t(batch, M, N, K, _p(A.view(np.uint16)), _p(B.view(np.uint16)), _p(out), alpha)
        return out.view(np.float16)

    def softmax_half(self, x_f16) -> np.ndarray:
        """rows of the last dimension, softmax_cuda's arithmetic (softmax.cu:4-40)."""
        x = np.ascontiguousarray(x_f16, np.float16)
        n = x.shape[-1]
        out = np.empty(x.shape, np.uint16)
        self.lib.orc_softmax_half.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.orc_softmax_half(x.size // n, n, _p(x.view(np.uint16)), _p(out))
        return out.view(np.float16)

    def rope_half(self, q_f16, k_f16, cos_f16, sin_f16, start_idx: int):
        """RotaryPosEmb_cuda_forward on copies of q, k [heads][len][hd]; cos / sin [positions][hd] (RotaryPosEmb.cu:4-34)."""
        q = np.array(q_f16, np.float16, copy=True); k = np.array(k_f16, np.float16, copy=True)
        heads, ln, hd = q.shape
        c = np.ascontiguousarray(cos_f16, np.float16); s_ = np.ascontiguousarray(sin_f16, np.float16)
        self.lib.orc_rope_half.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4
        self.lib.orc_rope_half(_p(q.view(np.uint16)), _p(k.view(np.uint16)), _p(c.view(np.uint16)), _p(s_.view(np.uint16)), heads, ln, hd, start_idx)
        return q, k

    def fp32_matmul_transposed(self, A, B, bias, M, N, K):
        A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.float32)
        bias = None if bias is None else np.ascontiguousarray(bias, np.float32)
        out = np.empty((M, N), np.float32)
        self.lib.orc_fp32_matmul_transposed(C.c_int(M), C.c_int(N), C.c_int(K), _p(A), _p(B), _p(bias), _p(out))
        return out


class Reference(_Lib):
    """oracle/_ref/libtce_ref.so: the reference's own code."""

    def __init__(self, path: str = REF_SO):
        super().__init__(path, "ref_")

    def sizeof_matmul_params(self):
        self.lib.ref_sizeof_matmul_params.restype = C.c_int
        return int(self.lib.ref_sizeof_matmul_params())

    def fp32_matmul_transposed(self, A, B, M, N, K, use_ref_backend=True):
        A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.float32)
        out = np.empty((M, N), np.float32)
        self.lib.ref_fp32_matmul_transposed(C.c_int(M), C.c_int(N), C.c_int(K), _p(A), _p(B), _p(out), C.c_int(int(use_ref_backend)))
        return out


def naive_mat_mul_int4_x86(lib, prefix, A, B_q4_3, scales, M, N, K, G=32):
    """QM_x86 branch of naive_mat_mul_int4 through either library (lib = CDLL, prefix 'orc_' or 'ref_')."""
    A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B_q4_3, np.uint8); sc = np.ascontiguousarray(scales, np.float32)
    out = np.empty((M, N), np.float32)
    getattr(lib, prefix + "naive_mat_mul_int4_x86")(C.c_int(M), C.c_int(N), C.c_int(K), C.c_int(G), _p(A), _p(B), _p(sc), _p(out))
    return out


def naive_mat_mul_int4_isa(lib, prefix, isa, A, B_bytes, scales, M, N, K, G):
    """The QM_ARM / QM_METAL branch of naive_mat_mul_int4 (isa = 'arm' | 'metal') through either library (prefix 'orc_' or 'ref_'): pure byte arithmetic, any bytes."""
    A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B_bytes, np.uint8); sc = np.ascontiguousarray(scales, np.float32)
    out = np.empty((M, N), np.float32)
    getattr(lib, f"{prefix}naive_mat_mul_int4_{isa}")(C.c_int(M), C.c_int(N), C.c_int(K), C.c_int(G), _p(A), _p(B), _p(sc), _p(out))
    return out


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def have_ref_avx() -> bool:
    if not os.path.exists(REF_AVX_SO):
        return False
    try:
        with open("/proc/cpuinfo") as f:
            flags = f.read()
        return " avx2 " in flags and " fma " in flags
    except OSError:
        return False


def _aligned(shape, dtype, align: int = 64) -> np.ndarray:
    """The reference allocates every operand with posix_memalign (llm/src/utils.cc allocate_aligned_memory) and its AVX
    kernels use aligned 256-bit loads; numpy only guarantees 16 bytes."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.empty(n + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


def _aligned_copy(a: np.ndarray, dtype) -> np.ndarray:
    out = _aligned(a.shape, dtype)
    out[...] = a
    return out


class ReferenceAVX:
    """Timed CPU baseline: the reference's W4A8 (group 32, QM_x86 layout) fast path.  Not a parity oracle."""

    def __init__(self, num_thread: int, path: str = REF_AVX_SO):
        self.lib = C.CDLL(path)
        self.lib.ref_avx_w4a8_g32.restype = C.c_int
        self.num_thread = int(num_thread)  # fixed for the life of the process (static pool in the reference)

    @staticmethod
    def pack_q4_3(codes: np.ndarray) -> np.ndarray:
        """quantize_methods.py:232-240: per 64 codes, byte e = code[e] | code[32+e] << 4."""
        N, K = codes.shape
        c = codes.reshape(N, K // 64, 2, 32)
        return np.ascontiguousarray((c[:, :, 0, :] | (c[:, :, 1, :] << 4)).astype(np.uint8).reshape(N, K // 2))

    def make_timed_call(self, A_f32, B_q4_3, scales_f32, M, N, K):
        """Returns a zero-argument callable with all (64-byte aligned) buffers pre-bound, for timing loops."""
        A = _aligned_copy(np.asarray(A_f32, np.float32), np.float32)
        B = _aligned_copy(np.asarray(B_q4_3, np.uint8), np.uint8)
        sc = _aligned_copy(np.asarray(scales_f32, np.float32), np.float32)
        off = _aligned(sc.shape, np.float32)
        off[...] = 0
        a8 = _aligned((M * K,), np.int8)
        asc = _aligned((M * K // 32,), np.float32)
        out = _aligned((M, N), np.float32)
        args = (C.c_int(M), C.c_int(N), C.c_int(K), C.c_int(self.num_thread), _p(A), _p(B), _p(sc), _p(off), _p(a8), _p(asc), _p(out))
        keep = (A, B, sc, off, a8, asc, out)
        fn = self.lib.ref_avx_w4a8_g32

        def call(_keep=keep):
            rc = fn(*args)
            if rc != 0:
                raise ValueError("K must be a multiple of 64")
            return out
        return call

    def w4a8(self, A_f32, B_q4_3, scales_f32, M, N, K):
        return self.make_timed_call(A_f32, B_q4_3, scales_f32, M, N, K)().copy()
