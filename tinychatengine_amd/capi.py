"""ctypes binding of libtce_hip.so (include/tce_matmul.h).  Plain pointers and sizes; torch is only used by callers
to own device memory and streams.  Loading fails loudly when the HIP library has not been built: there is no fallback."""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TCE_LIB_PATH") or os.path.join(_PKG, "lib", "libtce_hip.so")  # (TCE_LIB_PATH: another build of the same library, for A/B runs)

TCE_OK = 0
TCE_ERR_BAD_ARG = -1
TCE_ERR_UNSUPPORTED_GROUP = -2
TCE_ERR_UNSUPPORTED_SHAPE = -3
TCE_ERR_HIP = -4
TCE_ERR_UNSUPPORTED_KIND = -5
TCE_W4A16_GEMV_MAX_M = 8
TCE_MAX_GROUP = 4
TCE_W4_FORCE_GEMV = 1
TCE_W4_FORCE_GEMM = 2
TCE_W4_SILU_MUL_PAIRS = 8
TCE_W4_ADD_TO_C = 16
TCE_PLAN_CHAINED = 1
TCE_PLAN_TAGGED = 2
TCE_ABI_VERSION = 113  # include/tce_matmul.h TCE_VERSION: the struct layouts mirrored below
TCE_PLAN_OVERLAPPED = 4
TCE_PLAN_TUNED = 8
TCE_PLAN_INDEPENDENT = 16
TCE_W4_ZERO_POINT_IS_8 = 4
TCE_BIAS_NONE, TCE_BIAS_INT8, TCE_BIAS_FP32 = 0, 1, 2
TCE_OUT_INT8, TCE_OUT_FP32 = 0, 1

# every symbol include/tce_matmul.h declares (tests/test_boundary.py checks the .so exports exactly these)
EXPORTS = [
    "tce_w4a16_forward", "tce_w4a16_residual_rmsnorm_workspace_bytes", "tce_w4a16_forward_residual_rmsnorm", "tce_w4a16_prepack_bytes", "tce_w4a16_prepack", "tce_w4a16_gemm_scratch_bytes", "tce_w4a16_gemm_scratch_faults", "tce_w4a16_describe_dispatch", "tce_w8a8_describe_dispatch", "tce_reset_last_error", "tce_bmm_f16t", "tce_rope_half", "tce_softmax_half", "tce_attention_decode_f16", "tce_attention_decode_workspace_bytes", "tce_attention_decode_describe", "tce_attention_decode_step_f16", "tce_attention_decode_step_gqa_f16", "tce_attention_decode_describe_gqa", "tce_attention_decode_step_pos_f16", "tce_attention_prefill_f16", "tce_attention_prefill_workspace_bytes", "tce_opt_attention_decode", "tce_prefetch", "tce_add_half", "tce_silu_mul_half", "tce_rmsnorm_half", "tce_w4a16_forward_group_rmsnorm", "tce_w4a16_forward_group", "tce_w4a16_forward_independent", "tce_w4a16_describe_independent", "tce_w4a16_forward_independent_gather", "tce_w4a16_check_zero_point_8", "tce_w4a16_awq_fp16acc", "tce_w4a16_awq_workspace_bytes",
    "tce_w4a16_gemm_awq", "tce_w4a16_shard", "tce_comm_create", "tce_comm_export", "tce_comm_connect", "tce_comm_connect_local", "tce_allgather_f16", "tce_comm_rccl_unique_id", "tce_comm_rccl_init", "tce_allgather_rows_workspace_bytes", "tce_allgather_rows_f16", "tce_comm_status", "tce_comm_set_timeout_ms", "tce_comm_reset", "tce_comm_device", "tce_comm_destroy", "tce_w8a8_matmul", "tce_opt_softmax_q", "tce_opt_kv_append", "tce_layernorm_q", "tce_layernorm_q_w8a8_group", "tce_plan_create", "tce_plan_create_ex", "tce_plan_is_chained", "tce_plan_geometry", "tce_plan_status", "tce_plan_launch_geometry", "tce_plan_launch", "tce_plan_n_launches",
    "tce_w4a16_forward_v2", "tce_w8a8_matmul_v2", "tce_w8a8_scratch_bytes", "tce_attention_decode_step_deferred_f16", "tce_w4a16_forward_deferred_attention", "tce_plan_destroy", "tce_version", "tce_last_error", "tce_build_info", "tce_w4a16_set_gemv_config", "tce_w4a16_set_gemv_i8",
    "tce_w4a16_set_gemm_config", "tce_w4a16_algorithmic_bytes", "tce_w4a16_gemv_variant", "tce_w4a16_gemm_variant",
    "tce_w4a16_set_debug_mode", "tce_attention_set_tuning", "tce_w8a8_set_tuning", "tce_w4a16_set_debug_buffer", "tce_w4a16_check_zero_point_8_async", "tce_host_alloc", "tce_host_free", "tce_malloc", "tce_free", "tce_memcpy", "tce_synchronize", "tce_device_count",
]


class W4A16Desc(C.Structure):
    """struct tce_w4a16_desc"""
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("group_size", C.c_int32),
        ("A", C.c_void_p), ("qweight", C.c_void_p), ("scales", C.c_void_p), ("zeros", C.c_void_p), ("C", C.c_void_p),
        ("lda", C.c_int32), ("ldc", C.c_int32), ("scales_stride", C.c_int32), ("zeros_stride", C.c_int32),
        ("flags", C.c_int32), ("reserved", C.c_int32),
        ("rmsnorm_gamma", C.c_void_p), ("rmsnorm_eps", C.c_float), ("reserved2", C.c_int32),
        ("prepacked", C.c_void_p), ("scratch", C.c_void_p),
    ]


class W8A8Desc(C.Structure):
    """struct tce_w8a8_desc"""
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("batch", C.c_int32),
        ("A", C.c_void_p), ("B", C.c_void_p), ("bias", C.c_void_p), ("C", C.c_void_p),
        ("strideA", C.c_int64), ("strideB", C.c_int64), ("strideC", C.c_int64),
        ("alpha", C.c_float), ("beta", C.c_float), ("q_min", C.c_int32), ("q_max", C.c_int32),
        ("bias_kind", C.c_int32), ("out_kind", C.c_int32), ("b_per_row", C.c_int32), ("accumulate", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("reserved2", C.c_int32),
    ]


class AttnDeferred(C.Structure):
    """struct tce_attention_deferred: what a deferred attention step leaves for the linear that consumes it"""
    _fields_ = [("slots", C.c_int32), ("chunk", C.c_int32), ("heads", C.c_int32), ("stride", C.c_int32), ("part", C.c_void_p)]


class W4A16DescV2(C.Structure):
    """struct tce_w4a16_desc_v2: the size-prefixed form (ABI 110)"""
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("desc", W4A16Desc)]


class W8A8DescV2(C.Structure):
    """struct tce_w8a8_desc_v2"""
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("desc", W8A8Desc), ("scratch", C.c_void_p)]


def w4a16_forward_v2(desc: W4A16Desc, stream: int | None, extra_zero_bytes: int = 0) -> int:
    """tce_w4a16_forward_v2; extra_zero_bytes: pretend to be a LATER host whose descriptor has that many more (zeroed) bytes."""
    buf = (C.c_ubyte * (C.sizeof(W4A16DescV2) + extra_zero_bytes))()
    v2 = W4A16DescV2.from_buffer(buf)
    v2.struct_size = C.sizeof(W4A16DescV2) + extra_zero_bytes
    v2.desc = desc
    return lib().tce_w4a16_forward_v2(C.byref(v2), C.c_void_p(stream or 0))


_w8a8_scratch: dict = {}


def w8a8_scratch(device) -> "torch.Tensor":
    """tce_w8a8_desc_v2.scratch for `device`: one zeroed area shared by the harness's calls (ordered by its one stream)."""
    import torch
    key = str(device)
    if key not in _w8a8_scratch:
        L = lib()
        L.tce_w8a8_scratch_bytes.restype = C.c_size_t
        _w8a8_scratch[key] = torch.zeros(int(L.tce_w8a8_scratch_bytes()), dtype=torch.uint8, device=device)
    return _w8a8_scratch[key]


def w8a8_matmul_v2(desc: W8A8Desc, stream: int | None, scratch_ptr: int | None = None) -> int:
    v2 = W8A8DescV2(struct_size=C.sizeof(W8A8DescV2), reserved0=0, desc=desc, scratch=scratch_ptr)
    return lib().tce_w8a8_matmul_v2(C.byref(v2), C.c_void_p(stream or 0))


class TceError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libtce_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """The loaded library.  Raises if it has not been built (python -m tinychatengine_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m tinychatengine_amd.build` "
                "(hipcc, gfx950).  tinychatengine_amd has no CPU or PyTorch fallback."
            )
        try:
            # FIRST: the PyTorch wheel ships its own HIP runtime.  Loaded before it, this library would bring the system's copy into the process and the two
            # runtimes do not share a device (hipStreamCreate: "no ROCm-capable device" -- seen with build() and smoke() called in one process).
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.tce_version.restype = C.c_int
        # the descriptors are plain structs without a size field: a library built from another header revision would read the mirrored
        # structs below at the wrong offsets -- refuse it here, loudly, instead
        if int(L.tce_version()) != TCE_ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {int(L.tce_version())}, this wrapper mirrors {TCE_ABI_VERSION} (include/tce_matmul.h: TCE_VERSION): rebuild with `python -m tinychatengine_amd.build`")
        L.tce_last_error.restype = C.c_char_p
        L.tce_build_info.restype = C.c_char_p
        L.tce_w4a16_forward.argtypes = [C.POINTER(W4A16Desc), C.c_void_p]
        L.tce_w4a16_forward_group.argtypes = [C.POINTER(W4A16Desc), C.c_int, C.c_void_p]
        L.tce_w4a16_forward_independent.argtypes = [C.POINTER(W4A16Desc), C.c_int, C.POINTER(C.c_int), C.c_void_p]
        L.tce_w4a16_describe_independent.argtypes = [C.POINTER(W4A16Desc), C.c_int, C.c_char_p, C.c_int]
        L.tce_w4a16_forward_independent_gather.argtypes = [C.POINTER(W4A16Desc), C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        L.tce_w4a16_prepack_bytes.argtypes = [C.c_int] * 3
        L.tce_w4a16_prepack_bytes.restype = C.c_size_t
        L.tce_w4a16_gemm_scratch_bytes.argtypes = []
        L.tce_w4a16_gemm_scratch_bytes.restype = C.c_size_t
        L.tce_w4a16_prepack.argtypes = [C.POINTER(W4A16Desc), C.c_void_p, C.c_void_p]
        L.tce_prefetch.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
        L.tce_bmm_f16t.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ushort, C.c_void_p]
        L.tce_rope_half.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
        L.tce_softmax_half.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
        L.tce_attention_decode_f16.argtypes = [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_ushort, C.c_void_p]
        L.tce_attention_decode_workspace_bytes.argtypes = [C.c_int] * 3
        L.tce_attention_decode_workspace_bytes.restype = C.c_size_t
        L.tce_attention_decode_describe.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.tce_attention_decode_step_f16.argtypes = [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_ushort, C.c_void_p]
        L.tce_attention_decode_step_gqa_f16.argtypes = [C.c_void_p] * 8 + [C.c_int] * 5 + [C.c_ushort, C.c_void_p]
        L.tce_attention_decode_step_pos_f16.argtypes = [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_ushort, C.c_void_p]
        L.tce_attention_decode_step_deferred_f16.argtypes = [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_ushort, C.POINTER(AttnDeferred), C.c_void_p]
        L.tce_w4a16_forward_deferred_attention.argtypes = [C.POINTER(W4A16Desc), C.POINTER(AttnDeferred), C.c_void_p, C.c_int, C.c_void_p]
        L.tce_opt_attention_decode.argtypes = [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_float, C.c_float, C.c_void_p]
        L.tce_attention_prefill_workspace_bytes.restype = C.c_size_t
        L.tce_attention_prefill_workspace_bytes.argtypes = [C.c_int] * 3
        L.tce_attention_prefill_f16.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + \
            [C.c_int] * 6 + [C.c_ushort, C.c_void_p]
        L.tce_attention_decode_describe_gqa.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.tce_add_half.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
        L.tce_silu_mul_half.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
        L.tce_rmsnorm_half.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.tce_w4a16_forward_group_rmsnorm.argtypes = [C.POINTER(W4A16Desc), C.c_int, C.c_void_p, C.c_float, C.c_void_p]
        L.tce_w4a16_check_zero_point_8.argtypes = [C.c_void_p, C.c_longlong]
        L.tce_w4a16_awq_fp16acc.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5
        L.tce_w4a16_awq_workspace_bytes.argtypes = [C.c_int] * 3
        L.tce_w4a16_awq_workspace_bytes.restype = C.c_size_t
        L.tce_w4a16_gemm_awq.argtypes = [C.c_int] * 4 + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        L.tce_w8a8_matmul.argtypes = [C.POINTER(W8A8Desc), C.c_void_p]
        L.tce_opt_softmax_q.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.tce_opt_kv_append.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
        L.tce_w4a16_shard.argtypes = [C.POINTER(W4A16Desc), C.c_int, C.c_int, C.POINTER(W4A16Desc)]
        L.tce_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.tce_comm_export.argtypes = [C.c_void_p, C.c_void_p]
        L.tce_comm_connect.argtypes = [C.c_void_p, C.c_void_p]
        L.tce_comm_connect_local.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.tce_allgather_f16.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.tce_comm_status.argtypes = [C.c_void_p]
        L.tce_comm_set_timeout_ms.argtypes = [C.c_void_p, C.c_int]
        L.tce_comm_reset.argtypes = [C.c_void_p]
        L.tce_comm_device.argtypes = [C.c_void_p]
        L.tce_comm_destroy.argtypes = [C.c_void_p]
        L.tce_comm_destroy.restype = None
        L.tce_layernorm_q.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.tce_layernorm_q_w8a8_group.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(W8A8Desc), C.c_int, C.c_void_p, C.c_void_p]
        L.tce_plan_create.argtypes = [C.POINTER(W4A16Desc), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_void_p)]
        L.tce_plan_create_ex.argtypes = [C.POINTER(W4A16Desc), C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.tce_plan_is_chained.argtypes = [C.c_void_p]
        L.tce_plan_status.argtypes = [C.c_void_p]
        L.tce_plan_geometry.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        L.tce_plan_launch.argtypes = [C.c_void_p, C.c_void_p]
        L.tce_plan_n_launches.argtypes = [C.c_void_p]
        L.tce_plan_destroy.argtypes = [C.c_void_p]
        L.tce_plan_destroy.restype = None
        L.tce_w4a16_set_gemv_config.argtypes = [C.c_int] * 4
        L.tce_w4a16_set_gemm_config.argtypes = [C.c_int] * 2
        L.tce_w4a16_set_gemv_i8.argtypes = [C.c_int] * 2
        L.tce_w4a16_residual_rmsnorm_workspace_bytes.restype = C.c_size_t
        L.tce_w4a16_forward_residual_rmsnorm.argtypes = [C.POINTER(W4A16Desc), C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tce_comm_rccl_unique_id.argtypes = [C.c_void_p]
        L.tce_comm_rccl_init.argtypes = [C.c_void_p, C.c_void_p]
        L.tce_allgather_rows_workspace_bytes.argtypes = [C.c_int, C.c_int]
        L.tce_allgather_rows_workspace_bytes.restype = C.c_size_t
        L.tce_allgather_rows_f16.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.tce_w4a16_algorithmic_bytes.argtypes = [C.c_int] * 4
        L.tce_w4a16_algorithmic_bytes.restype = C.c_int64
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != TCE_OK:
        raise TceError(rc, lib().tce_last_error().decode())


def last_error() -> str:
    return lib().tce_last_error().decode()


def describe_attention_step(heads: int, keys: int, kv_heads: int | None = None) -> dict:
    """How tce_attention_decode_step_f16 / _gqa_f16 cuts a context of `keys` keys: {"chunks", "keys-per-chunk", "waves", "workgroups", "combine"} (no GPU needed)."""
    buf = C.create_string_buffer(160)
    check(lib().tce_attention_decode_describe_gqa(heads, heads if kv_heads is None else kv_heads, keys, buf, len(buf)))
    out = {}
    for tok in buf.value.decode().split():
        k, v = tok.split("=")
        out[k] = int(v) if v.isdigit() else v
    return out


def describe_dispatch(desc: W4A16Desc) -> str:
    buf = C.create_string_buffer(128)
    check(lib().tce_w4a16_describe_dispatch(C.byref(desc), buf, 128))
    return buf.value.decode()


def w4a16_forward(desc: W4A16Desc, stream: int | None) -> int:
    return lib().tce_w4a16_forward(C.byref(desc), C.c_void_p(stream or 0))


def w4a16_forward_residual_rmsnorm(desc: W4A16Desc, gamma_ptr: int, eps: float, xn_ptr: int, workspace_ptr: int, stream: int | None) -> int:
    """o_proj / down_proj + residual add + the RMSNorm that follows, one launch (tce_w4a16_forward_residual_rmsnorm)."""
    return lib().tce_w4a16_forward_residual_rmsnorm(C.byref(desc), C.c_void_p(gamma_ptr), float(eps), C.c_void_p(xn_ptr), C.c_void_p(workspace_ptr), C.c_void_p(stream or 0))


def w4a16_forward_group(descs: list[W4A16Desc], stream: int | None) -> int:
    arr = (W4A16Desc * len(descs))(*descs)
    return lib().tce_w4a16_forward_group(arr, len(descs), C.c_void_p(stream or 0))


TCE_MAX_INDEPENDENT = 8


def w4a16_forward_independent(descs: list[W4A16Desc], stream: int | None) -> int:
    """tce_w4a16_forward_independent: linears that share nothing, one launch where the library has one.  Returns the number of kernel launches made (raises on error)."""
    arr = (W4A16Desc * len(descs))(*descs)
    n = C.c_int(0)
    check(lib().tce_w4a16_forward_independent(arr, len(descs), C.byref(n), C.c_void_p(stream or 0)))
    return n.value


def describe_w8a8_dispatch(desc: W8A8Desc, with_scratch: bool = False) -> str:
    """tce_w8a8_describe_dispatch: the kernel form tce_w8a8_matmul would run for this descriptor (no GPU needed)."""
    buf = C.create_string_buffer(128)
    L = lib()
    L.tce_w8a8_describe_dispatch.argtypes = [C.POINTER(W8A8Desc), C.c_int, C.c_char_p, C.c_int]
    check(L.tce_w8a8_describe_dispatch(C.byref(desc), 1 if with_scratch else 0, buf, 128))
    return buf.value.decode()


def describe_independent(descs: list[W4A16Desc]) -> str:
    """tce_w4a16_describe_independent: how the library would run these linears through tce_w4a16_forward_independent (no GPU needed)."""
    arr = (W4A16Desc * len(descs))(*descs)
    buf = C.create_string_buffer(128)
    check(lib().tce_w4a16_describe_independent(arr, len(descs), buf, len(buf)))
    return buf.value.decode()


def w8a8_matmul(desc: W8A8Desc, stream: int | None) -> int:
    return lib().tce_w8a8_matmul(C.byref(desc), C.c_void_p(stream or 0))


def algorithmic_bytes(M: int, N: int, K: int, G: int) -> int:
    return int(lib().tce_w4a16_algorithmic_bytes(M, N, K, G))


def set_gemv_config(rows: int = 0, waves_n: int = 0, waves_k: int = 0, depth: int = 0) -> None:
    check(lib().tce_w4a16_set_gemv_config(rows, waves_n, waves_k, depth))


def set_gemv_i8(mode: int = 0, tiles_per_wave: int = 0) -> None:
    """The decode kernel on pre-packed copies: mode 0 automatic, 1 off; tiles_per_wave 0 the rule, 1, 2."""
    check(lib().tce_w4a16_set_gemv_i8(mode, tiles_per_wave))


def attention_set_tuning(waves_per_workgroup: int = 0, workgroups: int = 0, heads_per_workgroup: int = 0) -> None:
    """The fast decode attention step's cut, by name (0 = the fitted rule): waves per workgroup 4 / 8 / 16, workgroups 32..8192, query heads per workgroup 1 / 2 / 4."""
    check(lib().tce_attention_set_tuning(waves_per_workgroup, workgroups, heads_per_workgroup))


def w8a8_set_tuning(quartets_per_tile: int = 0, big_tiles: int = 0, deep_pipeline: int = 0) -> None:
    """tce_w8a8_matmul's kernel choice, by name (0 = the rules): quartets per 64x64 tile 1 / 2 / 4; the 128-row tiles 1..4 forced, 9 off; the deep-pipeline 64x64 tile 1 / 2 / 4 forced, 9 off."""
    check(lib().tce_w8a8_set_tuning(quartets_per_tile, big_tiles, deep_pipeline))


def set_gemm_config(m_tiles: int = 0, n_tiles: int = 0) -> None:
    check(lib().tce_w4a16_set_gemm_config(m_tiles, n_tiles))


def gemv_variants() -> list[tuple[int, int, int, int]]:
    """(rows_per_wave, waves_n, waves_k, depth) of every compiled GEMV kernel variant."""
    out, i = [], 0
    r, n, k, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    while lib().tce_w4a16_gemv_variant(i, C.byref(r), C.byref(n), C.byref(k), C.byref(d)) == TCE_OK:
        out.append((r.value, n.value, k.value, d.value))
        i += 1
    return out


def gemm_variants() -> list[tuple[int, int]]:
    out, i = [], 0
    m, n = C.c_int(), C.c_int()
    while lib().tce_w4a16_gemm_variant(i, C.byref(m), C.byref(n)) == TCE_OK:
        out.append((m.value, n.value))
        i += 1
    return out


class Plan:
    """tce_plan: a fixed sequence of W4A16 launches captured into one hipGraph (one decode token's linears)."""

    def __init__(self, launches: list[list[W4A16Desc]], chained: bool = False, tagged: bool = False, overlapped: bool = False, tuned: bool = False, independent: bool = False):
        """tagged (chained: accepted synonym): TCE_PLAN_TAGGED -- one persistent kernel walks the list, the plan's data flow
        ordered by polling tagged output words (include/tce_matmul.h).  self.tagged tells whether that form was built.
        tuned: TCE_PLAN_TUNED -- the decode launches' geometries are timed on this device at creation (stream-ordered plans).
        independent: TCE_PLAN_INDEPENDENT -- every group is a set of linears that share nothing (up to TCE_MAX_INDEPENDENT; tce_w4a16_forward_independent)."""
        flat = [d for g in launches for d in g]
        self._descs = (W4A16Desc * len(flat))(*flat)
        self._groups = (C.c_int32 * len(launches))(*[len(g) for g in launches])
        self._h = C.c_void_p()
        check(lib().tce_plan_create_ex(self._descs, self._groups, len(launches), (TCE_PLAN_TAGGED if tagged else 0) | (TCE_PLAN_CHAINED if chained else 0) | (TCE_PLAN_OVERLAPPED if overlapped else 0) | (TCE_PLAN_TUNED if tuned else 0) | (TCE_PLAN_INDEPENDENT if independent else 0), C.byref(self._h)))
        self.n_launches = len(launches)
        self.kind = int(lib().tce_plan_is_chained(self._h))  # 0 stream-ordered, 2 token kernel (fp16 body), 3 overlapped launches, 4 token kernel on the int8-contraction body (round 6)
        self.chained = self.kind != 0
        self.tagged = self.kind in (2, 4)
        self.overlapped = self.kind == 3

    def launch(self, stream: int | None) -> None:
        check(lib().tce_plan_launch(self._h, C.c_void_p(stream or 0)))

    def geometry(self) -> dict:
        """Chained plans: what the token kernel runs with."""
        v = [C.c_int() for _ in range(4)]
        check(lib().tce_plan_geometry(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("rows", "depth", "waves", "workgroups"), (x.value for x in v)))

    def launch_geometries(self) -> list[tuple[int, int, int, int]]:
        """Per launch: the (rows, waves_n, waves_k, depth) a TCE_PLAN_TUNED plan chose, (0, 0, 0, 0) where the dispatcher's rule stayed."""
        out = []
        for i in range(self.n_launches):
            v = [C.c_int() for _ in range(4)]
            check(lib().tce_plan_launch_geometry(self._h, i, *[C.byref(x) for x in v]))
            out.append(tuple(x.value for x in v))
        return out

    def status(self) -> None:
        """Synchronise and raise if a chained launch ever gave up waiting for its predecessor."""
        check(lib().tce_plan_status(self._h))

    def close(self) -> None:
        if self._h:
            lib().tce_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """tce_comm: this rank's peer-write window + the mapped windows of the other ranks (include/tce_matmul.h, csrc/comm.hip)."""

    def __init__(self, rank: int, world: int, max_vector_elems: int, slots: int = 4):
        self.rank, self.world = rank, world
        h = C.c_void_p()
        check(lib().tce_comm_create(rank, world, max_vector_elems, slots, C.byref(h)))
        self.handle = h

    def export(self) -> bytes:
        buf = C.create_string_buffer(64)
        check(lib().tce_comm_export(self.handle, buf))
        return buf.raw

    def connect(self, handles: list[bytes]) -> None:
        table = C.create_string_buffer(b"".join(handles), 64 * self.world)
        check(lib().tce_comm_connect(self.handle, table))

    @staticmethod
    def connect_local(comms: list["Comm"]) -> None:
        arr = (C.c_void_p * len(comms))(*[c.handle for c in comms])
        for c in comms:
            check(lib().tce_comm_connect_local(c.handle, arr))

    def allgather(self, slot: int, src_ptr: int, dst_ptr: int, n_total: int, stream: int | None) -> None:
        check(lib().tce_allgather_f16(self.handle, slot, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), n_total, C.c_void_p(stream or 0)))

    def forward_independent_gather(self, descs: list, gathered: int, slot: int, dst_ptr: int, stream: int | None) -> int:
        """tce_w4a16_forward_independent_gather: the linears of `descs` as one launch, linear `gathered`'s slice exchanged with the other ranks inside it (the complete
        vector lands at dst_ptr).  Returns the number of kernel launches made."""
        arr = (W4A16Desc * len(descs))(*descs)
        n = C.c_int(0)
        check(lib().tce_w4a16_forward_independent_gather(arr, len(descs), gathered, self.handle, slot, C.c_void_p(dst_ptr), C.byref(n), C.c_void_p(stream or 0)))
        return n.value

    @staticmethod
    def rccl_unique_id() -> bytes:
        """Rank 0: the opaque blob every rank passes to rccl_init (exchanged by the host like the IPC handles)."""
        buf = C.create_string_buffer(128)
        check(lib().tce_comm_rccl_unique_id(buf))
        return buf.raw

    def rccl_init(self, unique_id: bytes) -> None:
        """Collective: the RCCL side of this communicator (exchanges beyond the peer-write kernel's 64 KiB-per-rank regime)."""
        check(lib().tce_comm_rccl_init(self.handle, C.create_string_buffer(unique_id, 128)))

    def allgather_rows(self, slot: int, src_ptr: int, dst_ptr: int, m: int, n_total: int, workspace_ptr: int | None, stream: int | None, ldd: int = 0) -> None:
        check(lib().tce_allgather_rows_f16(self.handle, slot, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), m, n_total, ldd, C.c_void_p(workspace_ptr or 0), C.c_void_p(stream or 0)))

    def status(self) -> int:
        return int(lib().tce_comm_status(self.handle))

    def set_timeout_ms(self, ms: int) -> None:
        check(lib().tce_comm_set_timeout_ms(self.handle, int(ms)))

    def reset(self) -> None:
        """Re-arm after a timed-out wait (every rank, with no exchange in flight anywhere)."""
        check(lib().tce_comm_reset(self.handle))

    @property
    def device(self) -> int:
        return int(lib().tce_comm_device(self.handle))

    def close(self) -> None:
        if self.handle:
            lib().tce_comm_destroy(self.handle)
            self.handle = None
