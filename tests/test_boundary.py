"""CPU (no GPU): the drop-in boundary itself.

* libtce_hip.so loads and exports every symbol include/tce_matmul.h declares;
* descriptor structs have the sizes the header implies; argument validation returns the documented codes before any
  HIP call (so it is testable without a GPU);
* the C++ adapter's matmul_params has the reference's exact layout (compared with the reference build's own offsets);
* the product package never imports, links or opens anything under oracle/ and has no CPU fallback.
"""
import ctypes as C
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from tinychatengine_amd import build as B
    B.build()
    B.build_adapter()
    from tinychatengine_amd import capi
    return capi


def _declared_symbols():
    src = open(os.path.join(REPO, "include", "tce_matmul.h")).read() + open(os.path.join(REPO, "include", "tce_tuning.h")).read()  # (round 6: the tuning entry points have their own header)
    return sorted(set(re.findall(r"TCE_API\s+[\w\s\*]+?\b(tce_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(built.EXPORTS) == declared, "capi.EXPORTS must list exactly what the header declares"
    out = subprocess.check_output(["nm", "-D", "--defined-only", built.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (tce_\w+)", out))
    assert set(declared) <= exported, f"missing from the .so: {set(declared) - exported}"
    lib = built.lib()
    for s in declared:
        assert hasattr(lib, s)
    assert lib.tce_version() == 113
    assert b"gfx950" in lib.tce_build_info()


def test_library_contains_gfx950_code_objects_only(built):
    blob = open(built.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"gfx1100", b"sm_80"):
        assert other not in blob


def test_descriptor_sizes(built):
    assert C.sizeof(built.W4A16Desc) == 112 and C.sizeof(built.W8A8Desc) == 120
    assert C.sizeof(built.W4A16DescV2) == 120 and C.sizeof(built.W8A8DescV2) == 136


def test_size_prefixed_descriptors_validate_their_size(built):
    """ABI 110: a descriptor smaller than the first size-prefixed layout, or one whose extra bytes are not zero, is refused before anything touches a device."""
    capi = built
    L = capi.lib()
    v2 = capi.W4A16DescV2()
    v2.struct_size = C.sizeof(capi.W4A16DescV2) - 8
    assert L.tce_w4a16_forward_v2(C.byref(v2), None) == -1 and b"struct_size" in L.tce_last_error()
    buf = (C.c_ubyte * (C.sizeof(capi.W4A16DescV2) + 16))()
    big = capi.W4A16DescV2.from_buffer(buf)
    big.struct_size = C.sizeof(capi.W4A16DescV2) + 16
    buf[C.sizeof(capi.W4A16DescV2) + 3] = 1  # a later host set a field this library does not know
    assert L.tce_w4a16_forward_v2(C.byref(big), None) == -1 and b"does not know" in L.tce_last_error()
    w = capi.W8A8DescV2()
    w.struct_size = 12
    assert L.tce_w8a8_matmul_v2(C.byref(w), None) == -1
    assert L.tce_w4a16_forward_v2(None, None) == -1
    #  an uninitialised size is refused instead of being walked (0xFFFFFFFF would read 4 GiB past the caller's struct); reserved0 must be zero
    v2 = capi.W4A16DescV2()
    v2.struct_size = 0xFFFFFFFF
    assert L.tce_w4a16_forward_v2(C.byref(v2), None) == -1 and b"cap" in L.tce_last_error()
    v2.struct_size = C.sizeof(capi.W4A16DescV2)
    v2.reserved0 = 7
    assert L.tce_w4a16_forward_v2(C.byref(v2), None) == -1 and b"reserved0" in L.tce_last_error()
    w = capi.W8A8DescV2()
    w.struct_size = 4097
    assert L.tce_w8a8_matmul_v2(C.byref(w), None) == -1 and b"cap" in L.tce_last_error()


def test_argument_validation_needs_no_gpu(built):
    capi = built
    buf = (C.c_char * 4096)()
    p = C.addressof(buf)
    p16 = (p + 15) & ~15
    d = capi.W4A16Desc(M=1, N=16, K=256, group_size=100, A=p16, qweight=p16, scales=p16, zeros=p16, C=p16)
    assert capi.w4a16_forward(d, None) == capi.TCE_ERR_UNSUPPORTED_GROUP
    assert "Unsupported group size: 100" in capi.last_error()  # the reference's message (gemv_cuda.cu:255)
    d.group_size = 128
    d.K = 200
    assert capi.w4a16_forward(d, None) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    d.K = 256
    d.A = p16 + 2
    assert capi.w4a16_forward(d, None) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    d.A = p16
    d.M = 0
    assert capi.w4a16_forward(d, None) == capi.TCE_ERR_BAD_ARG
    d.M = 1
    d.C = None
    assert capi.w4a16_forward(d, None) == capi.TCE_ERR_BAD_ARG
    assert capi.lib().tce_w4a16_forward_group(None, 1, None) == capi.TCE_ERR_BAD_ARG
    w = capi.W8A8Desc(M=1, N=1, K=1, batch=1, A=p16, B=p16, bias=None, C=p16, bias_kind=capi.TCE_BIAS_INT8, out_kind=capi.TCE_OUT_FP32)
    assert capi.w8a8_matmul(w, None) == capi.TCE_ERR_UNSUPPORTED_KIND
    w.bias_kind, w.out_kind = capi.TCE_BIAS_INT8, capi.TCE_OUT_INT8
    assert capi.w8a8_matmul(w, None) == capi.TCE_ERR_BAD_ARG  # bias_kind set, bias null
    w.bias_kind, w.q_min, w.q_max = capi.TCE_BIAS_NONE, 5, 3
    assert capi.w8a8_matmul(w, None) == capi.TCE_ERR_BAD_ARG
    assert capi.algorithmic_bytes(1, 4096, 4096, 128) == 8_732_672
    assert capi.algorithmic_bytes(1, 11008, 4096, 128) == 23_455_232
    assert capi.lib().tce_w4a16_awq_workspace_bytes(64, 256, 128) == 64 * 32 * 4 + 64 * 4 + 64 * 8 * 2
    with pytest.raises(capi.TceError):
        capi.set_gemv_config(3, 3, 3, 3)
    assert (4, 4, 1, 2) in capi.gemv_variants() and (4, 2) in capi.gemm_variants()


def test_attention_and_glue_argument_validation_needs_no_gpu(built):
    """The attention / OPT glue entry points refuse bad arguments before any HIP call (host pointers are never dereferenced on these paths)."""
    capi = built
    L = capi.lib()
    buf = (C.c_char * 8192)()
    p = (C.addressof(buf) + 15) & ~15
    vp = C.c_void_p
    assert int(L.tce_attention_prefill_workspace_bytes(32, 100, 128)) == 32 * 100 * 128 * 2 and int(L.tce_attention_prefill_workspace_bytes(32, 100, 64)) == 0
    pre = lambda **kw: L.tce_attention_prefill_f16(vp(kw.get("qkv", p)), kw.get("ld_qkv", 0), vp(p), vp(p), None, None, None, 0, 1, vp(kw.get("out", p)), kw.get("ld_out", 0), vp(kw.get("ws", p)),
                                                   kw.get("heads", 4), kw.get("kv", 2), kw.get("hd", 128), 64, kw.get("pos", 0), kw.get("m", 8), 0x2DA8, None)
    assert pre(kv=3) == capi.TCE_ERR_BAD_ARG and "do not divide" in capi.last_error()
    assert pre(hd=64) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert pre(pos=60) == capi.TCE_ERR_BAD_ARG                 # pos + m > max_keys
    assert pre(ws=None) == capi.TCE_ERR_BAD_ARG
    assert pre(qkv=p + 2) == capi.TCE_ERR_UNSUPPORTED_SHAPE    # 16-byte pieces
    assert pre(ld_qkv=100) == capi.TCE_ERR_BAD_ARG             # shorter than a row
    assert pre(out=p + 4) == capi.TCE_ERR_UNSUPPORTED_SHAPE    # 8-byte stores
    assert pre(ld_out=4 * 128 + 2) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert L.tce_attention_decode_step_gqa_f16(vp(p), vp(p), vp(p), None, None, None, vp(p), vp(p), 6, 4, 128, 64, 0, 0x2DA8, None) == capi.TCE_ERR_BAD_ARG
    assert L.tce_opt_softmax_q(vp(p), vp(p), vp(p), 2, 2, 8, 4, None) == capi.TCE_ERR_BAD_ARG            # ld_probs < tgz
    assert L.tce_opt_softmax_q(vp(p), vp(p), vp(p), 2, 2, 9000, 0, None) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert L.tce_opt_kv_append(vp(p), vp(p), vp(p), vp(p), 2, 64, 4, 62, 64, None) == capi.TCE_ERR_BAD_ARG  # pos + sq > max_keys
    # the one-launch OPT attention: heads of 64 or 128 (OPT-125M / 1.3B, OPT-6.7B), a decode path (m <= 8), 16-byte pieces
    oa = lambda hd=64, m=1, pos=0, mk=64, ld=0, q=p: L.tce_opt_attention_decode(vp(q), vp(p), vp(p), vp(p), vp(p), vp(p), vp(p), 2, hd, m, pos, mk, ld, 1.0e-3, 7.8e-3, None)
    assert oa(hd=96) == capi.TCE_ERR_UNSUPPORTED_SHAPE and "64 or 128" in capi.last_error()
    assert oa(m=9) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert oa(pos=60, m=8) == capi.TCE_ERR_BAD_ARG               # pos + m > max_keys
    assert oa(mk=72, pos=1) == capi.TCE_ERR_UNSUPPORTED_SHAPE    # max_keys % 16
    assert oa(hd=128, ld=200) == capi.TCE_ERR_UNSUPPORTED_SHAPE  # ld < heads * 128
    assert oa(q=p + 8) == capi.TCE_ERR_UNSUPPORTED_SHAPE         # 16-byte aligned pointers
    assert L.tce_opt_attention_decode(None, vp(p), vp(p), vp(p), vp(p), vp(p), vp(p), 2, 64, 1, 0, 64, 0, 1.0e-3, 7.8e-3, None) == capi.TCE_ERR_BAD_ARG
    # tce_layernorm_q: n % 4, n <= 8192; the LayerNormQ + linears group: a decode path, k % 16
    assert L.tce_layernorm_q(vp(p), vp(p), vp(p), vp(p), 2, 6, None) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert L.tce_layernorm_q(vp(p), vp(p), vp(p), vp(p), 2, 8196, None) == capi.TCE_ERR_UNSUPPORTED_SHAPE
    assert L.tce_layernorm_q(vp(p), vp(p), vp(p), vp(p), 0, 64, None) == capi.TCE_ERR_BAD_ARG


def test_adapter_exports_the_reference_member_functions(built):
    from tinychatengine_amd import build as B
    out = subprocess.check_output(["nm", "-D", "--defined-only", "-C", B.ADAPTER_LIB_PATH], text=True)
    for m in ["gemv_forward_cuda(matmul_params const*)", "naive_mat_mul_fp16_int4(matmul_params const*)",
              "gemm_forward_cuda(matmul_params const*, int)", "mat_mul_accelerator_int8_fast_2x2_32unroll(matmul_params const*)",
              "mat_mul_accelerator_int8_fast_32unroll_over_column(matmul_params const*)",
              "mat_mul_accelerator_int8_fast_2x2_32unroll_nobias(matmul_params const*)",
              "mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_batch(matmul_params const*)",
              "mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32(matmul_params const*)",
              "mat_mul_accelerator_int8_fast_2x2_32unroll_nobias_ofp32_batch(matmul_params const*)",
              "mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32(matmul_params const*)",
              "mat_mul_accelerator_int8_fast_2x2_32unroll_bfp32_ofp32_over_column(matmul_params const*)",
              "mat_mul_accelerator_int4_fast(matmul_params const*)", "mat_mul_accelerator_int4_fast_no_offset(matmul_params const*)"]:
        assert f"matmul::MatmulOperator::{m}" in out, m


def test_adapter_descriptor_layout_is_the_references(built):
    """Same names, same offsets: the adapter's matmul_params is binary-compatible with kernels/matmul.h:52-92."""
    from tinychatengine_amd import build as B
    a = C.CDLL(B.ADAPTER_LIB_PATH)
    a.tce_adapter_layout.restype = C.c_long
    mine = [a.tce_adapter_layout(i) for i in range(18)]
    assert mine[0] == 416 and mine[1] == 80
    expected = [416, 80, 80, 160, 240, 320, 328, 332, 368, 376, 384, 392, 16, 32, 40, 64, 76, 400]  # from the reference build
    assert mine == expected
    from oracle import oracle as O
    if O.have_ref():
        r = C.CDLL(O.REF_SO)
        if hasattr(r, "ref_layout"):
            r.ref_layout.restype = C.c_long
            assert [r.ref_layout(i) for i in range(18)] == mine


def test_product_never_touches_the_oracle_and_has_no_cpu_fallback():
    pkg = os.path.join(REPO, "tinychatengine_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cc")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f"{f} imports oracle"
                assert "libtce_oracle" not in text and "libtce_ref" not in text, f"{f} opens an oracle library"
    bench = open(os.path.join(REPO, "bench.py")).read()
    oracle_imports = [m.start() for m in re.finditer(r"from oracle import|import oracle", bench)]
    assert oracle_imports, "bench.py's cpu_baseline leg uses the oracle"
    leg = bench.index("def cpu_baseline_worker")
    nxt = bench.index("\ndef main", leg)
    assert all(leg < pos < nxt for pos in oracle_imports), "oracle may only be imported inside the cpu_baseline functions"


def test_missing_library_fails_loudly(built, monkeypatch):
    capi = built
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libtce_hip.so")
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        capi.lib()


def test_the_product_library_holds_no_diagnostic_instantiation(built):
    """Round 6: libtce_hip.so is built without -DTCE_LAB -- the kernels with parts of a loop switched off (outputs meaningless), the stream-only /
    timestamp / arithmetic-only forms of the decode kernels and the token kernel's stamped form exist in libtce_hip_lab.so only (python -m tinychatengine_amd.build
    --lab).  The build writes the product's kernel list next to the library; held here: the list is what the objects hold, none of its entries is a diagnostic
    instantiation, and the modes that select one are refused by the product library."""
    import re
    from tinychatengine_amd import build as B
    path = B.LIB_PATH + ".kernels.txt"
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(B.LIB_PATH):
        objs = [os.path.join(B.LIB_DIR, s.replace(".hip", ".o")) for s in B.HIP_SOURCES]
        B.write_kernel_list(objs, B.LIB_PATH)
    lines = [ln.strip() for ln in open(path) if ln.strip() and not ln.startswith("#")]
    assert len(lines) > 300
    bad = []
    for ln in lines:
        name = ln.split(": ", 1)[1]
        m = re.match(r"void w4a16_gemm_pk\w*_kernel<(.*?)>\(", name)
        if m:  # <KS, LG, ABL, NS> or <LG, ABL>: ABL = loop parts switched off
            args = [a.strip() for a in m.group(1).split(",")]
            abl = args[2] if name.startswith("void w4a16_gemm_pk_kernel<") else args[1]
            if abl != "0":
                bad.append(ln)
        m = re.match(r"void w4a16_gemv_kernel<(.*?)>\(", name)
        if m and m.group(1).split(",")[6].strip() in ("1", "2", "4"):  # MODE: 1 stream only, 2 timestamps, 4 arithmetic only
            bad.append(ln)
        m = re.match(r"void w4a16_gemv_stream_kernel<(.*?)>\(", name)
        if m and m.group(1).split(",")[3].strip() != "0":
            bad.append(ln)
        if name.startswith("void w4a16_gemv_i8_token_kernel<true>"):
            bad.append(ln)
    assert not bad, "diagnostic instantiations in the product library:\n" + "\n".join(bad[:10])
    L = built.lib()
    for mode in (601, 655, 2607, 26001, 1, 2, 4, 7702):
        assert L.tce_w4a16_set_debug_mode(mode) == -1 and b"lab build" in L.tce_last_error(), mode
    assert L.tce_w4a16_set_debug_mode(600) == 0 and L.tce_w4a16_set_debug_mode(0) == 0
