"""The drop-in, proven with the reference's OWN level-2 callers.

oracle/Makefile target `l2link` compiles, UNMODIFIED and where they lie under /root/reference,
    llm/src/ops/cuda/linear.cu, llm/src/ops/W8A8B8O8Linear{,ReLU}.cc, W8A8BFP32OFP32Linear.cc, BMM_S8T_S8N_{F32T,S8T}.cc, llm/src/utils.cc
(host C++, their own -DQM_CUDA flavour, <cuda*.h> names from oracle/cuda_shim/), builds the HIP adapter against the reference's own
kernels/matmul.h (-DTCE_ADAPTER_USE_REFERENCE_HEADER) and links both with oracle/l2_harness.cc into oracle/_ref/l2_harness.

CPU part (here): the link succeeds, every matmul::MatmulOperator member those objects reference is defined by the adapter, the
binary loads.  GPU part (-m gpu; the prebuilt binary travels to the GPU box like the other oracle/_ref files): the reference's
Linear_half_int4::forward / W8A8B8O8Linear::forward / ... run from those objects on files written by quantize.py / numpy in the
reference's on-disk formats, and their outputs are compared with the oracle (W4A16: north-star tolerance; int8: bit-exact).
"""
import os
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(REPO, "oracle", "_ref", "l2_harness")
REF = "/root/reference"
ALPHA, BETA = 0.0005035400390625, 0.02130126953125  # llm/tests/non_cuda/test_ops.cc:179

L2_OBJECTS = ["llm/src/ops/cuda/linear.cu", "llm/src/ops/W8A8B8O8Linear.cc", "llm/src/ops/W8A8B8O8LinearReLU.cc", "llm/src/ops/W8A8BFP32OFP32Linear.cc",
              "llm/src/ops/BMM_S8T_S8N_F32T.cc", "llm/src/ops/BMM_S8T_S8N_S8T.cc"]


def _build():
    if os.path.isdir(os.path.join(REF, "kernels")):
        from tinychatengine_amd import build as B
        B.build()
        subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "l2link"])
    if not os.path.exists(HARNESS):
        pytest.skip("oracle/_ref/l2_harness not built (needs /root/reference at build time)")


def test_reference_l2_callers_link_against_the_adapter(tmp_path):
    _build()
    r = subprocess.run([HARNESS, "symbols"], capture_output=True, text=True)
    assert r.returncode == 0 and "sizeof(matmul_params)=416" in r.stdout, r.stdout + r.stderr
    if not os.path.isdir(os.path.join(REF, "kernels")):
        return
    # what the reference's objects ask of the backend == what the adapter defines (no member is missing, none is re-typed in the harness)
    inc = ["-I", os.path.join(REPO, "oracle", "cuda_shim"), "-I", f"{REF}/llm/include", "-I", f"{REF}/llm/include/nn_modules", "-I", f"{REF}/kernels",
           "-I", f"{REF}/llm/half-2.2.0/include"]
    wanted = set()
    for src in L2_OBJECTS:
        obj = tmp_path / (os.path.basename(src) + ".o")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-DQM_CUDA", "-w", *inc, "-x", "c++", "-c", os.path.join(REF, src), "-o", str(obj)])
        for line in subprocess.check_output(["nm", "-uC", str(obj)], text=True).splitlines():
            if "matmul::MatmulOperator::" in line:
                wanted.add(line.split("U ", 1)[1].strip())
    assert len(wanted) == 10, wanted  # gemv_forward_cuda, naive_mat_mul_fp16_int4 and the eight int8 members
    defined = {ln.split(" T ", 1)[1].strip() for ln in subprocess.check_output(["nm", "-C", "--defined-only", HARNESS], text=True).splitlines()
               if " T matmul::MatmulOperator::" in ln}
    assert wanted <= defined, wanted - defined
    harness_src = open(os.path.join(REPO, "oracle", "l2_harness.cc")).read()
    assert "MatmulOperator::" not in harness_src.split("#include", 1)[1], "the harness must not define operator members itself"


def test_qm_hip_patch_applies_and_the_patched_flavour_builds(tmp_path):
    """tinychatengine_amd/adapter/reference_qm_hip.patch (the header diffs of INTEGRATION.md section 2) applies cleanly to the
    reference tree, and under -DQM_HIP -- real HIP headers, no shim -- hipcc compiles the unmodified llm/src/ops/cuda/linear.cu and
    the adapter against the PATCHED kernels/matmul.h; the two link into one shared object with nothing of matmul:: left undefined."""
    import shutil
    if not os.path.isdir(os.path.join(REF, "kernels")):
        pytest.skip("needs /root/reference")
    from tinychatengine_amd import build as B
    B.build()
    files = ["kernels/matmul.h", "llm/include/common.h", "llm/include/utils.h", "llm/include/operators.h", "llm/include/ops/linear.h"]
    for f in files:
        os.makedirs(tmp_path / os.path.dirname(f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), tmp_path / f)
    patch = os.path.join(REPO, "tinychatengine_amd", "adapter", "reference_qm_hip.patch")
    subprocess.check_call(["patch", "-p1", "-s", "-d", str(tmp_path), "-i", patch])
    inc = ["-I", str(tmp_path / "llm/include"), "-I", str(tmp_path / "kernels"), "-I", f"{REF}/llm/include", "-I", f"{REF}/llm/include/nn_modules",
           "-I", f"{REF}/kernels", "-I", f"{REF}/llm/half-2.2.0/include"]
    hip = ["hipcc", "-std=c++17", "-O1", "-DQM_HIP", "-w", "--offload-arch=gfx950", "-fPIC", "-x", "hip"]
    subprocess.check_call([*hip, "-c", f"{REF}/llm/src/ops/cuda/linear.cu", "-o", str(tmp_path / "linear.o"), *inc])
    adir = os.path.join(REPO, "tinychatengine_amd", "adapter")
    subprocess.check_call([*hip, "-DTCE_ADAPTER_USE_REFERENCE_HEADER", "-c", os.path.join(adir, "matmul_operator_hip.cc"), "-o", str(tmp_path / "adapter.o"),
                           "-I", os.path.join(REPO, "include"), "-I", adir, *inc])
    so = tmp_path / "libqmhip.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-o", str(so), str(tmp_path / "linear.o"), str(tmp_path / "adapter.o"),
                           "-L", os.path.join(REPO, "tinychatengine_amd", "lib"), "-ltce_hip"])
    undefined = [ln for ln in subprocess.check_output(["nm", "-uC", str(so)], text=True).splitlines() if "matmul::" in ln]
    assert not undefined, undefined


# ------------------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def _run(args):
    r = subprocess.run([HARNESS, *[str(a) for a in args]], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"l2_harness {args}: rc={r.returncode}\n{r.stdout}\n{r.stderr}"


@gpu
@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (1, 512, 11008), (4, 256, 4096), (33, 384, 4096)])
def test_reference_Linear_half_int4_forward_on_the_hip_backend(tmp_path, oracle, M, N, K):
    """llm/src/ops/cuda/linear.cu:5-40 (unmodified object) -> adapter -> libtce_hip.so.  K = 11008: padded scale / zero rows."""
    import torch
    from conftest import w4a16_close
    from tinychatengine_amd import quantize as Q
    assert torch.cuda.is_available()
    if not os.path.exists(HARNESS):
        pytest.skip("oracle/_ref/l2_harness not built")
    g = torch.Generator().manual_seed(N + K)
    w = torch.empty(N, K).normal_(0, 0.02, generator=g)
    qw, sc, zp = Q.quantize_q4_6(w, 128)
    d = str(tmp_path)
    Q.save_linear_q4_6(d, qw, sc, zp)
    x = torch.empty(M, K).normal_(0, 1, generator=g).to(torch.float16).numpy()
    x.tofile(os.path.join(d, "x.bin"))
    _run(["w4a16", d, M, N, K])
    got = np.fromfile(os.path.join(d, "out.bin"), dtype=np.float16).reshape(M, N)
    ref32 = oracle.w4a16_gemv_q4_6_mt(x, qw.numpy().view(np.uint32), sc.numpy(), zp.numpy().view(np.uint32), M, N, K, 128)
    assert not np.isnan(got.astype(np.float32)).any()
    ok, worst = w4a16_close(got, ref32)
    assert ok, worst


def _int8_files(d, B, M, N, K, seed, bias_kind):
    rng = np.random.default_rng(seed)
    x = rng.integers(-128, 128, (B, M, K), dtype=np.int8)
    w = rng.integers(-128, 128, (N, K), dtype=np.int8)
    x.tofile(os.path.join(d, "x.bin"))
    w.tofile(os.path.join(d, "weight.bin"))
    np.array([ALPHA], np.float32).tofile(os.path.join(d, "alpha.bin"))
    np.array([BETA], np.float32).tofile(os.path.join(d, "beta.bin"))
    if bias_kind == "int8":
        b = rng.integers(-128, 128, N, dtype=np.int8)
        b.tofile(os.path.join(d, "bias_int8.bin"))
    else:
        b = rng.standard_normal(N).astype(np.float32)
        b.tofile(os.path.join(d, "bias.bin"))
    return x, w, b


@gpu
@pytest.mark.parametrize("B,M,N,K", [(1, 108, 768, 768), (1, 1, 3072, 768), (2, 5, 64, 96)])
def test_reference_W8A8_linears_on_the_hip_backend(tmp_path, oracle, B, M, N, K):
    """W8A8B8O8Linear.cc:38-78, W8A8B8O8LinearReLU.cc:40-78, W8A8BFP32OFP32Linear.cc:35-73 (unmodified objects): bit-exact."""
    import torch
    assert torch.cuda.is_available()
    if not os.path.exists(HARNESS):
        pytest.skip("oracle/_ref/l2_harness not built")
    d = str(tmp_path)
    for cmd, qmin in (("w8a8", -128), ("w8a8relu", 0)):
        x, w, b = _int8_files(d, B, M, N, K, seed=M + N, bias_kind="int8")
        _run([cmd, d, B, M, N, K])
        got = np.fromfile(os.path.join(d, "out.bin"), dtype=np.int8).reshape(B, M, N)
        for i in range(B):
            assert np.array_equal(got[i], oracle.int8_matmul_bias_i8(x[i], w, b, ALPHA, BETA, qmin, 127, M, N, K)), (cmd, i)
    x, w, b = _int8_files(d, B, M, N, K, seed=M + N + 1, bias_kind="fp32")
    _run(["w8a8fp32", d, B, M, N, K])
    got = np.fromfile(os.path.join(d, "out.bin"), dtype=np.float32).reshape(B, M, N)
    for i in range(B):
        assert np.array_equal(got[i].view(np.uint32), oracle.int8_matmul_bias_f32(x[i], w, b, ALPHA, M, N, K).view(np.uint32))


@gpu
@pytest.mark.parametrize("B,M,N,K", [(12, 108, 108, 64), (12, 108, 64, 108 + 4), (12, 1, 40, 64)])
def test_reference_int8_BMMs_on_the_hip_backend(tmp_path, oracle, B, M, N, K):
    """BMM_S8T_S8N_F32T.cc:12-63 / BMM_S8T_S8N_S8T.cc:12-62 (unmodified objects); m == 1 with b > 1 takes the *_batch members."""
    import torch
    assert torch.cuda.is_available()
    if not os.path.exists(HARNESS):
        pytest.skip("oracle/_ref/l2_harness not built")
    d = str(tmp_path)
    rng = np.random.default_rng(B + M + N)
    x = rng.integers(-128, 128, (B, M, K), dtype=np.int8)
    w = rng.integers(-128, 128, (B, N, K), dtype=np.int8)
    x.tofile(os.path.join(d, "x.bin"))
    w.tofile(os.path.join(d, "weight.bin"))
    np.array([0.0013], np.float32).tofile(os.path.join(d, "alpha.bin"))
    a = float(np.float32(0.0013))
    _run(["bmm_f32", d, B, M, N, K])
    got = np.fromfile(os.path.join(d, "out.bin"), dtype=np.float32).reshape(B, M, N)
    for h in range(B):
        assert np.array_equal(got[h].view(np.uint32), oracle.int8_matmul_nobias_f32(x[h], w[h], a, M, N, K).view(np.uint32)), h
    _run(["bmm_s8", d, B, M, N, K])
    got = np.fromfile(os.path.join(d, "out.bin"), dtype=np.int8).reshape(B, M, N)
    for h in range(B):
        assert np.array_equal(got[h], oracle.int8_matmul_nobias_i8(x[h], w[h], a, -128, 127, M, N, K)), h
