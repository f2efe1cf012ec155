"""GPU: the C++ drop-in boundary end to end -- adapter_selftest fills a (poisoned) matmul_params like the reference's L2
wrappers, calls matmul::MatmulOperator members, and compares with expected values computed here by the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cpp_adapter_selftest(oracle, tmp_path):
    import torch
    assert torch.cuda.is_available()
    from tinychatengine_amd import build as B
    B.build_adapter()
    rng = np.random.default_rng(2024)
    # case 1: FP16Linear_int4 (test_ops.cu:671-724 uses m=1, n=32000, k=4096; here n=2048 keeps the oracle instant)
    M, N, K, G = 1, 2048, 4096, 128
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    qw, sc, zp, _, _ = oracle.quantize_q4_6(w, G)
    x = rng.standard_normal((M, K)).astype(np.float16)
    ref32, _ = oracle.w4a16_gemv_q4_6(x, qw, sc, zp, M, N, K, G)
    blob = struct.pack("5i", M, N, K, G, zp.shape[1]) + qw.tobytes() + sc.tobytes() + zp.tobytes() + x.tobytes() + ref32.tobytes()
    # case 2: W8A8B8O8LinearReLU 108x768x3072 with the reference's alpha/beta (test_ops.cc:177-209)
    M, N, K = 108, 3072, 768
    al, be = 0.0005035400390625, 0.02130126953125
    A = rng.integers(-128, 128, (M, K), dtype=np.int8)
    Bm = rng.integers(-128, 128, (N, K), dtype=np.int8)
    bias = rng.integers(-128, 128, N, dtype=np.int8)
    exp = oracle.int8_matmul_bias_i8(A, Bm, bias, al, be, 0, 127, M, N, K)
    blob += struct.pack("3i2f", M, N, K, al, be) + A.tobytes() + Bm.tobytes() + bias.tobytes() + exp.tobytes()
    # case 3: the adapter's tensor cache.  Expected outputs are the library's own through the C ABI (bit-exact comparison in the
    # self-test: the cached zero-point-8 fast path and the general path must both equal what a direct call returns)
    from tinychatengine_amd import capi
    M, N, K, G = 1, 64, 512, 128
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    qw, sc, zp8, _, _ = oracle.quantize_q4_6(w, G)
    nib = rng.integers(0, 16, (N, zp8.shape[1] * 8), dtype=np.uint32)
    zpr = (nib.reshape(N, -1, 8) << (np.arange(8, dtype=np.uint32) * 4)).sum(axis=2).astype(np.uint32)
    x = rng.standard_normal((M, K)).astype(np.float16)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    outs = []
    for z in (zp8, zpr):
        tq, ts, tz, tx = t(qw.view(np.int32)), t(sc.view(np.float16)), t(z.view(np.int32)), t(x)
        o = torch.zeros((M, N), dtype=torch.float16, device=dev)
        # the adapter runs decode batches on the packed copy it builds at first sight of a weight tensor (round 4): the expected bits come from the same kernel family
        from tinychatengine_amd.linear import Linear_half_int4
        lin = Linear_half_int4(tq, ts, tz, G).prepack()
        d = lin.desc(tx, o)
        assert capi.describe_dispatch(d).startswith("gemv-i8")
        capi.check(capi.w4a16_forward(d, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        ref32, _ = oracle.w4a16_gemv_q4_6(x, qw, sc, z, M, N, K, G)
        from conftest import w4a16_close
        assert w4a16_close(o.cpu().numpy(), ref32)[0]
        outs.append(o.cpu().numpy())
    assert not np.array_equal(outs[0], outs[1])
    blob += struct.pack("5i", M, N, K, G, zp8.shape[1]) + qw.tobytes() + sc.tobytes() + zp8.tobytes() + zpr.tobytes() + x.tobytes() + outs[0].tobytes() + outs[1].tobytes()
    path = tmp_path / "vectors.bin"
    path.write_bytes(blob)
    r = subprocess.run([B.ADAPTER_TEST_PATH, str(path)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("Passed!") == 5 and "Fail!" not in r.stdout  # (round 6: + the GEMM scratch fault count)


def test_adapter_bench_tiny_runs_every_leg():
    """adapter/adapter_bench.cc (the drop-in path on the clock: MatmulOperator::gemv_forward_cuda per linear, null stream, eager, the reference's glue order) on the
    tiny model: all three legs run, every launch is counted (11 per block + final norm + lm_head), the logits are finite, the adapter holds one packed copy per linear."""
    import json
    from tinychatengine_amd import build as B
    B.build_adapter()
    r = subprocess.run([B.ADAPTER_BENCH_PATH, "--model", "tiny", "--tokens", "20", "--warmup", "3", "--keys", "96"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["non_finite_logits"] == 0
    for leg in ("adapter", "capi_eager", "capi_graph"):
        assert d[leg]["launches_per_token"] == 11 * 2 + 2 and d[leg]["tokens_per_s"] > 0 and d[leg]["ms_per_token_events"] > 0
    assert d["adapter_device_bytes"] > 0


def test_adapter_never_synchronises_in_steady_state(tmp_path):
    """the operator is asynchronous and never syncs.  Round 4's adapter read the zero-point verdict back with a blocking copy on first sight of a tensor;
    round 5 polls a pinned word.  rocprofv3's HIP-API trace of adapter_bench (adapter leg only, 3 + 10 tokens): between the first hipEventRecord and the second --
    the timed tokens -- there is no hipDeviceSynchronize / hipStreamSynchronize / hipMemcpy* at all."""
    import csv
    import glob
    import shutil
    from tinychatengine_amd import build as B
    B.build_adapter()
    if not shutil.which("rocprofv3"):
        pytest.skip("rocprofv3 not on PATH")
    out = tmp_path / "trace"
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(["rocprofv3", "--hip-runtime-trace", "--output-format", "csv", "-d", str(out), "--", B.ADAPTER_BENCH_PATH, "--model", "tiny", "--tokens", "10", "--warmup", "3",
                        "--keys", "96", "--legs", "adapter"], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    files = glob.glob(str(out / "**" / "*hip_api_trace.csv"), recursive=True)
    assert files, os.listdir(out)
    rows = sorted(csv.DictReader(open(files[0])), key=lambda x: int(x["Start_Timestamp"]))
    names = [x["Function"] for x in rows]
    rec = [i for i, n in enumerate(names) if n == "hipEventRecord"]
    assert len(rec) >= 2
    timed = names[rec[0] + 1:rec[1]]
    assert sum(n in ("hipLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel", "hipExtLaunchKernel") for n in timed) >= 10 * 24
    blocking = [n for n in timed if "Synchronize" in n or n.startswith("hipMemcpy") or n.startswith("hipMemset") or "Malloc" in n or "Free" in n]
    assert not blocking, sorted(set(blocking))
