"""Generate tests/golden/cuda_gemv_golden.npz: outputs of the REFERENCE'S OWN CUDA GEMV KERNEL -- kernels/cuda/gemv_cuda.cu,
matmul::MatmulOperator::gemv_forward_cuda -> gemv_kernel_g128, the hot-path kernel this repository replaces -- on seeded inputs at the
BASELINE decode shapes.  The kernel cannot run on a device here; oracle/_ref/glue_harness runs its unmodified source on the CPU through
the host emulation in oracle/cuda_emul/ (a block's 32 x 4 threads as concurrent OS threads, __shfl_down_sync as a warp exchange):
same operations in the same order as on the device, fp32 accumulation, one fp16 rounding at the end.

Only the OUTPUTS are stored (a few tens of KB); the inputs are regenerated from the recorded seeds by gemv_case() below, which the
GPU test imports -- weights N(0, 0.02^2) quantized by the oracle's q4_6 quantizer (itself pinned against the reference's Python
quantizer, tests/test_oracle.py), activations N(0, 1).

    make -C oracle glue && python tests/golden/make_cuda_gemv_golden.py        (build container only: needs /root/reference)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

CASES = [  # (M, N, K, seed): the four per-block linears of the BASELINE-named model, its lm_head, a batch of 2, the 13B down_proj's K
    (1, 4096, 4096, 101), (1, 12288, 4096, 102), (1, 11008, 4096, 103), (1, 4096, 11008, 104), (1, 32000, 4096, 105), (2, 512, 4096, 106),
    (1, 1024, 13824, 107),
]


def gemv_case(oracle, M, N, K, seed):
    """The inputs of a case: (activations fp16 [M][K], qweight u32 [N][K/8], scales fp16 [N][zw*8], zeros u32 [N][zw])."""
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((N, K)) * 0.02).astype(np.float32)
    qw, sc, zp, _, _ = oracle.quantize_q4_6(w, 128)
    a = rng.standard_normal((M, K)).astype(np.float16)
    return a, np.ascontiguousarray(qw, np.uint32), np.ascontiguousarray(sc, np.float16), np.ascontiguousarray(zp, np.uint32)


def run_reference_kernel(a, qw, sc, zp, M, N, K):
    harness = os.path.join(REPO, "oracle", "_ref", "glue_harness")
    with tempfile.TemporaryDirectory() as d:
        paths = {}
        for name, arr in (("x", a), ("q", qw), ("s", sc), ("z", zp)):
            paths[name] = os.path.join(d, name + ".bin")
            arr.tofile(paths[name])
        out = os.path.join(d, "o.bin")
        subprocess.run([harness, "gemv", str(M), str(N), str(K), paths["x"], paths["q"], paths["s"], paths["z"], out], check=True, timeout=1800)
        return np.fromfile(out, np.float16).reshape(M, N)


def main():
    from oracle.oracle import Oracle
    orc = Oracle()
    store = {"cases": np.array(CASES, np.int64)}
    for (M, N, K, seed) in CASES:
        a, qw, sc, zp = gemv_case(orc, M, N, K, seed)
        out = run_reference_kernel(a, qw, sc, zp, M, N, K)
        store[f"out_{M}_{N}_{K}"] = out
        ref32, _ = orc.w4a16_gemv_q4_6(a, qw, sc, zp, M, N, K, 128)
        same = float((out.view(np.uint16) == ref32.astype(np.float16).view(np.uint16)).mean())
        print(f"{M} x {N} x {K}: reference CUDA kernel == fp16(oracle fp32) on {same * 100:.2f} % of the outputs", flush=True)
    np.savez_compressed(os.path.join(HERE, "cuda_gemv_golden.npz"), **store)


if __name__ == "__main__":
    main()
