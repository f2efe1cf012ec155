"""Host-side check of the index arithmetic shared by the prepack kernel and the 128-row MFMA GEMM
(tinychatengine_amd/csrc/w4a16_mfma_layout.hpp): compiled with g++ and run here, no GPU involved."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mfma_layout_index_arithmetic(tmp_path):
    exe = tmp_path / "test_mfma_layout"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(REPO, "tinychatengine_amd", "csrc"),
                           os.path.join(REPO, "tests", "host", "test_mfma_layout.cc"), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "mfma layout ok" in r.stdout, r.stdout + r.stderr
