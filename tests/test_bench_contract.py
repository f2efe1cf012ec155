"""The committed bench lines (profiles/r1/bench_n1*.json, written by bench.py on the GPU box) carry every key the driver's contract names, with the
types and relations it checks (value = whole-job rate, roofline.frac = achieved / peak, cpu_baseline on the reference's own code)."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r1", "bench_n1*.json")))


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_bench_line_contract(path):
    text = open(path).read().strip().splitlines()
    line = json.loads(text[-1])  # the JSON line is the last line of stdout
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert line["unit"] == "tokens/s" and line["higher_is_better"] is True and line["n_gpus"] == 1
    assert line["data"] == "synthetic" and line["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 1000.0 / line["ms_per_step"]) / line["value"] < 0.02
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / r["avg_launch_us"] / 1e3) / r["achieved"] < 1e-2
    if r.get("traffic") is not None:  # PMC bytes per launch: no wasted re-reads
        assert 0.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.1
    if "cpu_baseline" in line:
        c = line["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == line["unit"] and c["sample"]


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus N` the way the driver invokes it (no launcher, WORLD_SIZE unset) must start the N ranks itself (it used to abort with
    "--gpus N needs torch.distributed.run").  Without a GPU every rank stops at bench.py's own "needs an MI355X" -- which proves the re-exec under
    torch.distributed.run happened, with the right world size, and that the ranks got past the argument / rendezvous-variable path."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher path (the GPU box runs the real thing)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=300, env=env)
    err = r.stderr
    assert "starting the ranks" in err and "--nproc-per-node=2" in err, err[-2000:]
    assert "needs torch.distributed.run" not in err
    assert err.count("bench.py needs an MI355X") >= 1, err[-2000:]  # the ranks ran bench.py's main() (both print it; the launcher may cut the second short)
    assert r.returncode != 0  # and the job fails loudly: no CPU fallback
