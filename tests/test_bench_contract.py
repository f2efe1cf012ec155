"""The bench line the driver reads: every key the contract names, with the types and relations it checks (value = whole-job rate, roofline.frac =
achieved / peak, cpu_baseline on the reference's own code) -- and a hard SIZE budget: the driver keeps the last 8 KB of the job's output, and round 4's 21.7 KB
line came back as `parsed: null`.  Checked on (i) the committed lines, newest round first, (ii) bench.compact_line applied to the largest
full record in the repository, (iii) the real print path under `--gpus 1` and `--gpus 2` (bench.py --selftest-emit: no GPU work)."""
import glob
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROUNDS = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*")), key=lambda p: int(re.sub(r"\D", "", os.path.basename(p)) or 0))
LINES = [p for r in ROUNDS for p in sorted(glob.glob(os.path.join(r, "bench_n1*.json")))]
NEWEST = [p for p in sorted(glob.glob(os.path.join(ROUNDS[-1], "bench_n1.json")))] if ROUNDS else []
FULL_R4 = os.path.join(ROOT, "profiles", "r4", "bench_n1.json")  # a 21.7 KB full record: the worst case the compactor has seen
MAX_LINE = 8192


def check_contract(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert line["unit"] == "tokens/s" and line["higher_is_better"] is True
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


@pytest.mark.parametrize("path", LINES, ids=[os.path.relpath(p, ROOT) for p in LINES])
def test_bench_line_contract(path):
    text = open(path).read().strip().splitlines()
    line = json.loads(text[-1])  # the JSON line is the last line of stdout
    assert line["n_gpus"] == 1
    check_contract(line)


@pytest.mark.parametrize("path", NEWEST, ids=[os.path.relpath(p, ROOT) for p in NEWEST])
def test_newest_committed_line_fits_the_driver_tail(path):
    """From round 5 on the committed line IS what bench.py printed: it must fit the tail the driver keeps and carry roofline + cpu_baseline."""
    if int(re.sub(r"\D", "", os.path.basename(os.path.dirname(path)))) < 5:
        pytest.skip("rounds 1-4 committed the full record (that was the bug)")
    text = open(path).read().strip().splitlines()[-1]
    assert len(text) <= MAX_LINE, len(text)
    line = json.loads(text)
    assert "roofline" in line and "cpu_baseline" in line and "details_file" in line


def test_compact_line_of_the_largest_record():
    import bench
    full = json.loads(open(FULL_R4).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000
    c = bench.compact_line(full, "gpurun_out/bench_details_n1.json")
    text = bench.dump_line(c)
    assert len(text) <= bench.LINE_BUDGET <= MAX_LINE - 1024  # headroom: the driver's tail also holds stderr
    check_contract(c)
    assert c["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert len(c["roofline"]["shapes"]) == len(full["roofline"]["shapes"]) == 12
    assert c["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert "dropped_for_size" not in c
    # a smaller budget drops whole optional blocks, lowest priority first, never a contract key
    small = bench.compact_line(full, "x.json", budget=3600)
    assert len(bench.dump_line(small)) <= 3600 and small.get("dropped_for_size")
    check_contract(small)
    # blown-up prose fields cannot push the line over: they are cut, not copied
    fat = json.loads(json.dumps(full))
    fat["config"]["issue"] = "x" * 5000
    fat["roofline"]["kernel"] = "y" * 5000
    fat["cpu_baseline"]["sample"] = "z" * 5000
    assert len(bench.dump_line(bench.compact_line(fat, "x.json"))) <= bench.LINE_BUDGET


@pytest.mark.parametrize("gpus", [1, 2])
def test_the_line_is_the_last_stdout_line_and_fits(gpus, tmp_path):
    """The real print path (rank 0 prints after every rank flushed and met), launched the way the driver launches it: `python bench.py --gpus N`."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TCE_BENCH_DETAILS"] = str(tmp_path / "details.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--selftest-emit", FULL_R4], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert any("selftest noise" in l for l in lines[:-1])
    assert len(lines[-1]) <= MAX_LINE
    line = json.loads(lines[-1])
    assert line["n_gpus"] == gpus
    check_contract(line)
    # what the driver does: keep the last 8 KB of the output, parse the last line
    tail = (r.stdout + "\n\n---- stderr ----\n/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n")[-8192:]
    last = [l for l in tail.splitlines() if l.startswith("{")][-1]
    assert json.loads(last)["value"] == line["value"]
    full = json.load(open(env["TCE_BENCH_DETAILS"]))
    assert "other_configs" in full and len(json.dumps(full)) > 20000  # nothing is lost: the side file holds the full record


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus N` the way the driver invokes it (no launcher, WORLD_SIZE unset) must start the N ranks itself (it used to abort with
    "--gpus N needs torch.distributed.run").  Without a GPU every rank stops at bench.py's own "needs an MI355X" -- which proves the re-exec under
    torch.distributed.run happened, with the right world size, and that the ranks got past the argument / rendezvous-variable path."""
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
