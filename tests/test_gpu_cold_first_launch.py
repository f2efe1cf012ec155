"""Every forced form of the pre-packed GEMM as the FIRST execution of its kernel in a fresh process, behind a launch of another form (round 5).

A kernel's first execution differs from every later one: cold instruction cache, LDS and registers holding another kernel's leftovers.  Tests that run one process and
repeat launches on the same data cannot see a bug that needs that state -- round 5 met one (csrc/w4a16_gemm_pk.hip, the note in `rescale`).  Here each group size gets
fresh processes (scripts/probes/pk_form2_g32_repeat.py) that run form 1 and then every other form once, a NaN-poisoning launch (csrc/testkit_poison.hip) in front of
each, against the oracle."""
import ast
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES_128 = [61, 62, 63, 64, 66, 67, 68, 2669, 2670, 2671, 2672, 2673, 2674, 2675, 60]
MODES_OTHER = [61, 62, 63, 64, 60]


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,G", [(192, 200, 512, 32), (192, 200, 512, 64), (384, 520, 2048, 128), (260, 300, 1408, 64)])
def test_first_execution_of_every_gemm_form(M, N, K, G):
    modes = MODES_128 if G == 128 else MODES_OTHER
    for attempt in range(2):  # (the failure this test was written for showed in 6 of 6 fresh processes)
        env = dict(os.environ, REPS="1", POISON="1")
        r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "probes", "pk_form2_g32_repeat.py"), str(M), str(N), str(K), str(G)] + [str(m) for m in modes],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        rows = [ln for ln in r.stdout.splitlines() if ln[:1].isdigit()]
        assert rows, r.stdout[-2000:]
        row = ast.literal_eval(rows[0].split(" ", 1)[1])
        bad = {m: v[:3] for m, v in row.items() if v[0] > 1.0}
        assert not bad, f"attempt {attempt}, {M}x{N}x{K} g{G}: first executions off the oracle: {bad}"
