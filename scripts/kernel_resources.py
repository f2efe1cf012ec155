#!/usr/bin/env python3
"""Registers / spills / LDS / occupancy of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), one row per kernel.

    python scripts/kernel_resources.py tinychatengine_amd/csrc/w4a16_gemv_i8.hip [filter-substring] [-- extra hipcc flags]
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(src, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage",
           "-I", os.path.join(REPO, "include"), "-I", os.path.join(REPO, "tinychatengine_amd", "csrc"), *extra, "-c", src, "-o", "/dev/null"]
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        raise SystemExit(r.returncode)
    rows, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: (?:[^:]*:\d+:\d+: )?\s*([A-Za-z ]+(?:\[bytes/\w+\])?): (\S+)", ln)
        if "Function Name:" in ln:
            cur = {"name": ln.split("Function Name:")[1].split()[0]}
            rows.append(cur)
        elif cur is not None and m:
            cur[m.group(1).strip()] = m.group(2)
    return rows


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


if __name__ == "__main__":
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        extra = args[args.index("--") + 1:]
        args = args[:args.index("--")]
    flt = args[1] if len(args) > 1 else ""
    for r in resources(args[0], extra):
        nm = demangle(r["name"])
        nm = re.sub(r"tce::\(anonymous namespace\)::", "", nm).split("(")[0]
        if flt and flt not in nm:
            continue
        print(f"{nm:90s} vgpr {r.get('VGPRs', '?'):>4} agpr {r.get('AGPRs', '?'):>4} spill {r.get('VGPRs Spill', '?'):>3} scratch {r.get('ScratchSize [bytes/lane]', '?'):>4} "
              f"sgpr {r.get('SGPRs', '?'):>4} occ {r.get('Occupancy [waves/SIMD]', '?'):>2} lds {r.get('LDS Size [bytes/block]', '?')}")
