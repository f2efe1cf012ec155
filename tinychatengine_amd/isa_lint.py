"""Build-time ISA lint over the gfx950 code objects of the library (round 6).

Why it exists: round 5 shipped a kernel that returned a wrong result about once in 10^3 launches under load -- one accumulator register's lanes 48-63 without their
history -- from device code the compiler's own hazard recognizer accepts.  Round 6 traced it to ONE instruction form and reproduced it in a 150-line standalone
kernel (scripts/probes/pk_opsel_repro.hip; record and numbers: profiles/r6/pk_lost_lanes_rule.md):

    RULE 1   v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 whose LOW half takes the HIGH register of the second source pair (`op_sel:[0,1]`, `op_sel:[0,1,0]`) compute the
             low result in lanes 48-63 as if that register held 0.0 -- intermittently, and only while ANOTHER wave on the same SIMD is executing MFMAs: 2e-7 .. 4e-4 of the
             products by the length of the idle stretch in front of the instruction (a GEMM's post-loop block: 1 launch in 10^2 .. 10^3; with 64 idle wait states
             in front of it 45-100 % of the launches).  Never seen (0 of 2.7e10 products each, beside 1e4 .. 1e7 failures of the bad forms in the same runs, four boxes):
             the same operations selecting through `op_sel_hi` only (`op_sel_hi:[1,0]`: both halves take the LOW register -- what the fixed sources emit), `op_sel:[1,0]`
             and `op_sel:[1,1]`, v_pk_mov_b32 with any op_sel, plain VOP2 / VOP3 multiplies, one wave per SIMD, a SIMD without MFMA traffic.
             The lint refuses EVERY low-half op_sel on the three arithmetic instructions (nothing needs the two forms that measured clean) and lets v_pk_mov_b32 pass.

hipcc emits the bad form when it packs two scalar f32 operations whose shared operand lives in the odd register of a 64-bit pair (it folds the broadcast into op_sel).
Nothing in the toolchain knows the rule, so the build enforces it: any object holding such an instruction is refused (build.py deletes it), and the source is changed
until the compiler stops producing it (csrc/w4a16_gemm_pk.hip, `e_last`).

    python -m tinychatengine_amd.isa_lint [object files ...]      (default: every object of tinychatengine_amd/lib)
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
OBJDUMP = os.path.join(LLVM_BIN, "llvm-objdump")

# the instruction classes RULE 1 covers: VOP3P arithmetic on 64-bit register pairs (v_pk_mov_b32 measured clean with every op_sel: profiles/r6/pk_opsel_repro_forms.jsonl)
PAIR_OPS = ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32")
_OPSEL_LOW = re.compile(r"\bop_sel:\[([01](?:,[01])*)\]")


def disassemble(obj: str) -> str:
    """Disassembly of the gfx950 code object embedded in a host object file (hipcc -c output)."""
    d = tempfile.mkdtemp(prefix="tce_lint_")
    try:
        o = os.path.join(d, "x.o")
        shutil.copy(obj, o)
        subprocess.run([OBJDUMP, "--offloading", o], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d, check=False)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        if not co:
            return ""
        return subprocess.run([OBJDUMP, "-d", os.path.join(d, co[0])], capture_output=True, text=True, check=True).stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)


def kernels(text: str) -> dict[str, list[str]]:
    """symbol -> its instructions (mnemonic and operands, comments stripped)"""
    out: dict[str, list[str]] = {}
    cur = None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None or "\t" not in ln:
            continue
        t = ln.split("\t", 1)[1].split("//")[0].strip()
        if t:
            cur.append(t)
    return out


def demangle(names: list[str]) -> list[str]:
    if not names:
        return []
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.splitlines() if r.returncode == 0 else names
    return [n.replace("tce::(anonymous namespace)::", "").replace("(anonymous namespace)::", "") for n in out]


def rule1_violations(instructions: list[str]) -> list[str]:
    bad = []
    for t in instructions:
        if not t.startswith(PAIR_OPS):
            continue
        m = _OPSEL_LOW.search(t)
        if m and "1" in m.group(1):
            bad.append(t)
    return bad


def lint_object(obj: str) -> tuple[list[str], list[tuple[str, str, int]]]:
    """(kernel symbols of the object, violations as (kernel, example instruction, count))"""
    ks = kernels(disassemble(obj))
    viol = []
    for name, ins in ks.items():
        bad = rule1_violations(ins)
        if bad:
            viol.append((name, bad[0], len(bad)))
    return list(ks.keys()), viol


def format_violations(obj: str, viol: list[tuple[str, str, int]]) -> str:
    names = demangle([v[0] for v in viol])
    lines = [f"{os.path.basename(obj)}: {len(viol)} kernel(s) hold a packed f32 instruction whose LOW half selects a HIGH source register (isa_lint.py RULE 1):"]
    for (_, ex, n), nice in zip(viol, names):
        lines.append(f"    {n:4d} x  {ex}    in {nice.split('(')[0]}")
    return "\n".join(lines)


if __name__ == "__main__":
    lib_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
    objs = sys.argv[1:] or sorted(os.path.join(lib_dir, f) for f in os.listdir(lib_dir) if f.endswith(".o"))
    rc = 0
    for o in objs:
        names, viol = lint_object(o)
        if viol:
            rc = 1
            print(format_violations(o, viol))
        else:
            print(f"{os.path.basename(o)}: {len(names)} kernels, clean")
    sys.exit(rc)
