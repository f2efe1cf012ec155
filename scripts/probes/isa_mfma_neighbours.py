#!/usr/bin/env python3
"""Round 5 probe (no GPU): where, in the device code of csrc/w4a16_gemm_pk.hip, a packed-f32 vector instruction stands DIRECTLY in front of an MFMA that does not read its
result while the MFMA that does read it (as its accumulator input) follows within a few wait states -- the shape of the one site the failing groups-of-32 kernel had and its
clean builds did not (profiles/r5/pk_form2_g32_first_launch.txt).  Also prints, per kernel, the histogram of wait states between a vector write and the MFMA reading it.

    python scripts/probes/isa_mfma_neighbours.py [object file = tinychatengine_amd/lib/w4a16_gemm_pk.o] [max wait states = 6]"""
import collections, os, re, shutil, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return [int(m.group(1))] if m else []


def disassemble(obj):
    d = tempfile.mkdtemp()
    try:
        o = os.path.join(d, "x.o")
        shutil.copy(obj, o)
        subprocess.run([OBJDUMP, "--offloading", o], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        return subprocess.run([OBJDUMP, "-d", os.path.join(d, co[0])], capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)


def kernels(text):
    out, cur = {}, None
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None or "\t" not in ln:
            continue
        t = ln.split("\t", 1)[1].split("//")[0].strip()
        if not t:
            continue
        parts = t.split(None, 1)
        ops = [o.strip().split()[0] if o.strip() else "" for o in (parts[1] if len(parts) > 1 else "").split(",")]
        cur.append((parts[0], ops, t))
    return out


def wait_states(i):
    return int(i[1][0]) + 1 if i[0] == "s_nop" else 1


def straddles(ins, maxd):
    hits = []
    for k, (op, ops, t) in enumerate(ins[:-2]):
        if not (op.startswith("v_pk_") and "f32" in op) or not ins[k + 1][0].startswith("v_mfma"):
            continue
        w = set(regs(ops[0]))
        if w & set(sum((regs(o) for o in ins[k + 1][1][1:4]), [])):
            continue
        d = 0
        for j in range(k + 1, min(len(ins), k + 40)):
            if j > k + 1 and ins[j][0].startswith("v_mfma") and w & set(regs(ins[j][1][3])):
                if d <= maxd:
                    hits.append((d, t, ins[j][2]))
                break
            d += wait_states(ins[j])
    return hits


def write_to_mfma(ins):
    hist = collections.Counter()
    for k, (op, ops, t) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        need, d = set(regs(ops[3])), 0
        for j in range(k - 1, max(-1, k - 16), -1):
            pop, pops, _ = ins[j]
            if pop.startswith("v_") and not pop.startswith(("v_mfma", "v_cmp")) and pops and set(regs(pops[0])) & need:
                hist[(pop.split("_e")[0] if pop.endswith(("_e32", "_e64")) else pop, d)] += 1
                break
            d += wait_states(ins[j])
    return hist


if __name__ == "__main__":
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tinychatengine_amd", "lib", "w4a16_gemm_pk.o")
    maxd = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    for name, ins in kernels(disassemble(obj)).items():
        hits = straddles(ins, maxd)
        if not hits:
            continue
        nice = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("tce::(anonymous namespace)::", "").split("(")[0]
        h = write_to_mfma(ins)
        close = {f"{op}@{d}": n for (op, d), n in sorted(h.items()) if d <= 3}
        print(f"{nice}: {len(hits)} site(s), wait states {sorted(set(x[0] for x in hits))}; vector write -> accumulator read at <= 3 wait states: {close}")
        print(f"      e.g. {hits[0][1]}  ...  {hits[0][2]}")
