#!/usr/bin/env python3
"""Round 5 probe tooling: builds of the library whose GEMM device code is the compiler's own assembly with a hand edit applied -- single-site experiments on the
two-quartet groups-of-32 kernel (profiles/r5/pk_form2_g32_first_launch.txt).

    python scripts/probes/pk_asm_variant.py name[:edit[,edit...]] ...        -> tinychatengine_amd/lib/abl/libtce_<name>.so   (load with TCE_LIB_PATH)

edits (applied to w4a16_gemm_pk_kernel<2, 5, 0, 1> only):
    shiftN        N s_nop 0 at the kernel's entry (moves its code by 4 N bytes)
    gapA:N        s_nop N-1 between the packed multiply that stands directly in front of an unrelated MFMA and that MFMA (site A)
    gapB:N        s_nop N-1 directly in front of the tile's own MFMA (site B: more wait states behind the tile's rescale products)
    gapC:N        s_nop N-1 behind the unrelated MFMA, in front of the unpacked multiplies
The pipeline is hipcc's own, spelled out: device assembly -> object -> code object -> bundle -> host object with the bundle embedded."""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
SRC = os.path.join(REPO, "tinychatengine_amd", "csrc", "w4a16_gemm_pk.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-Wno-unused-function", "-w", "-I", os.path.join(REPO, "include"), "-I", os.path.dirname(SRC)]
WORK = "/tmp/asmx"
KERNEL = "_ZN3tce12_GLOBAL__N_120w4a16_gemm_pk_kernelILi2ELi5ELi0ELi1EEEvNS0_10PkGemmArgsE"


def pad(n):
    """n wait states as an EVEN number of s_nop instructions (whole 8-byte units: the code behind the pad keeps its position modulo 8)"""
    assert n >= 2 and n % 2 == 0, "an even number of wait states, at least two"
    out, half = [], n // 2
    for part in (half, half):
        while part > 0:
            out.append(f"\ts_nop {min(part, 16) - 1}")
            part -= min(part, 16)
    if len(out) % 2: out.append("\ts_nop 0")
    return out


def sh(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout[-3000:])
        raise SystemExit(1)


def device_asm():
    os.makedirs(WORK, exist_ok=True)
    out = os.path.join(WORK, "pk_dev.s")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(SRC):
        sh(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-S", SRC, "-o", out])
    return open(out).read().split("\n")


def kernel_range(lines):
    b = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    e = next(i for i in range(b, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return b, e


def find_site(lines, b, e):
    """site: v_pk_mul_f32 X / v_mfma (not reading X) / ... / v_mfma reading X as its third source within a few instructions, with unpacked v_mul_f32_e64 behind the first MFMA"""
    ins = [(i, lines[i].split(";")[0].strip()) for i in range(b, e) if lines[i].startswith("\t") and not lines[i].strip().startswith((".", ";"))]
    sites = []
    for k, (i, t) in enumerate(ins[:-8]):
        if not t.startswith("v_pk_mul_f32"): continue
        nxt = ins[k + 1][1]
        if not nxt.startswith("v_mfma"): continue
        if not (ins[k + 2][1].startswith("v_mul_f32_e64") and ins[k + 3][1].startswith("v_mul_f32_e64")): continue
        dst = re.match(r"v_pk_mul_f32 v\[(\d+):(\d+)\]", t)
        lo = int(dst.group(1))
        for kk in range(k + 2, k + 9):
            tt = ins[kk][1]
            if tt.startswith("v_mfma"):
                m = re.findall(r"v\[(\d+):(\d+)\]", tt)
                if m and int(m[-1][0]) <= lo <= int(m[-1][1]):
                    sites.append((i, ins[k + 1][0], ins[kk][0]))
                break
    return sites


def apply(lines, edits):
    b, e = kernel_range(lines)
    out = list(lines)
    for ed in edits:
        if ed.startswith("shift"):
            n = int(ed[5:])
            bb, _ = kernel_range(out)
            first = next(i for i in range(bb + 1, len(out)) if out[i].startswith("\t") and not out[i].strip().startswith((".", ";")))
            out[first:first] = ["\ts_nop 0"] * n
        elif ed.startswith("gap"):
            which, n = ed[3], int(ed.split(":")[1])
            bb, ee = kernel_range(out)
            sites = find_site(out, bb, ee)
            if not sites:
                raise SystemExit(f"no site found for {ed}")
            for (pk, mf1, mf2) in reversed(sites):
                at = {"A": mf1, "B": mf2, "C": mf1 + 1}[which]
                out[at:at] = pad(n)
            print(f"  {ed}: {len(sites)} site(s)", flush=True)
        elif ed.startswith("post:"):  # N wait states behind the loop's exit (in front of the first instruction that reads an accumulator after the last MFMA)
            n = int(ed.split(":")[1])
            bb, ee = kernel_range(out)
            last_mfma = max(i for i in range(bb, ee) if out[i].strip().startswith("v_mfma"))
            at = next(i for i in range(last_mfma, ee) if out[i].strip().startswith("v_pk_mul_f32") or out[i].strip().startswith("v_mul_f32"))
            pads = pad(n)
            out[at:at] = pads
            print(f"  {ed}: in front of line {at - bb} of the kernel: {out[at + len(pads)].strip()}", flush=True)
        elif ed.startswith("postafter:"):  # N wait states behind the post-loop multiplies (in front of the wait + barrier that follow them)
            n = int(ed.split(":")[1])
            bb, ee = kernel_range(out)
            last_mfma = max(i for i in range(bb, ee) if out[i].strip().startswith("v_mfma"))
            at = next(i for i in range(last_mfma, ee) if out[i].strip().startswith("s_barrier"))
            at = max(i for i in range(last_mfma, at) if out[i].strip().startswith("v_pk_mul_f32")) + 1
            pads = pad(n)
            out[at:at] = pads
            print(f"  {ed}: behind line {at - bb - 1} of the kernel: {out[at - 1].strip()}", flush=True)
        elif ed.startswith("postvalu:"):  # N independent vector moves (a busy vector ALU instead of an idle one) behind the loop's exit
            n = int(ed.split(":")[1])
            bb, ee = kernel_range(out)
            last_mfma = max(i for i in range(bb, ee) if out[i].strip().startswith("v_mfma"))
            at = next(i for i in range(last_mfma, ee) if out[i].strip().startswith("v_pk_mul_f32") or out[i].strip().startswith("v_mul_f32"))
            out[at:at] = ["\tv_mov_b32_e32 v232, v233"] * n
            print(f"  {ed}: {n} v_mov in front of {out[at + n].strip()}", flush=True)
        elif ed in ("unpk", "unpk1", "spread"):  # the post-loop packed multiplies as two plain ones each (unpk: all of them; unpk1: those that take the scale pair's HIGH register for both products) / one wait state between them
            bb, ee = kernel_range(out)
            last_mfma = max(i for i in range(bb, ee) if out[i].strip().startswith("v_mfma"))
            bar = next(i for i in range(last_mfma, ee) if out[i].strip().startswith("s_barrier"))
            cnt = 0
            for i in range(bar, last_mfma, -1):
                t = out[i].split(";")[0].strip()
                m = re.match(r"v_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] (op_sel_hi:\[1,0\]|op_sel:\[0,1\])$", t)
                if not m: continue
                d0, s0, e0 = int(m.group(1)), int(m.group(3)), int(m.group(5))
                hi = m.group(7) == "op_sel:[0,1]"
                if ed == "spread":
                    out[i + 1:i + 1] = ["\ts_nop 0"]
                    cnt += 1
                    continue
                if ed == "unpk1" and not hi: continue
                e = e0 + 1 if hi else e0
                assert d0 != s0 + 1, t
                out[i:i + 1] = [f"\tv_mul_f32_e32 v{d0}, v{s0}, v{e}", f"\tv_mul_f32_e32 v{d0 + 1}, v{s0 + 1}, v{e}"]
                cnt += 1
            print(f"  {ed}: {cnt} instruction(s)", flush=True)
        elif ed == "hipair":  # the post-loop multiplies that take the scale pair's HIGH register for both products read a pair {hi, hi} through the LOW-register form instead (no op_sel:[0,1])
            bb, ee = kernel_range(out)
            last_mfma = max(i for i in range(bb, ee) if out[i].strip().startswith("v_mfma"))
            bar = next(i for i in range(last_mfma, ee) if out[i].strip().startswith("s_barrier"))
            first = None
            cnt = 0
            for i in range(last_mfma, bar):
                t = out[i].split(";")[0].strip()
                m = re.match(r"(v_pk_mul_f32 v\[\d+:\d+\], v\[\d+:\d+\]), v\[(\d+):(\d+)\] op_sel:\[0,1\]$", t)
                if not m: continue
                if first is None: first, hi = i, int(m.group(3))
                out[i] = f"\t{m.group(1)}, v[232:233] op_sel_hi:[1,0]"
                cnt += 1
            out[first:first] = [f"\tv_mov_b32_e32 v232, v{hi}", f"\tv_mov_b32_e32 v233, v{hi}"]
            print(f"  {ed}: {cnt} instruction(s)", flush=True)
        elif ed.startswith("allB:"):  # s_nop N-1 in front of EVERY MFMA of the loop whose accumulator input was written by a vector instruction fewer than `within` wait states earlier
            n = int(ed.split(":")[1])
            within = int(ed.split(":")[2]) if ed.count(":") > 1 else 8
            bb, ee = kernel_range(out)
            ins = [i for i in range(bb, ee) if out[i].startswith("\t") and not out[i].strip().startswith((".", ";"))]
            hits = []
            for k, i in enumerate(ins):
                t = out[i].split(";")[0].strip()
                if not t.startswith("v_mfma"): continue
                m = re.findall(r"v\[(\d+):(\d+)\]", t)
                lo, hi = int(m[-1][0]), int(m[-1][1])
                d = 0
                for kk in range(k - 1, max(0, k - 14), -1):
                    tt = out[ins[kk]].split(";")[0].strip()
                    mm = re.match(r"(v_pk_mul_f32|v_mul_f32_e64|v_mul_f32_e32) v\[?(\d+)", tt)
                    if mm and lo <= int(mm.group(2)) <= hi:
                        if d < within: hits.append(i)
                        break
                    d += int(tt.split()[1]) + 1 if tt.startswith("s_nop") else 1
            for i in reversed(hits):
                out[i:i] = pad(n)
            print(f"  {ed}: {len(hits)} MFMA(s) padded", flush=True)
        else:
            raise SystemExit(f"unknown edit {ed}")
    return out


def build(name, edits):
    lines = apply(device_asm(), edits)
    s = os.path.join(WORK, f"pk_{name}.s")
    open(s, "w").write("\n".join(lines))
    o, hs, fb, ho = (os.path.join(WORK, f"pk_{name}.{x}") for x in ("dev.o", "hsaco", "hipfb", "host.o"))
    sh([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    sh([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hs, o])
    sh([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", f"-input={hs}", f"-output={fb}"])
    sh(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-host-only", "-c", SRC, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-o", ho])
    L = os.path.join(REPO, "tinychatengine_amd", "lib")
    os.makedirs(os.path.join(L, "abl"), exist_ok=True)
    objs = [os.path.join(L, f) for f in sorted(os.listdir(L)) if f.endswith(".o") and f != "w4a16_gemm_pk.o"]
    lib = os.path.join(L, "abl", f"libtce_{name}.so")
    sh(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, ho])
    print("built", lib, flush=True)


if __name__ == "__main__":
    for spec in sys.argv[1:]:
        name, _, ed = spec.partition(":")
        build(name, [x for x in ed.split(",") if x])
