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
                out[at:at] = [f"\ts_nop {n - 1}"]
            print(f"  {ed}: {len(sites)} site(s)", flush=True)
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
