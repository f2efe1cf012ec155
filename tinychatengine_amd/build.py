"""Build libtce_hip.so (the C-ABI library, include/tce_matmul.h) in-tree with hipcc for gfx950.

    python -m tinychatengine_amd.build [--force]

The library is written for MI355X only: one offload arch, no fallback path.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libtce_hip.so")
ADAPTER_LIB_PATH = os.path.join(LIB_DIR, "libtce_matmul_operator.so")
LAB_LIB_PATH = os.path.join(LIB_DIR, "libtce_hip_lab.so")
TESTKIT_LIB_PATH = os.path.join(LIB_DIR, "libtce_testkit.so")
ADAPTER_TEST_PATH = os.path.join(LIB_DIR, "adapter_selftest")
ADAPTER_BENCH_PATH = os.path.join(LIB_DIR, "adapter_bench")

HIP_SOURCES = ["tce_capi.hip", "w4a16_gemv.hip", "w4a16_gemv_i8.hip", "w4a16_gemv_i8_token.hip", "w4a16_gemv_stream.hip", "w4a16_gemv_ovl.hip", "w4a16_gemm.hip", "w4a16_gemm_dma.hip", "w4a16_gemm_pk.hip", "w4a16_skinny.hip", "w4a16_awq.hip", "w8a8_gemm.hip", "w8a8_lnq_fused.hip", "glue.hip", "attention_ops.hip", "attention_fast.hip", "attention_prefill.hip", "opt_attention.hip", "comm.hip"]
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
    "-ffp-contract=off",  # the int8 epilogue and the fp16-accumulate entry point need every rounding
    "-Wall", "-Wno-unused-function",
]
# per-file additions (experiments land here first)
EXTRA_FLAGS: dict[str, list[str]] = {}  # (-fno-slp-vectorize on the GEMM: no packed fp32 fma beside the MFMAs -- measured neutral)
# Kernels that must not spill a vector register: lnq_w8a8_wide_kernel keeps 16 requested weight pieces per lane in flight through its sums; ONE spilled register
# makes the compiler wait for all of them at the spill (measured: the OPT-6.7B layer 94 -> 103 us, found only by accident).  The build fails instead.
NO_VGPR_SPILL: dict[str, list[str]] = {"w8a8_lnq_fused.hip": ["lnq_w8a8_wide_kernel"],
                                        "w4a16_gemv_stream.hip": ["w4a16_gemv_token_kernel"],
                                        "w4a16_gemv_i8_token.hip": ["w4a16_gemv_i8_token_kernelILb0E"],  # (ILb1E: the lab build's instantiation with wall-clock stamps)
                                        # round 5: EVERY instantiation of the decode kernel, the general-zero-point forms included (round 4 let four of them spill 3-19 registers: each
                                        # spilled scale / zero-point load became load -> wait -> scratch store, i.e. the wave's requests went out one at a time)
                                        "w4a16_gemv_i8.hip": ["w4a16_gemv_i8_kernel"],
                                        # the 256-row prefill GEMM lands asm loads in ordinary variables that become valid at a counted wait: a spill of one of them in
                                        # between would store stale data (the product instantiation only; its timing-experiment variants may spill)
                                        "w4a16_gemm_pk.hip": ["w4a16_gemm_pk256_kernelILi7ELi0E", "w4a16_gemm_pk256x2_kernelILi7ELi0E", "w4a16_gemm_pkw_kernelILi7ELi0E", "w4a16_gemm_pkwx2_kernelILi7ELi0E", "w4a16_gemm_pkw3_kernelILi7ELi0E", "w4a16_gemm_pkw3x2_kernelILi7ELi0E", "w4a16_gemm_pkw512_kernelILi7ELi0E"]}  # (round 3: four forms spilled 3-34 registers under __launch_bounds__(1024))


def _lint(src: str, obj: str) -> list[str]:
    """ISA lint of one freshly compiled object (isa_lint.py: the rules measured on MI355X that the toolchain does not know).  A violating object is deleted and the build
    fails: the rule is enforced on every kernel of every translation unit, not sampled by a soak test."""
    from . import isa_lint
    names, viol = isa_lint.lint_object(obj)
    if viol:
        msg = isa_lint.format_violations(obj, viol)
        os.remove(obj)
        raise RuntimeError(msg + f"\n(change {src} until hipcc stops emitting the form: see isa_lint.py)")
    return names


def _check_spills(src: str, stderr_text: str) -> None:
    name = None
    for line in stderr_text.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif "VGPRs Spill:" in line and name:
            n = int(line.split("VGPRs Spill:")[1].split()[0])
            if n > 0 and any(k in name for k in NO_VGPR_SPILL.get(src, [])):
                raise RuntimeError(f"{src}: kernel {name} spills {n} vector registers (see NO_VGPR_SPILL in build.py)")


def write_kernel_list(objs: list[str], lib_path: str) -> str:
    """<library>.kernels.txt: every device kernel of the library, demangled, one per line (what tests/test_boundary.py holds the product library to: no diagnostic
    instantiation in it), with the library's size in the first line."""
    from . import isa_lint
    names = []
    for o in objs:
        ks = isa_lint.kernels(isa_lint.disassemble(o))
        names += [f"{os.path.basename(o)[:-2]}: {n}" for n in isa_lint.demangle(list(ks.keys()))]
    path = lib_path + ".kernels.txt"
    with open(path, "w") as f:
        f.write(f"# {os.path.basename(lib_path)}: {os.path.getsize(lib_path)} bytes, {len(names)} kernels\n")
        f.write("\n".join(sorted(names)) + "\n")
    return path


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libtce_hip.so cannot be built (there is no CPU fallback)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, lab: bool = False) -> str:
    """lab = False: libtce_hip.so, the product -- the kernels the dispatcher (and the forced-form settings the parity tests use) can reach.
    lab = True: libtce_hip_lab.so from the same sources with -DTCE_LAB: additionally the diagnostic instantiations (parts of a loop switched off, stream-only /
    timestamp / arithmetic-only forms, the token kernel's stamps) that scripts/ use for timelines and ablations; objects under lib/lab/.  TCE_LIB_PATH selects it."""
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "lab") if lab else LIB_DIR
    os.makedirs(obj_dir, exist_ok=True)
    lib_path = LAB_LIB_PATH if lab else LIB_PATH
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(REPO_DIR, "include", "tce_matmul.h"), os.path.join(REPO_DIR, "include", "tce_tuning.h")]
    objs, jobs = [], []
    hipcc = _hipcc()
    for src in HIP_SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(op, [sp] + headers):
            check = ["-Rpass-analysis=kernel-resource-usage"] if src in NO_VGPR_SPILL else []
            jobs.append([hipcc, *HIPCC_FLAGS, *(["-DTCE_LAB"] if lab else []), *EXTRA_FLAGS.get(src, []), *check, "-I", os.path.join(REPO_DIR, "include"), "-I", CSRC, "-c", sp, "-o", op])
        objs.append(op)
    if jobs:  # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            src = os.path.basename(cmd[-3])
            if src not in NO_VGPR_SPILL:
                subprocess.check_call(cmd)
                _lint(src, cmd[-1])
                return
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            rest, in_remark = [], False  # the compiler's other diagnostics pass through; the remarks and their source excerpts do not
            for ln in r.stderr.splitlines():
                if "kernel-resource-usage" in ln:
                    in_remark = True
                elif in_remark and ln.lstrip()[:1].isdigit() is False and ln.strip().startswith("|") or in_remark and "|" in ln[:8]:
                    pass
                else:
                    in_remark = False
                    rest.append(ln)
            if rest:
                print("\n".join(rest), file=sys.stderr, flush=True)
            if r.returncode != 0:
                raise subprocess.CalledProcessError(r.returncode, cmd)
            try:
                _check_spills(src, r.stderr)
            except RuntimeError:
                os.remove(cmd[-1])  # (the object must not count as built)
                raise
            _lint(src, cmd[-1])

        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(run, jobs))
    if force or _stale(lib_path, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        write_kernel_list(objs, lib_path)
    if lab:
        return lib_path
    # test infrastructure (tests/, scripts/probes/ only): a launch that fills every CU's registers and LDS with NaNs (csrc/testkit_poison.hip)
    tk_src = os.path.join(CSRC, "testkit_poison.hip")
    if force or _stale(TESTKIT_LIB_PATH, [tk_src]):
        cmd = [hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", tk_src, "-o", TESTKIT_LIB_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


def build_adapter(force: bool = False, verbose: bool = False) -> str:
    """libtce_matmul_operator.so (the C++ matmul::MatmulOperator members, plain host code) + adapter_selftest."""
    build(force=False, verbose=verbose)
    adir = os.path.join(PKG_DIR, "adapter")
    src = os.path.join(adir, "matmul_operator_hip.cc")
    hdrs = [os.path.join(adir, "tce_matmul_operator.h"), os.path.join(REPO_DIR, "include", "tce_matmul.h")]
    inc = ["-I", os.path.join(REPO_DIR, "include"), "-I", adir]
    rpath = "-Wl,-rpath,$ORIGIN"
    if force or _stale(ADAPTER_LIB_PATH, [src, LIB_PATH] + hdrs):
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", *inc, src, "-o", ADAPTER_LIB_PATH,
               "-L", LIB_DIR, "-ltce_hip", rpath]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    test_src = os.path.join(adir, "adapter_selftest.cc")
    if force or _stale(ADAPTER_TEST_PATH, [test_src, ADAPTER_LIB_PATH] + hdrs):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", *inc, test_src, "-o", ADAPTER_TEST_PATH, "-L", LIB_DIR,
               "-ltce_matmul_operator", "-ltce_hip", rpath]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # the drop-in path on the clock (adapter_bench.cc): a measurement tool -- links the HIP runtime directly for events and graph capture
    bench_src = os.path.join(adir, "adapter_bench.cc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if force or _stale(ADAPTER_BENCH_PATH, [bench_src, ADAPTER_LIB_PATH] + hdrs):
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wno-unused-result", *inc, "-I", os.path.join(rocm, "include"), bench_src, "-o", ADAPTER_BENCH_PATH, "-L", LIB_DIR,
               "-ltce_matmul_operator", "-ltce_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64", rpath, "-Wl,-rpath," + os.path.join(rocm, "lib")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return ADAPTER_LIB_PATH


if __name__ == "__main__":
    if "--lab" in sys.argv:  # the diagnostics build only (scripts/: timelines, ablations); the product is what every other invocation builds
        print(build(force="--force" in sys.argv, verbose=True, lab=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_adapter(force="--force" in sys.argv, verbose=True))
    for p_ in (LIB_PATH, LAB_LIB_PATH):
        if os.path.exists(p_ + ".kernels.txt"):
            print(open(p_ + ".kernels.txt").readline().strip())
