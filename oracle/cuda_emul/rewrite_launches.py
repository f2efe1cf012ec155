#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/cuda_emul): rewrites the CUDA launch syntax of a reference source file for the host emulation.

    rewrite_launches.py [--concurrent] [--static-shared] < reference.cu > oracle/_ref/gen/name.cc

`kernel<<<cfg...>>>(args...);` (possibly over several lines, with comments between the arguments) becomes
`tce_emul::launch(tce_emul::cfg(cfg...), [&] { kernel(args...); });` -- `launch_concurrent` with --concurrent (kernels that use
__syncthreads / warp shuffles: the threads of a block run as OS threads).  --static-shared turns a non-static `__shared__ T v;`
inside a kernel into `static T v;` (one object for the block; blocks run one after the other).  Nothing else is touched; the output
goes to oracle/_ref/ (git-ignored) and is deleted after the build."""
import re
import sys

src = sys.stdin.read()
fn = "tce_emul::launch_concurrent" if "--concurrent" in sys.argv else "tce_emul::launch"
pat = re.compile(r"([A-Za-z_][A-Za-z_0-9]*)\s*<<<(.*?)>>>\s*\((.*?)\)\s*;", re.S)
src = pat.sub(lambda m: f"{fn}(tce_emul::cfg({m.group(2)}), [&] {{ {m.group(1)}({m.group(3)}); }});", src)
if "--static-shared" in sys.argv:
    src = re.sub(r"(?m)^(\s*)__shared__ ", r"\1static ", src)
sys.stdout.write(src)
