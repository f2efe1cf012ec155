"""The diagnostics build: ``libtce_hip_lab.so`` = the library's sources compiled with ``-DTCE_LAB`` (``python -m tinychatengine_amd.build --lab``).

It holds, on top of the product's kernels, the instantiations that exist to be measured, not used: parts of the prefill GEMM's loop switched off (outputs meaningless),
the decode kernels' stream-only / timestamp / arithmetic-only forms, the token kernel with wall-clock stamps.  The product library refuses the settings that select them
(tests/test_boundary.py).  Scripts that need them call ``use_lab()`` BEFORE they import ``tinychatengine_amd.capi`` (which reads TCE_LIB_PATH once, at import)."""
import os
import sys


def use_lab(verbose: bool = False) -> str:
    if "tinychatengine_amd.capi" in sys.modules:
        raise RuntimeError("use_lab() must be called before tinychatengine_amd.capi is imported")
    from . import build
    path = build.build(lab=True, verbose=verbose)
    os.environ["TCE_LIB_PATH"] = path
    return path
