"""tinychatengine_amd -- MI355X (gfx950) implementation of TinyChatEngine's quantized-matmul hot path.

Only what the path needs lives here:

* ``csrc/``     hand-written HIP kernels + the C ABI (``include/tce_matmul.h``) -> ``lib/libtce_hip.so``
* ``adapter/``  C++ ``matmul::MatmulOperator`` methods (the reference's link-time plugin boundary) on top of the C ABI
* ``capi``      ctypes binding of the C ABI (device pointers in, nothing else)
* ``matmul``    Python mirror of the reference operator interface (``matmul_params`` / ``MatmulOperator``)
* ``quantize``  the weight formats the path consumes (q4_6, q4_5) -- llm/tools/quantize_methods.py
* ``linear``    the L2 callers (``Linear_half_int4`` ...) and their column-sharded multi-GPU form

There is no CPU fallback anywhere in this package: if ``libtce_hip.so`` is missing, importing ``capi`` raises.
"""
__version__ = "0.1.0"
