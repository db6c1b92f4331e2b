"""rdst_amd — MI355X (gfx950) implementation of the RDST data-parallel hot path.

The product path is GPU-only: modules are constructed on CPU like any nn.Module (the reference's
trainer builds the net and then calls ``.to(device)``), but their forward runs hand-written HIP
kernels from ``librdst_hip.so`` through the C ABI in ``include/rdst_hip.h`` and raises if the
library is missing or the input is not on a GPU.  There is no CPU fallback here; the CPU oracle
used by the tests lives in ``oracle/`` and is never imported by this package.
"""
__version__ = "0.1.0"
