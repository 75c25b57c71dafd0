"""bialign_amd -- MI355X (gfx950) engine for the BiAlign hot path: the 4-D, shift-banded
max-plus DP fill + traceback, behind the reference's Python API.

    bialign_amd.bialignment   BiAligner and helpers (drop-in for the reference's `bialignment` module)
    bialign_amd.cli           the `bialign.py` command line
    bialign_amd.batch         many pairs per launch; bialign_amd.distributed: one process per GPU
    bialign_amd.engine        thin ctypes face of the C ABI (include/bialign.h)
    bialign_amd.build         hipcc build of libbialign_hip.so (no CPU fallback exists)

Importing the package is light; the HIP library is loaded by `bialign_amd.engine` / `_lib`.
"""
__version__ = "0.1.0"          # this package
REFERENCE_API_VERSION = "0.3"  # BiAlign version whose API it mirrors (reference bialignment_nonpyx.py:3)
