"""``import bialignment`` drop-in: same module name and public names as the
reference's Cython extension (reference setup.py:11-18), served by the MI355X
engine in ``bialign_amd``.  Put this directory on PYTHONPATH in place of the
reference's ``src/``."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from bialign_amd.bialignment import *  # noqa: E402,F401,F403
from bialign_amd.bialignment import (AffineDPMatrices, BiAligner, SparseMatrix4D, __version__,  # noqa: E402,F401
                                     affine_score, argmin, guard_case)
