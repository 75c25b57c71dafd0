"""``import bialignment_nonpyx`` drop-in (reference setup.py:50)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from bialign_amd.molecule_io import *  # noqa: E402,F401,F403
from bialign_amd.molecule_io import __version__, blosum62, helix_yadd_a, helix_yadd_b  # noqa: E402,F401
