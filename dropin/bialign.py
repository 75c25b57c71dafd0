#!/usr/bin/env python3
"""``bialign.py`` command line drop-in (reference setup.py:51); see bialign_amd/cli.py."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from bialign_amd.cli import VERSION_STRING, add_bialign_parameters, bialign, main  # noqa: E402,F401

if __name__ == "__main__":
    main()
