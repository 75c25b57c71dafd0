#!/bin/bash
# Timing-experiment builds of the engine (results are WRONG by construction; never shipped).
# usage: tools/exp_build.sh <output .so> [DEFINE[=value] ...]     e.g.  tools/exp_build.sh /tmp/x.so BIALIGN_EXP=1
set -e
cd "$(dirname "$0")/.."
out="$1"; shift
python - "$out" "$@" <<'PY'
import sys
from bialign_amd.build import build
print(build(force=True, out=sys.argv[1], defines=sys.argv[2:]))
PY
