#!/bin/bash
# why is 1024 x 512 slower than this morning: product against the previous commit's library, full timing record
cd $GRAFT_REPO_ROOT
for lib in "" exp_libs/prev.so; do for n in 1024 3072; do
echo "== lib ${lib:-product} pairs $n"
BIALIGN_LIB_OVERRIDE=${lib:+$PWD/$lib} AB_PAIRS=$n AB_LEN=512 timeout -k 10 200 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from bialign_amd import synth
from bialign_amd.batch import make_batch
n, ln = int(os.environ["AB_PAIRS"]), int(os.environ["AB_LEN"])
b = make_batch(synth.protein_batch(n, ln), dict(synth.PROTEIN_PARAMS))
for fo in (True, False):
    ts = []
    for _ in range(6):
        b.run(fill_only=fo); ts.append(b.timing()["fill_ms"])
    print("fill_only", fo, "fill ms", " ".join(f"{t:.2f}" for t in ts), b.timing())
b.close()
PY
done; done
