#!/bin/bash
# relaxed cross-CU rule for sparse s=1 launches: default pick against forced shapes, 150..400 pairs x len 1024, and a few len 512 / 2048
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z
{
for shape in "150 1024" "200 1024" "300 1024" "350 1024" "400 1024" "150 512" "200 512" "60 2048" "100 2048"; do set -- $shape
  for team in "" 8 4 x4 x6 x8; do
    echo -n "pairs $1 len $2 BIALIGN_TEAM=$team: "; BIALIGN_TEAM=$team AB_PAIRS=$1 AB_LEN=$2 timeout -k 10 200 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from bialign_amd import synth
from bialign_amd.batch import make_batch
n, ln = int(os.environ["AB_PAIRS"]), int(os.environ["AB_LEN"])
b = make_batch(synth.protein_batch(n, ln), dict(synth.PROTEIN_PARAMS))
ts = []
for _ in range(5):
    b.run(fill_only=True); ts.append(b.timing()["fill_ms"])
t = b.timing()
print(f"fill {min(ts[2:]):7.2f} ms  waves/pair {t['waves_per_pair']}{'x' if t['cross_cu'] else ''}", flush=True)
b.close()
PY
  done
done
} 2>&1 | tee gpurun_out/r03z/sparse_s1.log
