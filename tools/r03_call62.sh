#!/bin/bash
# cross-CU teams: does it pay to stay at one wave per SIMD (<= 1024 one-wave workgroups) rather than between one and two?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03end
{
for shape in "117 1024 x8 x9 x11 x13" "200 512 4 x4 x5 x6" "256 512 4 x3 x4 x5 x6" "170 1024 8 x5 x6 x8 x10" "60 2048 x13 x17 x21 x25" "340 512 4 x3 x4 x5 x6"; do set -- $shape
  n=$1; len=$2; shift 2
  for team in "" "$@"; do
    echo -n "pairs $n len $len BIALIGN_TEAM=$team: "; BIALIGN_TEAM=$team AB_PAIRS=$n AB_LEN=$len timeout -k 10 200 python - <<'PY'
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
print(f"fill {min(ts[2:]):7.2f} ms  waves/pair {t['waves_per_pair']}{'x' if t['cross_cu'] else ''}  waves {n*t['waves_per_pair']}", flush=True)
b.close()
PY
  done
done
} 2>&1 | tee gpurun_out/r03end/xcu_wave_counts.log
