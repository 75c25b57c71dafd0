#!/bin/bash
# rounds-aware choice between the slim and the two-wave sweep: what the default policy picks now, per pair count (len 512 and 1024)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
{
for len in 512 1024; do for n in 117 300 470 768 1280 256 512; do
  [ $len = 512 ] && [ $n = 117 ] && continue
  echo -n "len $len: "; AB_PAIRS=$n AB_LEN=$len BIALIGN_TEAM= timeout -k 10 200 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from bialign_amd import synth
from bialign_amd.batch import make_batch
n, ln = int(os.environ["AB_PAIRS"]), int(os.environ["AB_LEN"])
b = make_batch(synth.protein_batch(n, ln), dict(synth.PROTEIN_PARAMS))
ts = []
for _ in range(6):
    b.run(fill_only=True); ts.append(b.timing()["fill_ms"])
t = b.timing()
print(f"pairs {n:5d}: fill {min(ts[2:]):7.2f} ms = {min(ts[2:])*1024/n:6.2f} per 1024 pairs  waves/pair {t['waves_per_pair']}{'x' if t['cross_cu'] else ''} chunks {b.info['nchunks']}", flush=True)
b.close()
PY
done; done
} 2>&1 | tee gpurun_out/r03w/auto_picks2.log
