#!/bin/bash
# team policy after the rounds rules: picks and fill times over pair counts (s=1 affine), then the policy-sensitive suites
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
{
for len in 512 1024; do for n in 117 256 300 384 470 512 768 1024 1280 2048 3072 4096; do
  [ $len = 1024 ] && [ $n -gt 2048 ] && continue
  echo -n "len $len: "; AB_PAIRS=$n AB_LEN=$len timeout -k 10 200 python - <<'PY'
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
} 2>&1 | tee gpurun_out/r03w/auto_picks3.log
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_score_only.py tests/test_gpu_xcu_residency.py tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r03w/tests_policy.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r03w/tests_policy.log
