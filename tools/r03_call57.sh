#!/bin/bash
# one-layer sweep with the lane-major exchange array: parity suites, then A/B against the previous library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_score_only.py tests/test_gpu_lean_trace.py tests/test_gpu_dense_mu2.py -x -q -m gpu > gpurun_out/r03z/tests_linear.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r03z/tests_linear.log
[ $rc -eq 0 ] || exit $rc
{
for lib in "" exp_libs/prev.so; do
  echo "== lib ${lib:-product (lane-major)}"
  for shape in "8192 512 0" "4238 512 1" "1525 512 2" "778 512 3" "1059 1024 1" "8192 128 2"; do set -- $shape
    BIALIGN_LIB_OVERRIDE=${lib:+$PWD/$lib} N=$1 LEN=$2 S=$3 timeout -k 10 200 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from bialign_amd import synth
from bialign_amd.batch import make_batch
n, ln, s = int(os.environ["N"]), int(os.environ["LEN"]), int(os.environ["S"])
b = make_batch(synth.protein_batch(n, ln), dict(synth.PROTEIN_PARAMS, max_shift=s, gap_opening_cost=0, gap_cost=-200, shift_cost=-250))
ts = []
for _ in range(6):
    b.run(fill_only=True); ts.append(b.timing()["fill_ms"])
t = b.timing()
print(f"one-layer {n:5d} x {ln:4d} s={s}: fill {min(ts[2:]):7.2f} ms  {b.info['cells']/min(ts[2:])/1e6:6.1f} Gcells/s  team {t['waves_per_pair']}{'x' if t['cross_cu'] else ''}", flush=True)
b.close()
PY
  done
done
} 2>&1 | tee gpurun_out/r03z/linear_ab.log
