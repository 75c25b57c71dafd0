#!/bin/bash
# fill_affine_kernel's exchange array lane-major (16-byte LDS accesses) against row-major (rounds 1-3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03w/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r03w/tests.log
[ $rc -eq 0 ] || exit $rc
{
for lib in "" exp_libs/rowmajor.so; do
  echo "== lib ${lib:-product (lane-major)}"
  export BIALIGN_LIB_OVERRIDE=${lib:+$PWD/$lib}
  CFG4_RUNS=4 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -2
  CFG4_PAIRS=512 CFG4_LEN=512 CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -1
  CFG4_PAIRS=512 CFG4_LEN=512 CFG4_S=3 CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -1
  BIALIGN_SLIM=0 AB_PAIRS=1024 AB_LEN=1024 AB_STEPS=6 timeout -k 10 300 python tools/ab_fill.py 2>&1 | tail -1
  S0_S=0 timeout -k 10 300 python tools/s0_time.py 2>&1 | sed -n 5,6p
done
} 2>&1 | tee gpurun_out/r03w/ab.log
