#!/bin/bash
# ghost ring piece-major at s>=2 (no bank conflicts among a ghost row's lanes): parity, then A/B against the previous library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03y
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_parity.py tests/test_gpu_score_only.py tests/test_gpu_lean_trace.py -x -q -m gpu > gpurun_out/r03y/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r03y/tests.log
[ $rc -eq 0 ] || exit $rc
{
for lib in "" exp_libs/prev.so; do
  echo "== lib ${lib:-product (piece-major ring)}"
  export BIALIGN_LIB_OVERRIDE=${lib:+$PWD/$lib}
  CFG4_RUNS=4 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -2
  CFG4_PAIRS=512 CFG4_LEN=512 CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -1
  CFG4_PAIRS=512 CFG4_LEN=512 CFG4_S=3 CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -1
  CFG4_PAIRS=128 CFG4_LEN=1024 CFG4_S=3 CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -1
  CFG4_PAIRS=64 CFG4_LEN=400 CFG4_S=4 CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -1
done
} 2>&1 | tee gpurun_out/r03y/ab.log
