#!/bin/bash
# ghost region for packed s>=2 sweeps: parity of the packed-record suites, then one config-4 chunk and 256 pairs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03q
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/r03q/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r03q/tests.log
[ $rc -eq 0 ] || exit $rc
CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/cfg4_chunk.log
CFG4_PAIRS=256 CFG4_RUNS=2 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/cfg4_full.log
CFG4_PAIRS=512 CFG4_LEN=512 CFG4_S=3 CFG4_RUNS=2 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/s3_512.log
