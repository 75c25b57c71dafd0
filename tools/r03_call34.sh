#!/bin/bash
# ghost ring (Pack<S>::GSIDE): parity of the packed-record and team suites, then config 4
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03q
timeout -k 10 1000 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_parity.py tests/test_gpu_xcu_residency.py -x -q -m gpu > gpurun_out/r03q/tests2.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r03q/tests2.log
[ $rc -eq 0 ] || exit $rc
CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/cfg4_chunk_ring.log
CFG4_PAIRS=256 CFG4_RUNS=2 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/cfg4_full_ring.log
CFG4_PAIRS=512 CFG4_LEN=512 CFG4_S=3 CFG4_RUNS=2 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/s3_512_ring.log
