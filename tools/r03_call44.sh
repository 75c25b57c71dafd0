#!/bin/bash
# ghost rows unpacked with v_pk_sub_u16: s=3 (product), and s=2 with low-half records (BIALIGN_LOWHALF_S2=1) against offsets
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03u
timeout -k 10 600 python -m pytest tests/test_gpu_packed_records.py -x -q -m gpu > gpurun_out/r03u/tests_pksub.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r03u/tests_pksub.log
[ $rc -eq 0 ] || exit $rc
{
echo "== product, s=3"
for shape in "512 512" "128 1024" "1384 128"; do set -- $shape
  CFG4_PAIRS=$1 CFG4_LEN=$2 CFG4_S=3 CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -1
done
for lib in "" exp_libs/lh2.so; do
  echo "== lib ${lib:-product}, config-4 chunk and 512 x 512 s=2"
  BIALIGN_LIB_OVERRIDE=${lib:+$PWD/$lib} CFG4_RUNS=4 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -2
  BIALIGN_LIB_OVERRIDE=${lib:+$PWD/$lib} CFG4_PAIRS=512 CFG4_LEN=512 CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -1
done
} 2>&1 | tee gpurun_out/r03u/pksub_ab.log
BIALIGN_LIB_OVERRIDE=$PWD/exp_libs/lh2.so timeout -k 10 600 python -m pytest tests/test_gpu_packed_records.py -x -q -m gpu > gpurun_out/r03u/tests_lh2.log 2>&1; echo "lh2 tests rc=$?"; tail -2 gpurun_out/r03u/tests_lh2.log
