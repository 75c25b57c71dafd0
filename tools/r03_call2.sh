#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/valu_rate.hip && timeout -k 10 300 /tmp/valu_rate > gpurun_out/r03b/valu_rate.txt 2>&1
cat gpurun_out/r03b/valu_rate.txt
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_packed_records.py::test_dump_layers_first_fill_overflows_and_replans tests/test_gpu_packed_records.py::test_batch_of_1024_len_512_properties tests/test_gpu_dropin.py::test_full_config5_one_gpu_share tests/test_gpu_dropin.py::test_config5_all_eight_shards_on_one_gpu -x -q -m gpu > gpurun_out/r03b/new_tests.log 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/r03b/new_tests.log
AB_LEN=1024 timeout -k 10 300 python tools/time_create.py > gpurun_out/r03b/time_create.log 2>&1; tail -3 gpurun_out/r03b/time_create.log
