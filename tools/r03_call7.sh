#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
AB_LEN=512 timeout -k 10 800 python tools/slim_matrix.py 2>&1 | tee gpurun_out/r03d/slim_matrix_len512.log
