#!/bin/bash
cd $GRAFT_REPO_ROOT
T="tests/test_gpu_dense_mu2.py::test_dense_full_layers_vs_oracle"
echo "== default"; timeout -k 10 300 python -m pytest "$T" -q -m gpu 2>&1 | tail -3
echo "== PACK=0"; BIALIGN_PACK=0 timeout -k 10 300 python -m pytest "$T" -q -m gpu 2>&1 | tail -3
echo "== pre-diet lib, OPT2=0"; BIALIGN_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build_exp/opt2_0.so timeout -k 10 300 python -m pytest "$T" -q -m gpu 2>&1 | tail -3
