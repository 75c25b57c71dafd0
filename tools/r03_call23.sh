#!/bin/bash
cd $GRAFT_REPO_ROOT
for p in 32 64 128 178 256; do echo "== parts $p"; BIALIGN_WIDE_PARTS=$p timeout -k 10 200 python tools/wide_time.py 2>&1 | grep "1000x1000"; done | tee gpurun_out/r03l/wide_parts.log
