#!/bin/bash
# Registers, scratch and occupancy of every kernel in one translation unit of the library (serial compile, so that the
# remarks of different kernels do not interleave):   tools/kernel_regs.sh <max_shift> <kind 0=affine 1=one-layer> [-D...]
S=${1:-2}; K=${2:-0}; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DBIALIGN_TU_S=$S -DBIALIGN_TU_KIND=$K "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$(dirname "$0")/../bialign_amd/csrc/bialign_inst.hip" -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|VGPRs Spill" |
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/Function Name: _ZN7bialign/@/; s/EEvNS_11DeviceBatchE//; s/ \[[a-zA-Z/]+\]//' |
  tr '\n' ' ' | tr '@' '\n' | sed -E 's/  +/ /g'
echo
