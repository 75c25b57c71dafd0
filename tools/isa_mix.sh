#!/bin/bash
# Instruction mix of one affine fill kernel's ISA.  usage: tools/isa_mix.sh <S> <out dir> [DEFINE ...]
# Writes <out dir>/s<S>.s; prints register use and the op histogram of kernel <S, beta<=0, TW=2>.
set -e
cd "$(dirname "$0")/.."
S=$1; OUT=$2; shift 2
mkdir -p "$OUT"
DEFS=""; for d in "$@"; do DEFS="$DEFS -D$d"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S $DEFS -DBIALIGN_TU_S=$S -DBIALIGN_TU_KIND=0 \
    bialign_amd/csrc/bialign_inst.hip -o "$OUT/s$S.s"
K="${KERNEL:-_ZN7bialign18fill_affine_kernelILi${S}ELb1ELi2ELb0ELb0ELb0ELb0ELb0EEEvNS_11DeviceBatchE}"
awk -v k="$K:" '$1==k{p=1} p{print} /^\.Lfunc_end/{if(p)exit}' "$OUT/s$S.s" > "$OUT/k$S.s"
grep -E "^\s+\.(sgpr|vgpr)_count|NumVgprs|ScratchSize|Occupancy" "$OUT/s$S.s" | head -0
grep -A40 "^\s*.amdhsa_kernel $K" "$OUT/s$S.s" | grep -E "next_free_vgpr|next_free_sgpr|accum_offset|private_segment_fixed" 
wc -l "$OUT/k$S.s"
