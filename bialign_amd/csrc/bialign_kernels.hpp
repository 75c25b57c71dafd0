// bialign_kernels.hpp -- device code of the BiAlign DP engine for gfx950 (MI355X).
//
// Mapping of the 4-D lattice onto a wavefront (see DESIGN.md for the derivation)
// ------------------------------------------------------------------------------
// Lattice point q = (i,j,k,l), band coordinates a = k-i, b = l-j in [-s,s],
// W = 2s+1.  A 64-lane wave sweeps strips of rows of one pair (one wave per pair,
// or a team of waves on interleaved strips, see fill_affine_kernel).  Lane
// L = il*W + aa holds the (row, a) pair  i = strip*RR + il - 1,  a = aa - s  for
// R = 64/W lane rows, of which il = 0 is a *ghost* row (the last row of the
// previous strip, replayed from the layers already in HBM) and il = 1..R-1 are
// the RR real rows of the strip.  At step g the lane works on column
// j = (g - 2*il - aa) mod P  and computes the W lattice points b = -s..s of that
// (i,j,a): all nine affine states each.  With this skew every predecessor named
// by the reference's case generator (pyx:255-296) was computed 1, 2 or 3 steps
// earlier by lane L-1, L-W+1, L-W or by the lane itself.  Values for rows i+1
// cross lanes through a per-wave LDS exchange array (written when a point is
// done, read at the start of the next step); lane L-1's values travel by a DPP
// wave shift; values needed 2 or 3 steps later wait in registers.  Within a wave
// the LDS is in-order, so the sweep needs no barrier; waves of a team only meet
// through the layers in HBM and one progress word each.
//
// Algebra (bit-exact regrouping of the reference's 15 cases per state, SURVEY.md
// section 7 / Appendix A).  Halves: Y=(0,1) X=(1,0) M=(1,1), state = 3*hU + hV
// in the reference's layer order (pyx:61-65).
//   open(h,T) = beta if T is a gap half and h != T else 0
//   f_T(v)    = max_h( open(h,T) + v[h] )
//   H2[U][V](p) = f_V( M[(U,.)][p] )     served to group 2, offset (0,0,V)
//   H3[U][V](p) = f_U( M[(.,V)][p] )     served to group 3, offset (U,0,0)
//   G [U][V](p) = f_U( H2[.][V](p) )     served to group 1, offset (U,V)
//   M[(U,V)][q] = max( c1 + G[U][V](q-(U,V)), c2 + H2[U][V](q-(0,0,V)),
//                      c3 + H3[U][V](q-(U,0,0)) )  over the guard-valid groups
// Guard-invalid predecessors (pyx:133-141) carry the sentinel SENT, far below
// any reachable value; a result still in the sentinel window means "no valid
// case" and becomes exactly -2^30 (pyx:299-303).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef BIALIGN_EXP
// timing experiments only (tools/exp_build.sh; results are wrong by construction):
// 1 = no layer stores, 2 = stores wrap inside 1 MiB per wave, 9 = no team hand-off waits, 3 = never take the
// interior step variant, 4 = always take it, 8 = the wide-band affine sweep stamps the phases of a level (tools/wide_phases.py)
#define BIALIGN_EXP 0
#endif
#ifndef BIALIGN_PADMAX
#define BIALIGN_PADMAX 2
#endif
#ifndef BIALIGN_SLIM_DPP  // fill_affine_slim_kernel: 1 = lane L-1's values by DPP (as fill_affine_kernel), 0 = by ds_bpermute
#define BIALIGN_SLIM_DPP 1  // like the rows': 21 VALU less, 21 LDS instructions more per step -- and 4.5 % slower (48.7 vs 46.6 ms)
#endif

#ifdef BIALIGN_WPE  // experiment: cap the affine sweep's registers so that this many waves fit a SIMD
#define BIALIGN_WPE_ATTR __attribute__((amdgpu_waves_per_eu(BIALIGN_WPE, BIALIGN_WPE)))
#else
#define BIALIGN_WPE_ATTR
#endif

#include "bialign_types.hpp"
#include "bialign_feed.hpp"
#include "bialign_fill_affine.hpp"
#include "bialign_fill_slim.hpp"
#include "bialign_fill_linear.hpp"
#include "bialign_wide.hpp"
#include "bialign_traceback.hpp"
#include "bialign_dump.hpp"
