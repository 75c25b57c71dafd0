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
// interior step variant, 4 = always take it
#define BIALIGN_EXP 0
#endif
#ifndef BIALIGN_OPT  // A/B switches of equivalent step code: 1 = fused DPP-min exchange, 2 = ghost rows by branch
#define BIALIGN_OPT 3
#endif

namespace bialign {

constexpr int32_t NEG = -(1 << 30);                 // reference -infinity
constexpr int32_t SENT = -(1 << 30) - (1 << 29);    // "guard failed" marker
constexpr int32_t THRESH = -(1 << 30) - (1 << 28);  // below: no valid case
constexpr int NCOL = 65;                            // 64 lanes + 1 sentinel column
typedef int v4i __attribute__((ext_vector_type(4)));

struct PairDesc {
  int32_t n, m;       // lengths of A, B
  int32_t NS, P, G;   // strips, column period, total steps of the sweep
  int32_t trace_cap;  // 2(n+m)+2
  int64_t seq_a, seq_b;   // offsets into the code arrays
  int64_t layer_off;      // dword offset of this pair's records in the chunk buffer
  int64_t trace_off;      // byte offset in the trace buffer
  int64_t mu2_off;        // dense-mu2 mode: start of this pair's n x m table
  int64_t scratch_off;    // lean traceback: dword offset of this pair's one-strip scratch records
};

// Lean traceback: where a pair's walk stands between two strips.
struct TraceState {
  int32_t i, j, k, l;     // current lattice point
  int32_t st, cur;        // its state and layer value
  int32_t d0, d1;         // running shifts (pyx:541-545)
  int32_t len;            // columns emitted so far (end -> start order)
  int32_t strip;          // strip the current point lies in
  int32_t started, done;  // 0/1
};

struct DeviceBatch {
  const PairDesc* pairs;
  const int32_t* order;  // launch order (block -> pair id)
  const uint8_t *seq_a, *cls_a, *seq_b, *cls_b;
  const int32_t *s1, *s2;
  int32_t k1, k2;
  int32_t beta, gamma, delta;
  int32_t* layers;      // chunk buffer
  int32_t* scores;      // [npairs]
  uint8_t* trace;       // trace buffer
  int32_t* trace_len;   // [npairs]
  int32_t* complete;    // [npairs]
  int32_t* errflag;     // [1] sticky device-side error (team protocol timeout)
  const int32_t* mu2_dense;  // dense-mu2 mode: mu2(k,l) tables (else nullptr: LOOKUP form)
  int32_t* prog;        // cross-CU teams: [pairs in launch][64] progress words, zeroed per launch
  int32_t team;         // cross-CU teams: workgroups (= waves) per pair
  int32_t* scratch;     // lean traceback: full records of resw_k strips per pair
  TraceState* tstate;   // lean traceback: [npairs]
  int32_t resw_k;       // lean traceback: strips re-swept (in parallel) and walked per round
};

template <int S>
struct Geo {
  static constexpr int W = 2 * S + 1;
  static constexpr int R = 64 / W;       // lane rows per wave (incl. ghost row)
  static constexpr int RR = R - 1;       // real lattice rows per strip
  static constexpr int LIVE = R * W;     // lanes in use
  static constexpr int MAXOFF = 2 * (R - 1) + (W - 1);
  static constexpr int PADB = S + 1;     // guard bytes around B codes in LDS
};

// Record geometry: one record per step, ND dwords for each of the SL = RR*W real lanes
// (ghost and idle lanes own no storage), as NCH4 chunks [chunk][slot][4 dwords] followed by a
// [slot][TAIL] tail, everything packed: a wave-wide store instruction writes one contiguous
// run of SL*16 bytes and consecutive instructions / steps continue where the last one ended,
// so every byte of a pair's region is written and L2 assembles full lines.
//   LEAN records (score-only batches): nobody will trace back, so a step keeps only what the
// next strip's ghost row replays -- the bottom real row, W slots -- in the same chunk layout.
template <int S, int NL, bool LEAN = false>
struct Rec {
  static constexpr int W = 2 * S + 1;
  static constexpr int SL = LEAN ? W : (64 / W - 1) * W;  // storage slots = real lanes (bottom row only if LEAN)
  static constexpr int ND = NL * W;
  static constexpr int NCH4 = ND / 4;
  static constexpr int TAIL = ND % 4;
  static constexpr int CH = SL * 4;            // dwords per chunk
  static constexpr int RECDW = LEAN ? (SL * ND + 3) / 4 * 4 : SL * ND;  // 16-byte pieces stay aligned
  __host__ __device__ static inline int64_t dword(int64_t g, int slot, int d) {
    return d < 4 * NCH4 ? g * RECDW + (d >> 2) * CH + slot * 4 + (d & 3)
                        : g * RECDW + NCH4 * CH + slot * TAIL + (d - 4 * NCH4);
  }
};

// dword index of layer value (state st) of lattice point (i, j, aa, bb).
template <int S, int NL>
__host__ __device__ inline int64_t cell_dword(const PairDesc& pd, int i, int j, int aa, int bb,
                                              int st) {
  constexpr int W = 2 * S + 1, RR = Geo<S>::RR;
  const int strip = i / RR, il = i - strip * RR + 1;
  const int64_t g = (int64_t)strip * pd.P + j + 2 * il + aa;
  return pd.layer_off + Rec<S, NL>::dword(g, (il - 1) * W + aa, bb * NL + st);
}


// ---------------------------------------------------------------------------
// Ghost-row feed.  The first lane row of a strip replays the last row of the
// previous strip, whose layers already sit in HBM (they are output anyway), so
// strips exchange nothing but what the sweep writes regardless.  Fetching them
// step by step would put an HBM round trip -- and, through the in-order vmcnt
// counter, the completion of every earlier layer store -- on each step's
// critical path.  Instead, once per BLK steps the whole wave moves the next
// block's 16-byte pieces HBM -> LDS with LDS-DMA (global_load_lds_dwordx4: per-lane
// source address, lane-linear destination, no VGPRs), one block ahead of use.
// The DMA is issued from inline asm so that hipcc's waitcnt pass never sees a
// pending load (it would drain the store queue with vmcnt(0) every step); the
// one counted wait per block is written by hand.  The ghost of step g replays
// record g - GOFF for every ghost lane alike, so the feed needs no lane state.
// ---------------------------------------------------------------------------
template <int S, int NL, bool LEAN = false>
struct GhostFeed {
  using R_ = Rec<S, NL, LEAN>;
  static constexpr int W = 2 * S + 1, R = 64 / W;
  static constexpr int NP = R_::NCH4 + (R_::TAIL ? 1 : 0);  // 16-byte pieces per (step, a)
#ifdef BIALIGN_BLK_OVERRIDE
  static constexpr int BLK = BIALIGN_BLK_OVERRIDE;
#else
  static constexpr int BLK = S <= 1 ? 8 : 4;  // steps per prefetch block (the ring is 2*BLK*W*NP*16 bytes of LDS)
#endif
  static constexpr int NPIECE = BLK * W * NP;
  static constexpr int ROUNDS = (NPIECE + 63) / 64;
  static constexpr int SLOTS = ROUNDS * 64;                  // pieces per ring half (lane-linear)
  static constexpr int RING_DW = 2 * SLOTS * 4;              // two halves, dwords
  static constexpr int MIN_GOFF = 2 * BLK + 8;               // records must be this old when read

  // DMA the pieces of ghost steps [h0, h0+BLK) of this wave's sweep into the ring half
  // at LDS byte address lds_base.  A ghost lane (0,aa) at local step h sits in local
  // strip q = floor((h-aa)/P) and replays record  h + (q(T-1)+w)P - GOFF  (T waves per
  // pair, this one sweeps strips w, w+T, ...; T=1,w=0 gives h - GOFF).  blk_q / blk_rem
  // = h0 div / mod P, kept incrementally by the caller.
  __device__ static __forceinline__ void issue(const int32_t* lay, int h0, int blk_q, int blk_rem,
                                               int P, int T, int w, int GOFF, int rec_last, int lane,
                                               uint32_t lds_base) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int q = min(r * 64 + lane, NPIECE - 1);
      const int t = q / (W * NP), rem = q - t * (W * NP);
      const int aa = rem / NP, c = rem - aa * NP;
      const int xr = blk_rem + t - aa;
      const int ql = blk_q + (xr >= P ? 1 : 0) - (xr < 0 ? 1 : 0);
      const int rec = min(max(h0 + t - GOFF + (ql * (T - 1) + w) * P, 0), rec_last);
      const int sl = LEAN ? aa : (R - 2) * W + aa;  // storage slot of the bottom real row
      const int32_t* p = lay + (int64_t)rec * R_::RECDW +
                         (c < R_::NCH4 ? c * R_::CH + sl * 4 : R_::NCH4 * R_::CH + sl * R_::TAIL);
      const uint32_t dst = lds_base + r * 1024;  // wave-uniform; lane l lands at dst + 16*l
      uint32_t keep;
      // sc1: served by L2, never by this CU's L1 (the records may come from the partner wave)
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
          "global_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(p), "s"(dst)
          : "memory");
    }
  }
  // Store instructions a storing step issues at least (chunks + tail), i.e. vector-memory
  // operations younger than the block's DMAs that each such step adds.
  static constexpr int STORES_PER_STEP = R_::NCH4 + (R_::TAIL ? 1 : 0);

  // Retire the DMAs of the block about to be consumed.  vmcnt retires in order, so waiting
  // until at most N operations are outstanding retires everything older than the N youngest:
  // the wait is correct iff MORE than N vector-memory operations were issued after the DMAs.
  // `younger` is the wave's own count of those (the stores of the block's steps; idle steps
  // and short records issue none), so the deepest wait it justifies is picked here -- the
  // store queue is never drained further than needed, and never less.  Afterwards every
  // store older than the block just finished is acknowledged too.
  __device__ static __forceinline__ void wait_block(int younger) {
    if (younger > 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    else if (younger > 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (younger > 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger > 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (younger > 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (younger > 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  __device__ static __forceinline__ void fetch(int (&out)[R_::ND], const v4i* half, int t, int aa) {
    const v4i* src = half + (t * W + aa) * NP;
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      const v4i v = src[c];
      if (4 * c + 0 < R_::ND) out[4 * c + 0 < R_::ND ? 4 * c + 0 : 0] = v.x;
      if (4 * c + 1 < R_::ND) out[4 * c + 1 < R_::ND ? 4 * c + 1 : 0] = v.y;
      if (4 * c + 2 < R_::ND) out[4 * c + 2 < R_::ND ? 4 * c + 2 : 0] = v.z;
      if (4 * c + 3 < R_::ND) out[4 * c + 3 < R_::ND ? 4 * c + 3 : 0] = v.w;
    }
  }
};

// ---------------------------------------------------------------------------
// Dense-mu2 feed (structure similarities that are not a
// small class table, e.g. from predicted base-pair probabilities).  A lane keeps the W
// values mu2(k, j-s .. j+s) of its row in registers and needs ONE new value per step,
// mu2(k, v+s) for the "virtual column" v that runs through the strip change (v = j, or
// j - P once j+s has left the molecule: then the value already belongs to the next
// strip's row).  Like the ghost feed, the values come by LDS-DMA one block of steps
// ahead (global_load_lds_dword, per-lane source address, lane-linear destination).
// ---------------------------------------------------------------------------
template <int S>
struct Mu2Feed {
  static constexpr int BLK = GhostFeed<S, 9>::BLK;
  static constexpr int RING_DW = 2 * BLK * 64;
  // this lane's columns at the BLK steps of the block are jj0, jj0+1, ... (before wrapping)
  __device__ static __forceinline__ void issue(const int32_t* tab, int n, int m, int P, int jj0,
                                               int strip, int T, int w, int il, int aa,
                                               uint32_t lds_base) {
    constexpr int RR = Geo<S>::RR;
#pragma unroll
    for (int t = 0; t < BLK; ++t) {
      int jf = jj0 + t, q = strip;
      if (jf >= P) { jf -= P; ++q; }
      int l = jf + S;
      if (l > m) { l = jf - P + S; ++q; }  // already the next strip's row
      const int k = (q * T + w) * RR + il - 1 + aa - S;
      const int kc = min(max(k, 1), n), lc = min(max(l, 1), m);
      const int32_t* p = tab + (int64_t)(kc - 1) * m + (lc - 1);
      const uint32_t dst = lds_base + t * 256;  // lane l lands at dst + 4*l
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
          "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(p), "s"(dst)
          : "memory");
    }
  }
};

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
// f_T for the three target halves; arguments are the values for source half Y, X, M.
__device__ __forceinline__ int fM(int y, int x, int m) { return imax(imax(y, x), m); }
__device__ __forceinline__ int fX(int y, int x, int m, int beta) { return imax(x, beta + imax(y, m)); }
__device__ __forceinline__ int fY(int y, int x, int m, int beta) { return imax(y, beta + imax(x, m)); }

// ---------------------------------------------------------------------------
// Affine fill (pyx:474-509).  One wave per pair.
// ---------------------------------------------------------------------------
// Can target state (hU,hV) at band column bb end up with no guard-valid case for
// some band row a when all four lattice coordinates are >= 1?  (Then only the
// band decides validity and the answer is static per (state, bb).)
template <int W>
__host__ __device__ constexpr bool can_be_empty(int hU, int hV, int bb) {
  const int u0 = hU >= 1, u1 = hU != 1, v0 = hV >= 1, v1 = hV != 1;
  for (int aa = 0; aa < W; ++aa) {
    const int a1 = aa + u0 - v0, b1 = bb + u1 - v1;  // group 1, offset (U,V)
    const int a2 = aa - v0, b2 = bb - v1;            // group 2, offset (0,0,V)
    const int a3 = aa + u0, b3 = bb + u1;            // group 3, offset (U,0,0)
    const bool g1 = a1 >= 0 && a1 < W && b1 >= 0 && b1 < W;
    const bool g2 = a2 >= 0 && a2 < W && b2 >= 0 && b2 < W;
    const bool g3 = a3 >= 0 && a3 < W && b3 >= 0 && b3 < W;
    if (!g1 && !g2 && !g3) return true;
  }
  return false;
}

template <bool V>
struct BoolTag {
  static constexpr bool value = V;
};

// BETA_NONPOS: gap_opening_cost <= 0 (every practical parameter set).  Then
// open(h,T) + v[h] <= v[T] + ... lets f_X, f_Y reuse f_M's max3:
//   f_X(v) = max(v[X], beta + max3(v))     (exact for beta <= 0 only)
//
// TEAM = T waves per pair.  Wave w sweeps strips w, w+T, w+2T, ... with the same
// record layout as a single wave would produce; the only coupling is the ghost feed,
// which now replays records the previous wave of the ring (w-1, or T-1 for wave 0)
// wrote.  Each wave publishes in a progress word how many of its steps have their
// stores acknowledged; a wave checks its predecessor's word once per ghost block
// before prefetching.  Wave w>=1 therefore trails wave w-1 by lag >= 2(R-1)+2*BLK+8
// steps, and wave 0 may lead wave T-1 by at most P-lag: the host picks T only if
// T*lag fits into P with room to spare (team_shape()).
//   XCU = false: the team is one workgroup of TW waves (T = TW), progress words in LDS.
//   XCU = true : the team is A.team one-wave workgroups on any CUs / XCDs (T = A.team,
//     block b -> pair b / T, wave b % T; all co-resident by construction of the grid).
//     Per-XCD L2s are not coherent, so every layer store is write-through (sc1), the ghost
//     DMAs and the progress words are sc1 accesses too, and a word is published only
//     after the stores it covers have left the wave's vector-memory queue.
typedef int v3i __attribute__((ext_vector_type(3)));

template <bool XCU>
__device__ __forceinline__ void store_chunk(int32_t* p, v4i v) {
  if (XCU)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else
    *reinterpret_cast<v4i*>(p) = v;
}

//   RESW (lean traceback): re-sweep ONE strip of a pair with the ghost row taken from the LEAN
//   records of the strip above and the full records written to a scratch area (record = step
//   within the strip).  Workgroup b handles pair b / K, strip TraceState::strip - b % K (K =
//   A.resw_k strips per round, independent of each other, each into its own scratch slot); the
//   strip the walk stands in is swept only up to the walk's column.  One wave per strip.
template <int S, bool BETA_NONPOS, int TW, bool XCU, bool DENSE = false, bool LEAN = false, bool RESW = false>
__global__ void __launch_bounds__(64 * TW) fill_affine_kernel(const DeviceBatch A) {
  static_assert(!XCU || TW == 1, "cross-CU teams are built from one-wave workgroups");
  static_assert(!RESW || (TW == 1 && !XCU && !LEAN), "strip re-sweeps: one wave, full records");
  using G_ = Geo<S>;
  using R_ = Rec<S, 9, LEAN>;
  constexpr int W = G_::W, R = G_::R, RR = G_::RR, PADB = G_::PADB;
  constexpr int XR = 12;  // exchange rows per point that go through LDS
  constexpr int NV = XR * W, ND = R_::ND, NCH4 = R_::NCH4, TAIL = R_::TAIL, RECDW = R_::RECDW;
  extern __shared__ __align__(16) int32_t smem[];

  const int T = XCU ? A.team : TW;                       // team size
  const int slot = XCU ? blockIdx.x / T : (RESW ? blockIdx.x / A.resw_k : blockIdx.x);  // pair of this launch
  const int pid = A.order[slot];
  const PairDesc pd = A.pairs[pid];
  const int n = pd.n, m = pd.m, P = pd.P;
  int Qbase = 0, jlim = m, kk = 0;  // RESW: the strip to sweep, the last column the walk can still reach
  if (RESW) {
    const TraceState ts0 = A.tstate[pid];
    if (ts0.done) return;
    kk = blockIdx.x - slot * A.resw_k;
    Qbase = (ts0.started ? ts0.strip : pd.NS - 1) - kk;
    if (Qbase < 0) return;
    jlim = (ts0.started && kk == 0) ? ts0.j : m;
  }
  const int L = threadIdx.x & 63;
  const int wl = TW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave in workgroup
  const int w = XCU ? (int)(blockIdx.x - slot * T) : wl;                              // wave in team
  const int il = L / W, aa = L - il * W;
  const bool live = L < R * W;
  const bool ghost = (il == 0);
  const int beta = A.beta, gamma = A.gamma, delta = A.delta;
  const int k1 = A.k1, k2 = A.k2;
  const int gD = gamma + delta, gg = 2 * gamma, ggdd = 2 * gamma + 2 * delta, dd = 2 * delta;

  // ---- LDS carve-up: per wave a ghost ring and an exchange array; shared: progress
  //      words, score tables, sequence codes
  using GF = GhostFeed<S, 9, LEAN || RESW>;  // a re-sweep replays LEAN records
  using MF = Mu2Feed<S>;
  constexpr int PERW = GF::RING_DW + NV * NCOL + (DENSE ? MF::RING_DW : 0);  // dwords per wave
  v4i* ring = reinterpret_cast<v4i*>(smem + wl * GF::RING_DW);   // ghost-row ring, two halves
  int32_t* xch = smem + TW * GF::RING_DW + wl * (NV * NCOL);     // [NV][NCOL] exchange array
  int32_t* mu2ring = smem + TW * (GF::RING_DW + NV * NCOL) + wl * MF::RING_DW;  // dense-mu2 ring
  volatile int32_t* prog_lds = smem + TW * PERW;                  // [16] (in-workgroup teams)
  int32_t* s1 = smem + TW * PERW + 16;                            // [k1*k1]
  int32_t* s2 = s1 + k1 * k1;                                   // [k2*k2]
  const int npad = (n + 3) & ~3, mpad = (m + 2 * PADB + 3) & ~3;
  uint8_t* sa = reinterpret_cast<uint8_t*>(s2 + k2 * k2);  // seq A codes, [i-1]
  uint8_t* ca = sa + npad;                                  // cls A,       [k-1]
  uint8_t* sb = ca + npad;                                  // seq B codes, [j-1+PADB]
  uint8_t* cb = sb + mpad;                                  // cls B,       [l-1+PADB]

  for (int t = threadIdx.x; t < TW * PERW; t += 64 * TW) smem[t] = SENT;
  if (threadIdx.x < 16) prog_lds[threadIdx.x] = 0;
  for (int t = threadIdx.x; t < k1 * k1; t += 64 * TW) s1[t] = A.s1[t];
  for (int t = threadIdx.x; t < k2 * k2; t += 64 * TW) s2[t] = A.s2[t];
  for (int t = threadIdx.x; t < n; t += 64 * TW) {
    sa[t] = A.seq_a[pd.seq_a + t];
    ca[t] = A.cls_a[pd.seq_a + t];
  }
  for (int t = threadIdx.x; t < m + 2 * PADB; t += 64 * TW) {
    const int src = t - PADB;
    const bool ok = src >= 0 && src < m;
    sb[t] = ok ? A.seq_b[pd.seq_b + src] : 0;
    cb[t] = ok ? A.cls_b[pd.seq_b + src] : 0;
  }
  __syncthreads();

  // ---- per-lane constants
  const int colLW = (live && il >= 1) ? L - W : 64;                    // (i-1, a)
  const int colLW1 = (live && il >= 1 && aa < W - 1) ? L - W + 1 : 64; // (i-1, a+1)
  const bool a_first = (aa == 0);  // no (i, a-1) inside the band: lane L-1 is another row
  const int lane_cap = a_first ? SENT : 0x7fffffff;  // min() with it = "sentinel where a-1 leaves the band"
  const int GOFF = P - 2 * (R - 1);  // steps between a bottom row and its ghost copy
  int32_t* const lay = A.layers + pd.layer_off;                       // records the ghost feed replays
  int32_t* const sto = RESW ? A.scratch + pd.scratch_off + (int64_t)kk * (m + G_::MAXOFF + 1) * RECDW : lay;  // records this sweep writes

  const int rec_last = pd.G - 1;     // last record of this pair
  // local steps of this wave: its strips are w, w+T, ... (NSw of them)
  const int NSw = RESW ? 1 : (pd.NS - w + T - 1) / T;
  const int H = NSw > 0 ? (NSw - 1) * P + (RESW ? jlim : m) + G_::MAXOFF + 1 : 0;

  // ---- per-lane sweep state
  int jj = -(2 * il + aa);  // column of this step (< 0: not started)
  int strip = 0;            // local strip index q; lattice strip = q*T + w
  int rec_base = w * P;     // record of local step h for this lane = h + rec_base
  int i = 0, s1row = 0, s2row = 0;
  bool act_row = false;
  auto set_row = [&](int q) {
    i = (Qbase + q * T + w) * RR + il - 1;
    const int k = i + aa - S;
    act_row = live && i >= 0 && i <= n && k >= 0 && k <= n;
    s1row = (i >= 1 && i <= n) ? sa[i - 1] * k1 : 0;
    s2row = (k >= 1 && k <= n) ? ca[k - 1] * k2 : 0;
  };
  set_row(0);

  // delay lines (values read one step after production, used later)
  int dA1[2][W], dA2[2][W];  // GMM, GMX from (i-1,a): used at age 3
  int dAx[2][W];             // GXM, GXX from (i-1,a): age 2
  int dB[4][W];              // GMY, H3M[0..2] from (i-1,a+1): age 2
  int dC[2][W];              // GYM, GYX from (i,a-1): age 2
  int selfv[4][W];           // GYY, H3Y[0..2] of this lane's previous column
  int pubC[W][8];            // GYM, GYX, H2M[0..2], H2X[0..2] of the previous column, for lane L+1
  int ghostM[ND];            // ghost row: the nine layers of its W points
#pragma unroll
  for (int bb = 0; bb < W; ++bb) {
    dA1[0][bb] = dA1[1][bb] = dA2[0][bb] = dA2[1][bb] = SENT;
    dAx[0][bb] = dAx[1][bb] = dC[0][bb] = dC[1][bb] = SENT;
#pragma unroll
    for (int x = 0; x < 4; ++x) dB[x][bb] = selfv[x][bb] = SENT;
#pragma unroll
    for (int x = 0; x < 8; ++x) pubC[bb][x] = SENT;
  }
#pragma unroll
  for (int d = 0; d < ND; ++d) ghostM[d] = SENT;
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int32_t*)smem) + wl * GF::RING_DW * 4;

  // ---- team protocol (T > 1): partner progress needed before prefetching the ghost
  //      block whose last local step is h_last
  int blk_q = 0, blk_rem = 0;  // (next block start) div / mod P
  bool team_failed = false;
  int32_t* const prog_glb = XCU ? A.prog + (int64_t)slot * 64 : nullptr;
  auto prog_get = [&](int idx) __attribute__((always_inline)) -> int {
    if (XCU) return __hip_atomic_load(prog_glb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return prog_lds[idx];
  };
  auto prog_put = [&](int v) __attribute__((always_inline)) {  // lane 0 only
    if (XCU)
      __hip_atomic_store(prog_glb + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      prog_lds[w] = v;
  };
  int seen_prog = -0x40000000;  // the partner's progress as last read
  auto wait_partner = [&](int h_last) __attribute__((always_inline)) {
    if ((!XCU && TW == 1) || T == 1 || team_failed || BIALIGN_EXP == 9) return;  // 9: timing experiment, no hand-off waits
    const int src = w == 0 ? T - 1 : w - 1;
    const int need = h_last + 2 * (R - 1) + 1 - (w == 0 ? P : 0);
    // progress only grows: what was seen last time usually covers this block too, and a look at
    // the partner's word is a round trip to HBM for cross-CU teams
    if (seen_prog >= need) return;
    // bounded spin: a protocol bug must surface as an error, never as a hung GPU
    for (int spin = 0; (seen_prog = prog_get(src)) < need; ++spin) {
      if (spin > (1 << 20)) {  // ~0.5 s; then fail fast: no further waits, host reports the error
        if (L == 0) atomicExch(A.errflag, 1);
        team_failed = true;
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
  };
  const uint32_t mu2_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int32_t*)smem) +
      (TW * (GF::RING_DW + NV * NCOL) + wl * MF::RING_DW) * 4;
  const int32_t* const mu2tab = DENSE ? A.mu2_dense + pd.mu2_off : nullptr;
  int mu2w[W];  // dense-mu2 mode: mu2(k, j-s .. j+s) of this lane's row
#pragma unroll
  for (int bb = 0; bb < W; ++bb) mu2w[bb] = 0;
  auto prefetch_block = [&](int h0, int half, int jj0) __attribute__((always_inline)) {
    // h0 = first local step of the block (this lane is then at column jj0, before wrapping);
    // blk_q/blk_rem describe h0
    wait_partner(h0 + GF::BLK - 1);
    GF::issue(lay, h0 + Qbase * P, blk_q, blk_rem, P, T, w, GOFF, rec_last, L, ring_lds + half * GF::SLOTS * 16);
    if (DENSE) MF::issue(mu2tab, n, m, P, jj0, Qbase + strip, T, w, il, aa, mu2_lds + half * MF::BLK * 256);
    blk_rem += GF::BLK;
    if (blk_rem >= P) { blk_rem -= P; ++blk_q; }
  };
  // block 0 must be in the ring before the first step (waves w >= 1 start on a real ghost row)
  prefetch_block(0, 0, jj);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int vm_younger = 0;  // store instructions issued since the last block's DMAs (wave-uniform)

  // One step of the sweep.  INTERIOR steps (every lane's lattice points have all
  // four coordinates >= 1 and lie inside the molecule columns; ~90 % of the
  // steps) know that only the band can invalidate a case, so the "no valid case"
  // test runs for the few (state, b) pairs where that is possible and the
  // out-of-lattice bookkeeping disappears; boundary steps take the general form.
  auto step = [&](auto interior_tag, int g) __attribute__((always_inline)) {
    constexpr bool INTERIOR = decltype(interior_tag)::value;
    // ---- 0. ghost feed: at a block boundary retire last block's DMAs (which also tells
    //         how far this wave's own stores are acknowledged), start the next block's
    //         (before this step's stores); then pick this step's ghost layers out of the ring
    const int gt = g & (GF::BLK - 1), ghalf = (g / GF::BLK) & 1;
    if (gt == 0) {
      GF::wait_block(vm_younger);
      // all stores of steps before the block just finished are acknowledged (BLK <= 8)
      if ((XCU || TW > 1) && L == 0) prog_put(g - 8);
      prefetch_block(g + GF::BLK, ghalf ^ 1, jj + GF::BLK);
      vm_younger = 0;
    }
    GF::fetch(ghostM, ring + ghalf * GF::SLOTS, gt, aa);

    // ---- 1. exchange reads: what the three source lanes published last step.  Rows of band
    //         column r are first needed by point r-1, so they are fetched two points ahead
    //         (all of them up front for W <= 3): a sliding window keeps registers flat in W.
    int inA[W][4], inB[W][8], inC[W][8];
    auto read_rows = [&](int r) __attribute__((always_inline)) {
#pragma unroll
      for (int x = 0; x < 4; ++x) inA[r][x] = xch[(r * XR + x) * NCOL + colLW];
#pragma unroll
      for (int x = 0; x < 8; ++x) inB[r][x] = xch[(r * XR + 4 + x) * NCOL + colLW1];
      // lane L-1 = (i, a-1) hands its values over in registers: one DPP wave shift fused with a min
      // against the lane's cap (the sentinel where a-1 leaves the band, INT_MAX elsewhere).  One asm
      // block per band column: the compiler's own DPP folding gives up once the consumers are sunk
      // behind the store branches.  s_nop 1 = the two wait states a DPP read needs after a VALU write
      // of its source (the hazard recogniser does not look inside asm); lane 0 reads out of range -> 0
      // with bound_ctrl, it is an a_first lane anyway.  H2[.][M] (x = 2..4) of the last band column has
      // no consumer: offset (0,0,M) from there would leave the band.
      if (BIALIGN_OPT & 1) {
        if (r + 1 < W) {
          asm("s_nop 1\n\t"
              "v_min_i32_dpp %0, %8, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %1, %9, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %2, %10, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %3, %11, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %4, %12, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %5, %13, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %6, %14, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %7, %15, %16 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
              : "=&v"(inC[r][0]), "=&v"(inC[r][1]), "=&v"(inC[r][2]), "=&v"(inC[r][3]), "=&v"(inC[r][4]),
                "=&v"(inC[r][5]), "=&v"(inC[r][6]), "=&v"(inC[r][7])
              : "v"(pubC[r][0]), "v"(pubC[r][1]), "v"(pubC[r][2]), "v"(pubC[r][3]), "v"(pubC[r][4]),
                "v"(pubC[r][5]), "v"(pubC[r][6]), "v"(pubC[r][7]), "v"(lane_cap));
        } else {
          asm("s_nop 1\n\t"
              "v_min_i32_dpp %0, %5, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %1, %6, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %2, %7, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %3, %8, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
              "v_min_i32_dpp %4, %9, %10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
              : "=&v"(inC[r][0]), "=&v"(inC[r][1]), "=&v"(inC[r][5]), "=&v"(inC[r][6]), "=&v"(inC[r][7])
              : "v"(pubC[r][0]), "v"(pubC[r][1]), "v"(pubC[r][5]), "v"(pubC[r][6]), "v"(pubC[r][7]), "v"(lane_cap));
          inC[r][2] = inC[r][3] = inC[r][4] = SENT;
        }
      } else {
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const int nb = __builtin_amdgcn_mov_dpp(pubC[r][x], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
          inC[r][x] = a_first ? SENT : nb;
        }
      }
    };
    read_rows(0);
    if (W > 1) read_rows(W > 1 ? 1 : 0);

    // ---- 2. score inputs of this column (pyx:260-261; LOOKUP form)
    const int jc = INTERIOR ? jj : min(max(jj, 0), m + 1);
    const int mu1 = s1[s1row + sb[jc - 1 + PADB]];
    int mu2[W];
    if (DENSE) {  // slide the window, take this step's new value from the ring
#pragma unroll
      for (int bb = 0; bb + 1 < W; ++bb) mu2w[bb] = mu2w[bb + 1];
      mu2w[W - 1] = mu2ring[(ghalf * MF::BLK + gt) * 64 + L];
#pragma unroll
      for (int bb = 0; bb < W; ++bb) mu2[bb] = mu2w[bb];
    } else {
#pragma unroll
      for (int bb = 0; bb < W; ++bb) mu2[bb] = s2[s2row + cb[jc + bb]];  // l-1+PADB = jc+bb
    }

    const bool tile_act = INTERIOR ? true : (act_row && jj >= 0 && jj <= m);
    const bool is_origin = INTERIOR ? false : (tile_act && i == 0 && jj == 0 && aa == S);
    const int c3M = mu1 + dd, c_Mg = mu1 + gD;

    // Layer stores (pyx:504: M[state][idx] = ...).  Every real lane owns a 16-byte slot in
    // each chunk of its record, read back only for lattice points that exist; out-of-lattice
    // rows store don't-care values there so that no byte of a record stays unwritten (a
    // line left partly unwritten costs an HBM read-modify-write).  Only fully idle steps skip
    // the store.  Each chunk is issued as soon as its four values exist, spreading the
    // stores over the step.
    const int rec = g + rec_base;  // lanes of two strips (straddling steps) hit two records
    const bool do_store = BIALIGN_EXP != 1 && live && (LEAN ? il == R - 1 : !ghost) &&
                          (INTERIOR || __builtin_amdgcn_ballot_w64(tile_act && !ghost) != 0) &&
                          ((!XCU && TW == 1) || rec <= rec_last);
    const int slot = LEAN ? aa : L - W;  // storage slot of a real lane
    if (__builtin_amdgcn_ballot_w64(do_store) != 0) vm_younger += GF::STORES_PER_STEP;
    int32_t* const dst = BIALIGN_EXP == 2
                             ? A.layers + ((int64_t)(blockIdx.x & 255) << 18) + (int64_t)(g & 31) * RECDW
                             : sto + (int64_t)rec * RECDW;

    // ---- 3. the W lattice points of this (i, j, a)
    int outv[ND];
    int h2y[3] = {SENT, SENT, SENT};  // H2[U][Y] of point bb-1 (same step, same lane)
#pragma unroll
    for (int bb = 0; bb < W; ++bb) {
      if (bb + 2 < W) read_rows(bb + 2 < W ? bb + 2 : 0);
      const int l = jj + bb - S;
      const bool act = INTERIOR ? true : (tile_act && l >= 0 && l <= m);
      const int mu2v = mu2[bb];
      const int c_MM = mu1 + mu2v, c_gM = mu2v + gD, c2M = mu2v + dd;

      auto cases = [&](int (&Tv)[9]) __attribute__((always_inline)) {
#pragma unroll
      for (int hU = 0; hU < 3; ++hU) {
#pragma unroll
        for (int hV = 0; hV < 3; ++hV) {
          // group 1: offset (U,V)
          int gin = SENT;
          bool ok1 = true;
          if (hU == 2 && hV == 2) gin = dA2[0][bb];
          if (hU == 2 && hV == 1) { ok1 = bb + 1 < W; if (ok1) gin = dA2[1][bb + 1 < W ? bb + 1 : 0]; }
          if (hU == 2 && hV == 0) gin = dB[0][bb];
          if (hU == 1 && hV == 2) { ok1 = bb >= 1; if (ok1) gin = dAx[0][bb >= 1 ? bb - 1 : 0]; }
          if (hU == 1 && hV == 1) gin = dAx[1][bb];
          if (hU == 1 && hV == 0) { ok1 = bb >= 1; if (ok1) gin = inB[bb >= 1 ? bb - 1 : 0][1]; }
          if (hU == 0 && hV == 2) gin = dC[0][bb];
          if (hU == 0 && hV == 1) { ok1 = bb + 1 < W; if (ok1) gin = dC[1][bb + 1 < W ? bb + 1 : 0]; }
          if (hU == 0 && hV == 0) gin = selfv[0][bb];
          const int c1 = (hU == 2 && hV == 2) ? c_MM
                         : (hU == 2)          ? c_Mg
                         : (hV == 2)          ? c_gM
                         : (hU == hV)         ? gg
                                              : ggdd;
          // group 2: offset (0,0,V)
          int h2in = SENT;
          bool ok2 = true;
          if (hV == 2) { ok2 = bb >= 1; if (ok2) h2in = inC[bb >= 1 ? bb - 1 : 0][2 + hU]; }
          if (hV == 1) h2in = inC[bb][5 + hU];
          if (hV == 0) { ok2 = bb >= 1; if (ok2) h2in = h2y[hU]; }
          const int c2 = (hV == 2) ? c2M : gD;
          // group 3: offset (U,0,0)
          int h3in = SENT;
          bool ok3 = true;
          if (hU == 2) { ok3 = bb + 1 < W; if (ok3) h3in = dB[1 + hV][bb + 1 < W ? bb + 1 : 0]; }
          if (hU == 1) h3in = inB[bb][5 + hV];
          if (hU == 0) { ok3 = bb + 1 < W; if (ok3) h3in = selfv[1 + hV][bb + 1 < W ? bb + 1 : 0]; }
          const int c3 = (hU == 2) ? c3M : gD;

          int t = SENT;
          bool any = false;
          if (ok1) { t = c1 + gin; any = true; }
          if (ok2) { t = any ? imax(t, c2 + h2in) : c2 + h2in; any = true; }
          if (ok3) { t = any ? imax(t, c3 + h3in) : c3 + h3in; any = true; }
          Tv[3 * hU + hV] = t;
        }
      }

      };
      // finalise: ghost rows take the stored layers; "no valid case" -> -2^30
      // (pyx:299-303); points outside the lattice carry the sentinel.
      int M[9];
      if (INTERIOR && (BIALIGN_OPT & 2)) {
        // ghost lanes keep what the ring delivered; the others compute in place under the
        // execution mask (no per-value select)
#pragma unroll
        for (int q = 0; q < 9; ++q) M[q] = ghostM[bb * 9 + q];
        if (!ghost) {
          int Tv[9];
          cases(Tv);
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            int tv = Tv[q];
            if (can_be_empty<W>(q / 3, q % 3, bb)) tv = tv < THRESH ? NEG : tv;
            M[q] = tv;
          }
        }
      } else if (INTERIOR) {
        int Tv[9];
        cases(Tv);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          int tv = ghost ? ghostM[bb * 9 + q] : Tv[q];
          if (can_be_empty<W>(q / 3, q % 3, bb)) tv = tv < THRESH ? NEG : tv;
          M[q] = tv;
        }
      } else {
        int Tv[9];
        cases(Tv);
        const int low = act ? NEG : SENT;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const int tv = ghost ? ghostM[bb * 9 + q] : Tv[q];
          const bool bad = (tv < THRESH) | !act;
          M[q] = bad ? low : tv;
        }
        if (bb == S) M[8] = is_origin ? 0 : M[8];  // pyx:483-485
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) outv[bb * 9 + q] = M[q];
      if (LEAN && bb == S) {  // score-only: the end cell (n,m,n,m) is all the host wants (pyx:509)
        if (live && !ghost && aa == S && i == n && jj == m) {
          int best = M[0];
#pragma unroll
          for (int q = 1; q < 9; ++q) best = imax(best, M[q]);
          A.scores[pid] = best;
        }
      }
      if (do_store) {
#pragma unroll
        for (int c = 0; c < NCH4; ++c) {
          if (4 * c + 3 >= bb * 9 && 4 * c + 3 < (bb + 1) * 9) {  // chunk c completes with this point
            v4i v;
            v.x = outv[4 * c]; v.y = outv[4 * c + 1]; v.z = outv[4 * c + 2]; v.w = outv[4 * c + 3];
            store_chunk<XCU>(dst + c * R_::CH + slot * 4, v);
          }
        }
        if (bb == W - 1) {
#pragma unroll
          for (int t = 0; t < TAIL; ++t) {
            if (XCU)
              __hip_atomic_store(dst + NCH4 * R_::CH + slot * TAIL + t, outv[4 * NCH4 + t], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
            else
              dst[NCH4 * R_::CH + slot * TAIL + t] = outv[4 * NCH4 + t];
          }
        }
      }

      // derived values for the successors
      int H2[3][3], H3[3][3], Gd[3][3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        H2[u][2] = fM(M[3 * u], M[3 * u + 1], M[3 * u + 2]);
        if (BETA_NONPOS) {
          const int bm = beta + H2[u][2];
          H2[u][0] = imax(M[3 * u], bm);
          H2[u][1] = imax(M[3 * u + 1], bm);
        } else {
          H2[u][0] = fY(M[3 * u], M[3 * u + 1], M[3 * u + 2], beta);
          H2[u][1] = fX(M[3 * u], M[3 * u + 1], M[3 * u + 2], beta);
        }
      }
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        H3[2][v] = fM(M[v], M[3 + v], M[6 + v]);
        Gd[2][v] = fM(H2[0][v], H2[1][v], H2[2][v]);
        if (BETA_NONPOS) {
          const int bm3 = beta + H3[2][v], bmg = beta + Gd[2][v];
          H3[0][v] = imax(M[v], bm3);
          H3[1][v] = imax(M[3 + v], bm3);
          Gd[0][v] = imax(H2[0][v], bmg);
          Gd[1][v] = imax(H2[1][v], bmg);
        } else {
          H3[0][v] = fY(M[v], M[3 + v], M[6 + v], beta);
          H3[1][v] = fX(M[v], M[3 + v], M[6 + v], beta);
          Gd[0][v] = fY(H2[0][v], H2[1][v], H2[2][v], beta);
          Gd[1][v] = fX(H2[0][v], H2[1][v], H2[2][v], beta);
        }
      }
      // publish (all reads of this step were issued above, LDS keeps order)
      int32_t* row = xch + (bb * XR) * NCOL + L;
      row[0 * NCOL] = Gd[2][2];
      row[1 * NCOL] = Gd[2][1];
      row[2 * NCOL] = Gd[1][2];
      row[3 * NCOL] = Gd[1][1];
      row[4 * NCOL] = Gd[2][0];
      row[5 * NCOL] = Gd[1][0];
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        row[(6 + v) * NCOL] = H3[2][v];
        row[(9 + v) * NCOL] = H3[1][v];
      }
      pubC[bb][0] = Gd[0][2];
      pubC[bb][1] = Gd[0][1];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        pubC[bb][2 + u] = H2[u][2];
        pubC[bb][5 + u] = H2[u][1];
      }
      selfv[0][bb] = Gd[0][0];
#pragma unroll
      for (int v = 0; v < 3; ++v) selfv[1 + v][bb] = H3[0][v];
#pragma unroll
      for (int u = 0; u < 3; ++u) h2y[u] = H2[u][0];

      // delay lines: index bb (bb-1 for GXM/GXX) has served its last consumer of this step
      dA2[0][bb] = dA1[0][bb];
      dA2[1][bb] = dA1[1][bb];
      dA1[0][bb] = inA[bb][0];
      dA1[1][bb] = inA[bb][1];
      dB[0][bb] = inB[bb][0];
#pragma unroll
      for (int v = 0; v < 3; ++v) dB[1 + v][bb] = inB[bb][2 + v];
      dC[0][bb] = inC[bb][0];
      dC[1][bb] = inC[bb][1];
      if (bb >= 1) {
        dAx[0][bb >= 1 ? bb - 1 : 0] = inA[bb >= 1 ? bb - 1 : 0][2];
        dAx[1][bb >= 1 ? bb - 1 : 0] = inA[bb >= 1 ? bb - 1 : 0][3];
      }
    }
    dAx[0][W - 1] = inA[W - 1][2];
    dAx[1][W - 1] = inA[W - 1][3];

    // ---- 6. advance
    ++jj;
    if (!RESW && jj == P) {  // a re-sweep ends inside its one strip
      jj = 0;
      ++strip;
      rec_base += (T - 1) * P;
      set_row(strip);
    }
  };

  // Two separate loops (not one loop with a branch inside): each keeps its loop-carried
  // registers where it likes; values only move at the rare hand-overs between runs.
  auto all_interior = [&]() __attribute__((always_inline)) {
    const bool lane_interior = !live || (jj >= S + 1 && jj <= m && i >= S + 1);
    return __builtin_amdgcn_ballot_w64(lane_interior) == ~0ull;
  };
  int g = 0;  // local step of this wave
  while (g < H) {
    while (g < H && (BIALIGN_EXP == 3 || (BIALIGN_EXP != 4 && !all_interior()))) {
      step(BoolTag<false>{}, g);
      ++g;
    }
    while (g < H && BIALIGN_EXP != 3 && (BIALIGN_EXP == 4 || all_interior())) {
      step(BoolTag<true>{}, g);
      ++g;
    }
  }
  if (XCU || TW > 1) {  // everything this wave wrote is acknowledged: release the partner for good
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (L == 0) prog_put(0x7fffffff);
  }
}

// |U0-V0| + |U1-V1| of a column / state given its two halves (pyx:97, 541-545)
__device__ __forceinline__ int shift_of(int hU, int hV) {
  return hU == hV ? 0 : ((hU == 2 || hV == 2) ? 1 : 2);
}

// ---------------------------------------------------------------------------
// Affine traceback (pyx:535-586).  One wave per pair: lane c < 15 owns candidate
// c of the case generator's order (pyx:275-296) -- nine sources of the full
// offset, then three of the structure-only and three of the sequence-only offset
// -- so a column costs one HBM round trip and a handful of instructions.  The
// tie-break of pyx:554-565 ("first candidate minimising [|d0|+|d1|, |d1|]" with
// the source state added as a one-step look-ahead) is a wave-min over the packed
// key (|d0|+|d1|, |d1|, c).
// ---------------------------------------------------------------------------
// Score tables and the pair's sequence codes staged in LDS for the tracebacks (every
// column needs mu1, mu2: two dependent global loads otherwise).
struct TraceInputs {
  const int32_t *s1, *s2;
  const uint8_t *sa, *ca, *sb, *cb;
};
__device__ __forceinline__ TraceInputs stage_trace_inputs(const DeviceBatch& A, const PairDesc& pd,
                                                          int32_t* smem) {
  const int k1 = A.k1, k2 = A.k2, n = pd.n, m = pd.m;
  int32_t* s1 = smem;
  int32_t* s2 = s1 + k1 * k1;
  uint8_t* sa = reinterpret_cast<uint8_t*>(s2 + k2 * k2);
  uint8_t* ca = sa + ((n + 3) & ~3);
  uint8_t* sb = ca + ((n + 3) & ~3);
  uint8_t* cb = sb + ((m + 3) & ~3);
  for (int t = threadIdx.x; t < k1 * k1; t += 64) s1[t] = A.s1[t];
  for (int t = threadIdx.x; t < k2 * k2; t += 64) s2[t] = A.s2[t];
  for (int t = threadIdx.x; t < n; t += 64) {
    sa[t] = A.seq_a[pd.seq_a + t];
    ca[t] = A.cls_a[pd.seq_a + t];
  }
  for (int t = threadIdx.x; t < m; t += 64) {
    sb[t] = A.seq_b[pd.seq_b + t];
    cb[t] = A.cls_b[pd.seq_b + t];
  }
  __syncthreads();
  return TraceInputs{s1, s2, sa, ca, sb, cb};
}

__device__ __forceinline__ int wave_min16(int v) {  // min over lanes 0..15, valid in every lane < 16
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) v = min(v, __shfl_xor(v, d, 16));
  return v;
}

//   STRIP (lean traceback): the walk continues from the pair's
//   TraceState through ONE strip -- the one fill_affine_kernel<.., RESW> has just re-swept into
//   the scratch records -- and stops when it steps into the strip above (whose bottom row, the
//   only row of it a candidate can touch from here, is in the LEAN records) or ends.
template <int S, bool DO_TRACE, bool STRIP = false>
__global__ void __launch_bounds__(64) traceback_affine_kernel(const DeviceBatch A, int npairs) {
  const int pid = A.order[blockIdx.x];
  const PairDesc pd = A.pairs[pid];
  const int n = pd.n, m = pd.m;
  const int beta = A.beta, gamma = A.gamma, delta = A.delta;
  const int32_t* lay = A.layers;
  const int c = threadIdx.x;  // candidate lane
  constexpr int BIG = 0x7fffffff;
  constexpr int W = 2 * S + 1, RR = Geo<S>::RR;
  extern __shared__ __align__(16) int32_t smem[];

  TraceState ts{};
  if (STRIP) {
    ts = A.tstate[pid];
    if (ts.done) return;
  }
  const int Q = STRIP ? (ts.started ? ts.strip : pd.NS - 1) : 0;
  const int Qlo = STRIP ? max(Q - A.resw_k + 1, 0) : 0;  // strips Qlo..Q sit in the scratch slots Q-sp
  const int64_t sstride = (int64_t)(m + Geo<S>::MAXOFF + 1) * Rec<S, 9>::RECDW;
  // layer value (state ss) of lattice point (pi, pj, a, b)
  auto cell = [&](int pi, int pj, int a, int b, int ss) -> int {
    if (!STRIP) return lay[cell_dword<S, 9>(pd, pi, pj, a, b, ss)];
    const int sp = pi / RR, ilp = pi - sp * RR + 1;
    if (sp >= Qlo)  // inside a re-swept strip: record = step within the strip
      return A.scratch[pd.scratch_off + (Q - sp) * sstride +
                       Rec<S, 9>::dword(pj + 2 * ilp + a, (ilp - 1) * W + a, b * 9 + ss)];
    // bottom row of the strip above (ilp == RR): LEAN record of its global step
    return lay[pd.layer_off + Rec<S, 9, true>::dword((int64_t)sp * pd.P + pj + 2 * ilp + a, a, b * 9 + ss)];
  };

  int i = n, j = m, k = n, l = m, d0 = 0, d1 = 0, len = 0, complete = 0;
  int st = 0, cur = 0;
  if (!STRIP || !ts.started) {
    // pyx:573-582: best end layer, first one with the least shift
    const int endv = c < 9 ? cell(n, m, S, S, c) : -BIG;
    const int best = __builtin_amdgcn_readfirstlane(-wave_min16(-endv));
    if (c == 0) A.scores[pid] = best;
    if (!DO_TRACE) return;
    const int skey = (c < 9 && endv == best) ? (shift_of(c / 3, c % 3) << 4 | c) : BIG;
    st = __builtin_amdgcn_readfirstlane(wave_min16(skey)) & 15;
    cur = best;
  } else {
    i = ts.i; j = ts.j; k = ts.k; l = ts.l; st = ts.st; cur = ts.cur; d0 = ts.d0; d1 = ts.d1; len = ts.len;
  }
  const TraceInputs in = stage_trace_inputs(A, pd, smem);
  const uint8_t *sa = in.sa, *ca = in.ca, *sb = in.sb, *cb = in.cb;

  uint8_t* out = A.trace + pd.trace_off;
  bool finished = true;  // STRIP: false when the walk merely left this strip
  // lane-constant part of the candidate: its group and, for groups 2/3, the free half h
  const int grp = c < 9 ? 1 : (c < 12 ? 2 : 3);
  const int hfree = grp == 2 ? 2 - (c - 9) : 2 - (c - 12);  // h = M, X, Y in the generator's order
  while (true) {
    if (i == 0 && j == 0 && k == 0 && l == 0 && st == 8) { complete = 1; break; }
    if (STRIP && i < Qlo * RR) { finished = false; break; }  // above the re-swept strips: next round
    const int hU = st / 3, hV = st - 3 * hU;
    const int u0 = hU >= 1, u1 = hU != 1, v0 = hV >= 1, v1 = hV != 1;
    const int mu1 = (i >= 1 && j >= 1) ? in.s1[sa[i - 1] * A.k1 + sb[j - 1]] : 0;
    const int mu2 = (k >= 1 && l >= 1)
                        ? (A.mu2_dense ? A.mu2_dense[pd.mu2_off + (int64_t)(k - 1) * m + (l - 1)]
                                       : in.s2[ca[k - 1] * A.k2 + cb[l - 1]])
                        : 0;
    const int valU = hU == 2 ? mu1 : gamma, valV = hV == 2 ? mu2 : gamma;

    // this lane's candidate: offset, source state, score (pyx:84-131)
    const int o0 = grp == 2 ? 0 : u0, o1 = grp == 2 ? 0 : u1;
    const int o2 = grp == 3 ? 0 : v0, o3 = grp == 3 ? 0 : v1;
    const int ss = grp == 1 ? c : (grp == 2 ? 3 * hU + hfree : 3 * hfree + hV);
    const int ra = ss / 3, rb = ss - 3 * ra;
    const int openU = (hU != 2 && ra != hU) ? beta : 0, openV = (hV != 2 && rb != hV) ? beta : 0;
    const int sc = grp == 1   ? delta * shift_of(hU, hV) + valU + valV + openU + openV
                   : grp == 2 ? delta * (v0 + v1) + valV + openV
                              : delta * (u0 + u1) + valU + openU;
    const int pi = i - o0, pj = j - o1, pk = k - o2, pl = l - o3;
    const bool ok = c < 15 && pi >= 0 && pj >= 0 && pk >= 0 && pl >= 0 && abs(pk - pi) <= S &&
                    abs(pl - pj) <= S;  // pyx:133-141
    const int ld = ok ? cell(pi, pj, pk - pi + S, pl - pj + S, ss) : 0;
    // pyx:554-565: cases reproducing the cell; look-ahead adds the offset AND the source state
    const int r0 = ra >= 1, r1 = ra != 1, r2 = rb >= 1, r3 = rb != 1;
    const int t0 = d0 + (o0 - o2) + (r0 - r2), t1 = d1 + (o1 - o3) + (r1 - r3);
    const int key = (ok && ld + sc == cur) ? ((abs(t0) + abs(t1)) << 16 | abs(t1) << 8 | c) : BIG;
    const int kmin = __builtin_amdgcn_readfirstlane(wave_min16(key));
    if (kmin == BIG) break;  // pyx:570-571 -> "incomplete traceback"
    const int pick = kmin & 63;
    const int code = __builtin_amdgcn_readlane(o0 * 8 + o1 * 4 + o2 * 2 + o3, pick);
    st = __builtin_amdgcn_readlane(ss, pick);
    cur = __builtin_amdgcn_readlane(ld, pick);
    const int q0 = (code >> 3) & 1, q1 = (code >> 2) & 1, q2 = (code >> 1) & 1, q3 = code & 1;
    d0 += q0 - q2;  // pyx:566: only the offset moves the running shift
    d1 += q1 - q3;
    if (c == 0 && len < pd.trace_cap) out[len] = (uint8_t)code;
    ++len;
    i -= q0; j -= q1; k -= q2; l -= q3;
  }
  if (STRIP && !finished) {  // hand over to the next round
    if (c == 0) {
      TraceState nx;
      nx.i = i; nx.j = j; nx.k = k; nx.l = l; nx.st = st; nx.cur = cur; nx.d0 = d0; nx.d1 = d1;
      nx.len = len; nx.strip = Qlo - 1; nx.started = 1; nx.done = 0;
      A.tstate[pid] = nx;
    }
    return;
  }
  if (len > pd.trace_cap) len = pd.trace_cap;
  __builtin_amdgcn_s_waitcnt(0);  // lane 0's byte stores before the wave-wide reversal
  __syncthreads();
  for (int x = c; x < len / 2; x += 64) {  // pyx:586 reversed
    const uint8_t t = out[x];
    out[x] = out[len - 1 - x];
    out[len - 1 - x] = t;
  }
  if (c == 0) {
    A.trace_len[pid] = len;
    A.complete[pid] = complete;
    if (STRIP) {
      ts.done = 1;
      ts.started = 1;
      A.tstate[pid] = ts;
    }
  }
}

// ---------------------------------------------------------------------------
// Non-affine fill (pyx:443-471): one layer, thirteen cases (pyx:233-248).
// Same lane mapping and skew as the affine sweep.  A lane publishes only its W
// layer values per step; the three source lanes' values are read one step
// later and kept in registers for the cases that need them 2 or 3 steps later
// (age of offset o = o0 + o1 + o2).
// ---------------------------------------------------------------------------
template <int S, int TW, bool DENSE = false, bool LEAN = false, bool RESW = false>
__global__ void __launch_bounds__(64 * TW) fill_linear_kernel(const DeviceBatch A) {
  using G_ = Geo<S>;
  using R_ = Rec<S, 1, LEAN>;
  constexpr int W = G_::W, R = G_::R, RR = G_::RR, PADB = G_::PADB;
  constexpr int NV = W, ND = R_::ND, NCH4 = R_::NCH4, TAIL = R_::TAIL, RECDW = R_::RECDW;
  extern __shared__ __align__(16) int32_t smem[];

  constexpr int T = TW;  // team = the workgroup's waves (see fill_affine_kernel)
  static_assert(!RESW || (TW == 1 && !LEAN), "strip re-sweeps (see fill_affine_kernel): one wave, full records");
  const int pslot = RESW ? blockIdx.x / A.resw_k : blockIdx.x;
  const int pid = A.order[pslot];
  const PairDesc pd = A.pairs[pid];
  const int n = pd.n, m = pd.m, P = pd.P;
  int Qbase = 0, jlim = m, kk = 0;  // RESW: the strip to sweep, the last column the walk can still reach
  if (RESW) {
    const TraceState ts0 = A.tstate[pid];
    if (ts0.done) return;
    kk = blockIdx.x - pslot * A.resw_k;
    Qbase = (ts0.started ? ts0.strip : pd.NS - 1) - kk;
    if (Qbase < 0) return;
    jlim = (ts0.started && kk == 0) ? ts0.j : m;
  }
  const int L = threadIdx.x & 63;
  const int w = TW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int il = L / W, aa = L - il * W;
  const bool live = L < R * W;
  const bool ghost = (il == 0);
  const int gamma = A.gamma, delta = A.delta;
  const int k1 = A.k1, k2 = A.k2;
  const int gD = gamma + delta, gg = 2 * gamma;

  using GF = GhostFeed<S, 1, LEAN || RESW>;
  using MF = Mu2Feed<S>;
  constexpr int PERW = GF::RING_DW + NV * NCOL + (DENSE ? MF::RING_DW : 0);
  v4i* ring = reinterpret_cast<v4i*>(smem + w * GF::RING_DW);
  int32_t* xch = smem + TW * GF::RING_DW + w * (NV * NCOL);
  int32_t* mu2ring = smem + TW * (GF::RING_DW + NV * NCOL) + w * MF::RING_DW;
  volatile int32_t* prog = smem + TW * PERW;  // [16] steps with acknowledged stores
  int32_t* s1 = smem + TW * PERW + 16;
  int32_t* s2 = s1 + k1 * k1;
  const int npad = (n + 3) & ~3, mpad = (m + 2 * PADB + 3) & ~3;
  uint8_t* sa = reinterpret_cast<uint8_t*>(s2 + k2 * k2);
  uint8_t* ca = sa + npad;
  uint8_t* sb = ca + npad;
  uint8_t* cb = sb + mpad;

  for (int t = threadIdx.x; t < TW * PERW; t += 64 * TW) smem[t] = SENT;
  if (threadIdx.x < 16) prog[threadIdx.x] = 0;
  for (int t = threadIdx.x; t < k1 * k1; t += 64 * TW) s1[t] = A.s1[t];
  for (int t = threadIdx.x; t < k2 * k2; t += 64 * TW) s2[t] = A.s2[t];
  for (int t = threadIdx.x; t < n; t += 64 * TW) {
    sa[t] = A.seq_a[pd.seq_a + t];
    ca[t] = A.cls_a[pd.seq_a + t];
  }
  for (int t = threadIdx.x; t < m + 2 * PADB; t += 64 * TW) {
    const int src = t - PADB;
    const bool ok = src >= 0 && src < m;
    sb[t] = ok ? A.seq_b[pd.seq_b + src] : 0;
    cb[t] = ok ? A.cls_b[pd.seq_b + src] : 0;
  }
  __syncthreads();

  const int colLW = (live && il >= 1) ? L - W : 64;
  const int colLW1 = (live && il >= 1 && aa < W - 1) ? L - W + 1 : 64;
  const int colL1 = (live && il >= 1 && aa > 0) ? L - 1 : 64;
  const int GOFF = P - 2 * (R - 1);
  int32_t* const lay = A.layers + pd.layer_off;  // records the ghost feed replays
  int32_t* const sto = RESW ? A.scratch + pd.scratch_off + (int64_t)kk * (m + G_::MAXOFF + 1) * RECDW : lay;

  const int rec_last = pd.G - 1;
  const int NSw = RESW ? 1 : (pd.NS - w + T - 1) / T;  // this wave's strips: w, w+T, ...
  const int H = NSw > 0 ? (NSw - 1) * P + (RESW ? jlim : m) + G_::MAXOFF + 1 : 0;
  int jj = -(2 * il + aa);
  int strip = 0;         // local strip index q; lattice strip = q*T + w
  int rec_base = w * P;  // record of local step h for this lane = h + rec_base
  int i = 0, s1row = 0, s2row = 0;
  bool act_row = false;
  auto set_row = [&](int q) {
    i = (Qbase + q * T + w) * RR + il - 1;
    const int k = i + aa - S;
    act_row = live && i >= 0 && i <= n && k >= 0 && k <= n;
    s1row = (i >= 1 && i <= n) ? sa[i - 1] * k1 : 0;
    s2row = (k >= 1 && k <= n) ? ca[k - 1] * k2 : 0;
  };
  set_row(0);

  int lw1[W], lw2[W];  // (i-1,a):   value seen 1 step ago (age 2), 2 steps ago (age 3)
  int lwp1[W];         // (i-1,a+1): age 2
  int l11[W];          // (i,a-1):   age 2
  int selfM[W];        // own previous column
  int ghostM[ND];
#pragma unroll
  for (int bb = 0; bb < W; ++bb) lw1[bb] = lw2[bb] = lwp1[bb] = l11[bb] = selfM[bb] = ghostM[bb] = SENT;
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int32_t*)smem) + w * GF::RING_DW * 4;

  // team protocol, as in fill_affine_kernel (in-workgroup form)
  int blk_q = 0, blk_rem = 0;
  bool team_failed = false;
  auto wait_partner = [&](int h_last) __attribute__((always_inline)) {
    if (T == 1 || team_failed) return;
    const int src = w == 0 ? T - 1 : w - 1;
    const int need = h_last + 2 * (R - 1) + 1 - (w == 0 ? P : 0);
    for (int spin = 0; prog[src] < need; ++spin) {
      if (spin > (1 << 20)) {
        if (L == 0) atomicExch(A.errflag, 1);
        team_failed = true;
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
  };
  const uint32_t mu2_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int32_t*)smem) +
      (TW * (GF::RING_DW + NV * NCOL) + w * MF::RING_DW) * 4;
  const int32_t* const mu2tab = DENSE ? A.mu2_dense + pd.mu2_off : nullptr;
  int mu2w[W];
#pragma unroll
  for (int bb = 0; bb < W; ++bb) mu2w[bb] = 0;
  auto prefetch_block = [&](int h0, int half, int jj0) __attribute__((always_inline)) {
    wait_partner(h0 + GF::BLK - 1);
    GF::issue(lay, h0 + Qbase * P, blk_q, blk_rem, P, T, w, GOFF, rec_last, L, ring_lds + half * GF::SLOTS * 16);
    if (DENSE) MF::issue(mu2tab, n, m, P, jj0, Qbase + strip, T, w, il, aa, mu2_lds + half * MF::BLK * 256);
    blk_rem += GF::BLK;
    if (blk_rem >= P) { blk_rem -= P; ++blk_q; }
  };
  prefetch_block(0, 0, jj);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int vm_younger = 0;  // see the affine kernel

  for (int g = 0; g < H; ++g) {
    const int gt = g & (GF::BLK - 1), ghalf = (g / GF::BLK) & 1;
    if (gt == 0) {
      GF::wait_block(vm_younger);
      if (T > 1 && L == 0) prog[w] = g - 8;
      prefetch_block(g + GF::BLK, ghalf ^ 1, jj + GF::BLK);
      vm_younger = 0;
    }
    GF::fetch(ghostM, ring + ghalf * GF::SLOTS, gt, aa);
    int inLW[W], inLW1[W], inL1[W];
#pragma unroll
    for (int bb = 0; bb < W; ++bb) {
      inLW[bb] = xch[bb * NCOL + colLW];
      inLW1[bb] = xch[bb * NCOL + colLW1];
      inL1[bb] = xch[bb * NCOL + colL1];
    }
    const int jc = min(max(jj, 0), m + 1);
    const int mu1 = s1[s1row + sb[jc - 1 + PADB]];
    int mu2[W];
    if (DENSE) {
#pragma unroll
      for (int bb = 0; bb + 1 < W; ++bb) mu2w[bb] = mu2w[bb + 1];
      mu2w[W - 1] = mu2ring[(ghalf * MF::BLK + gt) * 64 + L];
#pragma unroll
      for (int bb = 0; bb < W; ++bb) mu2[bb] = mu2w[bb];
    } else {
#pragma unroll
      for (int bb = 0; bb < W; ++bb) mu2[bb] = s2[s2row + cb[jc + bb]];
    }

    const bool tile_act = act_row && jj >= 0 && jj <= m;
    const bool is_origin = tile_act && i == 0 && jj == 0 && aa == S;
    const int c4 = mu1 + delta, c12 = mu1 + gD;

    int outv[ND];
    int prev = SENT;  // value of point bb-1 of this step (case (0,0,0,1))
#pragma unroll
    for (int bb = 0; bb < W; ++bb) {
      const int l = jj + bb - S;
      const bool act = tile_act && l >= 0 && l <= m;
      const int mu2v = mu2[bb];
      const int c5 = mu2v + delta, c10 = mu2v + gD;
      // the thirteen cases in generator order (pyx:233-248); b-band violations are static
      int t = (mu1 + mu2v) + lw2[bb];                    // (1,1,1,1)
      t = imax(t, gg + lw1[bb]);                         // (1,0,1,0)
      t = imax(t, gg + selfM[bb]);                       // (0,1,0,1)
      if (bb + 1 < W) t = imax(t, c4 + lwp1[bb + 1 < W ? bb + 1 : 0]);   // (1,1,0,0)
      if (bb >= 1) t = imax(t, c5 + inL1[bb >= 1 ? bb - 1 : 0]);         // (0,0,1,1)
      t = imax(t, gD + inLW1[bb]);                       // (1,0,0,0)
      if (bb + 1 < W) t = imax(t, gD + selfM[bb + 1 < W ? bb + 1 : 0]);  // (0,1,0,0)
      t = imax(t, gD + inL1[bb]);                        // (0,0,1,0)
      if (bb >= 1) t = imax(t, gD + prev);               // (0,0,0,1)
      if (bb >= 1) t = imax(t, c10 + lw1[bb >= 1 ? bb - 1 : 0]);         // (1,0,1,1)
      t = imax(t, c10 + l11[bb]);                        // (0,1,1,1)
      if (bb + 1 < W) t = imax(t, c12 + lw2[bb + 1 < W ? bb + 1 : 0]);   // (1,1,1,0)
      t = imax(t, c12 + lwp1[bb]);                       // (1,1,0,1)

      const int tv = ghost ? ghostM[bb] : t;
      const bool bad = (tv < THRESH) | !act;
      int M = bad ? (act ? NEG : SENT) : tv;             // pyx:299-303
      if (bb == S) M = is_origin ? 0 : M;                // np.zeros origin (pyx:27, 464-465)
      outv[bb] = M;
      prev = M;
    }
#pragma unroll
    for (int bb = 0; bb < W; ++bb) {
      xch[bb * NCOL + L] = outv[bb];
      selfM[bb] = outv[bb];
      lw2[bb] = lw1[bb];
      lw1[bb] = inLW[bb];
      lwp1[bb] = inLW1[bb];
      l11[bb] = inL1[bb];
    }

    const int rec = g + rec_base;
    if (LEAN && live && !ghost && aa == S && i == n && jj == m) A.scores[pid] = outv[S];  // pyx:471
    const bool do_store = __builtin_amdgcn_ballot_w64(tile_act && !ghost) != 0 && live &&
                          (LEAN ? il == R - 1 : !ghost) && (T == 1 || rec <= rec_last);  // see the affine kernel
    if (__builtin_amdgcn_ballot_w64(do_store) != 0) vm_younger += GF::STORES_PER_STEP;
    if (do_store) {
      const int slot = LEAN ? aa : L - W;
      int32_t* dst = sto + (int64_t)rec * RECDW;
#pragma unroll
      for (int c = 0; c < NCH4; ++c) {
        v4i v;
        v.x = outv[4 * c]; v.y = outv[4 * c + 1]; v.z = outv[4 * c + 2]; v.w = outv[4 * c + 3];
        *reinterpret_cast<v4i*>(dst + c * R_::CH + slot * 4) = v;
      }
#pragma unroll
      for (int t = 0; t < TAIL; ++t) dst[NCH4 * R_::CH + slot * TAIL + t] = outv[4 * NCH4 + t];
    }
    ++jj;
    if (!RESW && jj == P) {
      jj = 0;
      ++strip;
      rec_base += (T - 1) * P;
      set_row(strip);
    }
  }
  if (T > 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (L == 0) prog[w] = 0x7fffffff;
  }
}

// Non-affine traceback (pyx:513-531): the first case, in generator order, that is
// guard-valid and reproduces the cell; stops when none does (the origin).  One wave
// per pair, lane c < 13 = case c; "first" = wave-min over the matching lane ids.
template <int S, bool DO_TRACE, bool STRIP = false>  // STRIP: see traceback_affine_kernel
__global__ void __launch_bounds__(64) traceback_linear_kernel(const DeviceBatch A, int npairs) {
  const int pid = A.order[blockIdx.x];
  const PairDesc pd = A.pairs[pid];
  const int n = pd.n, m = pd.m;
  const int gamma = A.gamma, delta = A.delta;
  const int32_t* lay = A.layers;
  const int c = threadIdx.x;
  constexpr int BIG = 0x7fffffff;
  constexpr int W = 2 * S + 1, RR = Geo<S>::RR;
  extern __shared__ __align__(16) int32_t smem[];
  TraceState ts{};
  if (STRIP) {
    ts = A.tstate[pid];
    if (ts.done) return;
  }
  const int Q = STRIP ? (ts.started ? ts.strip : pd.NS - 1) : 0;
  const int Qlo = STRIP ? max(Q - A.resw_k + 1, 0) : 0;
  const int64_t sstride = (int64_t)(m + Geo<S>::MAXOFF + 1) * Rec<S, 1>::RECDW;
  auto cell = [&](int pi, int pj, int a, int b) -> int {
    if (!STRIP) return lay[cell_dword<S, 1>(pd, pi, pj, a, b, 0)];
    const int sp = pi / RR, ilp = pi - sp * RR + 1;
    if (sp >= Qlo)
      return A.scratch[pd.scratch_off + (Q - sp) * sstride + Rec<S, 1>::dword(pj + 2 * ilp + a, (ilp - 1) * W + a, b)];
    return lay[pd.layer_off + Rec<S, 1, true>::dword((int64_t)sp * pd.P + pj + 2 * ilp + a, a, b)];
  };
  int cur = (STRIP && ts.started) ? ts.cur : cell(n, m, S, S);
  if (c == 0 && !(STRIP && ts.started)) A.scores[pid] = cur;  // pyx:471
  if (!DO_TRACE) return;
  const TraceInputs in = stage_trace_inputs(A, pd, smem);
  const uint8_t *sa = in.sa, *ca = in.ca, *sb = in.sb, *cb = in.cb;

  // offsets of the thirteen cases as bit masks o0*8+o1*4+o2*2+o3 (pyx:233-248), per lane
  constexpr int OFF[16] = {15, 10, 5, 12, 3, 8, 4, 2, 1, 11, 7, 14, 13, 0, 0, 0};
  int code_c = 0;
#pragma unroll
  for (int t = 0; t < 13; ++t)
    if (c == t) code_c = OFF[t];
  const int o0 = (code_c >> 3) & 1, o1 = (code_c >> 2) & 1, o2 = (code_c >> 1) & 1, o3 = code_c & 1;
  // score of case c as a*mu1 + b*mu2 + const (pyx:233-248)
  const int use1 = (c == 0 || c == 3 || c == 11 || c == 12), use2 = (c == 0 || c == 4 || c == 9 || c == 10);
  const int gD = gamma + delta;
  const int kconst = c == 0 ? 0 : (c <= 2 ? 2 * gamma : (c <= 4 ? delta : gD));

  uint8_t* out = A.trace + pd.trace_off;
  int i = n, j = m, k = n, l = m, len = 0;
  if (STRIP && ts.started) { i = ts.i; j = ts.j; k = ts.k; l = ts.l; len = ts.len; }
  bool finished = true;
  while (true) {
    if (STRIP && i < Qlo * RR) { finished = false; break; }
    const int mu1 = (i >= 1 && j >= 1) ? in.s1[sa[i - 1] * A.k1 + sb[j - 1]] : 0;
    const int mu2 = (k >= 1 && l >= 1)
                        ? (A.mu2_dense ? A.mu2_dense[pd.mu2_off + (int64_t)(k - 1) * m + (l - 1)]
                                       : in.s2[ca[k - 1] * A.k2 + cb[l - 1]])
                        : 0;
    const int sc = kconst + (use1 ? mu1 : 0) + (use2 ? mu2 : 0);
    const int pi = i - o0, pj = j - o1, pk = k - o2, pl = l - o3;
    const bool ok = c < 13 && pi >= 0 && pj >= 0 && pk >= 0 && pl >= 0 && abs(pk - pi) <= S && abs(pl - pj) <= S;
    const int ld = ok ? cell(pi, pj, pk - pi + S, pl - pj + S) : 0;
    const int key = (ok && ld + sc == cur) ? c : BIG;
    const int pick = __builtin_amdgcn_readfirstlane(wave_min16(key));
    if (pick == BIG) break;
    const int code = __builtin_amdgcn_readlane(code_c, pick);
    cur = __builtin_amdgcn_readlane(ld, pick);
    if (c == 0 && len < pd.trace_cap) out[len] = (uint8_t)code;
    ++len;
    i -= (code >> 3) & 1; j -= (code >> 2) & 1; k -= (code >> 1) & 1; l -= code & 1;
  }
  if (STRIP && !finished) {
    if (c == 0) {
      TraceState nx{};
      nx.i = i; nx.j = j; nx.k = k; nx.l = l; nx.cur = cur; nx.len = len;
      nx.strip = Qlo - 1; nx.started = 1; nx.done = 0;
      A.tstate[pid] = nx;
    }
    return;
  }
  if (len > pd.trace_cap) len = pd.trace_cap;
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int x = c; x < len / 2; x += 64) {
    const uint8_t t = out[x];
    out[x] = out[len - 1 - x];
    out[len - 1 - x] = t;
  }
  if (c == 0) {
    A.trace_len[pid] = len;
    A.complete[pid] = 1;
    if (STRIP) {
      ts.done = 1;
      ts.started = 1;
      A.tstate[pid] = ts;
    }
  }
}

// ---------------------------------------------------------------------------
// Layer dump in the reference layout (tests only).
// ---------------------------------------------------------------------------
template <int S, int NL>
__global__ void dump_layers_kernel(const DeviceBatch A, int pid, int32_t* out) {
  constexpr int W = 2 * S + 1;
  const PairDesc pd = A.pairs[pid];
  const int n = pd.n, m = pd.m;
  const int64_t cells = (int64_t)(n + 1) * (m + 1) * W * W;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < cells;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int bb = t % W, aa = (t / W) % W;
    const int j = (t / (W * W)) % (m + 1), i = t / ((int64_t)W * W * (m + 1));
    const int k = i + aa - S, l = j + bb - S;
    const bool ok = k >= 0 && k <= n && l >= 0 && l <= m;
    for (int q = 0; q < NL; ++q)
      out[q * cells + t] = ok ? A.layers[cell_dword<S, NL>(pd, i, j, aa, bb, q)] : 0;
  }
}

}  // namespace bialign
