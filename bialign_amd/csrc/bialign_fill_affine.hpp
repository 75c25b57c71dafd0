// bialign_fill_affine.hpp -- affine sweep (nine layers, fifteen cases).  Part of bialign_kernels.hpp (include that, not this).
#pragma once

namespace bialign {

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
//   XCU = true : the team is A.team workgroups of TW waves on any CUs / XCDs (T = A.team * TW,
//     block b -> pair b / A.team, waves (b % A.team) * TW ..; all co-resident by construction of the grid).
//     Per-XCD L2s are not coherent, so the layer stores of a wave whose successor runs on another CU are
//     write-through (sc1) -- every wave of a one-wave workgroup, the last wave of an eight-wave one -- the
//     ghost DMAs are sc1 loads, the progress words between CUs live in HBM (sc1 atomics), and a word is
//     published only after the stores it covers have left the wave's vector-memory queue.
typedef int v3i __attribute__((ext_vector_type(3)));

// wt (wave-uniform): write through to memory -- the records of a wave whose successor runs on another CU
template <bool XCU>
__device__ __forceinline__ void store_chunk(int32_t* p, v4i v, bool wt) {
  if (XCU && wt)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else
    *reinterpret_cast<v4i*>(p) = v;
}

//   RESW (lean traceback): re-sweep ONE strip of a pair with the ghost row taken from the LEAN
//   records of the strip above and the full records written to a scratch area (record = step
//   within the strip).  Workgroup b handles pair b / K, strip TraceState::strip - b % K (K =
//   A.resw_k strips per round, independent of each other, each into its own scratch slot); the
//   strip the walk stands in is swept only up to the walk's column.  One wave per strip.
//   PACK: interior steps store packed records, the others full records in the pair's second region (Pack<S>).
template <int S, bool BETA_NONPOS, int TW, bool XCU, bool DENSE = false, bool LEAN = false, bool RESW = false,
          bool PACK = false>
__global__ void __launch_bounds__(64 * TW) BIALIGN_WPE_ATTR fill_affine_kernel(const DeviceBatch A) {
  static_assert(!PACK || (!LEAN && !RESW && S >= 1), "packed records: full-storage sweeps with a band");
  using PK_ = Pack<S>;
  // Ghost rows of packed records: unpacked once per block by BLK*W lanes (s=1: 24 lanes every 8 steps, -4 % fill
  // time) or by the ghost lanes' wave in every step (s=2: blocks of 4 or 2 steps, the in-place rewrite at the block
  // boundary costs more latency than it saves issue slots: 168 vs 172 ms on config 4).
  constexpr bool PK_COOP = S == 1;
  static_assert(!XCU || TW == 1 || TW == 8, "cross-CU teams are built from one-wave or eight-wave workgroups");
  // DIET (eight waves of the s=2 kernel in one workgroup = two per SIMD on a whole CU): their arrays fit
  // 160 KB of LDS only with half-length ghost blocks and molecule A's codes left in global memory (they
  // are read once per strip, in set_row).
  constexpr bool DIET = TW == 8 && S == 2;
  static_assert(!DIET || (!DENSE && !RESW), "the diet variant exists for LOOKUP sweeps only");
  static_assert(!RESW || (TW == 1 && !XCU && !LEAN), "strip re-sweeps: one wave, full records");
  using G_ = Geo<S>;
  using R_ = Rec<S, 9, LEAN>;
  constexpr int W = G_::W, R = G_::R, RR = G_::RR, PADB = G_::PADB;
  constexpr int XR = 12;  // exchange rows per point that go through LDS
  constexpr int NV = XR * W, ND = R_::ND, NCH4 = R_::NCH4, TAIL = R_::TAIL, RECDW = R_::RECDW;
  extern __shared__ __align__(16) int32_t smem[];

  const int T = XCU ? A.team * TW : TW;                  // team size
  const int slot = XCU ? blockIdx.x / A.team : (RESW ? blockIdx.x / A.resw_k : blockIdx.x);  // pair of this launch
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
    jlim = ts0.started ? ts0.j : m;  // the walk never moves right: no strip of this round is entered beyond its column
  }
  const int L = threadIdx.x & 63;
  const int wl = TW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave in workgroup
  const int w = XCU ? (int)(blockIdx.x - slot * A.team) * TW + wl : wl;               // wave in team
  const int il = L / W, aa = L - il * W;
  const bool live = L < R * W;
  const bool ghost = (il == 0);
  // the lanes that compute in interior steps, as a laundered register: the compiler then keeps the predicate as a mask
  // instead of re-deriving it from the lane number at every point of every step
  int computing_v = ghost ? 0 : 1;
  asm volatile("" : "+v"(computing_v));
  const bool computing = computing_v != 0;
  const int beta = A.beta, gamma = A.gamma, delta = A.delta;
  const int k1 = A.k1, k2 = A.k2;
  const int gD = gamma + delta, gg = 2 * gamma, ggdd = 2 * gamma + 2 * delta, dd = 2 * delta;

  // ---- LDS carve-up: per wave a ghost ring and an exchange array; shared: progress
  //      words, score tables, sequence codes
  using GF = GhostFeed<S, 9, LEAN || RESW, DIET ? 2 : 0>;  // a re-sweep replays LEAN records
  using MF = Mu2Feed<S>;
  constexpr int PERW = GF::RING_DW + NV * NCOL + (DENSE ? MF::RING_DW : 0);  // dwords per wave
  v4i* ring = reinterpret_cast<v4i*>(smem + wl * GF::RING_DW);   // ghost-row ring, two halves
  int32_t* xch = smem + TW * GF::RING_DW + wl * (NV * NCOL);     // exchange array [NCOL lanes][NV]
  int32_t* mu2ring = smem + TW * (GF::RING_DW + NV * NCOL) + wl * MF::RING_DW;  // dense-mu2 ring
  volatile int32_t* prog_lds = smem + TW * PERW;                  // [16] (in-workgroup teams)
  int32_t* s1 = smem + TW * PERW + 16;                            // [k1*k1]
  int32_t* s2 = s1 + k1 * k1;                                   // [k2*k2]
  const int npad = (n + 3) & ~3, mpad = (m + 2 * PADB + 3) & ~3;
  uint8_t* sa = reinterpret_cast<uint8_t*>(s2 + k2 * k2);  // seq A codes, [i-1]   (DIET: not staged)
  uint8_t* ca = sa + npad;                                  // cls A,       [k-1]
  uint8_t* sb = DIET ? sa : ca + npad;                      // seq B codes, [j-1+PADB]
  uint8_t* cb = sb + mpad;                                  // cls B,       [l-1+PADB]

  for (int t = threadIdx.x; t < TW * PERW; t += 64 * TW) smem[t] = SENT;
  if (threadIdx.x < 16) prog_lds[threadIdx.x] = 0;
  for (int t = threadIdx.x; t < k1 * k1; t += 64 * TW) s1[t] = A.s1[t];
  for (int t = threadIdx.x; t < k2 * k2; t += 64 * TW) s2[t] = A.s2[t];
  if (!DIET)
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

  const int64_t pk_bnd_off = (int64_t)pd.G * PK_::RECDW;  // PACK: the pair's full records follow its packed ones
  static_assert(!PACK || (pack_corner(W, 3, 0) == can_be_empty<W>(1, 0, 0) && pack_corner(W, 4, 0) == can_be_empty<W>(1, 1, 0) &&
                          pack_corner(W, 1, W - 1) == can_be_empty<W>(0, 1, W - 1) && pack_corner(W, 8, W - 1) == can_be_empty<W>(2, 2, W - 1)),
                "pack_corner mirrors can_be_empty");
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
    if (DIET) {  // once per strip: two byte loads from global memory, consumed right here
      s1row = (i >= 1 && i <= n) ? A.seq_a[pd.seq_a + i - 1] * k1 : 0;
      s2row = (k >= 1 && k <= n) ? A.cls_a[pd.seq_a + k - 1] * k2 : 0;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      s1row = (i >= 1 && i <= n) ? sa[i - 1] * k1 : 0;
      s2row = (k >= 1 && k <= n) ? ca[k - 1] * k2 : 0;
    }
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
  // Cross-CU teams of multi-wave workgroups: only the last wave of a workgroup hands over to another CU
  // (its successor is wave 0 of the next workgroup): it alone writes through and publishes its progress in
  // HBM; the others meet their successor in this workgroup's LDS and L2, like an in-workgroup team.
  const bool to_other_cu = XCU && (TW == 1 || wl == TW - 1);   // this wave's records are read on another CU
  // (Writing through the bottom lane row only -- all the successor replays -- was tried and is both slower,
  //  54 -> 96 ms on a config-4 chunk, and wrong: lines then mix write-through and write-back sectors.)
  const bool wt_lane = to_other_cu;
  const bool from_other_cu = XCU && (TW == 1 || wl == 0);      // this wave's predecessor runs on another CU
  int32_t* const prog_glb = XCU ? A.prog + (int64_t)slot * PROG_WORDS : nullptr;
  auto prog_get = [&](int idx) __attribute__((always_inline)) -> int {
    if (from_other_cu) return __hip_atomic_load(prog_glb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return prog_lds[XCU ? wl - 1 : idx];
  };
  auto prog_put = [&](int v) __attribute__((always_inline)) {  // lane 0 only
    if (to_other_cu)
      __hip_atomic_store(prog_glb + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      prog_lds[XCU ? wl : w] = v;
  };
  int seen_prog = -0x40000000;  // the partner's progress as last read (INT_MAX once a hand-off has timed out: no further waits)
  auto wait_partner = [&](int h_last) __attribute__((always_inline)) {
    if ((!XCU && TW == 1) || T == 1 || BIALIGN_EXP == 9) return;  // 9: timing experiment, no hand-off waits
    const int src = w == 0 ? T - 1 : w - 1;
    const int need = h_last + 2 * (R - 1) + 1 - (w == 0 ? P : 0);
    // progress only grows: what was seen last time usually covers this block too, and a look at
    // the partner's word is a round trip to HBM for cross-CU teams
    if (seen_prog >= need) return;
    // bounded spin: a protocol bug must surface as an error, never as a hung GPU
    for (int spin = 0; (seen_prog = prog_get(src)) < need; ++spin) {
      if (spin > A.spin_limit) {  // ~1 s by default; then fail fast: no further waits, the host recovers or reports
        if (L == 0) atomicOr(A.errflag, 1);
        seen_prog = 0x7fffffff;
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
    if (PACK)
      GF::issue_packed(lay, pk_bnd_off, m, h0, blk_q, blk_rem, P, T, w, rec_last, L, ring_lds + half * GF::SLOTS * 16);
    else
      GF::issue(lay, h0 + Qbase * P, blk_q, blk_rem, P, T, w, GOFF, rec_last, L, ring_lds + half * GF::SLOTS * 16);
    if (DENSE) MF::issue(mu2tab, n, m, P, jj0, Qbase + strip, T, w, il, aa, mu2_lds + half * MF::BLK * 256);
    blk_rem += GF::BLK;
    if (blk_rem >= P) { blk_rem -= P; ++blk_q; }
  };
  // block 0 must be in the ring before the first step (waves w >= 1 start on a real ghost row)
  prefetch_block(0, 0, jj);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int vm_younger = 0;  // store instructions issued since the last block's DMAs (wave-uniform)
  int pk_all = 0;      // packed records: OR of every offset this lane has stored since the last range check

  // Exchange inputs of rows i-1 (what lanes L-W, L-W+1 published for each band column).  They are loop-carried:
  // row r is fetched from the exchange array at the END of a step, as soon as its last consumer of that step is
  // through, for the step after it -- the LDS round trip (publish -> read) overlaps the rest of the step instead
  // of standing at the head of the next one.  Likewise the score inputs of the next column: the two dependent LDS
  // reads (code byte, then table entry) leave the head of the step (-1 % fill time at len 1024; the sweep is bound by
  // VALU issue, profiles/r03a_headline_baseline).
  // Both early fetches cost registers (12 W for the exchange inputs, W + 3 for the scores): at max_shift >= 2 the
  // eight-wave kernels (256 registers) spill with them and config 4 ran at half speed -- there the rows are read at
  // the head of the step, two band columns ahead of use, as in round 2.
  constexpr bool PREF = S <= 1;
  int inA[W][4], inB[W][8];
  auto read_lds = [&](int r) __attribute__((always_inline)) {
    // a lane's NV values lie together (NV = 12 W dwords: 16-byte aligned, and eight neighbouring lanes' 16-byte pieces
    // fall into 32 different banks): what a source lane published for band column r comes back as three 16-byte reads
    const v4i a = *reinterpret_cast<const v4i*>(xch + colLW * NV + r * XR);
    const v4i b0 = *reinterpret_cast<const v4i*>(xch + colLW1 * NV + r * XR + 4);
    const v4i b1 = *reinterpret_cast<const v4i*>(xch + colLW1 * NV + r * XR + 8);
    inA[r][0] = a.x; inA[r][1] = a.y; inA[r][2] = a.z; inA[r][3] = a.w;
    inB[r][0] = b0.x; inB[r][1] = b0.y; inB[r][2] = b0.z; inB[r][3] = b0.w;
    inB[r][4] = b1.x; inB[r][5] = b1.y; inB[r][6] = b1.z; inB[r][7] = b1.w;
  };
  // A delay-line stage.  The copy is opaque to the compiler: coalescing it would keep the old value alive in the
  // register the next fetch wants, and the allocator then copies the freshly fetched values at the back-edge instead
  // -- behind an lgkmcnt wait, which is the stall the early fetch was to hide.  (Moving two stages as one register
  // pair buys nothing: on gfx950 v_mov_b64 costs a SIMD 4.3 cycles, two v_mov_b32 4.8 -- tools/valu_rate.hip.)
  auto dmov = [](int& dst, int src) __attribute__((always_inline)) {
    if (PREF) asm("v_mov_b32 %0, %1" : "=v"(dst) : "v"(src));
    else dst = src;
  };
  auto dmov2 = [&](int& d0, int& d1, int s0, int s1) __attribute__((always_inline)) {
    dmov(d0, s0);
    dmov(d1, s1);
  };
  int mu1n = 0, mu2n[W];
  auto lookup_mu = [&]() __attribute__((always_inline)) {  // score inputs of the column this lane works on next (LOOKUP form)
    const int jc = min(max(jj, 0), m + 1);
    mu1n = s1[s1row + sb[jc - 1 + PADB]];
#pragma unroll
    for (int bb = 0; bb < W; ++bb) mu2n[bb] = DENSE ? 0 : s2[s2row + cb[jc + bb]];  // l-1+PADB = jc+bb
  };
  if (PREF) {
#pragma unroll
    for (int r = 0; r < W; ++r) read_lds(r);
  }
  // ... and, one step further ahead, the two codes an interior step's end looks up (column jj + 1 of the same row)
  int sbn = 0, cbn = 0;
  auto fetch_codes = [&]() __attribute__((always_inline)) {
    const int jc1 = min(max(jj + 1, 0), m + 1);
    sbn = sb[jc1 - 1 + PADB];
    cbn = DENSE ? 0 : cb[jc1 + W - 1];
  };
  if (PREF) {
    lookup_mu();
    fetch_codes();
  }

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
      // the DMAs just retired were issued at step g - BLK ahead of that step's stores, and vmcnt retires in
      // order: the stores of all steps before g - BLK are acknowledged
      if ((XCU || TW > 1) && L == 0) prog_put(g - GF::BLK);
      if (PACK && (g & 15) == 0) {  // every 16 steps: an offset since then that does not fit 16 bits (or collides with the -2^30 mark): the host falls back
        const bool bad = live && !ghost && (unsigned)pk_all > 0xffffu;
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && L == 0) atomicOr(A.errflag, 2);
        pk_all = 0;
      }
      if (PACK && PK_COOP) {
        // The block that has just landed holds, per (step t, band row a), the ghost row's source as it lies in
        // HBM: a packed lane record if that step of the strip above was interior (record phase c + 2(R-1)), else
        // a full one.  Lane t*W + a unpacks its entry in place, once per block -- the ghost lanes then read
        // plain layer values every step, and only BLK*W lanes of one step in BLK pay for the decode.
        const int c0 = __builtin_amdgcn_readfirstlane(jj), q0 = __builtin_amdgcn_readfirstlane(strip);
        if (L < GF::BLK * W) {
          const int t = L / W, ai = L - t * W;
          int ph = c0 + t, qst = q0 * T + w;  // lane 0's phase and strip at step g + t
          if (ph >= P) { ph -= P; qst += T; }
          const int ts = ph + 2 * (R - 1), over = ts >= P ? 1 : 0;
          if (PK_::interior(qst - 1 + over, ts - over * P, m)) {
            v4i* pc = ring + ghalf * GF::SLOTS + (t * W + ai) * GF::NP;
            int raw[4 * PK_::NPC], dec[4 * GF::NP];
#pragma unroll
            for (int c = 0; c < PK_::NPC; ++c) {
              const v4i v = pc[c];
              raw[4 * c] = v.x; raw[4 * c + 1] = v.y; raw[4 * c + 2] = v.z; raw[4 * c + 3] = v.w;
            }
#pragma unroll
            for (int d = 0; d < 4 * GF::NP; ++d) {
              const int h = PK_::hw(d < ND ? d : 0);
              const unsigned word = (unsigned)raw[d < ND && d != PK_::ANCHOR ? h >> 1 : 0];
              const unsigned e = d == PK_::ANCHOR ? 0x8000u : PK_::offset_of((h & 1) ? word >> 16 : word & 0xffffu, raw[0]);
              const int v = raw[0] + (int)e;
              dec[d] = d >= ND ? 0 : (pack_corner(W, d % 9, d / 9) ? (e == 0xffffu ? NEG : v) : v);
            }
#pragma unroll
            for (int c = 0; c < GF::NP; ++c) {
              v4i v;
              v.x = dec[4 * c]; v.y = dec[4 * c + 1]; v.z = dec[4 * c + 2]; v.w = dec[4 * c + 3];
              pc[c] = v;
            }
          }
        }
      }
      prefetch_block(g + GF::BLK, ghalf ^ 1, jj + GF::BLK);
      vm_younger = 0;
      if (A.prio_mode) {
        // Issue arbitration favours the oldest wave of a SIMD: the first workgroup dispatched to a CU finishes early
        // and leaves its SIMD to a lone wave that cannot fill it (round 1 saw first-finish 14.8 ms, last 21.3).  Rotating
        // four priority levels over the workgroups of a CU by age (blockIdx / 256 CUs) every 128 steps keeps the waves of
        // a SIMD level: -8 % fill time at len 1024, -5.6 % at config 2, -5 % on config 4 (profiles/r02k_priority).
        const int lvl = ((g >> 7) + (int)(blockIdx.x >> 8)) & 3;
        if (lvl == 0) __builtin_amdgcn_s_setprio(0);
        else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
        else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
      }
    }
    bool ghost_packed = false;
    if (PACK && !PK_COOP) {  // per step: is the ghost row's source record (phase c + 2(R-1) of the strip above) a packed one?
      const int c0 = __builtin_amdgcn_readfirstlane(jj), q0 = __builtin_amdgcn_readfirstlane(strip);
      const int ts = c0 + 2 * (R - 1), over = ts >= P ? 1 : 0;
      ghost_packed = PK_::interior(q0 * T + w - 1 + over, ts - over * P, m);
    }
    if (PACK && !PK_COOP && ghost_packed) {
      int raw[4 * PK_::NPC];
      GF::template fetch_pieces<PK_::NPC>(raw, ring + ghalf * GF::SLOTS, gt, aa);
      unsigned off2[PK_::NDW];  // the two offsets each record dword holds
#pragma unroll
      for (int x = 1; x < PK_::NDW; ++x) off2[x] = PK_::offsets_of((unsigned)raw[x], raw[0]);
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const int h = PK_::hw(d);
        const unsigned word = off2[d != PK_::ANCHOR ? h >> 1 : 1];
        const unsigned e = d == PK_::ANCHOR ? 0x8000u : ((h & 1) ? word >> 16 : word & 0xffffu);
        const int v = raw[0] + (int)e;
        ghostM[d] = pack_corner(W, d % 9, d / 9) ? (e == 0xffffu ? NEG : v) : v;
      }
    } else {
      GF::fetch(ghostM, ring + ghalf * GF::SLOTS, gt, aa);
    }

    // ---- 1. exchange reads: what the three source lanes published last step.  Rows of band
    //         column r are first needed by point r-1, so they are fetched two points ahead
    //         (all of them up front for W <= 3): a sliding window keeps registers flat in W.
    int inC[W][8];
    auto read_rows = [&](int r) __attribute__((always_inline)) {
      if (!PREF) read_lds(r);
      // lane L-1 = (i, a-1) hands its values over in registers: one DPP wave shift fused with a min
      // against the lane's cap (the sentinel where a-1 leaves the band, INT_MAX elsewhere).  One asm
      // block per band column: the compiler's own DPP folding gives up once the consumers are sunk
      // behind the store branches.  s_nop 1 = the two wait states a DPP read needs after a VALU write
      // of its source (the hazard recogniser does not look inside asm; dropping it "because the sources are a
      // step old" broke the dense s=2,3 kernels in round 3); lane 0 reads out of range -> 0
      // with bound_ctrl, it is an a_first lane anyway.  H2[.][M] (x = 2..4) of the last band column has
      // no consumer: offset (0,0,M) from there would leave the band.
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
    };
    read_rows(0);
    if (W > 1) read_rows(W > 1 ? 1 : 0);

    // ---- 2. score inputs of this column (pyx:260-261; LOOKUP form)
    const int jc = INTERIOR ? jj : min(max(jj, 0), m + 1);
    const int mu1 = PREF ? mu1n : s1[s1row + sb[jc - 1 + PADB]];
    int mu2[W];
    if (DENSE) {  // slide the window, take this step's new value from the ring
#pragma unroll
      for (int bb = 0; bb + 1 < W; ++bb) mu2w[bb] = mu2w[bb + 1];
      mu2w[W - 1] = mu2ring[(ghalf * MF::BLK + gt) * 64 + L];
#pragma unroll
      for (int bb = 0; bb < W; ++bb) mu2[bb] = mu2w[bb];
    } else if (PREF) {
#pragma unroll
      for (int bb = 0; bb < W; ++bb) mu2[bb] = mu2n[bb];
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
    const int pad_idx = L < W ? L : (L >= R * W ? W + (L - R * W) : 64);  // spare lanes: ghost row, idle lanes
    const bool pad_lane = !LEAN && pad_idx < R_::SLP - R_::SL;             // ... that own a pad slot of every chunk
    const bool do_store = BIALIGN_EXP != 1 && ((live && (LEAN ? il == R - 1 : !ghost)) || pad_lane) &&
                          (INTERIOR || __builtin_amdgcn_ballot_w64(tile_act && !ghost) != 0) &&
                          ((!XCU && TW == 1) || rec <= rec_last);
    const int slot = LEAN ? aa : (pad_lane ? R_::SL + pad_idx : L - W);  // storage slot of this lane
    if (INTERIOR && BIALIGN_EXP != 1) vm_younger += PACK ? PK_::NPC : GF::STORES_PER_STEP;  // (an interior step always stores)
    else if (__builtin_amdgcn_ballot_w64(do_store) != 0) vm_younger += GF::STORES_PER_STEP;
    int32_t* dst = BIALIGN_EXP == 2
                       ? A.layers + ((int64_t)(blockIdx.x & 255) << 18) + (int64_t)(g & 31) * RECDW
                       : sto + (int64_t)rec * RECDW;
    int32_t* const dstp = lay + (int64_t)rec * PK_::RECDW;  // PACK, interior steps: the packed record
    if (PACK && !INTERIOR) {  // full record of a non-interior step: the pair's second region, by (step-strip, phase)
      const int tl = jj + 2 * il + aa;  // = rec - (lane's strip) * P
      const int over = tl >= P ? 1 : 0;
      dst = lay + pk_bnd_off + PK_::bidx(strip * T + w + over, tl - over * P, P, m) * RECDW;
    }

    // ---- 3. the W lattice points of this (i, j, a)
    int outv[ND];
    int pk_base = 0, pk_min = 0, pk_max = 0, pk_e[PACK ? ND : 1];  // packed records: base, running extremes of the step's values, what the record takes
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
          if (hU < 2 && hV < 2 && ok2 && ok3) {
            // both gap-gap groups cost gamma + Delta: max(c1 + g, gD + h2, gD + h3) = gD + max(g + (c1 - gD), h2, h3),
            // exact in integers, one add less (c1 - gD is wave-uniform: gamma - Delta or gamma + Delta)
            const int inner = ok1 ? imax(imax(gin + (c1 - gD), h2in), h3in) : imax(h2in, h3in);
            t = gD + inner;
          } else {
            if (ok1) { t = c1 + gin; any = true; }
            if (ok2) { t = any ? imax(t, c2 + h2in) : c2 + h2in; any = true; }
            if (ok3) { t = any ? imax(t, c3 + h3in) : c3 + h3in; any = true; }
          }
          Tv[3 * hU + hV] = t;
        }
      }

      };
      // finalise: ghost rows take the stored layers; "no valid case" -> -2^30
      // (pyx:299-303); points outside the lattice carry the sentinel.
      int M[9];
      bool isneg[9] = {};  // "no valid case" per corner state, as the finalisation found it (computing lanes)
      if (INTERIOR) {
        // ghost lanes keep what the ring delivered; the others compute in place under the
        // execution mask (no per-value select)
#pragma unroll
        for (int q = 0; q < 9; ++q) M[q] = ghostM[bb * 9 + q];
        if (computing) {
          int Tv[9];
          cases(Tv);
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            int tv = Tv[q];
            if (can_be_empty<W>(q / 3, q % 3, bb)) {
              isneg[q] = tv < THRESH;
              tv = isneg[q] ? NEG : tv;
            }
            M[q] = tv;
          }
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
      if (PACK && INTERIOR) {
        // Packed record (Pack<S>): dword 0 = base, then the low halves of all values but the anchor (whose offset is
        // 0x8000 by construction).  The range of what the lanes that hold lattice points store is checked through the
        // running minimum and maximum of the step.
        if (bb == 0) {
          pk_base = M[8] - 0x8000;
          pk_min = pk_max = M[8];
        }
        {
          int xs[9];
          int nx = 0;
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            if (bb * 9 + q == PK_::ANCHOR) continue;
            int x = M[q];
            if (pack_corner(W, q, bb)) {  // offset 0xffff is the -2^30 mark here: base + 0xffff stands in for the value, in the
              // record and in the running minimum alike (it is the largest value a record can hold; a finite value that
              // happens to equal -2^30 is not marked and fails the range check)
              const bool ng = isneg[q];
              x = ng ? pk_base + 0xffff : x;
            }
            pk_e[PACK ? bb * 9 + q : 0] = x;  // (the record takes the low half)
            xs[nx++] = x;
          }
          // range: every stored value within [base, base + 0xfffe]  <=>  min - base >= 0 and max - base + 1 <= 0xffff;
          // the maximum of a point's nine values is its G[M][M], computed below anyway
#pragma unroll
          for (int t = 0; t + 1 < nx; t += 2) pk_min = imin(imin(pk_min, xs[t]), xs[t + 1]);
          if (nx & 1) pk_min = imin(pk_min, xs[nx - 1]);
        }
        if (do_store) {
#pragma unroll
          for (int c = 0; c < PK_::NPC; ++c) {
            constexpr int NDWc = PK_::NDW;
            // (issued together after the last point: the packed sweep is bound by issue, not by the store queue -- spreading
            //  them over the step, which paid 6 % with full records, now costs 1-2.5 %: config-4 chunk 78.5 vs 76.5 ms)
            if (bb == W - 1) {
              int dw[4];
#pragma unroll
              for (int x = 0; x < 4; ++x) {
                const int d = 4 * c + x;  // dword of the lane record
                const int lo = d < NDWc ? PK_::val(2 * d) : 0, hi = d < NDWc ? PK_::val(2 * d + 1) : 0;
                // low halves of two offsets into one dword: one v_perm_b32
                dw[x] = d == 0 ? pk_base
                               : (d < NDWc ? (int)__builtin_amdgcn_perm((unsigned)pk_e[PACK ? hi : 0], (unsigned)pk_e[PACK ? lo : 0], 0x05040100u) : 0);
              }
              if (c < PK_::NCH) {
                v4i v;
                v.x = dw[0]; v.y = dw[1]; v.z = dw[2]; v.w = dw[3];
                store_chunk<XCU>(dstp + c * R_::CH + slot * 4, v, wt_lane);
              } else if (slot < PK_::TSLOTS) {  // the tail piece: TAILDW dwords per lane
                int32_t* tp = dstp + PK_::NCH * R_::CH + slot * PK_::TAILDW;
#pragma unroll
                for (int x = 0; x < PK_::TAILDW; ++x) {
                  if (XCU && wt_lane) __hip_atomic_store(tp + x, dw[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  else tp[x] = dw[x];
                }
              }
            }
          }
        }
      }
      if (do_store && !(PACK && INTERIOR)) {
#pragma unroll
        for (int c = 0; c < NCH4; ++c) {
          if (4 * c + 3 >= bb * 9 && 4 * c + 3 < (bb + 1) * 9) {  // chunk c completes with this point
            v4i v;
            v.x = outv[4 * c]; v.y = outv[4 * c + 1]; v.z = outv[4 * c + 2]; v.w = outv[4 * c + 3];
            store_chunk<XCU>(dst + c * R_::CH + slot * 4, v, wt_lane);
          }
        }
        if (LEAN && bb == W - 1) {
#pragma unroll
          for (int t = 0; t < TAIL; ++t) {
            if (wt_lane)
              __hip_atomic_store(dst + NCH4 * R_::CH + slot * TAIL + t, outv[4 * NCH4 + t], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
            else
              dst[NCH4 * R_::CH + slot * TAIL + t] = outv[4 * NCH4 + t];
          }
        }
      }
      if (!LEAN && bb == W - 1 && !(PACK && INTERIOR)) {
        // the tail is stored by ALL 64 lanes (Rec::TAILSLOTS): the spare ones fill the record up to its end
        const bool wave_stores = BIALIGN_EXP != 1 && (INTERIOR || __builtin_amdgcn_ballot_w64(tile_act && !ghost) != 0) &&
                                 ((!XCU && TW == 1) || rec <= rec_last);
        if (wave_stores) {
          const int tslot = (live && !ghost) ? L - W : R_::SL + (L < W ? L : W + (L - R * W));
#pragma unroll
          for (int t = 0; t < TAIL; ++t) {
            if (wt_lane)
              __hip_atomic_store(dst + NCH4 * R_::CH + tslot * TAIL + t, outv[4 * NCH4 + t], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
            else
              dst[NCH4 * R_::CH + tslot * TAIL + t] = outv[4 * NCH4 + t];
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
      if (PACK && INTERIOR) {
        pk_max = imax(pk_max, Gd[2][2]);
        if (bb == W - 1) {  // rows outside the lattice hold don't-care values; the OR runs on across steps, tested every 16
          const int acc = pk_all | (pk_max - pk_base + 1) | (pk_min - pk_base);
          pk_all = act_row ? acc : 0;
        }
      }
      // publish (all reads of this step were issued above, LDS keeps order)
      v4i* mine = reinterpret_cast<v4i*>(xch + L * NV + bb * XR);
      mine[0] = v4i{Gd[2][2], Gd[2][1], Gd[1][2], Gd[1][1]};
      mine[1] = v4i{Gd[2][0], Gd[1][0], H3[2][0], H3[2][1]};
      mine[2] = v4i{H3[2][2], H3[1][0], H3[1][1], H3[1][2]};
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
      dmov2(dA2[0][bb], dA2[1][bb], dA1[0][bb], dA1[1][bb]);
      dmov2(dA1[0][bb], dA1[1][bb], inA[bb][0], inA[bb][1]);
      dmov2(dB[0][bb], dB[1][bb], inB[bb][0], inB[bb][2]);
      dmov2(dB[2][bb], dB[3][bb], inB[bb][3], inB[bb][4]);
      dC[0][bb] = inC[bb][0];
      dC[1][bb] = inC[bb][1];
      if (bb >= 1) {
        dmov2(dAx[0][bb >= 1 ? bb - 1 : 0], dAx[1][bb >= 1 ? bb - 1 : 0], inA[bb >= 1 ? bb - 1 : 0][2], inA[bb >= 1 ? bb - 1 : 0][3]);
        if (PREF) read_lds(bb >= 1 ? bb - 1 : 0);  // row bb-1 has served this step: fetch what it holds for the next
      }
    }
    dmov2(dAx[0][W - 1], dAx[1][W - 1], inA[W - 1][2], inA[W - 1][3]);
    if (PREF) read_lds(W - 1);

    // ---- 6. advance
    ++jj;
    if (!RESW && !INTERIOR && jj == P) {  // a re-sweep ends inside its one strip; an interior step never ends a strip (phase <= m - S)
      jj = 0;
      ++strip;
      rec_base += (T - 1) * P;
      set_row(strip);
    }
    if (PREF) {
      if (INTERIOR) {  // same row, next column, everything inside the molecule: the window slides by one, and the
        mu1n = s1[s1row + sbn];  // two codes it needs were fetched at the end of the step before
        if (!DENSE) {
#pragma unroll
          for (int bb = 0; bb + 1 < W; ++bb) dmov(mu2n[bb], mu2[bb + 1]);
          mu2n[W - 1] = s2[s2row + cbn];
        }
      } else {
        lookup_mu();
      }
      fetch_codes();
    }
  };

  // Two separate loops (not one loop with a branch inside): each keeps its loop-carried
  // registers where it likes; values only move at the rare hand-overs between runs.
  // Interior steps by record number (Pack<S>::interior; lane 0 = ghost row, a = -s, carries the wave's phase):
  // phase c in [LO, m - S] puts every lane inside the molecule columns with all its band points, strip >= Q0
  // puts every lane row at i >= S + 1.  Readers of packed records apply the same rule to find a cell.
  // (Deciding the kind of a whole run of steps at once -- counted loops instead of a test per step -- costs
  //  ~40 registers in hipcc's allocation and spills the eight-wave s=2 kernels: measured, not kept.)
  auto all_interior = [&]() __attribute__((always_inline)) {
    const int c0 = __builtin_amdgcn_readfirstlane(jj), q0 = __builtin_amdgcn_readfirstlane(strip);
    return PK_::interior(Qbase + q0 * T + w, c0, m);
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
  if (PACK && __builtin_amdgcn_ballot_w64(live && !ghost && (unsigned)pk_all > 0xffffu) != 0 && L == 0) atomicOr(A.errflag, 2);
  if (XCU || TW > 1) {  // everything this wave wrote is acknowledged: release the partner for good
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (L == 0) prog_put(0x7fffffff);
  }
}

}  // namespace bialign
