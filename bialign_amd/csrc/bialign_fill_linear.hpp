// bialign_fill_linear.hpp -- one-layer sweep (thirteen cases).  Part of bialign_kernels.hpp (include that, not this).
#pragma once

namespace bialign {

// ---------------------------------------------------------------------------
// Non-affine fill (pyx:443-471): one layer, thirteen cases (pyx:233-248).
// Same lane mapping and skew as the affine sweep.  A lane publishes only its W
// layer values per step; the three source lanes' values are read one step
// later and kept in registers for the cases that need them 2 or 3 steps later
// (age of offset o = o0 + o1 + o2).
// ---------------------------------------------------------------------------
//   XCU: the team is A.team one-wave workgroups on any CUs (progress words in HBM, write-through
//   stores), as in fill_affine_kernel -- for a handful of long pairs, e.g. one pair from the CLI.
template <int S, int TW, bool DENSE = false, bool LEAN = false, bool RESW = false, bool XCU = false>
__global__ void __launch_bounds__(64 * TW) fill_linear_kernel(const DeviceBatch A) {
  static_assert(!XCU || (TW == 1 && !RESW), "cross-CU teams are built from one-wave workgroups");
  using G_ = Geo<S>;
  using R_ = Rec<S, 1, LEAN>;
  constexpr int W = G_::W, R = G_::R, RR = G_::RR, PADB = G_::PADB;
  constexpr int NV = W, ND = R_::ND, NCH4 = R_::NCH4, TAIL = R_::TAIL, RECDW = R_::RECDW;
  extern __shared__ __align__(16) int32_t smem[];

  const int T = XCU ? A.team : TW;  // team = the workgroup's waves, or A.team one-wave workgroups (see fill_affine_kernel)
  static_assert(!RESW || (TW == 1 && !LEAN), "strip re-sweeps (see fill_affine_kernel): one wave, full records");
  const int pslot = XCU ? blockIdx.x / T : (RESW ? blockIdx.x / A.resw_k : blockIdx.x);
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
    jlim = ts0.started ? ts0.j : m;  // the walk never moves right: no strip of this round is entered beyond its column
  }
  const int L = threadIdx.x & 63;
  const int wl = TW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave in workgroup
  const int w = XCU ? (int)(blockIdx.x - pslot * T) : wl;                              // wave in team
  const int il = L / W, aa = L - il * W;
  const bool live = L < R * W;
  const bool ghost = (il == 0);
  const int gamma = A.gamma, delta = A.delta;
  const int k1 = A.k1, k2 = A.k2;
  const int gD = gamma + delta, gg = 2 * gamma;

  using GF = GhostFeed<S, 1, LEAN || RESW>;
  using MF = Mu2Feed<S>;
  constexpr int PERW = GF::RING_DW + NV * NCOL + (DENSE ? MF::RING_DW : 0);
  v4i* ring = reinterpret_cast<v4i*>(smem + wl * GF::RING_DW);
  int32_t* xch = smem + TW * GF::RING_DW + wl * (NV * NCOL);
  int32_t* mu2ring = smem + TW * (GF::RING_DW + NV * NCOL) + wl * MF::RING_DW;
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
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int32_t*)smem) + wl * GF::RING_DW * 4;

  // team protocol, as in fill_affine_kernel
  int blk_q = 0, blk_rem = 0;
  bool team_failed = false;
  int32_t* const prog_glb = XCU ? A.prog + (int64_t)pslot * PROG_WORDS : nullptr;
  auto prog_get = [&](int idx) __attribute__((always_inline)) -> int {
    if (XCU) return __hip_atomic_load(prog_glb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return prog[idx];
  };
  auto prog_put = [&](int v) __attribute__((always_inline)) {  // lane 0 only
    if (XCU)
      __hip_atomic_store(prog_glb + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      prog[w] = v;
  };
  int seen_prog = -0x40000000;  // the partner's progress as last read (it only grows)
  auto wait_partner = [&](int h_last) __attribute__((always_inline)) {
    if (T == 1 || team_failed) return;
    const int src = w == 0 ? T - 1 : w - 1;
    const int need = h_last + 2 * (R - 1) + 1 - (w == 0 ? P : 0);
    if (seen_prog >= need) return;
    for (int spin = 0; (seen_prog = prog_get(src)) < need; ++spin) {
      if (spin > A.spin_limit) {
        if (L == 0) atomicOr(A.errflag, 1);
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
      if (T > 1 && L == 0) prog_put(g - GF::BLK);  // see the affine kernel
      prefetch_block(g + GF::BLK, ghalf ^ 1, jj + GF::BLK);
      vm_younger = 0;
      if (A.prio_mode) {  // wave priorities rotate over the workgroups of a CU by age: see fill_affine_kernel
        const int lvl = ((g >> 7) + (int)(blockIdx.x >> 8)) & 3;
        if (lvl == 0) __builtin_amdgcn_s_setprio(0);
        else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
        else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
      }
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
    const int pad_idx = L < W ? L : (L >= R * W ? W + (L - R * W) : 64);
    const bool pad_lane = !LEAN && pad_idx < R_::SLP - R_::SL;  // spare lanes owning a pad slot (Rec::SLP)
    const bool do_store = __builtin_amdgcn_ballot_w64(tile_act && !ghost) != 0 &&
                          ((live && (LEAN ? il == R - 1 : !ghost)) || pad_lane) &&
                          (T == 1 || rec <= rec_last);  // see the affine kernel
    if (__builtin_amdgcn_ballot_w64(do_store) != 0) vm_younger += GF::STORES_PER_STEP;
    if (do_store) {
      const int slot = LEAN ? aa : (pad_lane ? R_::SL + pad_idx : L - W);
      int32_t* dst = sto + (int64_t)rec * RECDW;
#pragma unroll
      for (int c = 0; c < NCH4; ++c) {
        v4i v;
        v.x = outv[4 * c]; v.y = outv[4 * c + 1]; v.z = outv[4 * c + 2]; v.w = outv[4 * c + 3];
        store_chunk<XCU>(dst + c * R_::CH + slot * 4, v, true);
      }
      if (LEAN) {
#pragma unroll
        for (int t = 0; t < TAIL; ++t) {
          if (XCU)
            __hip_atomic_store(dst + NCH4 * R_::CH + slot * TAIL + t, outv[4 * NCH4 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            dst[NCH4 * R_::CH + slot * TAIL + t] = outv[4 * NCH4 + t];
        }
      }
    }
    if (!LEAN && __builtin_amdgcn_ballot_w64(tile_act && !ghost) != 0 && (T == 1 || rec <= rec_last)) {
      // tail by all 64 lanes (Rec::TAILSLOTS), as in the affine kernel
      const int tslot = (live && !ghost) ? L - W : R_::SL + (L < W ? L : W + (L - R * W));
      int32_t* dst = sto + (int64_t)rec * RECDW;
#pragma unroll
      for (int t = 0; t < TAIL; ++t) {
        if (XCU)
          __hip_atomic_store(dst + NCH4 * R_::CH + tslot * TAIL + t, outv[4 * NCH4 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          dst[NCH4 * R_::CH + tslot * TAIL + t] = outv[4 * NCH4 + t];
      }
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
    if (L == 0) prog_put(0x7fffffff);
  }
}

}  // namespace bialign
