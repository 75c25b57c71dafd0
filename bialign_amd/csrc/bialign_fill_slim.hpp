// bialign_fill_slim.hpp -- the affine sweep at THREE waves per SIMD.  Part of bialign_kernels.hpp (include that, not this).
//
// Why.  tools/valu_rate.hip (profiles/r03b_valu_rate): on gfx950 one wave issues a vector instruction every 4.75 cycles
// at best, whatever runs beside it, and a SIMD takes three such streams before the per-wave interval grows (v_add_u32:
// 4.75 / 4.75 / 4.75 / 7.8 cycles per instruction with 1 / 2 / 3 / 4 waves on the SIMD; v_max_i32, every VOP3 form, DPP:
// 5.1 / 8.9 / 8.9 / 12.9).  The packed sweep of fill_affine_kernel issues ~380 vector instructions per step and holds 188
// registers = two waves per SIMD: both of its waves issue at the two-wave rate, the third slot is empty
// (profiles/r03a_headline_baseline: VALU 40 % of a wave's cycles, HBM at 0.57 of peak).  A third wave costs the other two
// nothing per instruction -- if it fits: <= 168 registers and a third of the CU's LDS less per wave.
//
// How it fits (same lattice mapping, same records, same ghost feed, same results as fill_affine_kernel<S,true,TW,...,PACK>):
//  * Rows i-1 reach a lane through ds_bpermute_b32 -- the LDS crossbar, no LDS storage -- instead of the per-wave
//    exchange array [12 W][65] (9.4 KB at s=1): a three-wave workgroup then needs 3 x 6 KB of ghost ring + 6.5 KB of
//    tables and codes.  The exchange happens where the early fetch of fill_affine_kernel sits: at the end of a point,
//    into the registers its consumers of this step have just left.  Where a source must read as the sentinel (a+1
//    outside the band) the lane addresses lane 63, which owns no lattice point: it runs as a ghost lane whose "stored
//    layers" are a block of sentinels, so everything it derives is the sentinel (beta <= 0) -- no select per value.
//  * A ghost lane's layers are fetched from the ring nine at a time, inside the branch only ghost lanes take, not 27
//    up front by every lane; a packed lane record leaves as soon as a 16-byte chunk of it is complete.
//  * A workgroup is TWELVE waves = PPW pairs x teams of TW waves (PPW x TW = 12: 12 x 1, 6 x 2, 4 x 3, 2 x 6, 1 x 12), one
//    workgroup per CU.  A workgroup's waves are dealt round-robin over the four SIMDs, so every SIMD gets exactly
//    three; three-wave workgroups of one pair each measured 35 % SLOWER than twelve one-wave workgroups per CU
//    (profiles/r03d_slim: 16.4 vs 12.2 ms per 1024 pairs at len 512 -- their waves pile onto three of the four SIMDs).
//    The score tables are staged once per CU, each pair's codes once.
// Instantiated for LOOKUP scores, beta <= 0, packed records.
#pragma once

namespace bialign {

//   LEAN (score-only batches and the sweep of the memory-lean traceback): the step keeps only what the next strip's
//   ghost row replays -- the bottom real row, as plain int32 records (Rec<S,9,true>) -- and the lane that computes
//   (n,m,n,m) writes the score.  No stores to speak of, so this form is where the third wave pays most.
template <int S, int TW, int PPW, bool LEAN = false>
__global__ void __launch_bounds__(64 * TW * PPW, 3) fill_affine_slim_kernel(const DeviceBatch A) {
  static_assert(S >= 1, "packed records need a band");
  using PK_ = Pack<S>;
  using G_ = Geo<S>;
  using R_ = Rec<S, 9, LEAN>;
  constexpr int W = G_::W, R = G_::R, RR = G_::RR, PADB = G_::PADB;
  constexpr int ND = R_::ND, NCH4 = R_::NCH4, TAIL = R_::TAIL, RECDW = R_::RECDW;
  static_assert(R * W < 64, "lane 63 must be free: it is the sentinel source of the exchange");
  extern __shared__ __align__(16) int32_t smem[];

  constexpr int T = TW;                   // team size
  constexpr int NW = TW * PPW;            // waves per workgroup
  const int L = threadIdx.x & 63;
  const int wv = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave in workgroup
  const int pw = wv / TW, w = wv - pw * TW;            // pair of the workgroup, wave in its team
  const int slot = blockIdx.x * PPW + pw;              // pair of this launch
  const bool have_pair = slot < A.launch_pairs;
  const int pid = A.order[have_pair ? slot : 0];
  const PairDesc pd = A.pairs[pid];
  const int n = pd.n, m = pd.m, P = pd.P;
  const int il = L / W, aa = L - il * W;
  const bool live = L < R * W;
  const bool ghost = (il == 0);
  // lanes whose layers come out of LDS: the ghost row, and lane 63 (sentinels).  Kept as a laundered register: as a
  // predicate the compiler can see through (L < W || L >= R W) it is re-derived at every point of every step -- three
  // compares and two nested exec regions where one compare and one region do
  int ringfed_v = (ghost || !live) ? 1 : 0;
  asm volatile("" : "+v"(ringfed_v));
  const bool ringfed = ringfed_v != 0;
  int storing_v;  // lanes that own storage slots: the real rows (LEAN: the bottom one) and, in full records, the pad lanes
  {
    const int pad_idx0 = L < W ? L : (L >= R * W ? W + (L - R * W) : 64);
    storing_v = ((live && (LEAN ? il == R - 1 : !ghost)) || (!LEAN && pad_idx0 < R_::SLP - R_::SL)) ? 1 : 0;
  }
  asm volatile("" : "+v"(storing_v));
  const bool storing = storing_v != 0;
  const int beta = A.beta, gamma = A.gamma, delta = A.delta;
  const int k1 = A.k1, k2 = A.k2;
  const int gD = gamma + delta, gg = 2 * gamma, ggdd = 2 * gamma + 2 * delta, dd = 2 * delta;

  // ---- LDS carve-up: per wave a ghost ring; shared: a block of sentinels, progress words, score tables; per pair its codes
  using GF = GhostFeed<S, 9, LEAN, 0>;
  constexpr int PERW = GF::RING_DW;
  constexpr int SENTBLK = 4 * GF::NP;                                 // one (step, a) entry worth of sentinels
  v4i* ring = reinterpret_cast<v4i*>(smem + wv * GF::RING_DW);        // ghost-row ring, two halves
  int32_t* sentblk = smem + NW * PERW;
  volatile int32_t* prog_lds = smem + NW * PERW + SENTBLK + pw * TW;  // [16]: this team's words
  int32_t* s1 = smem + NW * PERW + SENTBLK + 16;                      // [k1*k1]
  int32_t* s2 = s1 + k1 * k1;                                         // [k2*k2]
  uint8_t* codes = reinterpret_cast<uint8_t*>(s2 + k2 * k2);          // per pair: seq A, cls A, seq B, cls B (A.slim_code_bytes each pair)
  const int npad = (n + 3) & ~3, mpad = (m + 2 * PADB + 3) & ~3;
  uint8_t* sa = codes + (size_t)pw * A.slim_code_bytes;    // seq A codes, [i-1]
  uint8_t* ca = sa + npad;                                  // cls A,       [k-1]
  uint8_t* sb = ca + npad;                                  // seq B codes, [j-1+PADB]
  uint8_t* cb = sb + mpad;                                  // cls B,       [l-1+PADB]

  for (int t = threadIdx.x; t < NW * PERW + SENTBLK; t += 64 * NW) smem[t] = SENT;
  if (threadIdx.x < 16) smem[NW * PERW + SENTBLK + threadIdx.x] = 0;
  for (int t = threadIdx.x; t < k1 * k1; t += 64 * NW) s1[t] = A.s1[t];
  for (int t = threadIdx.x; t < k2 * k2; t += 64 * NW) s2[t] = A.s2[t];
  if (have_pair) {  // each team stages its pair's codes
    const int tt = threadIdx.x - pw * (64 * TW);
    for (int t = tt; t < n; t += 64 * TW) {
      sa[t] = A.seq_a[pd.seq_a + t];
      ca[t] = A.cls_a[pd.seq_a + t];
    }
    for (int t = tt; t < m + 2 * PADB; t += 64 * TW) {
      const int src = t - PADB;
      const bool ok = src >= 0 && src < m;
      sb[t] = ok ? A.seq_b[pd.seq_b + src] : 0;
      cb[t] = ok ? A.cls_b[pd.seq_b + src] : 0;
    }
  }
  __syncthreads();
  if (!have_pair) return;  // (the last workgroup of a launch may hold fewer pairs)

  // ---- per-lane constants
  // ds_bpermute addresses (4 x source lane): (i-1, a) = lane L - W; (i-1, a+1) = lane L - W + 1, or -- where a+1 leaves
  // the band -- lane 63, whose derived values are all sentinels.  Ghost lanes and lane 63 read whatever: nobody uses it.
  const int addrA = ((L - W) & 63) * 4;
  const int addrB = (live && aa < W - 1 ? ((L - W + 1) & 63) : 63) * 4;
  // lane L-1 = (i, a-1), or lane 63 (sentinels) where a-1 leaves the band
  const int addrC = ((live && aa > 0) ? L - 1 : 63) * 4;
  const bool a_first = (aa == 0);
  const int lane_cap = a_first ? SENT : 0x7fffffff;  // min() with it = "sentinel where a-1 leaves the band"
  int32_t* const lay = A.layers + pd.layer_off;
  const int64_t pk_bnd_off = (int64_t)pd.G * PK_::RECDW;  // the pair's full records follow its packed ones
  const int rec_last = pd.G - 1;
  const int NSw = (pd.NS - w + T - 1) / T;  // strips of this wave: w, w+T, ...
  const int H = NSw > 0 ? (NSw - 1) * P + m + G_::MAXOFF + 1 : 0;

  // ---- per-lane sweep state
  int jj = -(2 * il + aa);  // column of this step (< 0: not started)
  int strip = 0;            // local strip index q; lattice strip = q*T + w
  int rec_base = w * P;     // record of local step h for this lane = h + rec_base
  int i = 0, s1row = 0, s2row = 0;
  bool act_row = false;
  auto set_row = [&](int q) {
    i = (q * T + w) * RR + il - 1;
    const int k = i + aa - S;
    act_row = live && i >= 0 && i <= n && k >= 0 && k <= n;
    s1row = (i >= 1 && i <= n) ? sa[i - 1] * k1 : 0;
    s2row = (k >= 1 && k <= n) ? ca[k - 1] * k2 : 0;
  };
  set_row(0);

  // delay lines (values exchanged one step after production, used later)
  int dA1[2][W], dA2[2][W];  // GMM, GMX from (i-1,a): used at age 3
  int dAx[2][W];             // GXM, GXX from (i-1,a): age 2
  int dB[4][W];              // GMY, H3M[0..2] from (i-1,a+1): age 2
  int dC[2][W];              // GYM, GYX from (i,a-1): age 2
  int selfv[4][W];           // GYY, H3Y[0..2] of this lane's previous column
  int pubC[W][8];            // GYM, GYX, H2M[0..2], H2X[0..2] of the previous column, for lane L+1 (DPP form only)
  int inC[W][8];             // ... as received from lane L-1 (bpermute form: loop-carried, exchanged at the end of a step)
  int inA[W][4], inB[W][8];  // what (i-1,a) and (i-1,a+1) derived one step ago (loop-carried: exchanged at the end of a step)
#pragma unroll
  for (int bb = 0; bb < W; ++bb) {
    dA1[0][bb] = dA1[1][bb] = dA2[0][bb] = dA2[1][bb] = SENT;
    dAx[0][bb] = dAx[1][bb] = dC[0][bb] = dC[1][bb] = SENT;
#pragma unroll
    for (int x = 0; x < 4; ++x) dB[x][bb] = selfv[x][bb] = inA[bb][x] = SENT;
#pragma unroll
    for (int x = 0; x < 8; ++x) pubC[bb][x] = inB[bb][x] = inC[bb][x] = SENT;
  }
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int32_t*)smem) + wv * GF::RING_DW * 4;

  // ---- team protocol (as fill_affine_kernel, in-workgroup form)
  int blk_q = 0, blk_rem = 0;  // (next block start) div / mod P
  int seen_prog = -0x40000000;  // the partner's progress as last read (INT_MAX once a hand-off has timed out: no further waits)
  auto wait_partner = [&](int h_last) __attribute__((always_inline)) {
    if (T == 1) return;
    const int src = w == 0 ? T - 1 : w - 1;
    const int need = h_last + 2 * (R - 1) + 1 - (w == 0 ? P : 0);
    if (seen_prog >= need) return;
    for (int spin = 0; (seen_prog = prog_lds[src]) < need; ++spin) {
      if (spin > A.spin_limit) {
        if (L == 0) atomicOr(A.errflag, 1);
        seen_prog = 0x7fffffff;
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
  };
  auto prefetch_block = [&](int h0, int half) __attribute__((always_inline)) {
    wait_partner(h0 + GF::BLK - 1);
    // (the lane number is laundered: everything the feed derives from it -- piece, band row, chunk, 64-bit source bases
    //  for three rounds -- would otherwise be hoisted out of the sweep and sit in a dozen registers the step needs)
    int Lv = L;
    asm volatile("" : "+v"(Lv));
    if (LEAN) GF::issue(lay, h0, blk_q, blk_rem, P, T, w, P - 2 * (R - 1), rec_last, Lv, ring_lds + half * GF::SLOTS * 16);
    else GF::issue_packed(lay, pk_bnd_off, m, h0, blk_q, blk_rem, P, T, w, rec_last, Lv, ring_lds + half * GF::SLOTS * 16);
    blk_rem += GF::BLK;
    if (blk_rem >= P) { blk_rem -= P; ++blk_q; }
  };
  prefetch_block(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int vm_younger = 0;
  int pk_all = 0;  // OR of every offset this lane has stored since the last range check

  // delay-line stage, opaque to the compiler (see fill_affine_kernel)
  auto dmov = [](int& dst, int src) __attribute__((always_inline)) { asm("v_mov_b32 %0, %1" : "=v"(dst) : "v"(src)); };
  auto bperm = [](int addr, int v) __attribute__((always_inline)) -> int { return __builtin_amdgcn_ds_bpermute(addr, v); };

  // score inputs one step ahead, codes two (see fill_affine_kernel)
  int mu1n = 0, mu2n[W], sbn = 0, cbn = 0;
  auto lookup_mu = [&]() __attribute__((always_inline)) {
    const int jc = min(max(jj, 0), m + 1);
    mu1n = s1[s1row + sb[jc - 1 + PADB]];
#pragma unroll
    for (int bb = 0; bb < W; ++bb) mu2n[bb] = s2[s2row + cb[jc + bb]];
  };
  auto fetch_codes = [&]() __attribute__((always_inline)) {
    const int jc1 = min(max(jj + 1, 0), m + 1);
    sbn = sb[jc1 - 1 + PADB];
    cbn = cb[jc1 + W - 1];
  };
  lookup_mu();
  fetch_codes();

  auto step = [&](auto interior_tag, int g) __attribute__((always_inline)) {
    constexpr bool INTERIOR = decltype(interior_tag)::value;
    // ---- 0. ghost feed, block boundary work
    const int gt = g & (GF::BLK - 1), ghalf = (g / GF::BLK) & 1;
    if (gt == 0) {
      GF::wait_block(vm_younger);
      if (TW > 1 && L == 0) prog_lds[w] = g - GF::BLK;
      if (!LEAN && (g & 15) == 0) {  // every 16 steps: an offset since then that does not fit 16 bits (or collides with the -2^30 mark): the host falls back
        const bool bad = live && !ghost && (unsigned)pk_all > 0xffffu;  // (lanes that hold lattice points in interior steps)
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && L == 0) atomicOr(A.errflag, 2);
        pk_all = 0;
      }
      if (!LEAN) {  // the block that has just landed: lane t*W + a unpacks its entry in place (fill_affine_kernel, PK_COOP)
        const int c0 = __builtin_amdgcn_readfirstlane(jj), q0 = __builtin_amdgcn_readfirstlane(strip);
        if (L < GF::BLK * W) {
          const int t = L / W, ai = L - t * W;
          int ph = c0 + t, qst = q0 * T + w;
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
      prefetch_block(g + GF::BLK, ghalf ^ 1);
      vm_younger = 0;
      if (A.prio_mode) {  // rotate four priority levels over the workgroups of a CU by age (fill_affine_kernel)
        const int lvl = ((g >> 7) + (wv >> 2)) & 3;  // (waves wv, wv + 4, wv + 8 share a SIMD)
        if (lvl == 0) __builtin_amdgcn_s_setprio(0);
        else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
        else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
      }
    }
    // where this lane's stored layers lie in LDS: its (step, a) entry of the ring, or the sentinel block (lane 63)
    const int32_t* const gsrc = live ? reinterpret_cast<const int32_t*>(ring + ghalf * GF::SLOTS + (gt * W + aa) * GF::NP)
                                     : sentblk;

    // ---- 1. lane L-1 = (i, a-1) hands its values over in registers: DPP wave shift fused with the cap (s_nop 1: the
    //         wait states a DPP read needs, see fill_affine_kernel)
    auto read_dpp = [&](int r) __attribute__((always_inline)) {
      if (!BIALIGN_SLIM_DPP) return;
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
    read_dpp(0);

    // ---- 2. score inputs of this column
    const int mu1 = mu1n;
    int mu2[W];
#pragma unroll
    for (int bb = 0; bb < W; ++bb) mu2[bb] = mu2n[bb];

    const bool tile_act = INTERIOR ? true : (act_row && jj >= 0 && jj <= m);
    const bool is_origin = INTERIOR ? false : (tile_act && i == 0 && jj == 0 && aa == S);
    const int c3M = mu1 + dd, c_Mg = mu1 + gD;

    // ---- layer stores (as fill_affine_kernel: every real lane stores every active step, spare lanes fill the pad slots)
    const int rec = g + rec_base;
    const int pad_idx = L < W ? L : (L >= R * W ? W + (L - R * W) : 64);
    const bool pad_lane = pad_idx < R_::SLP - R_::SL;
    constexpr bool PACKED = INTERIOR && !LEAN;  // this step writes a packed record
    // (in an interior step every lane's record exists -- rec <= rec_last -- so who stores is a lane constant)
    const bool do_store = INTERIOR ? storing
                                   : (((live && (LEAN ? il == R - 1 : !ghost)) || (!LEAN && pad_lane)) &&
                                      __builtin_amdgcn_ballot_w64(tile_act && !ghost) != 0 && (TW == 1 || rec <= rec_last));
    const int slot_ = LEAN ? aa : (pad_lane ? R_::SL + pad_idx : L - W);
    if (INTERIOR) vm_younger += PACKED ? PK_::NPC : GF::STORES_PER_STEP;  // (an interior step always stores)
    else if (__builtin_amdgcn_ballot_w64(do_store) != 0) vm_younger += GF::STORES_PER_STEP;
    int32_t* dst = lay + (int64_t)rec * RECDW;  // LEAN: the one region; else (boundary steps) the full record in the pair's second region
    int32_t* const dstp = lay + (int64_t)rec * PK_::RECDW;  // interior steps: the packed record
    if (!INTERIOR && !LEAN) {
      const int tl = jj + 2 * il + aa;
      const int over = tl >= P ? 1 : 0;
      dst = lay + pk_bnd_off + PK_::bidx(strip * T + w + over, tl - over * P, P, m) * RECDW;
    }

    // ---- 3. the W lattice points of this (i, j, a)
    int outv[PACKED ? 1 : ND];
    int pk_base = 0, pk_min = 0, pk_max = 0, pk_e[PACKED ? ND : 1];  // packed records: base, running extremes, the values
    int h2y[3] = {SENT, SENT, SENT};
    int defer[3] = {SENT, SENT, SENT};  // GXM, GXX, GXY of the previous point: their registers are busy for one more point
    int deferC[3] = {SENT, SENT, SENT}; // H2M[0..2] of the previous point, likewise
#pragma unroll
    for (int bb = 0; bb < W; ++bb) {
      if (bb + 1 < W) read_dpp(bb + 1 < W ? bb + 1 : 0);
      const int l = jj + bb - S;
      const bool act = INTERIOR ? true : (tile_act && l >= 0 && l <= m);
      const int mu2v = mu2[bb];
      const int c_MM = mu1 + mu2v, c_gM = mu2v + gD, c2M = mu2v + dd;

      auto cases = [&](int (&Tv)[9]) __attribute__((always_inline)) {
#pragma unroll
        for (int hU = 0; hU < 3; ++hU) {
#pragma unroll
          for (int hV = 0; hV < 3; ++hV) {
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
            const int c1 = (hU == 2 && hV == 2) ? c_MM : (hU == 2) ? c_Mg : (hV == 2) ? c_gM : (hU == hV) ? gg : ggdd;
            int h2in = SENT;
            bool ok2 = true;
            if (hV == 2) { ok2 = bb >= 1; if (ok2) h2in = inC[bb >= 1 ? bb - 1 : 0][2 + hU]; }
            if (hV == 1) h2in = inC[bb][5 + hU];
            if (hV == 0) { ok2 = bb >= 1; if (ok2) h2in = h2y[hU]; }
            const int c2 = (hV == 2) ? c2M : gD;
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
      // finalise: ring-fed lanes (ghost row; lane 63) take their stored layers -- nine values, fetched inside the
      // branch only they take -- the others compute; "no valid case" -> -2^30 (pyx:299-303)
      int M[9];
      bool isneg[9] = {};  // "no valid case" per corner state, as the finalisation found it (computing lanes)
      if (INTERIOR) {
        if (ringfed) {
#pragma unroll
          for (int q = 0; q < 9; ++q) M[q] = gsrc[bb * 9 + q];
        } else {
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
        int Gm[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) Gm[q] = SENT;
        if (ghost) {
#pragma unroll
          for (int q = 0; q < 9; ++q) Gm[q] = gsrc[bb * 9 + q];
        }
        int Tv[9];
        cases(Tv);
        const int low = act ? NEG : SENT;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const int tv = ghost ? Gm[q] : Tv[q];
          const bool bad = (tv < THRESH) | !act;
          M[q] = bad ? low : tv;
        }
        if (bb == S) M[8] = is_origin ? 0 : M[8];  // pyx:483-485
      }
      if (!PACKED) {
#pragma unroll
        for (int q = 0; q < 9; ++q) outv[PACKED ? 0 : bb * 9 + q] = M[q];
      }
      if (LEAN && bb == S) {  // score-only: the end cell (n,m,n,m) is all the host wants (pyx:509)
        if (live && !ghost && aa == S && i == n && jj == m) {
          int best = M[0];
#pragma unroll
          for (int q = 1; q < 9; ++q) best = imax(best, M[q]);
          A.scores[pid] = best;
        }
      }
      if (PACKED) {
        // packed record (Pack<S>): dword 0 = base, then the low halves of all values but the anchor; a piece of it leaves
        // as soon as its last value exists
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
            pk_e[PACKED ? bb * 9 + q : 0] = x;  // (the record takes the low half)
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
            const int dlast = 4 * c + 3 < NDWc - 1 ? 4 * c + 3 : NDWc - 1;  // last lane-record dword of piece c
            const int last = PK_::val(2 * dlast + 1);                        // ... and the last value it holds
            if (last >= bb * 9 && last < (bb + 1) * 9) {
              int dw[4];
#pragma unroll
              for (int x = 0; x < 4; ++x) {
                const int d = 4 * c + x;
                const int lo = d < NDWc ? PK_::val(2 * d) : 0, hi = d < NDWc ? PK_::val(2 * d + 1) : 0;
                dw[x] = d == 0 ? pk_base
                               : (d < NDWc ? (int)__builtin_amdgcn_perm((unsigned)pk_e[PACKED ? hi : 0], (unsigned)pk_e[PACKED ? lo : 0], 0x05040100u) : 0);
              }
              if (c < PK_::NCH) {
                v4i v;
                v.x = dw[0]; v.y = dw[1]; v.z = dw[2]; v.w = dw[3];
                *reinterpret_cast<v4i*>(dstp + c * R_::CH + slot_ * 4) = v;
              } else if (slot_ < PK_::TSLOTS) {  // the tail piece: TAILDW dwords per lane
                int32_t* tp = dstp + PK_::NCH * R_::CH + slot_ * PK_::TAILDW;
#pragma unroll
                for (int x = 0; x < PK_::TAILDW; ++x) tp[x] = dw[x];
              }
            }
          }
        }
      } else {
        if (do_store) {
#pragma unroll
          for (int c = 0; c < NCH4; ++c) {
            if (4 * c + 3 >= bb * 9 && 4 * c + 3 < (bb + 1) * 9) {  // chunk c completes with this point
              v4i v;
              v.x = outv[PACKED ? 0 : 4 * c]; v.y = outv[PACKED ? 0 : 4 * c + 1];
              v.z = outv[PACKED ? 0 : 4 * c + 2]; v.w = outv[PACKED ? 0 : 4 * c + 3];
              *reinterpret_cast<v4i*>(dst + c * R_::CH + slot_ * 4) = v;
            }
          }
          if (LEAN && bb == W - 1) {
#pragma unroll
            for (int t = 0; t < TAIL; ++t) dst[NCH4 * R_::CH + slot_ * TAIL + t] = outv[PACKED ? 0 : 4 * NCH4 + t];
          }
        }
        if (!LEAN && bb == W - 1) {  // the tail is stored by ALL 64 lanes (Rec::TAILSLOTS)
          const bool wave_stores = __builtin_amdgcn_ballot_w64(tile_act && !ghost) != 0 && (TW == 1 || rec <= rec_last);
          if (wave_stores) {
            const int tslot = (live && !ghost) ? L - W : R_::SL + (L < W ? L : W + (L - R * W));
#pragma unroll
            for (int t = 0; t < TAIL; ++t) dst[NCH4 * R_::CH + tslot * TAIL + t] = outv[PACKED ? 0 : 4 * NCH4 + t];
          }
        }
      }

      // derived values for the successors (beta <= 0: f_X(v) = max(v[X], beta + max3(v)))
      int H2[3][3], H3[3][3], Gd[3][3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        H2[u][2] = fM(M[3 * u], M[3 * u + 1], M[3 * u + 2]);
        const int bm = beta + H2[u][2];
        H2[u][0] = imax(M[3 * u], bm);
        H2[u][1] = imax(M[3 * u + 1], bm);
      }
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        H3[2][v] = fM(M[v], M[3 + v], M[6 + v]);
        Gd[2][v] = fM(H2[0][v], H2[1][v], H2[2][v]);
        const int bm3 = beta + H3[2][v], bmg = beta + Gd[2][v];
        H3[0][v] = imax(M[v], bm3);
        H3[1][v] = imax(M[3 + v], bm3);
        Gd[0][v] = imax(H2[0][v], bmg);
        Gd[1][v] = imax(H2[1][v], bmg);
      }
      if (PACKED) {
        pk_max = imax(pk_max, Gd[2][2]);
        if (bb == W - 1) {  // rows outside the lattice hold don't-care values; the OR runs on across steps, tested every 16
          const int acc = pk_all | (pk_max - pk_base + 1) | (pk_min - pk_base);
          pk_all = act_row ? acc : 0;
        }
      }
      if (BIALIGN_SLIM_DPP) {
      pubC[bb][0] = Gd[0][2];
      pubC[bb][1] = Gd[0][1];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        pubC[bb][2 + u] = H2[u][2];
        pubC[bb][5 + u] = H2[u][1];
      }
      }
      selfv[0][bb] = Gd[0][0];
#pragma unroll
      for (int v = 0; v < 3; ++v) selfv[1 + v][bb] = H3[0][v];
#pragma unroll
      for (int u = 0; u < 3; ++u) h2y[u] = H2[u][0];

      // delay lines: index bb (bb-1 for GXM/GXX) has served its last consumer of this step
      dmov(dA2[0][bb], dA1[0][bb]);
      dmov(dA2[1][bb], dA1[1][bb]);
      dmov(dA1[0][bb], inA[bb][0]);
      dmov(dA1[1][bb], inA[bb][1]);
      dmov(dB[0][bb], inB[bb][0]);
#pragma unroll
      for (int v = 0; v < 3; ++v) dmov(dB[1 + v][bb], inB[bb][2 + v]);
      if (BIALIGN_SLIM_DPP) {
        dC[0][bb] = inC[bb][0];
        dC[1][bb] = inC[bb][1];
      } else {
        // BIALIGN_SLIM_DPP=0 (measured, not shipped: 48.7 against 46.6 ms -- the LDS pipe has no room for 21 more
        // instructions per step): lane L-1's values travel like the rows'.  GYM, GYX wait one more step in dC; H2X is
        // used in this point only; H2M still serves the next point of this step and follows one point later
        dmov(dC[0][bb], inC[bb][0]);
        dmov(dC[1][bb], inC[bb][1]);
        inC[bb][0] = bperm(addrC, Gd[0][2]);
        inC[bb][1] = bperm(addrC, Gd[0][1]);
#pragma unroll
        for (int u = 0; u < 3; ++u) inC[bb][5 + u] = bperm(addrC, H2[u][1]);
        if (bb >= 1) {
#pragma unroll
          for (int u = 0; u < 3; ++u) inC[bb >= 1 ? bb - 1 : 0][2 + u] = bperm(addrC, deferC[u]);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) deferC[u] = H2[u][2];
      }
      // exchange (for the next step): what lanes (i-1, a) and (i-1, a+1) derived for this band column.  The registers
      // of GXM, GXX (inA[.][2..3]) and GXY (inB[.][1]) still serve the NEXT point of this step: those three follow
      // one point later.
      inA[bb][0] = bperm(addrA, Gd[2][2]);
      inA[bb][1] = bperm(addrA, Gd[2][1]);
      inB[bb][0] = bperm(addrB, Gd[2][0]);
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        inB[bb][2 + v] = bperm(addrB, H3[2][v]);
        inB[bb][5 + v] = bperm(addrB, H3[1][v]);
      }
      if (bb >= 1) {
        dmov(dAx[0][bb >= 1 ? bb - 1 : 0], inA[bb >= 1 ? bb - 1 : 0][2]);
        dmov(dAx[1][bb >= 1 ? bb - 1 : 0], inA[bb >= 1 ? bb - 1 : 0][3]);
        inA[bb >= 1 ? bb - 1 : 0][2] = bperm(addrA, defer[0]);
        inA[bb >= 1 ? bb - 1 : 0][3] = bperm(addrA, defer[1]);
        inB[bb >= 1 ? bb - 1 : 0][1] = bperm(addrB, defer[2]);
      }
      defer[0] = Gd[1][2];
      defer[1] = Gd[1][1];
      defer[2] = Gd[1][0];
    }
    dmov(dAx[0][W - 1], inA[W - 1][2]);
    dmov(dAx[1][W - 1], inA[W - 1][3]);
    inA[W - 1][2] = bperm(addrA, defer[0]);
    inA[W - 1][3] = bperm(addrA, defer[1]);
    inB[W - 1][1] = bperm(addrB, defer[2]);

    // ---- advance
    ++jj;
    if (!INTERIOR && jj == P) {  // (an interior step never ends a strip: its phase is at most m - S)
      jj = 0;
      ++strip;
      rec_base += (T - 1) * P;
      set_row(strip);
    }
    if (INTERIOR) {  // same row, next column, inside the molecule: the window slides by one
      mu1n = s1[s1row + sbn];
#pragma unroll
      for (int bb = 0; bb + 1 < W; ++bb) dmov(mu2n[bb], mu2[bb + 1]);
      mu2n[W - 1] = s2[s2row + cbn];
    } else {
      lookup_mu();
    }
    fetch_codes();
  };

  auto all_interior = [&]() __attribute__((always_inline)) {
    const int c0 = __builtin_amdgcn_readfirstlane(jj), q0 = __builtin_amdgcn_readfirstlane(strip);
    return PK_::interior(q0 * T + w, c0, m);
  };
  int g = 0;  // local step of this wave
  while (g < H) {
    while (g < H && !all_interior()) {
      step(BoolTag<false>{}, g);
      ++g;
    }
    // (a run counted once instead of tested per step makes hipcc spill 74 registers here, as it did in fill_affine_kernel)
    while (g < H && all_interior()) {
      step(BoolTag<true>{}, g);
      ++g;
    }
  }
  if (!LEAN && __builtin_amdgcn_ballot_w64(live && !ghost && (unsigned)pk_all > 0xffffu) != 0 && L == 0) atomicOr(A.errflag, 2);
  if (TW > 1) {  // everything this wave wrote is acknowledged: release the partner for good
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (L == 0) prog_lds[w] = 0x7fffffff;
  }
}

}  // namespace bialign
