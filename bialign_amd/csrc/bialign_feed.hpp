// bialign_feed.hpp -- ghost-row and dense-mu2 feeds (block LDS-DMA).  Part of bialign_kernels.hpp (include that, not this).
#pragma once

namespace bialign {

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
//   BLKO: 0 = the default block length of this max_shift, else the block length (the diet
//   variant of the s=2 kernel halves its ring to fit eight waves' arrays into one CU's LDS).
template <int S, int NL, bool LEAN = false, int BLKO = 0>
struct GhostFeed {
  using R_ = Rec<S, NL, LEAN>;
  static constexpr int W = 2 * S + 1, R = 64 / W;
  static constexpr int NP = R_::NCH4 + (R_::TAIL ? 1 : 0);  // 16-byte pieces per (step, a)
#ifdef BIALIGN_BLK_OVERRIDE
  static constexpr int BLK = BIALIGN_BLK_OVERRIDE;
#else
  static constexpr int BLK = BLKO ? BLKO : (S <= 1 ? 8 : 4);  // steps per prefetch block (the ring is 2*BLK*W*NP*16 bytes of LDS)
#endif
  static constexpr int NPIECE = BLK * W * NP;
  // Order of the pieces in a ring half.  s <= 1: [step][a][piece] (a lane's record contiguous; the in-place unpacking of the
  // s=1 sweeps and the slim kernel's dword reads rely on it; 7 pieces = 28 dwords apart, the W = 3 ghost lanes fall into
  // different banks).  s >= 2: [piece][step][a] -- the W lanes of a ghost row read W neighbouring 16-byte pieces; with the
  // record contiguous they were NP * 4 dwords apart, 48 at s=2 (a = 0 and a = 4 in the same banks) and 64 at s=3 (all seven in
  // the same banks: every ghost read of the wave took seven passes).
  static constexpr bool PIECE_MAJOR = S >= 2;
  __host__ __device__ static constexpr int slot(int t, int aa, int c) {
    return PIECE_MAJOR ? (c * BLK + t) * W + aa : (t * W + aa) * NP + c;
  }
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
      int t, aa, c;
      if (PIECE_MAJOR) {
        c = q / (BLK * W);
        const int rem = q - c * (BLK * W);
        t = rem / W;
        aa = rem - t * W;
      } else {
        t = q / (W * NP);
        const int rem = q - t * (W * NP);
        aa = rem / NP;
        c = rem - aa * NP;
      }
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
  // The same for a sweep with packed records (Pack<S>): the source of a piece is the packed record of an
  // interior step (NPC pieces per lane; the surplus lanes of a round re-read piece 0) or the full record of
  // any other step in the pair's second region (starting at dword bnd_off).  Not used by re-sweeps (Qbase = 0).
  __device__ static __forceinline__ void issue_packed(const int32_t* lay, int64_t bnd_off, int m, int h0, int blk_q,
                                                      int blk_rem, int P, int T, int w, int rec_last, int lane,
                                                      uint32_t lds_base) {
    using PK = Pack<S>;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int q = min(r * 64 + lane, NPIECE - 1);
      int t, aa, c;
      if (PIECE_MAJOR) {
        c = q / (BLK * W);
        const int rem = q - c * (BLK * W);
        t = rem / W;
        aa = rem - t * W;
      } else {
        t = q / (W * NP);
        const int rem = q - t * (W * NP);
        aa = rem / NP;
        c = rem - aa * NP;
      }
      const int xr = blk_rem + t - aa;
      const int ql = blk_q + (xr >= P ? 1 : 0) - (xr < 0 ? 1 : 0);
      const int jj = xr - (xr >= P ? P : 0) + (xr < 0 ? P : 0);  // the ghost lane's column
      const int ts = jj + 2 * (R - 1) + aa;                      // the bottom lane row of the strip above was there at this t
      const int cs = ts >= P ? ts - P : ts, qs = ql * T + w - 1 + (ts >= P ? 1 : 0);
      const int64_t rec = (int64_t)(ql * T + w - 1) * P + ts;
      const bool valid = rec >= 0 && rec <= rec_last;
      const int sl = (R - 2) * W + aa;  // storage slot of the bottom real row
      const int32_t* p;
      if (valid && PK::interior(qs, cs, m))  // pieces 0 .. NCH-1: the 16-byte chunks; piece NCH: the tail (TAILDW dwords, read 16 bytes wide)
        p = lay + rec * PK::RECDW + (c < PK::NCH ? c * R_::CH + sl * 4 : (c == PK::NCH && PK::TAILDW ? PK::NCH * R_::CH + sl * PK::TAILDW : sl * 4));
      else
        p = lay + bnd_off + (valid ? PK::bidx(qs, cs, P, m) : 0) * R_::RECDW +
            (c < R_::NCH4 ? c * R_::CH + sl * 4 : R_::NCH4 * R_::CH + sl * R_::TAIL);
      const uint32_t dst = lds_base + r * 1024;
      uint32_t keep;
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
    else if (younger > 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (younger > 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger > 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger > 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (younger > 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (younger > 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  // the first NPK pieces only (a packed lane record)
  template <int NPK>
  __device__ static __forceinline__ void fetch_pieces(int (&out)[4 * NPK], const v4i* half, int t, int aa) {
#pragma unroll
    for (int c = 0; c < NPK; ++c) {
      const v4i v = half[slot(t, aa, c)];
      out[4 * c] = v.x; out[4 * c + 1] = v.y; out[4 * c + 2] = v.z; out[4 * c + 3] = v.w;
    }
  }

  __device__ static __forceinline__ void fetch(int (&out)[R_::ND], const v4i* half, int t, int aa) {
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      const v4i v = half[slot(t, aa, c)];
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

}  // namespace bialign
