// bialign_types.hpp -- constants, batch descriptors, sweep geometry and record layout.  Part of bialign_kernels.hpp (include that, not this).
#pragma once

namespace bialign {

constexpr int32_t NEG = -(1 << 30);                 // reference -infinity
constexpr int32_t SENT = -(1 << 30) - (1 << 29);    // "guard failed" marker
constexpr int32_t THRESH = -(1 << 30) - (1 << 28);  // below: no valid case
constexpr int NCOL = 65;                            // 64 lanes + 1 sentinel column
constexpr int PROG_WORDS = 256;                     // cross-CU teams: progress words per pair = largest team
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
  int32_t* prog;        // cross-CU teams: [pairs in launch][PROG_WORDS] progress words, zeroed per launch
  int32_t team;         // cross-CU teams: workgroups (= waves) per pair
  int32_t* scratch;     // lean traceback: full records of resw_k strips per pair
  TraceState* tstate;   // lean traceback: [npairs]
  int32_t resw_k;       // lean traceback: strips re-swept (in parallel) and walked per round
  int32_t wide_s;       // wide-band path (max_shift beyond the tiled kernels): the band half-width
  int32_t prio_mode;    // 1 = rotate wave priorities by workgroup age (fill_affine_kernel); BIALIGN_PRIO=0 switches it off
  int32_t spin_limit;   // team hand-off: polls of the partner's progress word before a wave gives up (error flag)
  int32_t* wide_ring;            // wide-band affine sweep: derived values of the last WIDE_RING levels, all pairs of the launch
  const int64_t* wide_ring_off;  // ... [pairs in launch]: dword offset of a pair's ring
  int32_t wide_score_only;       // ... 1: no layers are stored, the last level's point writes the score
  int32_t launch_pairs;     // fill_affine_slim_kernel: pairs of this launch (a workgroup holds several)
  int32_t slim_code_bytes;  // fill_affine_slim_kernel: LDS bytes of one pair's sequence and class codes
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
// [slot][TAIL] tail (64 slots, see TAILSLOTS), everything packed: a wave-wide store instruction writes one contiguous
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
  // A chunk holds SLP >= SL slots: where rounding SL up to a multiple of 8 costs at most two slots (s=2: 55 -> 56,
  // s=4: 54 -> 56) the spare lanes write them too and every chunk store covers whole 128-byte lines (-1..2 %).
  static constexpr int SLP = (!LEAN && (SL + 7) / 8 * 8 - SL <= BIALIGN_PADMAX) ? (SL + 7) / 8 * 8 : SL;
  static constexpr int CH = SLP * 4;           // dwords per chunk
  // The tail has 64 slots, not SL: the spare lanes (ghost row, idle lanes) store don't-care values into the
  // last ones, so that a record ends on a 64-lane boundary -- at s=1 (6 x 960 + 768 B) it is 51 whole
  // 128-byte lines and consecutive records of a wave never share a line (3 % of the fill time).
  static constexpr int TAILSLOTS = LEAN ? SL : 64;
  static constexpr int RECDW = LEAN ? (SL * ND + 3) / 4 * 4 : NCH4 * CH + TAILSLOTS * TAIL;  // 16-byte pieces stay aligned
  __host__ __device__ static inline int64_t dword(int64_t g, int slot, int d) {
    return d < 4 * NCH4 ? g * RECDW + (d >> 2) * CH + slot * 4 + (d & 3)
                        : g * RECDW + NCH4 * CH + slot * TAIL + (d - 4 * NCH4);
  }
};

// ---------------------------------------------------------------------------
// Packed records (affine sweeps at max_shift 1..3, full storage).  In INTERIOR steps -- every lane's
// lattice points have all four coordinates >= 1 and lie inside the molecule -- each of a lane's ND
// layer values is either exactly -2^30 at one of six compile-time-known (state, b) positions
// (can_be_empty<W>) or lies within a few thousand of the lane's first value, so the lane record shrinks
// from ND dwords to  base + (ND - 1) halfwords  (base = M[(1,1,1,1)] of the lane's first point - 0x8000, so
// THAT value's offset is 0x8000 by construction and is not stored).  A halfword is the LOW HALF OF THE VALUE ITSELF
// (round 3; round 2 stored the offset value - base) -- no subtraction per value in the sweep: the
// offset is (halfword - low half of base) mod 2^16, exact while 0 <= value - base < 2^16;
// offset 0xffff at a can_be_empty position = -2^30: 14 dwords instead of 27 at s=1 (round 2 stored all ND offsets: 16),
// 24 instead of 45 at s=2, 32 instead of 63 at s=3 (36 in round 2).  The sweep verifies the range of every offset it stores; the first one that does
// not fit raises the device flag and the host repeats the batch with full records.  All other steps
// (strip changes, the first strip(s), the lattice border) keep full records in a second region of the
// pair's storage.  Which steps are interior is a function of the record number alone, so readers
// (ghost feed, tracebacks, dump) find a cell without any index:
//   record r = Q*P + t,  t = j + 2*il + aa  (Q = strip of the lane, t may reach into the next period)
//   phase c = r mod P, step-strip Qs = r div P;   interior  <=>  Qs >= Q0  and  LO <= c <= HI
// Layout of a packed record: NCH chunks [chunk][slot][4 dwords] like a full record's, then -- where the
// lane record is not a whole number of 16-byte chunks -- a tail [slot][TAILDW dwords].
// ---------------------------------------------------------------------------
// the six (state, band column) positions whose value can be exactly -2^30 in an interior step (= can_be_empty<W>)
__host__ __device__ constexpr bool pack_corner(int W, int st, int bb) {
  return (bb == 0 && (st == 3 || st == 5 || st == 6)) || (bb == W - 1 && (st == 1 || st == 2 || st == 7));
}

template <int S>
struct Pack {
  using G_ = Geo<S>;
  using R_ = Rec<S, 9>;
  static constexpr int ND = R_::ND;
  static constexpr int ANCHOR = 8;                     // value index (band column 0, state (1,1,1,1)) the base is taken from
  static constexpr int NHW = ND + 1;                   // halfwords of a lane record: base (2) + the other ND - 1 offsets
  static constexpr int NDW = (NHW + 1) / 2;            // ... dwords
  // whole 16-byte chunks, and a tail piece of TAILDW dwords per lane: 8 bytes at s=1 (14 dwords = 3 chunks + 2), none
  // at s=3 (32 dwords); a 12-byte tail (s=2: 23 dwords) is rounded up to a chunk -- three scalar dword stores per
  // lane (write-through ones in cross-CU teams) cost far more than the four bytes (config 4: 329 vs 165 ms)
  static constexpr int NCH = NDW / 4 + (NDW % 4 == 3 ? 1 : 0);
  static constexpr int TAILDW = NDW % 4 == 3 ? 0 : NDW % 4;
  static constexpr int NPC = NCH + (TAILDW ? 1 : 0);   // 16-byte pieces the ghost feed moves per lane record
  // tail slots: the storing lanes', rounded up so that records stay 16-byte aligned
  static constexpr int TSLOTS = TAILDW == 0 ? 0 : (TAILDW == 2 ? (R_::SL + 1) / 2 * 2 : (R_::SL + 3) / 4 * 4);
  static constexpr int RECDW = NCH * R_::CH + TSLOTS * TAILDW;
  static_assert(RECDW % 4 == 0, "packed records must stay 16-byte aligned");
  // halfword of value v (v != ANCHOR) in the lane record, and the value a halfword h >= 2 holds
  __host__ __device__ static constexpr int hw(int v) { return 2 + v - (v > ANCHOR ? 1 : 0); }
  __host__ __device__ static constexpr int val(int h) { return h - 2 < ANCHOR ? h - 2 : h - 1; }
  // dword offset of lane-record dword d of storage slot `slot` inside a packed record
  __host__ __device__ static constexpr int dwpos(int slot, int d) {
    return d < 4 * NCH ? (d >> 2) * R_::CH + slot * 4 + (d & 3) : NCH * R_::CH + slot * TAILDW + (d - 4 * NCH);
  }
  static constexpr int LO = S + 1 + G_::MAXOFF;       // first phase at which every lane has passed column S
  static constexpr int Q0 = (S + 2 + G_::RR - 1) / G_::RR;  // first strip whose ghost row is row >= S+1
  __host__ __device__ static inline int hi(int m) { return m - S; }  // last phase whose points all lie inside the molecule
  __host__ __device__ static inline bool interior(int qs, int c, int m) { return qs >= Q0 && c >= LO && c <= m - S; }
  __host__ __device__ static inline int nbs(int P, int m) { return P - (m - S - LO + 1); }  // full records per strip >= Q0
  // index of the full record of a non-interior step among the pair's full records
  __host__ __device__ static inline int64_t bidx(int qs, int c, int P, int m) {
    if (qs < Q0) return (int64_t)qs * P + c;
    return (int64_t)Q0 * P + (int64_t)(qs - Q0) * nbs(P, m) + (c < LO ? c : c - (m - S + 1) + LO);
  }
  // dwords of the pair's full-record region; records 0 .. G-1
  __host__ __device__ static inline int64_t full_records(int G, int P, int m) {
    const int nst = (G + P - 1) / P;  // step-strips touched
    return nst <= Q0 ? (int64_t)nst * P : (int64_t)Q0 * P + (int64_t)(nst - Q0) * nbs(P, m);
  }
  // dwords of a pair's storage in packed form: G packed records (those of non-interior steps stay unused), then
  // its full records
  __host__ __device__ static inline int64_t pair_dwords(int G, int P, int m) {
    return (int64_t)G * RECDW + full_records(G, P, m) * R_::RECDW;
  }
  // dwords a packed sweep writes (about: idle steps at the very end write nothing in either form)
  __host__ __device__ static inline int64_t written_dwords(int G, int P, int m) {
    const int64_t f = full_records(G, P, m);
    return (G - f) * RECDW + f * R_::RECDW;
  }
  // A halfword holds the LOW HALF OF THE VALUE ITSELF (round 3; round 2 stored the offset value - base): the sweep packs
  // without a subtraction per value and checks the range through a running minimum and maximum.  Saves the s=1 sweeps
  // 2.5 % (47.1 -> 45.9 ms at the headline shape).  At s=2 and 3, where every lane unpacks the ghost row in every step,
  // it only pays with the two-at-a-time unpacking below (config-4 chunk: offsets 73.4 ms, low halves unpacked value by
  // value 75, two at a time 72.0).
  // offset of a stored halfword h against the record's base
  __host__ __device__ static inline uint32_t offset_of(uint32_t h, int base) { return (h - (uint32_t)base) & 0xffffu; }
  // ... and of both halfwords of a record dword at once (device code: one v_pk_sub_u16 for two values instead of a
  // subtraction and a mask each; the caller picks a half, which folds into the addition of the base as an SDWA operand)
  typedef unsigned short pk_u16x2 __attribute__((ext_vector_type(2)));
  __device__ static inline uint32_t offsets_of(uint32_t word, int base) {
    const pk_u16x2 w = __builtin_bit_cast(pk_u16x2, word);
    const unsigned short b = (unsigned short)base;
    return __builtin_bit_cast(uint32_t, (pk_u16x2)(w - pk_u16x2{b, b}));
  }
  // value (state st of band column bb) of a lane slot in a packed record at p
  __host__ __device__ static inline int decode(const int32_t* p, int slot, int bb, int st, bool corner) {
    const int v = bb * 9 + st;
    const int base = p[slot * 4];  // both loads issued together: one memory round trip per cell
    if (v == ANCHOR) return base + 0x8000;
    const int h = hw(v);
    const uint32_t word = (uint32_t)p[dwpos(slot, h >> 1)];
    const uint32_t e = offset_of((h & 1) ? word >> 16 : word & 0xffffu, base);
    return (corner && e == 0xffffu) ? NEG : base + (int)e;
  }
};

// Layer value (state st) of lattice point (i, j, aa, bb) of a pair swept with packed records.
template <int S>
__host__ __device__ inline int packed_cell(const int32_t* layers, const PairDesc& pd, int i, int j, int aa, int bb, int st) {
  using PK = Pack<S>;
  constexpr int W = 2 * S + 1, RR = Geo<S>::RR;
  const int strip = i / RR, il = i - strip * RR + 1;
  const int t = j + 2 * il + aa, over = t >= pd.P ? 1 : 0;
  const int slot = (il - 1) * W + aa;
  const int32_t* base = layers + pd.layer_off;
  if (PK::interior(strip + over, t - over * pd.P, pd.m))
    return PK::decode(base + ((int64_t)strip * pd.P + t) * PK::RECDW, slot, bb, st, pack_corner(W, st, bb));
  return base[(int64_t)pd.G * PK::RECDW + PK::bidx(strip + over, t - over * pd.P, pd.P, pd.m) * Rec<S, 9>::RECDW +
              Rec<S, 9>::dword(0, slot, bb * 9 + st)];
}

// dword index of layer value (state st) of lattice point (i, j, aa, bb).
template <int S, int NL>
__host__ __device__ inline int64_t cell_dword(const PairDesc& pd, int i, int j, int aa, int bb,
                                              int st) {
  constexpr int W = 2 * S + 1, RR = Geo<S>::RR;
  const int strip = i / RR, il = i - strip * RR + 1;
  const int64_t g = (int64_t)strip * pd.P + j + 2 * il + aa;
  return pd.layer_off + Rec<S, NL>::dword(g, (il - 1) * W + aa, bb * NL + st);
}

}  // namespace bialign
