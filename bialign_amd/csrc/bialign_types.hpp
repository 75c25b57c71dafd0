// bialign_types.hpp -- constants, batch descriptors, sweep geometry and record layout.  Part of bialign_kernels.hpp (include that, not this).
#pragma once

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
  int32_t wide_s;       // wide-band path (max_shift beyond the tiled kernels): the band half-width
  int32_t spin_limit;   // team hand-off: polls of the partner's progress word before a wave gives up (error flag)
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
