// bialign_traceback.hpp -- tracebacks of both recurrences.  Part of bialign_kernels.hpp (include that, not this).
#pragma once

namespace bialign {

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
  // four DPP row rotations (a row = 16 lanes) instead of four trips through the LDS crossbar
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x128 /* row_ror:8 */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124 /* row_ror:4 */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x122 /* row_ror:2 */, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x121 /* row_ror:1 */, 0xf, 0xf, false));
  return v;
}

//   STRIP (lean traceback): the walk continues from the pair's
//   TraceState through ONE strip -- the one fill_affine_kernel<.., RESW> has just re-swept into
//   the scratch records -- and stops when it steps into the strip above (whose bottom row, the
//   only row of it a candidate can touch from here, is in the LEAN records) or ends.
//   WIDE (max_shift beyond the tiled kernels, bialign_wide.hpp): the band half-width is the runtime
//   value A.wide_s and the layers lie in the reference's own order; S is then a dummy (0).
//   PACK: the sweep stored packed records in interior steps (Pack<S>).
template <int S, bool DO_TRACE, bool STRIP = false, bool WIDE = false, bool PACK = false>
__global__ void __launch_bounds__(64) traceback_affine_kernel(const DeviceBatch A, int npairs) {
  static_assert(!WIDE || !STRIP, "no lean traceback on the wide-band path");
  static_assert(!PACK || (!WIDE && !STRIP), "packed records: full-storage tiled sweeps");
  const int SR = WIDE ? A.wide_s : S;  // band half-width
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
    if (WIDE) return lay[pd.layer_off + wide_dword(m, 2 * SR + 1, 9, pi, pj, a, b, ss)];
    if (PACK) return packed_cell<S>(lay, pd, pi, pj, a, b, ss);
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
    const int endv = c < 9 ? cell(n, m, SR, SR, c) : -BIG;
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
    const bool ok = c < 15 && pi >= 0 && pj >= 0 && pk >= 0 && pl >= 0 && abs(pk - pi) <= SR &&
                    abs(pl - pj) <= SR;  // pyx:133-141
    const int ld = ok ? cell(pi, pj, pk - pi + SR, pl - pj + SR, ss) : 0;
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

// Non-affine traceback (pyx:513-531): the first case, in generator order, that is
// guard-valid and reproduces the cell; stops when none does (the origin).  One wave
// per pair, lane c < 13 = case c; "first" = wave-min over the matching lane ids.
template <int S, bool DO_TRACE, bool STRIP = false, bool WIDE = false>  // STRIP, WIDE: see traceback_affine_kernel
__global__ void __launch_bounds__(64) traceback_linear_kernel(const DeviceBatch A, int npairs) {
  static_assert(!WIDE || !STRIP, "no lean traceback on the wide-band path");
  const int SR = WIDE ? A.wide_s : S;
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
    if (WIDE) return lay[pd.layer_off + wide_dword(m, 2 * SR + 1, 1, pi, pj, a, b, 0)];
    if (!STRIP) return lay[cell_dword<S, 1>(pd, pi, pj, a, b, 0)];
    const int sp = pi / RR, ilp = pi - sp * RR + 1;
    if (sp >= Qlo)
      return A.scratch[pd.scratch_off + (Q - sp) * sstride + Rec<S, 1>::dword(pj + 2 * ilp + a, (ilp - 1) * W + a, b)];
    return lay[pd.layer_off + Rec<S, 1, true>::dword((int64_t)sp * pd.P + pj + 2 * ilp + a, a, b)];
  };
  int cur = (STRIP && ts.started) ? ts.cur : cell(n, m, SR, SR);
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
    const bool ok = c < 13 && pi >= 0 && pj >= 0 && pk >= 0 && pl >= 0 && abs(pk - pi) <= SR && abs(pl - pj) <= SR;
    const int ld = ok ? cell(pi, pj, pk - pi + SR, pl - pj + SR) : 0;
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

}  // namespace bialign
