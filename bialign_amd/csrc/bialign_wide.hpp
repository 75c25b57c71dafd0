// bialign_wide.hpp -- the recurrences for ANY max_shift (runtime band width).  Part of bialign_kernels.hpp.
//
// The tiled sweep (bialign_fill_affine.hpp / bialign_fill_linear.hpp) keeps a whole band row of
// a lattice point in one lane's registers, which ends at max_shift 5 (~470 registers).  The
// reference accepts any band width (pyx:25-35, bialign.py:83), so wider bands take this path:
// a plain anti-diagonal wavefront.  Every predecessor of pyx:225-296 lowers the coordinate sum
//   D = i + j + k + l
// by at least one, so all lattice points of one D are independent: one workgroup per pair walks
// D = 0 .. 2(n+m); within a level thread t takes (i, a, b) triples (j follows from D; only
// every other b has the right parity).  Layers live in HBM in the reference's own order
// [i][j][a][b][state] (pyx:38, state fastest), read back through L2 after a workgroup barrier
// with agent-scope fences.  Straightforward and slow next to the tiled sweep -- one CU per
// pair, every predecessor re-read from L2 -- but exact for any band, and a band this wide is
// ~170+ cells per (i,j): the reference needs hours for what this does in seconds.
#pragma once

namespace bialign {

// dword index of state st of lattice point (i, j, aa, bb) in a wide-band pair's region.  The nine states of the affine
// recurrence sit in a 12-dword cell: 16-byte aligned, so a point's layers leave as three dwordx4 stores (an
// unaligned dwordx4 store is not something to rely on: the first version with 9-dword cells read back garbage).
__host__ __device__ inline int wide_pitch(int NL) { return NL == 9 ? 12 : NL; }
__host__ __device__ inline int64_t wide_dword(int m, int W, int NL, int i, int j, int aa, int bb, int st) {
  return ((((int64_t)i * (m + 1) + j) * W + aa) * W + bb) * wide_pitch(NL) + st;
}
__host__ __device__ inline int64_t wide_pair_dwords(int n, int m, int S, int NL) {
  const int64_t W = 2 * S + 1;
  return (int64_t)(n + 1) * (m + 1) * W * W * wide_pitch(NL);
}

struct WideCtx {
  int n, m, S, W;
  int beta, gamma, delta;
  const int32_t *s1, *s2;
  int k1, k2;
  const uint8_t *sa, *ca, *sb, *cb;  // this pair's codes (global memory)
  const int32_t* mu2tab;             // dense mu2 table of this pair or nullptr
  int32_t* lay;                      // this pair's layers
  __device__ __forceinline__ int mu1(int i, int j) const {  // pyx:435-436 (never contributes at i=0 or j=0)
    return (i >= 1 && j >= 1) ? s1[sa[i - 1] * k1 + sb[j - 1]] : 0;
  }
  __device__ __forceinline__ int mu2(int k, int l) const {  // pyx:438-440
    if (k < 1 || l < 1) return 0;
    return mu2tab ? mu2tab[(int64_t)(k - 1) * m + (l - 1)] : s2[ca[k - 1] * k2 + cb[l - 1]];
  }
  __device__ __forceinline__ bool valid(int pi, int pj, int pk, int pl) const {  // pyx:133-141
    return pi >= 0 && pj >= 0 && pk >= 0 && pl >= 0 && abs(pk - pi) <= S && abs(pl - pj) <= S;
  }
};

// One pair is swept by A.team workgroups of WIDE_THREADS threads (the "parts" of the pair: block b -> pair b / team,
// part b % team) on any CUs / XCDs; team = 1 for batches with many pairs.  Per-XCD L2s are not coherent, so layer
// values travel like the tiled sweep's cross-CU teams do: write-through stores and sc1 loads, and a level is
// closed by a counter in HBM (A.prog, one word per pair, zeroed per launch) that every part bumps once its
// stores are acknowledged.  All parts of a launch must be resident at once (the host sizes the grid from the
// runtime's occupancy figure); a part that waits longer than the spin limit raises the device flag and the host
// repeats the run with one workgroup per pair.
constexpr int WIDE_THREADS = 512;

// Visit every lattice point of level D with the threads of all parts of the pair.
template <typename F>
__device__ __forceinline__ void wide_for_level(const WideCtx& c, int D, int part, int parts, F&& f) {
  const int S = c.S, W = c.W, HW = (W + 1) / 2;
  // j = (D - 2i - (aa-S) - (bb-S)) / 2 must lie in [0, m]
  const int ilo = max(0, (D - 2 * c.m - 2 * S + 1) >> 1), ihi = min(c.n, (D + 2 * S) >> 1);
  const int items = (ihi - ilo + 1) * W * HW;
  for (int t = part * WIDE_THREADS + threadIdx.x; t < items; t += parts * WIDE_THREADS) {
    const int hb = t % HW, aa = (t / HW) % W, i = ilo + t / (HW * W);
    const int bb = 2 * hb + ((D - aa) & 1);  // aa + bb must have D's parity
    if (bb >= W) continue;
    const int rem = D - 2 * i - (aa - S) - (bb - S);
    if (rem < 0 || rem > 2 * c.m) continue;
    const int j = rem >> 1, k = i + aa - S, l = j + bb - S;
    if (k < 0 || k > c.n || l < 0 || l > c.m) continue;
    f(i, j, k, l, aa, bb);
  }
}

// nine consecutive layer values (one cell's states), from L2 or beyond, never from this CU's L1; the compiler
// keeps all of a point's loads in flight and waits once before the first use
__device__ __forceinline__ void wide_load9(const int32_t* p, int (&v)[9]) {
#pragma unroll
  for (int q = 0; q < 9; ++q) v[q] = __hip_atomic_load(p + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wide_store(int32_t* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dword ... sc1: write-through
}

#if BIALIGN_EXP == 8
__device__ long long wide_exp_polls, wide_exp_own;  // (every part's thread 0 adds to them: read for part 0 only by dividing, roughly)
#endif
// Close a level: every part's stores are acknowledged, then all parts of the pair have said so -- each in a word of its
// own (flags[part] = levels closed; a write-through store, nothing to serialise), and thread t of every part watches
// part t's word: one round trip sees all of them, where a shared counter took one atomic per part on one address.
__device__ __forceinline__ void wide_level_sync(const DeviceBatch& A, int32_t* flags, int level, int part, int parts, bool& failed) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (parts > 1) {
    if (threadIdx.x == 0) __hip_atomic_store(flags + part, level + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)threadIdx.x < parts && !failed) {
#if BIALIGN_EXP == 8
      const long long ts0 = __builtin_amdgcn_s_memtime();
#endif
      for (int spin = 0; __hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= level; ++spin) {
#if BIALIGN_EXP == 8
        if (threadIdx.x == 0) wide_exp_polls += 1;
#endif
        if (spin > A.spin_limit) {  // a partner was never scheduled: fail fast, the host recovers
          atomicOr(A.errflag, 1);
          failed = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
#if BIALIGN_EXP == 8
      if (threadIdx.x == 0) wide_exp_own += __builtin_amdgcn_s_memtime() - ts0;  // thread 0 watches part 0 (its own part for part 0)
#endif
    }
    __syncthreads();
  }
}

// A pair's sequence / class codes and the score tables, staged in LDS when they fit (the two dependent lookups of
// mu1 and mu2 are then off the level's chain of memory round trips); else the context keeps its global pointers.
constexpr int WIDE_STAGE_CODES = 8192, WIDE_STAGE_TAB = 1024;
struct WideStage {
  uint8_t codes[4][WIDE_STAGE_CODES];
  int32_t tab[2][WIDE_STAGE_TAB];
};
__device__ __forceinline__ void wide_stage(WideCtx& c, WideStage& st) {
  if (c.n <= WIDE_STAGE_CODES && c.m <= WIDE_STAGE_CODES && c.k1 * c.k1 <= WIDE_STAGE_TAB && c.k2 * c.k2 <= WIDE_STAGE_TAB) {
    for (int t = threadIdx.x; t < c.n; t += WIDE_THREADS) { st.codes[0][t] = c.sa[t]; st.codes[1][t] = c.ca[t]; }
    for (int t = threadIdx.x; t < c.m; t += WIDE_THREADS) { st.codes[2][t] = c.sb[t]; st.codes[3][t] = c.cb[t]; }
    for (int t = threadIdx.x; t < c.k1 * c.k1; t += WIDE_THREADS) st.tab[0][t] = c.s1[t];
    for (int t = threadIdx.x; t < c.k2 * c.k2; t += WIDE_THREADS) st.tab[1][t] = c.s2[t];
    __syncthreads();
    c.sa = st.codes[0]; c.ca = st.codes[1]; c.sb = st.codes[2]; c.cb = st.codes[3];
    c.s1 = st.tab[0]; c.s2 = st.tab[1];
  }
}

// ---------------------------------------------------------------------------
// Affine fill (pyx:474-509) with the tiled sweep's algebra (bialign_kernels.hpp: bit-exact regrouping of the fifteen
// cases per state).  Beside its nine layer values a point writes 27 DERIVED values
//   G[U][V] = f_U(H2[.][V]),  H2[U][V] = f_V(M[(U,.)]),  H3[U][V] = f_U(M[(.,V)]),   f_T(v) = max_h(open(h,T) + v[h])
// into a ring of the last WIDE_RING levels (a predecessor lowers D by 1..4), and a target state (U,V) then needs ONE
// value from each of three predecessor cells -- G[U][V](q-(U,V)), H2[U][V](q-(0,0,V)), H3[U][V](q-(U,0,0)) -- instead
// of nine, three and three layer values: 27 four-byte loads per point instead of 135 (round 2).  The ring is a
// structure of arrays, [D mod WIDE_RING][component][i][a][b/2] (within a level b has one parity; components: G at 3U+V,
// H2 at 9+3V+U, H3 at 18+3U+V): neighbouring threads hold neighbouring (i, a, b), and a predecessor by a fixed offset
// is a neighbour again, so one load or store instruction of a wave touches four or five 64-byte lines (making i the
// fastest index instead -- runs of 64 -- was measured and is 3-10 % slower: the score lookups then diverge) -- with the
// 28-dword slot per point that the first version of this sweep used it touched sixty-four, and the level was bound by
// the address path of the CU (27 gathers + 7 scattered stores per wave), not by latency.
// With A.wide_score_only the layers themselves are not stored at all: the last level's one point writes the score.
// ---------------------------------------------------------------------------
constexpr int WIDE_RING = 5, WIDE_SLOT = 27;
__host__ __device__ inline int64_t wide_ring_level_dwords(int n, int S) {
  const int64_t W = 2 * S + 1;
  return (int64_t)(n + 1) * W * ((W + 1) / 2) * WIDE_SLOT;
}
// f_T for target half Y, X, M over the values of source halves (y, x, m); exact for any sign of beta
__device__ __forceinline__ int wfY(int y, int x, int m, int beta) { return max(y, beta + max(x, m)); }
__device__ __forceinline__ int wfX(int y, int x, int m, int beta) { return max(x, beta + max(y, m)); }
__device__ __forceinline__ int wfM(int y, int x, int m) { return max(max(y, x), m); }

template <int UNUSED = 0>  // (a template so that only bialign_wide.hip instantiates it)
__global__ void __launch_bounds__(WIDE_THREADS) fill_wide_affine_kernel(const DeviceBatch A, int S) {
  const int parts = A.team, slot = blockIdx.x / parts, part = blockIdx.x - slot * parts;
  const int pid = A.order[slot];
  const PairDesc pd = A.pairs[pid];
  WideCtx c;
  c.n = pd.n; c.m = pd.m; c.S = S; c.W = 2 * S + 1;
  c.beta = A.beta; c.gamma = A.gamma; c.delta = A.delta;
  c.s1 = A.s1; c.s2 = A.s2; c.k1 = A.k1; c.k2 = A.k2;
  c.sa = A.seq_a + pd.seq_a; c.ca = A.cls_a + pd.seq_a; c.sb = A.seq_b + pd.seq_b; c.cb = A.cls_b + pd.seq_b;
  c.mu2tab = A.mu2_dense ? A.mu2_dense + pd.mu2_off : nullptr;
  c.lay = A.layers + pd.layer_off;
  const int n = c.n, m = c.m, W = c.W, HW = (W + 1) / 2;
  const int beta = c.beta, gamma = c.gamma, delta = c.delta;
  int32_t* const flags = A.prog + (int64_t)slot * PROG_WORDS;  // [parts] levels closed, zeroed per launch
  int32_t* const ring = A.wide_ring + A.wide_ring_off[slot];
  __shared__ WideStage stage;
  wide_stage(c, stage);
  const int64_t LV = (int64_t)(n + 1) * W * HW, lvl = wide_ring_level_dwords(n, S);  // points per level, dwords per level (27 LV)
  const bool keep_layers = !A.wide_score_only;
  bool failed = false;  // (a level barrier timed out for this thread: no further waits)

  // component 0 of lattice point (i, a, b) in the ring level at `base`; component v lies v * LV dwords further
  auto rpoint = [&](int32_t* base, int i, int aa, int bb) -> int32_t* {
    return base + (int64_t)(i * W + aa) * HW + (bb >> 1);
  };
#if BIALIGN_EXP == 8  // timing experiment: where a level's time goes (thread 0 of part 0 prints the averages)
  long long tph[5] = {0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
  auto stamp = [&](int ph) { const long long t = __builtin_amdgcn_s_memtime(); tph[ph] += t - tlast; tlast = t; };
#endif
  for (int D = 0; D <= 2 * (n + m); ++D) {
    int32_t* rb[WIDE_RING];  // ring levels of D, D-1, .. D-4
#pragma unroll
    for (int d = 0; d < WIDE_RING; ++d) rb[d] = ring + ((D - d + WIDE_RING) % WIDE_RING) * lvl;
    wide_for_level(c, D, part, parts, [&](int i, int j, int k, int l, int aa, int bb) {
      int M[9];
      if (D == 0) {  // pyx:483-485
#pragma unroll
        for (int q = 0; q < 9; ++q) M[q] = q == 8 ? 0 : NEG;
      } else {
        // the fifteen predecessor cells by offset code o0*8 + o1*4 + o2*2 + o3; an invalid one (pyx:133-141) is not read
        bool ok[16];
        const int32_t* src[16];
        ok[0] = false;
        src[0] = ring;
#pragma unroll
        for (int code = 1; code < 16; ++code) {
          const int o0 = (code >> 3) & 1, o1 = (code >> 2) & 1, o2 = (code >> 1) & 1, o3 = code & 1;
          const int pi = i - o0, pj = j - o1, pk = k - o2, pl = l - o3;
          ok[code] = c.valid(pi, pj, pk, pl);
          src[code] = ok[code] ? rpoint(rb[o0 + o1 + o2 + o3], pi, pk - pi + S, pl - pj + S) : ring;
        }
        int g[9], h2[9], h3[9];  // [3*hU + hV]: the one value each group contributes to target state (hU, hV)
#pragma unroll
        for (int hU = 0; hU < 3; ++hU) {
#pragma unroll
          for (int hV = 0; hV < 3; ++hV) {
            const int u0 = hU >= 1, u1 = hU != 1, v0 = hV >= 1, v1 = hV != 1;
            const int c1 = u0 * 8 + u1 * 4 + v0 * 2 + v1, c2 = v0 * 2 + v1, c3 = u0 * 8 + u1 * 4;
            g[3 * hU + hV] = __hip_atomic_load(src[c1] + (3 * hU + hV) * LV, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            h2[3 * hU + hV] = __hip_atomic_load(src[c2] + (9 + 3 * hV + hU) * LV, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            h3[3 * hU + hV] = __hip_atomic_load(src[c3] + (18 + 3 * hU + hV) * LV, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        const int mu1 = c.mu1(i, j), mu2 = c.mu2(k, l);
#if BIALIGN_EXP == 8
        stamp(0);  // address arithmetic, loads issued
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stamp(1);  // loads back
#endif
#pragma unroll
        for (int hU = 0; hU < 3; ++hU) {
#pragma unroll
          for (int hV = 0; hV < 3; ++hV) {
            const int u0 = hU >= 1, u1 = hU != 1, v0 = hV >= 1, v1 = hV != 1;
            const int valU = hU == 2 ? mu1 : gamma, valV = hV == 2 ? mu2 : gamma;
            const int c1 = u0 * 8 + u1 * 4 + v0 * 2 + v1, c2 = v0 * 2 + v1, c3 = u0 * 8 + u1 * 4;
            const int sh = hU == hV ? 0 : ((hU == 2 || hV == 2) ? 1 : 2);
            bool any = false;
            int best = NEG;  // pyx:299-303: no valid case -> exactly NEG
            auto take = [&](int v) { best = any ? (v > best ? v : best) : v; any = true; };
            if (ok[c1]) take(g[3 * hU + hV] + delta * sh + valU + valV);        // group 1 (pyx:275-279)
            if (ok[c2]) take(h2[3 * hU + hV] + delta * (v0 + v1) + valV);       // group 2 (pyx:284-290)
            if (ok[c3]) take(h3[3 * hU + hV] + delta * (u0 + u1) + valU);       // group 3 (pyx:291-296)
            M[3 * hU + hV] = best;
          }
        }
      }
      // derived values for the successors
      int H2[3][3], H3[3][3], G[3][3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        H2[u][0] = wfY(M[3 * u], M[3 * u + 1], M[3 * u + 2], beta);
        H2[u][1] = wfX(M[3 * u], M[3 * u + 1], M[3 * u + 2], beta);
        H2[u][2] = wfM(M[3 * u], M[3 * u + 1], M[3 * u + 2]);
      }
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        H3[0][v] = wfY(M[v], M[3 + v], M[6 + v], beta);
        H3[1][v] = wfX(M[v], M[3 + v], M[6 + v], beta);
        H3[2][v] = wfM(M[v], M[3 + v], M[6 + v]);
        G[0][v] = wfY(H2[0][v], H2[1][v], H2[2][v], beta);
        G[1][v] = wfX(H2[0][v], H2[1][v], H2[2][v], beta);
        G[2][v] = wfM(H2[0][v], H2[1][v], H2[2][v]);
      }
      int32_t* r = rpoint(rb[0], i, aa, bb);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          wide_store(r + (3 * u + v) * LV, G[u][v]);
          wide_store(r + (9 + 3 * v + u) * LV, H2[u][v]);  // H2 at 9 + 3V + U
          wide_store(r + (18 + 3 * u + v) * LV, H3[u][v]);
        }
      }
      if (keep_layers) {
        // the sweep itself never reads a layer back (its state is the ring): plain write-back stores, 16-byte aligned
        v4i* out = reinterpret_cast<v4i*>(c.lay + wide_dword(m, W, 9, i, j, aa, bb, 0));
        out[0] = v4i{M[0], M[1], M[2], M[3]};
        out[1] = v4i{M[4], M[5], M[6], M[7]};
        out[2] = v4i{M[8], 0, 0, 0};
      } else if (D == 2 * (n + m)) {  // score-only: the end cell (n,m,n,m) is all the host wants (pyx:509)
        int best = M[0];
#pragma unroll
        for (int q = 1; q < 9; ++q) best = max(best, M[q]);
        A.scores[pid] = best;
      }
    });
#if BIALIGN_EXP == 8
    stamp(2);  // compute, stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(3);  // stores acknowledged
#endif
    wide_level_sync(A, flags, D, part, parts, failed);
#if BIALIGN_EXP == 8
    stamp(4);  // barrier
#endif
  }
#if BIALIGN_EXP == 8
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double L = 2.0 * (n + m) + 1;
    printf("wide level phases (s_memtime ticks per level, part 0 thread 0 of %d parts): issue %.0f  loads %.0f  compute+stores %.0f  store-ack %.0f  barrier %.0f\n",
           parts, tph[0] / L, tph[1] / L, tph[2] / L, tph[3] / L, tph[4] / L);
    printf("  thread 0 of each part watching part 0's flag: %.1f polls, %.0f ticks per level and part\n",
           (double)wide_exp_polls / L / parts, (double)wide_exp_own / L / parts);
    wide_exp_polls = wide_exp_own = 0;
  }
#endif
}

// ---------------------------------------------------------------------------
// Non-affine fill (pyx:443-471): thirteen cases (pyx:233-248), one layer.
// ---------------------------------------------------------------------------
template <int UNUSED = 0>
__global__ void __launch_bounds__(WIDE_THREADS) fill_wide_linear_kernel(const DeviceBatch A, int S) {
  const int parts = A.team, slot = blockIdx.x / parts, part = blockIdx.x - slot * parts;
  const int pid = A.order[slot];
  const PairDesc pd = A.pairs[pid];
  WideCtx c;
  c.n = pd.n; c.m = pd.m; c.S = S; c.W = 2 * S + 1;
  c.beta = A.beta; c.gamma = A.gamma; c.delta = A.delta;
  c.s1 = A.s1; c.s2 = A.s2; c.k1 = A.k1; c.k2 = A.k2;
  c.sa = A.seq_a + pd.seq_a; c.ca = A.cls_a + pd.seq_a; c.sb = A.seq_b + pd.seq_b; c.cb = A.cls_b + pd.seq_b;
  c.mu2tab = A.mu2_dense ? A.mu2_dense + pd.mu2_off : nullptr;
  c.lay = A.layers + pd.layer_off;
  const int n = c.n, m = c.m, W = c.W;
  const int gamma = c.gamma, delta = c.delta, gD = gamma + delta;
  constexpr int OFF[13] = {15, 10, 5, 12, 3, 8, 4, 2, 1, 11, 7, 14, 13};  // o0*8+o1*4+o2*2+o3, generator order
  int32_t* const flags = A.prog + (int64_t)slot * PROG_WORDS;
  bool failed = false;  // (a level barrier timed out for this thread: no further waits)
  __shared__ WideStage stage;
  wide_stage(c, stage);

  for (int D = 0; D <= 2 * (n + m); ++D) {
    wide_for_level(c, D, part, parts, [&](int i, int j, int k, int l, int aa, int bb) {
      int32_t* out = c.lay + wide_dword(m, W, 1, i, j, aa, bb, 0);
      if (D == 0) { wide_store(out, 0); return; }  // zero-initialised storage (pyx:452)
      int pv[13];
      bool ok[13];
#pragma unroll
      for (int t = 0; t < 13; ++t) {  // all thirteen predecessors in flight together
        const int o0 = (OFF[t] >> 3) & 1, o1 = (OFF[t] >> 2) & 1, o2 = (OFF[t] >> 1) & 1, o3 = OFF[t] & 1;
        const int pi = i - o0, pj = j - o1, pk = k - o2, pl = l - o3;
        ok[t] = c.valid(pi, pj, pk, pl);
        const int32_t* src = ok[t] ? c.lay + wide_dword(m, W, 1, pi, pj, pk - pi + S, pl - pj + S, 0) : out;
        pv[t] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const int mu1 = c.mu1(i, j), mu2 = c.mu2(k, l);
      bool any = false;
      int best = NEG;
#pragma unroll
      for (int t = 0; t < 13; ++t) {
        if (!ok[t]) continue;
        const int use1 = (t == 0 || t == 3 || t == 11 || t == 12), use2 = (t == 0 || t == 4 || t == 9 || t == 10);
        const int kconst = t == 0 ? 0 : (t <= 2 ? 2 * gamma : (t <= 4 ? delta : gD));
        const int v = pv[t] + kconst + (use1 ? mu1 : 0) + (use2 ? mu2 : 0);
        best = any ? (v > best ? v : best) : v;
        any = true;
      }
      wide_store(out, best);
    });
    wide_level_sync(A, flags, D, part, parts, failed);
  }
}

// Layers in the reference layout for the test hook (same order already; out-of-lattice band slots -> 0).
template <int NL>
__global__ void dump_wide_kernel(const DeviceBatch A, int S, int pid, int32_t* out) {
  const int W = 2 * S + 1;
  const PairDesc pd = A.pairs[pid];
  const int n = pd.n, m = pd.m;
  const int64_t cells = (int64_t)(n + 1) * (m + 1) * W * W;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < cells; t += (int64_t)gridDim.x * blockDim.x) {
    const int bb = t % W, aa = (t / W) % W;
    const int j = (t / (W * W)) % (m + 1), i = t / ((int64_t)W * W * (m + 1));
    const int k = i + aa - S, l = j + bb - S;
    const bool ok = k >= 0 && k <= n && l >= 0 && l <= m;
    for (int q = 0; q < NL; ++q) out[q * cells + t] = ok ? A.layers[pd.layer_off + t * wide_pitch(NL) + q] : 0;
  }
}

}  // namespace bialign
