/*
 * bialign_oracle.c -- CPU restatement of the BiAlign DP fill + traceback.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * engine in bialign_amd/csrc.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may build, load or call it.  Nothing in
 * bialign_amd/ links or imports it and the product path never falls back
 * to it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against fixtures in tests/golden/ that were produced by the compiled
 * reference (tests/golden/make_golden.py): scores, traces and complete
 * nine-layer dumps, including the two README known answers (6800, 48500).
 *
 * The code deliberately follows the reference's *literal* case enumeration
 * (one affine_score() call per case) and its lexicographic loop order, so it
 * is an independent check of the regrouped max-plus algebra used on the GPU.
 *
 * Reference (read-only, /root/reference/src/bialignment.pyx), cited per
 * function as pyx:LINE.
 *
 * Conventions
 *   n = len(A), m = len(B), s = max_shift, W = 2s+1
 *   mu1[(i)*(m+1)+(j)]  = reference mu1(i,j), 1-based, row/col 0 unused
 *   mu2[(k)*(m+1)+(l)]  = reference mu2(k,l), 1-based, row/col 0 unused
 *   layer element (i,j,k,l) lives at [i][j][k-i+s][l-j+s]      (pyx:27-41)
 *   affine layers: nine of them in itertools.product order      (pyx:61-65)
 *   trace byte = o0*8 + o1*4 + o2*2 + o3, start -> end order    (pyx:586)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BIALIGN_NEG (-(1 << 30)) /* pyx:303, pyx:484 */

/* pyx:61-65: states with (x0|x1) and (x2|x3), product order. */
static const int STATES[9][4] = {
    {0, 1, 0, 1}, {0, 1, 1, 0}, {0, 1, 1, 1}, {1, 0, 0, 1}, {1, 0, 1, 0},
    {1, 0, 1, 1}, {1, 1, 0, 1}, {1, 1, 1, 0}, {1, 1, 1, 1}};

/* position of a state in STATES: halves order (0,1) < (1,0) < (1,1). */
static inline int state_index(const int st[4]) {
  return 3 * (2 * st[0] + st[1] - 1) + (2 * st[2] + st[3] - 1);
}

typedef struct {
  int n, m, s, W;
  int beta, gamma, delta;
  const int32_t *mu1, *mu2;
} prob_t;

static inline int64_t cell_off(const prob_t *p, int i, int j, int k, int l) {
  return (((int64_t)i * (p->m + 1) + j) * p->W + (k - i + p->s)) * p->W +
         (l - j + p->s);
}
static inline int64_t layer_elems(const prob_t *p) {
  return (int64_t)(p->n + 1) * (p->m + 1) * p->W * p->W;
}

/* mu look-ups.  The reference evaluates mu at index 0 through Python's
 * negative-index wrap (pyx:407); such a value only ever enters a *match*
 * half, whose offset needs that index >= 1, so it never contributes. */
static inline int MU1(const prob_t *p, int i, int j) {
  return (i >= 1 && j >= 1) ? p->mu1[(int64_t)i * (p->m + 1) + j] : 0;
}
static inline int MU2(const prob_t *p, int k, int l) {
  return (k >= 1 && l >= 1) ? p->mu2[(int64_t)k * (p->m + 1) + l] : 0;
}

/* pyx:84-131 */
static int affine_score(const int src[4], const int x[4], int mu1, int mu2,
                        int beta, int gamma, int Delta) {
  int score = Delta * (abs(x[0] - x[2]) + abs(x[1] - x[3]));
  for (int half = 0; half < 2; ++half) {
    int a = 2 * half, b = 2 * half + 1;
    int mu = half == 0 ? mu1 : mu2;
    int xa = x[a], xb = x[b];
    if (xa && xb) {
      score += mu;
    } else if (xa && !xb) {
      score += gamma;
      if (!(src[a] == 1 && src[b] == 0)) score += beta;
    } else if (!xa && xb) {
      score += gamma;
      if (!(src[a] == 0 && src[b] == 1)) score += beta;
    }
  }
  return score;
}

/* pyx:133-141 */
static int guard_case(const int o[4], const int x[4], int s) {
  return x[0] - o[0] >= 0 && x[1] - o[1] >= 0 && x[2] - o[2] >= 0 &&
         x[3] - o[3] >= 0 && abs(x[2] - o[2] - (x[0] - o[0])) <= s &&
         abs(x[3] - o[3] - (x[1] - o[1])) <= s;
}

typedef struct {
  int src[4]; /* source state */
  int off[4]; /* offset = new column */
  int score;
} acase_t;

/* pyx:255-296: up to 9 + 3 + 3 cases, in the reference's order. */
static int affine_cases(const prob_t *p, const int state[4], const int idx[4],
                        acase_t out[15]) {
  int cnt = 0;
  int mu1 = MU1(p, idx[0], idx[1]);
  int mu2 = MU2(p, idx[2], idx[3]);
  if (guard_case(state, idx, p->s)) {
    for (int ss = 0; ss < 9; ++ss) {
      memcpy(out[cnt].src, STATES[ss], sizeof(int) * 4);
      memcpy(out[cnt].off, state, sizeof(int) * 4);
      out[cnt].score = affine_score(STATES[ss], state, mu1, mu2, p->beta,
                                    p->gamma, p->delta);
      ++cnt;
    }
  }
  static const int half_states[3][2] = {{1, 1}, {1, 0}, {0, 1}};
  int off[4] = {0, 0, state[2], state[3]};
  if (guard_case(off, idx, p->s)) {
    for (int hs = 0; hs < 3; ++hs) {
      int src[4] = {state[0], state[1], half_states[hs][0], half_states[hs][1]};
      memcpy(out[cnt].src, src, sizeof(src));
      memcpy(out[cnt].off, off, sizeof(off));
      out[cnt].score =
          affine_score(src, off, mu1, mu2, p->beta, p->gamma, p->delta);
      ++cnt;
    }
  }
  int off3[4] = {state[0], state[1], 0, 0};
  if (guard_case(off3, idx, p->s)) {
    for (int hs = 0; hs < 3; ++hs) {
      int src[4] = {half_states[hs][0], half_states[hs][1], state[2], state[3]};
      memcpy(out[cnt].src, src, sizeof(src));
      memcpy(out[cnt].off, off3, sizeof(off3));
      out[cnt].score =
          affine_score(src, off3, mu1, mu2, p->beta, p->gamma, p->delta);
      ++cnt;
    }
  }
  return cnt;
}

int64_t bialign_oracle_layer_elems(int n, int m, int s) {
  prob_t p = {n, m, s, 2 * s + 1, 0, 0, 0, 0, 0};
  return layer_elems(&p);
}

/* pyx:474-509.  layers: int32[9][layer_elems], zero-filled here like
 * np.zeros (pyx:27); returns max over the nine layers at (n,m,n,m). */
int bialign_oracle_affine_fill(int n, int m, int s, int beta, int gamma,
                               int delta, const int32_t *mu1,
                               const int32_t *mu2, int32_t *layers,
                               int32_t *score_out) {
  if (n < 1 || m < 1 || s < 0) return -1; /* pyx:407 raises IndexError */
  prob_t p = {n, m, s, 2 * s + 1, beta, gamma, delta, mu1, mu2};
  int64_t L = layer_elems(&p);
  memset(layers, 0, sizeof(int32_t) * 9 * L);
  for (int q = 0; q < 9; ++q) layers[q * L + cell_off(&p, 0, 0, 0, 0)] = BIALIGN_NEG;
  layers[8 * L + cell_off(&p, 0, 0, 0, 0)] = 0; /* pyx:485 */

  acase_t cs[15];
  for (int i = 0; i <= n; ++i)
    for (int j = 0; j <= m; ++j) {
      int klo = i - s > 0 ? i - s : 0, khi = i + s < n ? i + s : n;
      int llo = j - s > 0 ? j - s : 0, lhi = j + s < m ? j + s : m;
      for (int k = klo; k <= khi; ++k)
        for (int l = llo; l <= lhi; ++l) {
          if (i == 0 && j == 0 && k == 0 && l == 0) continue;
          int idx[4] = {i, j, k, l};
          for (int t = 0; t < 9; ++t) {
            int cnt = affine_cases(&p, STATES[t], idx, cs);
            int best = BIALIGN_NEG; /* pyx:299-303: empty -> -1<<30 */
            for (int c = 0; c < cnt; ++c) {
              const int *o = cs[c].off;
              int32_t v = layers[state_index(cs[c].src) * L +
                                 cell_off(&p, i - o[0], j - o[1], k - o[2], l - o[3])] +
                          cs[c].score; /* pyx:315-320 */
              if (c == 0 || v > best) best = v;
            }
            layers[t * L + cell_off(&p, i, j, k, l)] = best;
          }
        }
    }
  int best = layers[0 * L + cell_off(&p, n, m, n, m)];
  for (int t = 1; t < 9; ++t) {
    int v = layers[t * L + cell_off(&p, n, m, n, m)];
    if (v > best) best = v;
  }
  *score_out = best;
  return 0;
}

static void shift_by(const int x[4], int ts[3]) { /* pyx:541-545 */
  ts[0] += x[0] - x[2];
  ts[1] += x[1] - x[3];
  ts[2] = abs(ts[0]) + abs(ts[1]);
}

/* pyx:535-586.  trace_out receives start->end offsets (one byte each);
 * *complete_out = 0 reproduces the reference's "incomplete traceback"
 * warning condition (pyx:584-585). */
int bialign_oracle_affine_traceback(int n, int m, int s, int beta, int gamma,
                                    int delta, const int32_t *mu1,
                                    const int32_t *mu2, const int32_t *layers,
                                    uint8_t *trace_out, int cap, int *len_out,
                                    int *complete_out) {
  prob_t p = {n, m, s, 2 * s + 1, beta, gamma, delta, mu1, mu2};
  int64_t L = layer_elems(&p);
  int64_t endc = cell_off(&p, n, m, n, m);

  /* pyx:573-582: best layer, first one with the least |shift|. */
  int best = layers[endc];
  for (int t = 1; t < 9; ++t)
    if (layers[t * L + endc] > best) best = layers[t * L + endc];
  int sel = -1, selkey = 0;
  for (int t = 0; t < 9; ++t) {
    if (layers[t * L + endc] != best) continue;
    int ts[3] = {0, 0, 0};
    shift_by(STATES[t], ts);
    if (sel < 0 || ts[2] < selkey) {
      sel = t;
      selkey = ts[2];
    }
  }

  int state[4], idx[4] = {n, m, n, m}, total[3] = {0, 0, 0};
  memcpy(state, STATES[sel], sizeof(state));
  int len = 0, complete = 0;
  acase_t cs[15];
  for (;;) {
    /* pyx:551 (the first call passes a tuple, which never equals the
     * list [1,1,1,1]; harmless because idx != origin there) */
    if (idx[0] == 0 && idx[1] == 0 && idx[2] == 0 && idx[3] == 0 &&
        state[0] == 1 && state[1] == 1 && state[2] == 1 && state[3] == 1) {
      complete = 1;
      break;
    }
    int cnt = affine_cases(&p, state, idx, cs);
    int32_t cur = layers[state_index(state) * L +
                         cell_off(&p, idx[0], idx[1], idx[2], idx[3])];
    int pick = -1, k0 = 0, k1 = 0;
    for (int c = 0; c < cnt; ++c) {
      const int *o = cs[c].off;
      int32_t v = layers[state_index(cs[c].src) * L +
                         cell_off(&p, idx[0] - o[0], idx[1] - o[1], idx[2] - o[2],
                                  idx[3] - o[3])] +
                  cs[c].score;
      if (v != cur) continue;
      int tmp[3] = {total[0], total[1], total[2]};
      shift_by(cs[c].off, tmp); /* pyx:559 */
      shift_by(cs[c].src, tmp); /* pyx:560 */
      int c0 = tmp[2], c1 = abs(tmp[1]);
      if (pick < 0 || c0 < k0 || (c0 == k0 && c1 < k1)) { /* pyx:564 first min */
        pick = c;
        k0 = c0;
        k1 = c1;
      }
    }
    if (pick < 0) break; /* pyx:570-571 */
    shift_by(cs[pick].off, total);
    if (len >= cap) return -2;
    trace_out[len++] = (uint8_t)(cs[pick].off[0] * 8 + cs[pick].off[1] * 4 +
                                 cs[pick].off[2] * 2 + cs[pick].off[3]);
    for (int q = 0; q < 4; ++q) idx[q] -= cs[pick].off[q];
    memcpy(state, cs[pick].src, sizeof(state));
  }
  for (int a = 0, b = len - 1; a < b; ++a, --b) { /* pyx:586 reversed */
    uint8_t t = trace_out[a];
    trace_out[a] = trace_out[b];
    trace_out[b] = t;
  }
  *len_out = len;
  *complete_out = complete;
  return 0;
}

/* ---- non-affine recurrence (selected iff gap_opening_cost == 0) ------- */

/* pyx:233-248: the thirteen cases in generator order. */
static const int LIN_OFF[13][4] = {
    {1, 1, 1, 1}, {1, 0, 1, 0}, {0, 1, 0, 1}, {1, 1, 0, 0}, {0, 0, 1, 1},
    {1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}, {1, 0, 1, 1},
    {0, 1, 1, 1}, {1, 1, 1, 0}, {1, 1, 0, 1}};

static void linear_scores(const prob_t *p, const int idx[4], int sc[13]) {
  int mu1 = MU1(p, idx[0], idx[1]), mu2 = MU2(p, idx[2], idx[3]);
  int g = p->gamma, D = p->delta;
  sc[0] = mu1 + mu2;
  sc[1] = g + g;
  sc[2] = g + g;
  sc[3] = mu1 + D;
  sc[4] = mu2 + D;
  sc[5] = sc[6] = sc[7] = sc[8] = g + D;
  sc[9] = sc[10] = g + mu2 + D;
  sc[11] = sc[12] = g + mu1 + D;
}

/* pyx:443-471.  layer: int32[layer_elems], zero-filled (so the origin is 0). */
int bialign_oracle_linear_fill(int n, int m, int s, int gamma, int delta,
                               const int32_t *mu1, const int32_t *mu2,
                               int32_t *layer, int32_t *score_out) {
  if (n < 1 || m < 1 || s < 0) return -1;
  prob_t p = {n, m, s, 2 * s + 1, 0, gamma, delta, mu1, mu2};
  memset(layer, 0, sizeof(int32_t) * layer_elems(&p));
  int sc[13];
  for (int i = 0; i <= n; ++i)
    for (int j = 0; j <= m; ++j) {
      int klo = i - s > 0 ? i - s : 0, khi = i + s < n ? i + s : n;
      int llo = j - s > 0 ? j - s : 0, lhi = j + s < m ? j + s : m;
      for (int k = klo; k <= khi; ++k)
        for (int l = llo; l <= lhi; ++l) {
          if (i == 0 && j == 0 && k == 0 && l == 0) continue;
          int idx[4] = {i, j, k, l};
          linear_scores(&p, idx, sc);
          int best = BIALIGN_NEG, any = 0;
          for (int c = 0; c < 13; ++c) {
            const int *o = LIN_OFF[c];
            if (!guard_case(o, idx, s)) continue; /* pyx:469 */
            int v = layer[cell_off(&p, i - o[0], j - o[1], k - o[2], l - o[3])] +
                    sc[c]; /* pyx:310-313 */
            if (!any || v > best) best = v;
            any = 1;
          }
          layer[cell_off(&p, i, j, k, l)] = best;
        }
    }
  *score_out = layer[cell_off(&p, n, m, n, m)];
  return 0;
}

/* pyx:513-531: first matching case wins; stops when nothing matches. */
int bialign_oracle_linear_traceback(int n, int m, int s, int gamma, int delta,
                                    const int32_t *mu1, const int32_t *mu2,
                                    const int32_t *layer, uint8_t *trace_out,
                                    int cap, int *len_out) {
  prob_t p = {n, m, s, 2 * s + 1, 0, gamma, delta, mu1, mu2};
  int idx[4] = {n, m, n, m}, len = 0, sc[13];
  for (;;) {
    linear_scores(&p, idx, sc);
    int pick = -1;
    for (int c = 0; c < 13 && pick < 0; ++c) {
      const int *o = LIN_OFF[c];
      if (!guard_case(o, idx, s)) continue;
      int v = layer[cell_off(&p, idx[0] - o[0], idx[1] - o[1], idx[2] - o[2],
                             idx[3] - o[3])] +
              sc[c];
      if (v == layer[cell_off(&p, idx[0], idx[1], idx[2], idx[3])]) pick = c;
    }
    if (pick < 0) break;
    if (len >= cap) return -2;
    const int *o = LIN_OFF[pick];
    trace_out[len++] = (uint8_t)(o[0] * 8 + o[1] * 4 + o[2] * 2 + o[3]);
    for (int q = 0; q < 4; ++q) idx[q] -= o[q];
  }
  for (int a = 0, b = len - 1; a < b; ++a, --b) {
    uint8_t t = trace_out[a];
    trace_out[a] = trace_out[b];
    trace_out[b] = t;
  }
  *len_out = len;
  return 0;
}
