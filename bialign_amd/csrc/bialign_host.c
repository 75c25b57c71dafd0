/*
 * bialign_host.c -- host-side (CPU) helpers of the presentation layer, plain C.
 *
 * Not part of the DP engine: the GPU computes score and trace; turning an RNA
 * trace into the reference's text needs a maximum-expected-accuracy fold of the
 * consensus pair matrix (reference bialignment.pyx:836-886), an O(L^2 * candidates)
 * recursion that costs tens of seconds in Python at L ~ 2000 columns -- a thousand
 * times the DP itself once that runs on the GPU.
 * Same arithmetic (IEEE doubles, same operation order) and the same tie-breaking as
 * bialign_amd/presentation.py::mea, which it accelerates and is tested against.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* sbpp: (n+1)x(n+1) row-major doubles, 1-based, diagonal = unpaired probabilities.
 * out: n bytes, 0 = unpaired, 1 = opening, 2 = closing.  Returns 0, or -1 on
 * allocation failure.  *score receives F[1][n]. */
int bialign_host_mea(const double *sbpp, int n, double gamma, uint8_t *out, double *score) {
  const size_t w = (size_t)n + 2;
  double *F = (double *)calloc(w * w, sizeof(double));
  int32_t *T = (int32_t *)calloc(w * w, sizeof(int32_t));
  /* candidate lists per right end j, grown on demand */
  int32_t **ck = (int32_t **)calloc((size_t)n + 1, sizeof(int32_t *));
  double **cc = (double **)calloc((size_t)n + 1, sizeof(double *));
  int32_t *cn = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
  int32_t *cap = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
  int rc = 0;
  if (!F || !T || !ck || !cc || !cn || !cap) { rc = -1; goto done; }
#define SB(i, j) sbpp[(size_t)(i) * ((size_t)n + 1) + (j)]
#define PUSH(j, k, c)                                                          \
  do {                                                                         \
    if (cn[j] == cap[j]) {                                                     \
      cap[j] = cap[j] ? 2 * cap[j] : 8;                                        \
      ck[j] = (int32_t *)realloc(ck[j], sizeof(int32_t) * cap[j]);             \
      cc[j] = (double *)realloc(cc[j], sizeof(double) * cap[j]);               \
      if (!ck[j] || !cc[j]) { rc = -1; goto done; }                            \
    }                                                                          \
    ck[j][cn[j]] = (k);                                                        \
    cc[j][cn[j]] = (c);                                                        \
    ++cn[j];                                                                   \
  } while (0)
  for (int i = n; i >= 1; --i) {
    PUSH(i, i, SB(i, i));
    double *row = F + (size_t)i * w;
    for (int j = i; j <= n; ++j) {
      double val = row[j];
      int32_t arg = T[(size_t)i * w + j];
      for (int t = 0; t < cn[j]; ++t) { /* strict improvements only, in order of discovery */
        const double alt = row[ck[j][t] - 1] + cc[j][t];
        if (val < alt) { val = alt; arg = ck[j][t]; }
      }
      row[j] = val;
      T[(size_t)i * w + j] = arg;
      if (i + 3 >= j) continue;
      const double closed = F[(size_t)(i + 1) * w + (j - 1)] + 2 * gamma * SB(i, j);
      if (closed > row[j]) {
        PUSH(j, i, closed);
        row[j] = closed;
        T[(size_t)i * w + j] = i;
      }
    }
  }
  memset(out, 0, (size_t)n);
  {
    /* explicit stack of (i, j) intervals */
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * 4 * ((size_t)n + 2));
    if (!st) { rc = -1; goto done; }
    int sp = 0;
    st[sp++] = 1; st[sp++] = n;
    while (sp > 0) {
      const int j = st[--sp], i = st[--sp];
      if (i + 3 >= j) continue;
      const int k = T[(size_t)i * w + j];
      if (k == 0) continue;
      if (k == j) { st[sp++] = i; st[sp++] = j - 1; continue; }
      out[k - 1] = 1;
      out[j - 1] = 2;
      if (k != i) { st[sp++] = i; st[sp++] = k - 1; }
      st[sp++] = k + 1; st[sp++] = j - 1;
    }
    free(st);
  }
  if (score) *score = n >= 1 ? F[(size_t)1 * w + n] : 0.0;
done:
  if (ck) for (int j = 0; j <= n; ++j) free(ck[j]);
  if (cc) for (int j = 0; j <= n; ++j) free(cc[j]);
  free(ck); free(cc); free(cn); free(cap); free(F); free(T);
  return rc;
}
