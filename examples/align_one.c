/* align_one.c -- the C ABI (include/bialign.h) from plain C, no Python, no torch.
 *
 *   gcc -std=c11 -Iinclude examples/align_one.c -o /tmp/align_one -Lbialign_amd -lbialign_hip \
 *       -Wl,-rpath,$PWD/bialign_amd && /tmp/align_one
 *
 * Aligns the reference README's RNA toy (README.md:91-103 there): GCGGGGGAUAUCCCCAUCG /
 * GGGGAUAUCCCCAUCG with structures ...(((.....))).....  /  .(((.....)))....  and prints
 * "SCORE: 6800" plus the trace as one hex digit per column (bit 3..0 = rows seqA, seqB, strA, strB).
 * The host side prepares what bialign_amd/scoring.py prepares: residue codes, the three RNA structure
 * classes (unpaired / opens to the right / closes to the left with its partner before i-1) and the
 * two score tables. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bialign.h"

#define OK(call)                                                        \
  do {                                                                  \
    if ((call) != BIALIGN_OK) {                                         \
      fprintf(stderr, "%s: %s\n", #call, bialign_last_error());         \
      return 1;                                                         \
    }                                                                   \
  } while (0)

static void encode(const char* seq, const char* str, uint8_t* code, uint8_t* cls) {
  int n = (int)strlen(seq), stack[256], top = 0;
  for (int i = 0; i < n; ++i) {
    code[i] = (uint8_t)(strchr("ACGU", seq[i]) - "ACGU");
    cls[i] = 0; /* unpaired */
    if (str[i] == '(') stack[top++] = i;
    if (str[i] == ')') {
      int partner = stack[--top];
      cls[partner] = 1;                        /* "down": pairs with a later position */
      cls[i] = partner < i - 1 ? 2 : 0;        /* "up" needs the partner strictly before i-1 */
    }
  }
}

int main(void) {
  const char *seqA = "GCGGGGGAUAUCCCCAUCG", *strA = "...(((.....))).....";
  const char *seqB = "GGGGAUAUCCCCAUCG", *strB = ".(((.....)))....";
  int32_t n = (int32_t)strlen(seqA), m = (int32_t)strlen(seqB);
  uint8_t ca[64], sa[64], cb[64], sb[64];
  encode(seqA, strA, ca, sa);
  encode(seqB, strB, cb, sb);

  int32_t s1[16], s2[9];                       /* match 100 / mismatch 0; structure_weight 400 on equal classes */
  for (int x = 0; x < 16; ++x) s1[x] = (x / 4 == x % 4) ? 100 : 0;
  for (int x = 0; x < 9; ++x) s2[x] = (x / 3 == x % 3) ? 400 : 0;

  bialign_params prm = {.gap_opening_cost = -200, .gap_cost = -50, .shift_cost = -150, .max_shift = 1,
                        .recurrence = BIALIGN_REC_AUTO, .flags = 0};
  bialign_scoring sc = {.k1 = 4, .s1 = s1, .k2 = 3, .s2 = s2};
  int64_t zero = 0;
  bialign_pairs pr = {.npairs = 1, .len_a = &n, .len_b = &m, .off_a = &zero, .off_b = &zero,
                      .seq_a = ca, .cls_a = sa, .seq_b = cb, .cls_b = sb, .mu2_dense = NULL, .mu2_off = NULL};

  bialign_engine* eng = NULL;
  bialign_batch* b = NULL;
  OK(bialign_engine_create(0, &eng));
  OK(bialign_batch_create(eng, &prm, &sc, &pr, 0, &b));
  OK(bialign_batch_run(b, 0));

  bialign_batch_info info;
  OK(bialign_batch_get_info(b, &info));
  int32_t score = 0, len = 0, complete = 0;
  int64_t off = 0;
  uint8_t* trace = malloc((size_t)info.trace_bytes);
  OK(bialign_batch_get_scores(b, &score));
  OK(bialign_batch_get_traces(b, trace, &off, &len, &complete));
  printf("SCORE: %d\n", score);
  printf("TRACE (%d columns, %s): ", len, complete ? "complete" : "incomplete");
  for (int t = 0; t < len; ++t) printf("%x", trace[off + t]);
  printf("\n");
  free(trace);
  bialign_batch_destroy(b);
  bialign_engine_destroy(eng);
  return score == 6800 ? 0 : 2;
}
