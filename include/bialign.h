/*
 * bialign.h -- C ABI of libbialign_hip.so, the MI355X (gfx950) engine for the
 * BiAlign hot path: 4-D shift-banded DP fill + traceback.
 *
 * The reference has no FFI; its boundary for this path is the method surface
 * of `cdef class BiAligner` (reference src/bialignment.pyx:155).  Each entry
 * point below names the reference code it replaces.  Everything is plain C:
 * pointers + sizes, no C++/torch types, no exceptions across the boundary.
 * Every function that can fail returns 0 on success and a negative
 * BIALIGN_E_* code otherwise; bialign_last_error() then holds a message for
 * the calling thread.
 *
 * Units of work.  A *pair* is one (A, B) molecule pair, i.e. one BiAligner
 * instance of the reference (src/bialign.py:11).  A *batch* is a set of
 * independent pairs sharing one parameter set; the reference runs a batch of
 * one.  Pairs are given in LOOKUP form:
 *     mu1(i,j) = s1[seq_a[i-1] * k1 + seq_b[j-1]]     (pyx:405-412, 435-436)
 *     mu2(k,l) = s2[cls_a[k-1] * k2 + cls_b[l-1]]     (pyx:414-429, 438-440)
 * with uint8 codes prepared by the host side (bialign_amd/scoring.py).
 *
 * or, for mu2 only, in DENSE form (bialign_pairs.mu2_dense).
 *
 * The engine is GPU only.  There is no CPU fallback behind this ABI.
 */
#ifndef BIALIGN_H
#define BIALIGN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BIALIGN_ABI_VERSION 9

#define BIALIGN_OK 0
#define BIALIGN_E_INVALID (-1)     /* bad argument (message says which) */
#define BIALIGN_E_UNSUPPORTED (-2) /* e.g. reduced storage at max_shift above BIALIGN_MAX_SHIFT_TILED */
#define BIALIGN_E_DEVICE (-3)      /* HIP runtime error */
#define BIALIGN_E_NOMEM (-4)       /* a single pair does not fit the HBM budget */
#define BIALIGN_E_RANGE (-5)       /* scores could leave the int32 safety window */

/* max_shift: any band width the reference takes (pyx:25-35; bialign.py:83 has no upper bound).  Bands up to
 * BIALIGN_MAX_SHIFT_TILED run the tiled register/LDS sweep (the fast path, all storage modes); wider bands
 * run a plain anti-diagonal kernel over layers kept in the reference's own array order (one workgroup per
 * pair, full storage only).  BIALIGN_MAX_SHIFT merely bounds the index arithmetic. */
#define BIALIGN_MAX_SHIFT_TILED 5
#define BIALIGN_MAX_SHIFT 1024
/* Molecule length: the tiled sweep stages both molecules' codes (2 bytes per residue and molecule) next to
 * its exchange arrays in one workgroup's LDS, the tracebacks stage them next to the score tables: the sum
 * must fit 160 KiB, i.e. n + m below ~60 000 residues at small alphabets (BIALIGN_E_UNSUPPORTED beyond). */
#define BIALIGN_NEG_INF (-(1 << 30)) /* the reference's -infinity, pyx:303,484 */

/* bialign_params.recurrence */
#define BIALIGN_REC_AUTO 0
#define BIALIGN_REC_AFFINE 1
#define BIALIGN_REC_LINEAR 2

/* run flags */
#define BIALIGN_RUN_FILL_ONLY 1u /* optimize() without traceback() */
#define BIALIGN_RUN_ASYNC 2u     /* enqueue and return; bialign_batch_wait (or any result getter) completes the run */

/* bialign_params.flags.  SCORE_ONLY: the batch will only ever be asked for scores (optimize()
 * without a later traceback(), e.g. all-against-all scoring): the sweep keeps just the rows the
 * next strip needs instead of all layers -- 1/20 of the HBM footprint and traffic at max_shift 1 --
 * and bialign_batch_get_traces / bialign_batch_dump_layers fail with BIALIGN_E_INVALID.  Beyond
 * BIALIGN_MAX_SHIFT_TILED: the affine recurrence only (no layers at all are stored there, just a ring of the
 * last five anti-diagonal levels' derived values); BIALIGN_E_UNSUPPORTED for the non-affine one. */
#define BIALIGN_BATCH_SCORE_ONLY 1u
/* LEAN_TRACE: full results (scores and traces) from the same reduced storage: after the lean sweep
 * the traceback re-sweeps strips of lattice rows into a per-pair scratch area -- as many at a time as
 * keep the device busy and the HBM budget allows, at most a quarter of the pair's full layers -- and
 * walks through them.  A twelfth to a third of the HBM footprint of the default mode (hbm_budget_bytes
 * decides) for ~1.1-1.3x the time: for pairs whose layers would not fit otherwise.  max_shift <=
 * BIALIGN_MAX_SHIFT_TILED only (BIALIGN_E_UNSUPPORTED beyond). */
#define BIALIGN_BATCH_LEAN_TRACE 2u

typedef struct bialign_engine bialign_engine; /* one per (process, device) */
typedef struct bialign_batch bialign_batch;   /* inputs resident in HBM */

/* BiAligner parameters that reach the DP (pyx:186-188, 230, 259). */
typedef struct bialign_params {
  int32_t gap_opening_cost; /* beta; != 0 selects the affine recurrence (pyx:204-205, 444) */
  int32_t gap_cost;         /* gamma */
  int32_t shift_cost;       /* Delta */
  int32_t max_shift;        /* s >= 0 (see BIALIGN_MAX_SHIFT_TILED) */
  int32_t recurrence;       /* BIALIGN_REC_AUTO: affine iff gap_opening_cost != 0, as optimize()
                               dispatches (pyx:444); BIALIGN_REC_AFFINE = affine_optimize() called
                               directly (pyx:474); BIALIGN_REC_LINEAR = the 13-case recurrence */
  uint32_t flags;           /* BIALIGN_BATCH_* */
} bialign_params;

/* Score tables, row-major, values already scaled (nonpyx:33: x100). */
typedef struct bialign_scoring {
  int32_t k1;        /* sequence alphabet size, 1..256 */
  const int32_t* s1; /* k1*k1 */
  int32_t k2;        /* structure class count, 1..256 */
  const int32_t* s2; /* k2*k2 */
} bialign_scoring;

/* Host-side description of the pairs; copied to HBM by bialign_batch_create. */
typedef struct bialign_pairs {
  int32_t npairs;
  const int32_t* len_a; /* [npairs] n >= 1 */
  const int32_t* len_b; /* [npairs] m >= 1 */
  const int64_t* off_a; /* [npairs] start of pair p in seq_a / cls_a */
  const int64_t* off_b; /* [npairs] start of pair p in seq_b / cls_b */
  const uint8_t* seq_a; /* sequence codes of all A molecules, concatenated */
  const uint8_t* cls_a; /* structure classes, same indexing as seq_a */
  const uint8_t* seq_b;
  const uint8_t* cls_b;
  /* Optional DENSE form of mu2 (NULL = LOOKUP form above): for structure similarities that are
   * not a small class table -- the reference's RNA mode with *predicted* structures, where
   * mu2(k,l) = int(sw*(sqrt(upA upB)+sqrt(dnA dnB)+sqrt(unpA unpB))) of real-valued features
   * (pyx:416-423).  Pair p's table is mu2_dense[mu2_off[p] + (k-1)*m + (l-1)], k=1..n, l=1..m;
   * cls_a / cls_b are then ignored (may be NULL). */
  const int32_t* mu2_dense;
  const int64_t* mu2_off;
} bialign_pairs;

typedef struct bialign_batch_info {
  int32_t npairs;
  int32_t nchunks;        /* HBM-budgeted chunks the batch is processed in */
  int32_t affine;         /* 1 = nine-layer affine recurrence, 0 = one layer */
  int32_t max_shift;
  int64_t cells;          /* in-band lattice points of all pairs (the unit of the metric) */
  int64_t layer_bytes;    /* algorithmic bytes: 36 B (affine) or 4 B per cell */
  int64_t hbm_layer_bytes;/* allocated size of the largest chunk's layer buffer */
  int64_t trace_bytes;    /* capacity of the trace buffer, sum of 2(n+m)+2 */
  int32_t storage;        /* 0 = all layers, BIALIGN_BATCH_SCORE_ONLY, or BIALIGN_BATCH_LEAN_TRACE (asked for, or
                             chosen by the engine because a pair's full layers exceed the HBM budget) */
  int32_t reserved;
} bialign_batch_info;

typedef struct bialign_timing {
  double fill_ms;      /* HIP-event time of the fill kernels of the last run */
  double traceback_ms; /* ... of the traceback (or score-only) kernels */
  int32_t fill_launches;
  int32_t traceback_launches;
  int32_t waves_per_pair; /* team size of the last fill launch (team sweep) */
  int32_t cross_cu;       /* 1 if that team was spread over one-wave workgroups */
  int32_t recovered_runs; /* runs of this batch repeated with in-workgroup teams after a cross-CU team lost
                             co-residency (another tenant on the device); the results are those of the repeat */
  int32_t packed_records; /* 1 if the last run's sweeps stored packed layer records (affine, max_shift 1 or 2: base +
                             16-bit offsets in interior steps, decoded by the tracebacks; chosen by the engine, exact) */
} bialign_timing;

int bialign_abi_version(void);
/* 0 for a product build.  Non-zero: the library was compiled as a kernel TIMING experiment (-DBIALIGN_EXP=n,
   tools/exp_build.sh: sweeps without stores, without hand-off waits, ...) whose results are wrong by
   construction; a binding must refuse such a library (bialign_amd/_lib.py does).  Replaces nothing in the
   reference. */
int bialign_build_experiment(void);
/* Number of visible HIP devices (<0 on error). */
int bialign_device_count(void);
const char* bialign_last_error(void);

/* Replaces nothing in the reference (it has no device); one engine plays the
 * role of the Python process that owns a BiAligner. */
/* Engine: device selection + one HIP stream + event pool + the layer buffer of the last batch
 * (kept for the next one: allocating tens of GB costs far more than sweeping them).  An engine
 * and its batches are used from one thread at a time. */
int bialign_engine_create(int device, bialign_engine** out);
/* Safe in any order with bialign_batch_destroy: an engine with live batches goes when its last batch goes. */
void bialign_engine_destroy(bialign_engine* eng);
/* Give the cached layer buffer back to the device (e.g. before another library needs the HBM). */
int bialign_engine_trim(bialign_engine* eng);
/* Pre-allocate the cached layer buffer (bytes) and choose WHERE it lies: on MI355X the physical
 * region a large allocation lands in decides 10-20 % of the sweep's store rate, and a plain
 * streaming write over the buffer predicts it (profiles/r01e_placement).  Up to `tries` candidate
 * allocations are probed (two memset passes each); the fastest is kept for all later batches of
 * this engine.  For long-running users: costs `tries` large allocations once.  Needs twice the
 * buffer in free HBM while it runs, otherwise it just allocates.  *rate_gbps (may be NULL)
 * receives the kept buffer's probe rate. */
int bialign_engine_reserve(bialign_engine* eng, int64_t bytes, int tries, double* rate_gbps);

/* Upload a batch and allocate its DP storage: BiAligner.__init__ (pyx:179-197)
 * for the part that reaches the DP, plus AffineDPMatrices / SparseMatrix4D
 * allocation (pyx:478, 452).  hbm_budget_bytes = 0 lets the engine use ~85 %
 * of the free device memory; a smaller budget forces more chunks. */
int bialign_batch_create(bialign_engine* eng, const bialign_params* params,
                         const bialign_scoring* scoring, const bialign_pairs* pairs,
                         int64_t hbm_budget_bytes, bialign_batch** out);
void bialign_batch_destroy(bialign_batch* b);
int bialign_batch_get_info(const bialign_batch* b, bialign_batch_info* info);

/* BiAligner.optimize() (pyx:443-509) followed -- unless BIALIGN_RUN_FILL_ONLY --
 * by BiAligner.traceback() (pyx:513-586), for every pair, chunk by chunk.
 * Returns after the device work has completed. */
int bialign_batch_run(bialign_batch* b, uint32_t flags);
/* Completes a BIALIGN_RUN_ASYNC run: waits for the batch's kernels, collects kernel times and the
 * device error flag.  A no-op when nothing is pending.  Lets the host prepare the next batch
 * (encoding, bialign_batch_create: uploads go through their own stream) while this one sweeps. */
int bialign_batch_wait(bialign_batch* b);
int bialign_batch_get_timing(const bialign_batch* b, bialign_timing* t);

/* Optimal scores: the return value of optimize() (pyx:471, 509). */
int bialign_batch_get_scores(const bialign_batch* b, int32_t* scores /* [npairs] */);

/* Traces: the return value of traceback() (pyx:531, 586).  Pair p's columns are
 * trace[trace_off[p] .. trace_off[p] + trace_len[p]), start -> end, one byte per
 * column = o0*8 + o1*4 + o2*2 + o3.  complete[p] == 0 is the condition under
 * which the reference prints "WARNING: incomplete traceback" (pyx:584-585);
 * it is always 1 for the non-affine recurrence.  trace must hold
 * bialign_batch_info.trace_bytes bytes. */
int bialign_batch_get_traces(const bialign_batch* b, uint8_t* trace, int64_t* trace_off,
                             int32_t* trace_len, int32_t* complete);

/* Test / introspection hooks.
 * dump_layers re-runs the fill of one pair and writes its layers in the
 * reference's layout [layer][i][j][k-i+s][l-j+s] (pyx:27-41, 61-71), layers in
 * itertools.product order, cells the reference never writes left 0.
 * out must hold nlayers*(n+1)*(m+1)*(2s+1)^2 int32. */
int bialign_batch_dump_layers(bialign_batch* b, int32_t pair, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* BIALIGN_H */
