"""Randomised parity soak: HIP engine (every storage mode, LOOKUP/DENSE, team shapes) against the CPU
oracle on small random pairs.  usage: python tools/fuzz_gpu.py <seconds> [seed]   (GPU box; test infrastructure)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bialign_amd import synth
from bialign_amd.batch import make_batch
from bialign_amd.engine import trace_codes_to_columns
from oracle import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t_end = time.time() + budget
cases = pairs_done = 0
while time.time() < t_end:
    s = int(rng.integers(0, 6)) if rng.random() < 0.85 else int(rng.integers(6, 11))   # 6..10: the wide-band path
    affine = bool(rng.integers(0, 4))           # 3/4 affine
    beta = int(rng.integers(-400, 80)) if affine else 0
    if affine and beta == 0:
        beta = -7
    params = dict(synth.PROTEIN_PARAMS, max_shift=s, gap_opening_cost=beta, gap_cost=int(rng.integers(-300, 1)),
                  shift_cost=int(rng.integers(-400, 1)), structure_weight=int(rng.integers(0, 1200)))
    npairs = int(rng.integers(1, 9))
    big = rng.random() < float(os.environ.get("FUZZ_BIG", 0.15))   # multi-strip / team-capable shapes
    hi = (420 if big else 90) if s <= 5 else (60 if big else 28)
    if s == 2 and big and rng.random() < 0.5:
        hi, npairs = 900, min(npairs, 2)                              # room for eight-wave workgroups and teams of them
    # beyond the tiled band: full storage, or (round 3, affine only) score-only from the ring of derived values
    mode = ["full", "score_only", "lean_trace"][int(rng.integers(0, 3))] if s <= 5 else ("score_only" if affine and rng.random() < 0.4 else "full")
    # (3, 6, 12 and -- at max_shift 1 -- 2: the three-waves-per-SIMD kernel where the pairs' period admits the team)
    team = str(rng.choice(["", "", "2", "3", "4", "6", "8", "12", "x2", "x3", "x5", "x8", "h1", "h2"]))
    lo = 1
    if s == 1 and affine and big and team in ("2", "3", "6", "12") and rng.random() < (0.15 if team == "12" else 0.7):
        lo, hi = {"2": (270, 420), "3": (290, 420), "6": (500, 560), "12": (940, 1000)}[team]  # every pair long enough for the team
        npairs = 1 if team == "12" else min(npairs, 3)
    shapes = [(int(rng.integers(lo, hi)), int(rng.integers(lo, hi))) for _ in range(npairs)]
    pairs = [synth.protein_pair(int(rng.integers(1 << 30)), n, m) for n, m in shapes]
    dense = rng.random() < 0.3
    tabs = [rng.integers(-500, 1500, size=(n, m)).astype(np.int32) for n, m in shapes] if dense else None
    os.environ.pop("BIALIGN_TEAM", None)
    if team:
        os.environ["BIALIGN_TEAM"] = team
    os.environ["BIALIGN_RESW_K"] = str(int(rng.choice([1, 2, 3, 8, 32])))
    budget_b = int(rng.choice([0, 0, 3 << 20, 12 << 20]))
    try:
        b = make_batch(pairs, params, mu2_dense=tabs, score_only=(mode == "score_only"),
                       lean_trace=(mode == "lean_trace"), hbm_budget_bytes=budget_b)
    except Exception as e:                      # budget too small for a pair even in reduced storage
        if "budget" in str(e):
            continue
        raise
    b.run()
    scores = b.scores()
    traces = ok = None
    if mode != "score_only":
        traces, ok = b.traces()
    b.close()
    for t, (pair, (n, m)) in enumerate(zip(pairs, shapes)):
        mu1, mu2 = oracle.mu_tables(*pair, params)
        if dense:
            mu2 = np.zeros((n + 1, m + 1), dtype=np.int32)
            mu2[1:, 1:] = tabs[t]
        ref = oracle.solve_tables(n, m, params, mu1, mu2, want_trace=(mode != "score_only"))
        ctx = (cases, t, n, m, s, mode, team, dense, params)
        assert int(scores[t]) == ref["score"], ctx
        if traces is not None:
            assert trace_codes_to_columns(traces[t]) == oracle.trace_to_lists(ref["trace"]), ctx
            assert bool(ok[t]) == ref["complete"], ctx
        pairs_done += 1
    cases += 1
    if cases % 200 == 0:
        print(f"{cases} batches, {pairs_done} pairs ok, {t_end - time.time():.0f} s left", flush=True)
print(f"fuzz ok: {cases} batches, {pairs_done} pairs, all equal to the oracle")
