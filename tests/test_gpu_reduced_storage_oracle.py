"""Reduced-storage modes against the CPU ORACLE (not against the default GPU path): score-only
sweeps, memory-lean traceback and the automatic switch to it under a tiny HBM budget -- modes the
engine may select by itself (bialign_capi.hip, "served from reduced storage instead of failing").
Seeded and bounded: 72 batches of multi-strip shapes, max_shift 0..5, re-sweep widths 1 / 3 / 32,
LOOKUP and DENSE mu2, both recurrences, beta of either sign."""
import numpy as np
import pytest

from bialign_amd import synth

pytestmark = pytest.mark.gpu

RR = {0: 63, 1: 20, 2: 11, 3: 8, 4: 6, 5: 4}  # lattice rows per strip


def one_case(case, mode, k, monkeypatch):
    from oracle import oracle
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    rng = np.random.default_rng(7000 + case)
    s = case % 6
    affine = case % 5 != 4
    beta = (int(rng.choice([-300, -150, -1, 60])) if affine else 0)
    params = dict(synth.PROTEIN_PARAMS, max_shift=s, gap_opening_cost=beta,
                  gap_cost=int(rng.integers(-300, 1)), shift_cost=int(rng.integers(-400, 1)),
                  structure_weight=int(rng.integers(0, 1200)))
    hi = {0: 260, 1: 170, 2: 110, 3: 80, 4: 60, 5: 44}[s]     # several strips, oracle done in well under a second
    lo = 2 * RR[s] + 1 if s else 70
    shapes = [(int(rng.integers(lo, hi)), int(rng.integers(lo // 2, hi))) for _ in range(3)] + [(1, 1), (hi, 3)]
    pairs = [synth.protein_pair(int(rng.integers(1 << 30)), n, m) for n, m in shapes]
    dense = case % 3 == 1
    tabs = [rng.integers(-500, 1500, size=(n, m)).astype(np.int32) for n, m in shapes] if dense else None
    monkeypatch.setenv("BIALIGN_RESW_K", str(k))
    kw = dict(score_only=(mode == "score_only"), lean_trace=(mode == "lean_trace"))
    if mode == "auto_lean":   # a budget below one pair's full layers: the engine must switch to lean traceback itself
        big = max(range(len(shapes)), key=lambda t: shapes[t][0] * shapes[t][1])
        probe = make_batch([pairs[big]], params, mu2_dense=[tabs[big]] if dense else None)
        kw["hbm_budget_bytes"] = int(probe.info["hbm_layer_bytes"] * 0.6)
        probe.close()
    b = make_batch(pairs, params, mu2_dense=tabs, **kw)
    if mode == "auto_lean":
        assert b.info["storage"] == 2  # BIALIGN_BATCH_LEAN_TRACE, chosen by the engine
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
        ctx = (case, t, n, m, s, mode, k, dense, params)
        assert int(scores[t]) == ref["score"], ctx
        if traces is not None:
            assert trace_codes_to_columns(traces[t]) == oracle.trace_to_lists(ref["trace"]), ctx
            assert bool(ok[t]) == ref["complete"], ctx


@pytest.mark.parametrize("k", [1, 3, 32])
@pytest.mark.parametrize("case", range(12))
def test_lean_trace_vs_oracle(case, k, monkeypatch):
    one_case(case, "lean_trace", k, monkeypatch)


@pytest.mark.parametrize("case", range(12, 24))
def test_score_only_vs_oracle(case, monkeypatch):
    one_case(case, "score_only", 1, monkeypatch)


@pytest.mark.parametrize("k", [1, 32])
@pytest.mark.parametrize("case", range(24, 36))
def test_tiny_budget_switches_to_lean_trace_vs_oracle(case, k, monkeypatch):
    one_case(case, "auto_lean", k, monkeypatch)
