"""max_shift above the tiled kernels (BIALIGN_MAX_SHIFT_TILED = 5): the engine's anti-diagonal path
(bialign_wide.hpp) against vectors of the compiled reference (tests/golden/wide_band.json: scores,
traces, every layer cell at max_shift 6, 7, 8, 10) and against the CPU oracle on larger shapes,
batches, chunked batches and dense mu2.  The reference takes any band width (pyx:25-35)."""
import numpy as np
import pytest

from conftest import load_golden
from bialign_amd import synth
from test_gpu_parity import check_against_golden

pytestmark = pytest.mark.gpu

WIDE = load_golden("wide_band.json")
LIN = dict(gap_opening_cost=0, gap_cost=-200, shift_cost=-250)


@pytest.mark.parametrize("rec", WIDE, ids=[r["name"] for r in WIDE])
def test_golden_wide_band(rec):
    check_against_golden(rec)


@pytest.mark.parametrize("n,m,s,ov", [(40, 37, 6, {}), (25, 60, 9, {}), (33, 33, 16, {}), (50, 20, 6, LIN),
                                       (31, 44, 12, LIN), (70, 64, 6, dict(gap_opening_cost=60)), (6, 5, 40, {})])
def test_full_layers_vs_oracle(n, m, s, ov):
    from oracle import oracle
    from test_gpu_parity import gpu_solve
    sa, sb, ta, tb = synth.protein_pair(3000 + n + s, n, m)
    params = dict(synth.PROTEIN_PARAMS, max_shift=s, **ov)
    ref = oracle.solve(sa, sb, ta, tb, params)
    got = gpu_solve(dict(seqA=sa, seqB=sb, strA=ta, strB=tb, params=params), layers=True)
    assert got["score"] == ref["score"]
    assert got["trace"] == oracle.trace_to_lists(ref["trace"])
    assert got["complete"] == ref["complete"]
    for g, e in zip(oracle.band_values(got["layers"], n, m, s), oracle.band_values(ref["layers"], n, m, s)):
        np.testing.assert_array_equal(g, e)


@pytest.mark.parametrize("dense", [False, True])
def test_ragged_chunked_batch_vs_oracle(dense):
    from oracle import oracle
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    rng = np.random.default_rng(77)
    shapes = [(30, 28), (5, 41), (41, 5), (1, 1), (17, 18), (36, 36), (2, 30), (24, 9)]
    pairs = [synth.protein_pair(3100 + t, n, m) for t, (n, m) in enumerate(shapes)]
    tabs = [rng.integers(-500, 1500, size=(n, m)).astype(np.int32) for n, m in shapes] if dense else None
    params = dict(synth.PROTEIN_PARAMS, max_shift=7)
    for budget in (0, 16 << 20):   # (the largest pair's layers are 14.8 MB: 12-dword cells since round 3)
        b = make_batch(pairs, params, hbm_budget_bytes=budget, mu2_dense=tabs)
        if budget:
            assert b.info["nchunks"] > 1
        b.run()
        scores = b.scores()
        traces, ok = b.traces()
        b.close()
        for t, (pair, (n, m)) in enumerate(zip(pairs, shapes)):
            mu1, mu2 = oracle.mu_tables(*pair, params)
            if dense:
                mu2 = np.zeros((n + 1, m + 1), dtype=np.int32)
                mu2[1:, 1:] = tabs[t]
            ref = oracle.solve_tables(n, m, params, mu1, mu2)
            assert int(scores[t]) == ref["score"]
            assert trace_codes_to_columns(traces[t]) == oracle.trace_to_lists(ref["trace"])
            assert bool(ok[t]) == ref["complete"]


@pytest.mark.parametrize("parts", ["1", "3", "64"])
def test_parts_per_pair(parts, monkeypatch):
    """A pair's levels spread over several workgroups (level counter in HBM, write-through stores): same layers."""
    from oracle import oracle
    from test_gpu_parity import gpu_solve
    monkeypatch.setenv("BIALIGN_WIDE_PARTS", parts)
    n, m, s = 70, 64, 6
    sa, sb, ta, tb = synth.protein_pair(3300, n, m)
    params = dict(synth.PROTEIN_PARAMS, max_shift=s)
    ref = oracle.solve(sa, sb, ta, tb, params)
    got = gpu_solve(dict(seqA=sa, seqB=sb, strA=ta, strB=tb, params=params), layers=True)
    assert got["score"] == ref["score"] and got["trace"] == oracle.trace_to_lists(ref["trace"])
    for g, e in zip(oracle.band_values(got["layers"], n, m, s), oracle.band_values(ref["layers"], n, m, s)):
        np.testing.assert_array_equal(g, e)
    assert got["timing"]["cross_cu"] == (parts != "1")


def test_lost_co_residency_is_recovered(monkeypatch):
    """Spin limit zero: the level barrier gives up at once; the run is repeated with one workgroup per pair."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    monkeypatch.setenv("BIALIGN_XCU_SPIN_LIMIT", "0")
    params = dict(synth.PROTEIN_PARAMS, max_shift=7, gap_opening_cost=0, gap_cost=-200, shift_cost=-250)
    pairs = [synth.protein_pair(3400 + t, 50 + t, 60) for t in range(3)]
    for p in (params, dict(synth.PROTEIN_PARAMS, max_shift=6)):
        b = make_batch(pairs, p)
        b.run()
        t = b.timing()
        assert t["recovered_runs"] == 1 and not t["cross_cu"]
        scores = b.scores()
        traces, ok = b.traces()
        b.close()
        for k, pair in enumerate(pairs):
            ref = oracle.solve(*pair, p)
            assert int(scores[k]) == ref["score"]
            assert trace_codes_to_columns(traces[k], as_tuples=(p["gap_opening_cost"] == 0)) == \
                (oracle.trace_to_lists(ref["trace"]) if p["gap_opening_cost"] else [tuple(c) for c in oracle.trace_to_lists(ref["trace"])])


def test_reduced_storage_beyond_the_tiled_band():
    """Round 3: score-only batches of the affine recurrence exist for wide bands too (the ring of derived values is
    all the sweep keeps); the memory-lean traceback and the one-layer recurrence's score-only form are still refused."""
    from bialign_amd import _lib
    from bialign_amd.batch import make_batch
    pair = synth.protein_pair(3200, 20, 20)
    for params, kw in ((dict(synth.PROTEIN_PARAMS, max_shift=6), dict(lean_trace=True)),
                       (dict(synth.PROTEIN_PARAMS, max_shift=6, **LIN), dict(score_only=True))):
        with pytest.raises(_lib.BialignError) as e:
            make_batch([pair], params, **kw)
        assert e.value.code == _lib.E_UNSUPPORTED


@pytest.mark.parametrize("s,parts", [(6, None), (8, "3"), (12, "1"), (7, "64")])
def test_score_only_wide_band_vs_oracle(s, parts, monkeypatch):
    from oracle import oracle
    from bialign_amd import _lib
    from bialign_amd.batch import make_batch
    if parts:
        monkeypatch.setenv("BIALIGN_WIDE_PARTS", parts)
    pairs = [synth.protein_pair(3300 + t, 30 + 11 * t, 55 - 9 * t) for t in range(4)] + [synth.protein_pair(3310, 3, 40)]
    for params in (dict(synth.PROTEIN_PARAMS, max_shift=s), dict(synth.PROTEIN_PARAMS, max_shift=s, gap_opening_cost=70)):
        b = make_batch(pairs, params, score_only=True)
        full = make_batch(pairs, params)
        assert b.info["hbm_layer_bytes"] * 100 < full.info["hbm_layer_bytes"]
        b.run()
        full.run()
        np.testing.assert_array_equal(b.scores(), full.scores())
        assert [int(x) for x in b.scores()] == [oracle.solve(*p, params, want_trace=False)["score"] for p in pairs]
        with pytest.raises(_lib.BialignError):
            b.traces()
        b.close()
        full.close()


def test_cli_takes_max_shift_8():
    """The drop-in CLI with --max_shift 8 (the reference's option has no upper bound, bialign.py:83)."""
    import contextlib, io
    from bialign_amd import cli
    rec = next(r for r in WIDE if r["name"] == "rna_s72_22x24")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        cli.main([rec["seqA"], rec["seqB"], "--strA", rec["strA"], "--strB", rec["strB"], "--structure_weight", "400",
                  "--gap_opening_cost", "-200", "--gap_cost", "-50", "--shift_cost", "-150", "--max_shift", "8"])
    text = buf.getvalue()
    assert f"SCORE: {rec['score']}" in text
    for line in rec["decode"]["default"]["lines"]:
        assert line in text
