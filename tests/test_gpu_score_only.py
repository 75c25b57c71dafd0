"""Score-only batches (BIALIGN_BATCH_SCORE_ONLY): the sweep keeps only the rows the next strip
replays and writes the end-cell score itself.  Scores must equal the full path's and the oracle's."""
import numpy as np
import pytest

from bialign_amd import synth

pytestmark = pytest.mark.gpu

LIN = dict(gap_opening_cost=0, gap_cost=-200, shift_cost=-250)


def both_ways(pairs, params, **kw):
    from bialign_amd.batch import make_batch
    out = []
    for lean in (False, True):
        b = make_batch(pairs, params, score_only=lean, **kw)
        b.run()
        out.append((b.scores().copy(), dict(b.info), b.timing()))
        b.close()
    return out


@pytest.mark.parametrize("s", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("ov", [{}, LIN, dict(gap_opening_cost=100)], ids=["affine", "linear", "beta>0"])
def test_scores_equal_full_path_and_oracle(s, ov):
    from oracle import oracle
    rng = np.random.default_rng(100 + s)
    shapes = [(int(rng.integers(1, 90)), int(rng.integers(1, 90))) for _ in range(24)] + [(1, 1), (64, 64), (63, 1)]
    pairs = [synth.protein_pair(700 + t, n, m) for t, (n, m) in enumerate(shapes)]
    params = dict(synth.PROTEIN_PARAMS, max_shift=s, **ov)
    (full, info_f, _), (lean, info_l, _) = both_ways(pairs, params)
    np.testing.assert_array_equal(lean, full)
    for t in range(0, len(pairs), 3):
        assert int(lean[t]) == oracle.solve(*pairs[t], params, want_trace=False)["score"]
    assert info_l["hbm_layer_bytes"] * 3 < info_f["hbm_layer_bytes"]   # RR rows -> 1 row per strip (RR = 4 at s=5)


@pytest.mark.parametrize("n,m,s,ov,team", [
    (300, 310, 1, {}, "2"), (130, 420, 1, {}, "2"), (420, 400, 0, {}, "2"), (150, 400, 2, {}, "4"),
    (330, 650, 1, {}, "x8"), (700, 1300, 1, {}, "x16"), (300, 320, 1, LIN, "2"), (200, 400, 2, LIN, "4"),
    (330, 650, 1, LIN, "8"), (257, 129, 3, {}, None), (500, 480, 1, {}, None),
    (200, 470, 2, {}, "8"), (360, 810, 2, {}, "h2")])
def test_multi_strip_and_team_shapes(n, m, s, ov, team, monkeypatch):
    if team:
        monkeypatch.setenv("BIALIGN_TEAM", team)
    pairs = [synth.protein_pair(800 + t, n - t, m + t) for t in range(3)]
    params = dict(synth.PROTEIN_PARAMS, max_shift=s, **ov)
    (full, _, tf), (lean, _, tl) = both_ways(pairs, params)
    np.testing.assert_array_equal(lean, full)
    assert tl["waves_per_pair"] == tf["waves_per_pair"] and tl["cross_cu"] == tf["cross_cu"]
    if team:
        assert tl["waves_per_pair"] == (8 * int(team[1:]) if team[0] == "h" else int(team.lstrip("x")))


@pytest.mark.parametrize("team,n,m", [("2", 300, 310), ("3", 300, 320), ("6", 330, 650), ("12", 500, 1000)])
def test_three_waves_per_simd_sweep_score_only_and_lean_trace(team, n, m, monkeypatch):
    """s=1, teams of 2, 3, 6, 12: fill_affine_slim_kernel in its LEAN form (score-only batches and the sweep of the
    memory-lean traceback) against the oracle -- scores, and the traces the strip re-sweeps recover from its rows."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    monkeypatch.setenv("BIALIGN_TEAM", team)
    pairs = [synth.protein_pair(880 + t, n - 7 * t, m + 5 * t) for t in range(3)]
    params = dict(synth.PROTEIN_PARAMS)
    refs = [oracle.solve(*p, params) for p in pairs]
    b = make_batch(pairs, params, score_only=True)
    b.run()
    assert b.timing()["waves_per_pair"] == int(team)
    assert [int(x) for x in b.scores()] == [r["score"] for r in refs]
    b.close()
    b = make_batch(pairs, params, lean_trace=True)
    b.run()
    assert b.timing()["waves_per_pair"] == int(team)
    traces, ok = b.traces()
    assert [int(x) for x in b.scores()] == [r["score"] for r in refs]
    for t, r in enumerate(refs):
        assert trace_codes_to_columns(traces[t]) == oracle.trace_to_lists(r["trace"]) and bool(ok[t]) == r["complete"]
    b.close()


def test_dense_mu2_score_only():
    rng = np.random.default_rng(5)
    shapes = [(130, 75), (75, 130), (40, 50), (300, 280)]
    pairs = [synth.protein_pair(900 + t, n, m) for t, (n, m) in enumerate(shapes)]
    tabs = [rng.integers(-300, 900, size=(n, m)).astype(np.int32) for n, m in shapes]
    for params in (dict(synth.PROTEIN_PARAMS), dict(synth.PROTEIN_PARAMS, max_shift=2, **LIN)):
        (full, _, _), (lean, _, _) = both_ways(pairs, params, mu2_dense=tabs)
        np.testing.assert_array_equal(lean, full)


def test_chunked_score_only_and_refusals():
    from bialign_amd.batch import make_batch
    from bialign_amd._lib import BialignError
    pairs = [synth.protein_pair(950 + t, 120, 110) for t in range(40)]
    params = dict(synth.PROTEIN_PARAMS)
    b = make_batch(pairs, params)
    b.run(fill_only=True)
    want = b.scores().copy()
    b.close()
    b = make_batch(pairs, params, score_only=True, hbm_budget_bytes=1 << 20)
    assert b.info["nchunks"] > 1
    b.run()
    np.testing.assert_array_equal(b.scores(), want)
    with pytest.raises(BialignError):
        b.traces()
    with pytest.raises(BialignError):
        b.dump_layers(0)
    b.close()


def test_config2_shape_sample():
    """256 pairs of BASELINE config 2's shape: same scores, a twentieth of the layer memory."""
    pairs = synth.protein_batch(256, 512)
    (full, info_f, tf), (lean, info_l, tl) = both_ways(pairs, dict(synth.PROTEIN_PARAMS))
    np.testing.assert_array_equal(lean, full)
    assert info_f["layer_bytes"] / info_l["hbm_layer_bytes"] > 15   # (36 B per cell against what the lean sweep keeps)
    print(f"fill ms full {tf['fill_ms']:.2f} lean {tl['fill_ms']:.2f}")
