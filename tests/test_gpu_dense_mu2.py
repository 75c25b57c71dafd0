"""GPU parity of the DENSE-mu2 form of the C ABI (bialign_pairs.mu2_dense): per-pair integer
tables instead of class codes -- what RNA alignments with predicted (real-valued) structure
features need (reference bialignment.pyx:415-423).  Against the golden vectors the compiled
reference produced with injected fractional features, and against the oracle on random tables."""
import numpy as np
import pytest

from conftest import load_golden
from bialign_amd import synth

pytestmark = pytest.mark.gpu

FEATURES = load_golden("fractional_features.json")


def dense_solve(pairs, tables, params, layers_of=None, budget=0):
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    b = make_batch(pairs, params, mu2_dense=tables, hbm_budget_bytes=budget)
    b.run()
    scores = [int(v) for v in b.scores()]
    traces, ok = b.traces()
    out = dict(scores=scores, traces=[trace_codes_to_columns(t) for t in traces],
               complete=[bool(v) for v in ok], timing=b.timing(), info=dict(b.info))
    if layers_of is not None:
        out["layers"] = b.dump_layers(layers_of)
    b.close()
    return out


@pytest.mark.parametrize("rec", FEATURES, ids=[r["name"] for r in FEATURES])
def test_golden_fractional_features_through_the_c_abi(rec):
    from oracle import oracle
    n, m, p = len(rec["seqA"]), len(rec["seqB"]), rec["params"]
    tab = np.array(rec["mu2"], dtype=np.int32)
    got = dense_solve([(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"])], [tab], p, layers_of=0)
    assert got["scores"][0] == rec["score"]
    assert got["traces"][0] == rec["trace"]
    assert got["complete"][0] == rec["complete"]
    if "layers" in rec:
        vals = oracle.band_values(got["layers"], n, m, p["max_shift"])
        for g, e in zip(vals, rec["layers"]):
            np.testing.assert_array_equal(g, np.array(e, dtype=np.int64))


@pytest.mark.parametrize("rec", FEATURES, ids=[r["name"] for r in FEATURES])
def test_golden_fractional_features_through_bialigner(rec):
    """The drop-in class with the same injection the golden script used on the reference."""
    import contextlib
    import io
    from bialign_amd import bialignment as ba
    feats = {rec["seqA"]: rec["featuresA"], rec["seqB"]: rec["featuresB"]}

    class FeatureAligner(ba.BiAligner):
        def _preprocess_seq(self, sequence, structure):
            mol = super()._preprocess_seq(sequence, structure)
            f = feats[str(sequence)]
            mol["up"], mol["down"], mol["unp"] = f["up"], f["down"], f["unp"]
            return mol

    b = FeatureAligner(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"], **rec["params"])
    assert int(b.optimize()) == rec["score"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        trace = b.traceback()
    assert [[int(v) for v in col] for col in trace] == rec["trace"]
    assert ("WARNING" not in buf.getvalue()) == rec["complete"]
    n, m = len(rec["seqA"]), len(rec["seqB"])
    assert [[int(b.mu2(k, l)) for l in range(1, m + 1)] for k in range(1, n + 1)] == rec["mu2"]


def random_table(rng, n, m, lo=-300, hi=900):
    return rng.integers(lo, hi + 1, size=(n, m)).astype(np.int32)


def oracle_dense(pair, tab, params):
    from oracle import oracle
    sa, sb, ta, tb = pair
    n, m = len(sa), len(sb)
    mu1, _ = oracle.mu_tables(sa, sb, ta, tb, params)
    mu2 = np.zeros((n + 1, m + 1), dtype=np.int32)
    mu2[1:, 1:] = tab
    return oracle.solve_tables(n, m, params, mu1, mu2)


LIN = dict(gap_opening_cost=0, gap_cost=-200, shift_cost=-250)


@pytest.mark.parametrize("n,m,s,seed,ov,team", [
    (130, 75, 1, 61, {}, None), (75, 130, 2, 62, {}, None), (200, 190, 0, 63, {}, None),
    (61, 64, 3, 64, {}, None), (40, 50, 4, 65, {}, None), (33, 45, 5, 66, {}, None),
    (300, 310, 1, 67, {}, "2"), (130, 420, 1, 68, {}, "2"), (420, 400, 0, 69, {}, "2"),
    (150, 400, 2, 70, {}, "2"), (257, 129, 1, 71, dict(gap_opening_cost=100), None),
    (130, 75, 1, 72, LIN, None), (75, 130, 2, 73, LIN, None), (120, 110, 0, 74, LIN, None),
    (50, 60, 3, 75, LIN, None), (33, 45, 5, 76, LIN, None), (300, 320, 1, 77, LIN, "2"),
    (200, 400, 2, 78, LIN, "2"), (330, 650, 1, 79, LIN, "x6"), (200, 500, 2, 80, LIN, "x5")])
def test_dense_full_layers_vs_oracle(n, m, s, seed, ov, team, monkeypatch):
    """Random integer mu2 tables: every layer cell, score and trace against the oracle; single
    wave and teams of two (the dense form's largest team)."""
    from oracle import oracle
    if team:
        monkeypatch.setenv("BIALIGN_TEAM", team)
    rng = np.random.default_rng(seed)
    pair = synth.protein_pair(seed, n, m)
    params = dict(synth.PROTEIN_PARAMS, max_shift=s, **ov)
    tab = random_table(rng, n, m)
    ref = oracle_dense(pair, tab, params)
    got = dense_solve([pair], [tab], params, layers_of=0)
    if team:
        assert got["timing"]["waves_per_pair"] == int(team.lstrip("x")) and got["timing"]["cross_cu"] == team.startswith("x")
    assert got["scores"][0] == ref["score"]
    assert got["traces"][0] == oracle.trace_to_lists(ref["trace"])
    assert got["complete"][0] == ref["complete"]
    gv = oracle.band_values(got["layers"], n, m, s)
    rv = oracle.band_values(ref["layers"], n, m, s)
    for g, e in zip(gv, rv):
        np.testing.assert_array_equal(g, e)


@pytest.mark.parametrize("n,m,s,seed,team,lean", [(170, 400, 1, 81, "4", False), (330, 650, 1, 82, "x6", False), (100, 300, 2, 83, "4", False),
                                                   (200, 400, 2, 84, "x5", False), (90, 400, 3, 85, "x3", False),
                                                   (170, 400, 1, 86, "4", True), (330, 650, 1, 87, "x6", True)])
def test_dense_wider_teams(n, m, s, seed, team, lean, monkeypatch):
    """Dense mu2 with four waves per workgroup and with cross-CU teams (round 2; before: two waves at most)."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    monkeypatch.setenv("BIALIGN_TEAM", team)
    pair = synth.protein_pair(seed, n, m)
    tab = random_table(np.random.default_rng(seed), n, m, -400, 1300)
    params = dict(synth.PROTEIN_PARAMS, max_shift=s)
    ref = oracle_dense(pair, tab, params)
    if lean:   # score-only storage: the score against the oracle
        b = make_batch([pair], params, mu2_dense=[tab], score_only=True)
        b.run()
        t, score = b.timing(), int(b.scores()[0])
        b.close()
        assert score == ref["score"]
    else:
        got = dense_solve([pair], [tab], params, layers_of=0)
        t = got["timing"]
        assert got["scores"][0] == ref["score"] and got["traces"][0] == oracle.trace_to_lists(ref["trace"])
        assert got["complete"][0] == ref["complete"]
        for g, e in zip(oracle.band_values(got["layers"], n, m, s), oracle.band_values(ref["layers"], n, m, s)):
            np.testing.assert_array_equal(g, e)
    assert t["waves_per_pair"] == int(team.lstrip("x")) and t["cross_cu"] == team.startswith("x")


def test_dense_ragged_batch_and_chunking():
    from oracle import oracle
    rng = np.random.default_rng(80)
    shapes = [(40, 33), (5, 90), (90, 5), (64, 64), (1, 1), (17, 18), (100, 100), (2, 50), (1, 70), (70, 1)]
    pairs = [synth.protein_pair(200 + t, n, m) for t, (n, m) in enumerate(shapes)]
    tabs = [random_table(rng, n, m) for n, m in shapes]
    for params in (dict(synth.PROTEIN_PARAMS), dict(synth.PROTEIN_PARAMS, max_shift=2, **LIN)):
        for budget in (0, 5 << 20):
            got = dense_solve(pairs, tabs, params, budget=budget)
            if budget and params["gap_opening_cost"]:
                assert got["info"]["nchunks"] > 1
            for t, pair in enumerate(pairs):
                ref = oracle_dense(pair, tabs[t], params)
                assert got["scores"][t] == ref["score"]
                assert got["traces"][t] == oracle.trace_to_lists(ref["trace"])
                assert got["complete"][t] == ref["complete"]


def test_dense_fuzz_against_oracle():
    from oracle import oracle
    rng = np.random.default_rng(81)
    for it in range(60):
        s = int(rng.integers(0, 6))
        n, m = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        affine = bool(rng.integers(0, 2))
        params = dict(synth.PROTEIN_PARAMS, max_shift=s,
                      gap_opening_cost=int(rng.integers(-300, 60)) if affine else 0,
                      gap_cost=int(rng.integers(-300, 1)), shift_cost=int(rng.integers(-400, 1)))
        if affine and params["gap_opening_cost"] == 0:
            params["gap_opening_cost"] = -1
        pair = synth.protein_pair(300 + it, n, m)
        tab = random_table(rng, n, m, lo=int(rng.integers(-2000, 1)), hi=int(rng.integers(0, 3000)))
        ref = oracle_dense(pair, tab, params)
        got = dense_solve([pair], [tab], params)
        assert got["scores"][0] == ref["score"], (it, n, m, s, params)
        assert got["traces"][0] == oracle.trace_to_lists(ref["trace"]), (it, n, m, s, params)
        assert got["complete"][0] == ref["complete"]


def test_dense_equals_class_form_on_the_same_scores():
    """A dense table built from the class scores gives the class form's layers bit for bit."""
    from bialign_amd.batch import make_batch
    pair = synth.protein_pair(90, 300, 280)
    params = dict(synth.PROTEIN_PARAMS)
    sw = params["structure_weight"]
    tab = np.array([[sw if x == y else 0 for y in pair[3]] for x in pair[2]], dtype=np.int32)
    b = make_batch([pair], params)
    b.run()
    score = int(b.scores()[0])
    want = b.dump_layers(0), score
    b.close()
    got = dense_solve([pair], [tab], params, layers_of=0)
    np.testing.assert_array_equal(got["layers"], want[0])
    assert got["scores"][0] == want[1]


def test_dense_argument_errors():
    from bialign_amd.batch import make_batch
    pair = synth.protein_pair(91, 20, 22)
    with pytest.raises(ValueError):
        make_batch([pair], dict(synth.PROTEIN_PARAMS), mu2_dense=[np.zeros((20, 21), dtype=np.int32)])
    with pytest.raises(ValueError):
        make_batch([pair], dict(synth.PROTEIN_PARAMS), mu2_dense=[])
    big = np.full((20, 22), 1 << 27, dtype=np.int32)   # outside the sentinel-safe score range
    with pytest.raises(Exception) as e:
        make_batch([pair], dict(synth.PROTEIN_PARAMS), mu2_dense=[big])
    assert "safety window" in str(e.value)


def test_bialigner_with_external_base_pair_probabilities():
    """Predicted-structure mode without ViennaRNA: bppA / bppB in fold_compound.bpp() layout."""
    import contextlib
    import io
    import math
    from oracle import oracle
    from bialign_amd import bialignment as ba
    from test_host_mirror import random_bpp
    sa, sb, _, _ = synth.rna_pair(77, 44, 39)
    bpa, bpb = random_bpp(11, len(sa)), random_bpp(12, len(sb))
    for ov in (dict(max_shift=1), dict(max_shift=2, gap_opening_cost=0, gap_cost=-200, shift_cost=-250)):
        params = dict(synth.RNA_PARAMS, nameA="A", nameB="B", **ov)
        b = ba.BiAligner(sa, sb, None, None, bppA=bpa, bppB=bpb, **params)
        A, B, sw = b.molA, b.molB, params["structure_weight"]
        n, m = len(sa), len(sb)
        mu2 = np.zeros((n + 1, m + 1), dtype=np.int32)
        for k in range(1, n + 1):
            for l in range(1, m + 1):   # pyx:416-423, literally
                mu2[k, l] = int(sw * (math.sqrt(A["up"][k] * B["up"][l]) + math.sqrt(A["down"][k] * B["down"][l])
                                      + math.sqrt(A["unp"][k] * B["unp"][l])))
        mu1, _ = oracle.mu_tables(sa, sb, "." * n, "." * m, params)
        ref = oracle.solve_tables(n, m, params, mu1, mu2)
        assert int(b.optimize()) == ref["score"]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            trace = b.traceback()
        assert [[int(v) for v in col] for col in trace] == oracle.trace_to_lists(ref["trace"])
        lines = b.decode_trace(trace)
        assert len(lines) >= 4 and all(isinstance(x, str) for x in lines)
