"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and the
golden vectors of the compiled reference.  Bit-exact: scores, traces, layers."""
import numpy as np
import pytest

from conftest import load_golden
from bialign_amd import synth

pytestmark = pytest.mark.gpu

KNOWN = load_golden("known_answers.json")
SMALL = load_golden("small_layers.json")
MEDIUM = load_golden("medium_traces.json")


def supported(rec):
    from bialign_amd import _lib
    return rec["params"]["max_shift"] <= _lib.MAX_SHIFT


def gpu_solve(rec, layers=False):
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    b = make_batch([(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"])], rec["params"])
    b.run()
    score = int(b.scores()[0])
    traces, ok = b.traces()
    out = dict(score=score, trace=trace_codes_to_columns(traces[0]), complete=bool(ok[0]), timing=b.timing())
    if layers:
        out["layers"] = b.dump_layers(0)
    b.close()
    return out


def check_against_golden(rec):
    from oracle import oracle
    if not supported(rec):
        pytest.skip("max_shift not instantiated")
    want_layers = "layers" in rec
    got = gpu_solve(rec, layers=want_layers)
    assert got["score"] == rec["score"]
    assert got["trace"] == rec["trace"]
    assert got["complete"] == rec["complete"]
    if want_layers:
        n, m, s = len(rec["seqA"]), len(rec["seqB"]), rec["params"]["max_shift"]
        vals = oracle.band_values(got["layers"], n, m, s)
        for g, e in zip(vals, rec["layers"]):
            np.testing.assert_array_equal(g, np.array(e, dtype=np.int64))


@pytest.mark.parametrize("rec", KNOWN, ids=[r["name"] for r in KNOWN])
def test_known_answers(rec):
    check_against_golden(rec)


@pytest.mark.parametrize("rec", SMALL, ids=[r["name"] for r in SMALL])
def test_small_full_layers(rec):
    check_against_golden(rec)


@pytest.mark.parametrize("rec", MEDIUM, ids=[r["name"] for r in MEDIUM])
def test_medium_traces(rec):
    check_against_golden(rec)


@pytest.mark.parametrize("n,m,s,seed", [(130, 75, 1, 5), (75, 130, 2, 6), (200, 200, 0, 7),
                                         (61, 64, 3, 8), (257, 129, 1, 9), (40, 50, 4, 10), (33, 45, 5, 11)])
def test_full_layers_vs_oracle(n, m, s, seed):
    """Several strips / ragged shapes: every layer cell against the oracle."""
    full_layers_check(n, m, s, seed)


def full_layers_check(n, m, s, seed):
    from oracle import oracle
    sa, sb, ta, tb = synth.protein_pair(seed, n, m)
    params = dict(synth.PROTEIN_PARAMS, max_shift=s)
    rec = dict(seqA=sa, seqB=sb, strA=ta, strB=tb, params=params)
    ref = oracle.solve(sa, sb, ta, tb, params)
    got = gpu_solve(rec, layers=True)
    assert got["score"] == ref["score"]
    assert got["trace"] == oracle.trace_to_lists(ref["trace"])
    assert got["complete"] == ref["complete"]
    gv = oracle.band_values(got["layers"], n, m, s)
    rv = oracle.band_values(ref["layers"], n, m, s)
    for g, e in zip(gv, rv):
        np.testing.assert_array_equal(g, e)
    return got["timing"]


def test_ragged_batch_vs_oracle():
    """One launch, pairs of different lengths (ragged batch), chunked by a tiny budget."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    shapes = [(40, 33), (5, 90), (90, 5), (64, 64), (1, 1), (17, 18), (100, 100), (2, 50)]
    pairs = [synth.protein_pair(100 + t, n, m) for t, (n, m) in enumerate(shapes)]
    params = dict(synth.PROTEIN_PARAMS)
    for budget in (0, 5 << 20):
        b = make_batch(pairs, params, hbm_budget_bytes=budget)
        if budget:
            assert b.info["nchunks"] > 1
        b.run()
        scores = b.scores()
        traces, ok = b.traces()
        for t, (sa, sb, ta, tb) in enumerate(pairs):
            ref = oracle.solve(sa, sb, ta, tb, params)
            assert int(scores[t]) == ref["score"]
            assert trace_codes_to_columns(traces[t]) == oracle.trace_to_lists(ref["trace"])
            assert bool(ok[t]) == ref["complete"]
        b.close()


@pytest.mark.parametrize("n,m,s,seed,team", [(300, 300, 1, 21, 2), (130, 420, 1, 22, 2), (420, 400, 0, 23, 2),
                                              (330, 650, 1, 24, 8), (170, 400, 1, 25, 4), (100, 300, 2, 26, 4),
                                              (80, 300, 3, 27, 4)])
def test_team_sweep_full_layers(n, m, s, seed, team, monkeypatch):
    """T waves per pair on interleaved strips (forced): every layer cell, trace and score."""
    monkeypatch.setenv("BIALIGN_TEAM", str(team))
    t = full_layers_check(n, m, s, seed)
    assert t["waves_per_pair"] == team and not t["cross_cu"]


@pytest.mark.parametrize("n,m,s,seed,team", [(330, 650, 1, 31, 8), (700, 1300, 1, 32, 16), (130, 420, 1, 33, 2),
                                              (130, 300, 1, 35, 3), (210, 430, 1, 36, 5), (480, 930, 1, 37, 12),
                                              (200, 400, 2, 38, 7),
                                              (200, 500, 2, 34, 8), (420, 400, 0, 35, 2), (90, 400, 3, 36, 4),
                                              (50, 300, 4, 37, 3), (66, 330, 4, 38, 5), (40, 280, 5, 39, 3)])
def test_cross_cu_team_full_layers(n, m, s, seed, team, monkeypatch):
    """The team spread over one-wave workgroups on different CUs / XCDs (write-through stores,
    progress words in HBM): every layer cell, trace and score."""
    monkeypatch.setenv("BIALIGN_TEAM", "x%d" % team)
    t = full_layers_check(n, m, s, seed)
    assert t["waves_per_pair"] == team and t["cross_cu"]


@pytest.mark.parametrize("n,m,s,seed,team,hybrid", [(180, 440, 2, 51, 8, 0), (200, 470, 2, 52, 8, 1),
                                                     (360, 810, 2, 53, 16, 2), (540, 1180, 2, 54, 24, 3)])
def test_eight_wave_workgroups_s2_full_layers(n, m, s, seed, team, hybrid, monkeypatch):
    """The s=2 sweep with eight waves per workgroup (two per SIMD; half-length ghost blocks, molecule A's
    codes read from global memory), alone and as cross-CU teams of such workgroups: every layer cell."""
    monkeypatch.setenv("BIALIGN_TEAM", ("h%d" % hybrid) if hybrid else "8")
    t = full_layers_check(n, m, s, seed)
    assert t["waves_per_pair"] == team and t["cross_cu"] == (hybrid > 1)


def test_team_sweep_batch_matches_single_wave(monkeypatch):
    """Default policy picks two waves per pair for this batch; results equal the one-wave sweep."""
    from bialign_amd.batch import make_batch
    pairs = synth.protein_batch(48, 512)
    params = dict(synth.PROTEIN_PARAMS)
    out = {}
    for team in ("1", "2"):
        monkeypatch.setenv("BIALIGN_TEAM", team)
        b = make_batch(pairs, params)
        b.run()
        traces, ok = b.traces()
        out[team] = (b.scores().tolist(), [t.tolist() for t in traces], ok.tolist())
        b.close()
    assert out["1"] == out["2"]


def test_fuzz_against_oracle():
    """160 random small problems -- random lengths, max_shift 0..5, both recurrences, random
    (also positive / zero) costs, random match/mismatch or BLOSUM62, RNA and protein -- every layer
    cell, score, trace and completeness flag against the CPU oracle."""
    import random
    from oracle import oracle
    rng = random.Random(20261004)
    checked = 0
    for case in range(160):
        s = rng.choice([0, 1, 1, 2, 2, 3, 4, 5])
        n, m = rng.randint(1, 34), rng.randint(1, 34)
        rna = rng.random() < 0.4
        affine = rng.random() < 0.7
        params = dict(type="RNA" if rna else "Protein", max_shift=s,
                      gap_opening_cost=(rng.choice([-300, -150, -1, 50]) if affine else 0),
                      gap_cost=rng.choice([-200, -50, 0, 30]), shift_cost=rng.choice([-250, -150, -10, 0, 40]),
                      structure_weight=rng.choice([0, 100, 400, 800]),
                      simmatrix=None if rna or rng.random() < 0.3 else "BLOSUM62",
                      sequence_match_similarity=rng.choice([100, 300, 0]),
                      sequence_mismatch_similarity=rng.choice([0, -100, 100]))
        sa, sb, ta, tb = (synth.rna_pair if rna else synth.protein_pair)(5000 + case, n, m)
        rec = dict(seqA=sa, seqB=sb, strA=ta, strB=tb, params=params)
        ref = oracle.solve(sa, sb, ta, tb, params)
        got = gpu_solve(rec, layers=True)
        assert got["score"] == ref["score"], (case, params)
        assert got["trace"] == oracle.trace_to_lists(ref["trace"]), (case, params)
        assert got["complete"] == ref["complete"], (case, params)
        for g, e in zip(oracle.band_values(got["layers"], n, m, s), oracle.band_values(ref["layers"], n, m, s)):
            np.testing.assert_array_equal(g, e, err_msg=str((case, params)))
        checked += 1
    assert checked == 160


def test_dump_layers_of_any_pair_in_a_batch():
    """dump_layers re-fills one pair with that pair's own launch shape, wherever it sits in the batch."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    shapes = [(30, 20), (400, 420), (12, 12), (90, 300)]
    pairs = [synth.protein_pair(700 + t, n, m) for t, (n, m) in enumerate(shapes)]
    params = dict(synth.PROTEIN_PARAMS)
    b = make_batch(pairs, params)
    b.run()
    for t in (1, 3, 0):
        n, m = shapes[t]
        ref = oracle.solve(*pairs[t], params, want_trace=False)
        for g, e in zip(oracle.band_values(b.dump_layers(t), n, m, 1), oracle.band_values(ref["layers"], n, m, 1)):
            np.testing.assert_array_equal(g, e)
    b.close()


@pytest.mark.parametrize("n,m,s,seed,team", [(300, 320, 1, 41, 2), (330, 650, 1, 42, 8), (200, 400, 2, 43, 4),
                                              (420, 400, 0, 44, 2), (150, 300, 3, 45, 2),
                                              (330, 650, 1, 46, "x8"), (700, 1300, 1, 47, "x16"), (200, 500, 2, 48, "x7"),
                                              (90, 400, 4, 49, "x3"), (420, 400, 0, 50, "x2")])
def test_team_sweep_linear_full_layers(n, m, s, seed, team, monkeypatch):
    """The non-affine (13-case) fill with T waves per pair, in one workgroup or (xN) as one-wave workgroups on N
    CUs: the layer, trace and score."""
    from oracle import oracle
    monkeypatch.setenv("BIALIGN_TEAM", str(team))
    sa, sb, ta, tb = synth.protein_pair(seed, n, m)
    params = dict(synth.PROTEIN_PARAMS, max_shift=s, gap_opening_cost=0, gap_cost=-200, shift_cost=-250)
    rec = dict(seqA=sa, seqB=sb, strA=ta, strB=tb, params=params)
    ref = oracle.solve(sa, sb, ta, tb, params)
    got = gpu_solve(rec, layers=True)
    assert got["timing"]["waves_per_pair"] == int(str(team).lstrip("x")) and got["timing"]["cross_cu"] == str(team).startswith("x")
    assert got["score"] == ref["score"] and got["trace"] == oracle.trace_to_lists(ref["trace"])
    for g, e in zip(oracle.band_values(got["layers"], n, m, s), oracle.band_values(ref["layers"], n, m, s)):
        np.testing.assert_array_equal(g, e)
