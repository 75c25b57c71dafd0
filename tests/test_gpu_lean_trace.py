"""Memory-lean traceback (BIALIGN_BATCH_LEAN_TRACE): lean sweep, then per strip a re-sweep into a
scratch area and a walk through it.  Scores, traces and completeness flags must equal the default
path's (which the other GPU tests pin to the oracle and the golden vectors)."""
import numpy as np
import pytest

from conftest import load_golden
from bialign_amd import synth

pytestmark = pytest.mark.gpu


def run(pairs, params, **kw):
    from bialign_amd.batch import make_batch
    b = make_batch(pairs, params, **kw)
    b.run()
    traces, ok = b.traces()
    out = (b.scores().copy(), [t.tolist() for t in traces], [bool(v) for v in ok], dict(b.info), b.timing())
    b.close()
    return out


def same(pairs, params, **kw):
    full = run(pairs, params, **kw)
    lean = run(pairs, params, lean_trace=True, **kw)
    np.testing.assert_array_equal(lean[0], full[0])
    for t, (a, b) in enumerate(zip(lean[1], full[1])):
        assert a == b, f"pair {t}: traces differ"
    assert lean[2] == full[2]
    return full, lean


@pytest.mark.parametrize("s", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("beta", [-150, 100], ids=["beta<0", "beta>0"])
def test_random_shapes_equal_default_path(s, beta):
    rng = np.random.default_rng(40 + s)
    shapes = [(int(rng.integers(1, 140)), int(rng.integers(1, 140))) for _ in range(20)] + \
             [(1, 1), (64, 64), (63, 1), (1, 90), (200, 7), (7, 200)]
    pairs = [synth.protein_pair(1200 + t, n, m) for t, (n, m) in enumerate(shapes)]
    same(pairs, dict(synth.PROTEIN_PARAMS, max_shift=s, gap_opening_cost=beta))


@pytest.mark.parametrize("k", ["1", "2", "5", "32"])
@pytest.mark.parametrize("n,m,s", [(500, 480, 1), (300, 650, 1), (650, 300, 1), (420, 400, 0), (257, 300, 2),
                                    (200, 210, 3), (1500, 40, 1), (40, 1500, 1)])
def test_many_strips(n, m, s, k, monkeypatch):
    monkeypatch.setenv("BIALIGN_RESW_K", k)   # strips re-swept and walked per round
    pairs = [synth.protein_pair(1300 + t, n - 3 * t, m + 2 * t) for t in range(3)]
    full, lean = same(pairs, dict(synth.PROTEIN_PARAMS, max_shift=s))
    if n >= 250 and k == "1":  # many strips: one strip of scratch plus the bottom rows is a fraction of all layers
        assert lean[3]["hbm_layer_bytes"] * 3 < full[3]["hbm_layer_bytes"]


def test_golden_rna_and_protein_cases():
    for rec in load_golden("medium_traces.json") + load_golden("known_answers.json"):
        p = rec["params"]
        if p["gap_opening_cost"] == 0 or p["max_shift"] > 5:
            continue
        from bialign_amd.engine import trace_codes_to_columns
        got = run([(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"])], p, lean_trace=True)
        assert int(got[0][0]) == rec["score"]
        assert trace_codes_to_columns(np.array(got[1][0], dtype=np.uint8)) == rec["trace"]
        assert got[2][0] == rec["complete"]


def test_chunked_and_ragged():
    shapes = [(300, 280), (5, 90), (90, 5), (64, 64), (1, 1), (170, 180), (100, 100), (2, 50), (260, 30)]
    pairs = [synth.protein_pair(1400 + t, n, m) for t, (n, m) in enumerate(shapes)]
    params = dict(synth.PROTEIN_PARAMS)
    full = run(pairs, params)
    lean = run(pairs, params, lean_trace=True, hbm_budget_bytes=6 << 20)
    assert lean[3]["nchunks"] > 1
    np.testing.assert_array_equal(lean[0], full[0])
    assert lean[1] == full[1] and lean[2] == full[2]


LIN = dict(gap_opening_cost=0, gap_cost=-200, shift_cost=-250)


@pytest.mark.parametrize("k", ["1", "3", "32"])
@pytest.mark.parametrize("s", [0, 1, 2, 3, 4, 5])
def test_one_layer_recurrence(s, k, monkeypatch):
    monkeypatch.setenv("BIALIGN_RESW_K", k)
    rng = np.random.default_rng(60 + s)
    shapes = [(int(rng.integers(1, 140)), int(rng.integers(1, 140))) for _ in range(14)] + \
             [(1, 1), (64, 64), (63, 1), (1, 90), (400, 380), (30, 700), (700, 30)]
    pairs = [synth.protein_pair(1900 + t, n, m) for t, (n, m) in enumerate(shapes)]
    same(pairs, dict(synth.PROTEIN_PARAMS, max_shift=s, **LIN))


def test_one_layer_dense_and_golden():
    rng = np.random.default_rng(10)
    shapes = [(130, 75), (75, 130), (300, 280)]
    pairs = [synth.protein_pair(1950 + t, n, m) for t, (n, m) in enumerate(shapes)]
    tabs = [rng.integers(-300, 900, size=(n, m)).astype(np.int32) for n, m in shapes]
    same(pairs, dict(synth.PROTEIN_PARAMS, max_shift=2, **LIN), mu2_dense=tabs)
    from bialign_amd.engine import trace_codes_to_columns
    for rec in load_golden("medium_traces.json") + load_golden("known_answers.json"):
        p = rec["params"]
        if p["gap_opening_cost"] != 0 or p["max_shift"] > 5:
            continue
        got = run([(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"])], p, lean_trace=True)
        assert int(got[0][0]) == rec["score"]
        assert trace_codes_to_columns(np.array(got[1][0], dtype=np.uint8)) == rec["trace"]


def test_refusals():
    from bialign_amd.batch import make_batch
    from bialign_amd._lib import BialignError
    pair = synth.protein_pair(1, 30, 30)
    b = make_batch([pair], dict(synth.PROTEIN_PARAMS), lean_trace=True)
    b.run()
    with pytest.raises(BialignError):
        b.dump_layers(0)
    b.close()


def test_config2_shape_sample_and_timing():
    pairs = synth.protein_batch(256, 512)
    full, lean = same(pairs, dict(synth.PROTEIN_PARAMS))
    print(f"full: fill {full[4]['fill_ms']:.2f} tb {full[4]['traceback_ms']:.2f} ms, {full[3]['hbm_layer_bytes'] / 2**30:.1f} GiB | "
          f"lean: fill {lean[4]['fill_ms']:.2f} tb {lean[4]['traceback_ms']:.2f} ms, {lean[3]['hbm_layer_bytes'] / 2**30:.1f} GiB")


def test_engine_falls_back_to_lean_traceback_when_a_pair_exceeds_the_budget():
    """Instead of BIALIGN_E_NOMEM: same results from reduced storage (affine, LOOKUP form)."""
    from bialign_amd._lib import BATCH_LEAN_TRACE, BialignError
    from bialign_amd.batch import make_batch
    pairs = [synth.protein_pair(1500 + t, 400, 380) for t in range(2)]
    params = dict(synth.PROTEIN_PARAMS)
    full = run(pairs, params)
    assert full[3]["storage"] == 0
    lean = run(pairs, params, hbm_budget_bytes=12 << 20)   # full layers of one pair: ~50 MB
    assert lean[3]["storage"] == BATCH_LEAN_TRACE
    np.testing.assert_array_equal(lean[0], full[0])
    assert lean[1] == full[1] and lean[2] == full[2]
    lin = dict(params, **LIN)
    full_l, lean_l = run(pairs, lin), run(pairs, lin, hbm_budget_bytes=2 << 20)   # one-layer: ~5.6 MB per pair
    assert lean_l[3]["storage"] == BATCH_LEAN_TRACE
    np.testing.assert_array_equal(lean_l[0], full_l[0])
    assert lean_l[1] == full_l[1]
    with pytest.raises(BialignError):   # not even the reduced storage of one pair fits: still an error
        make_batch(pairs, params, hbm_budget_bytes=64 << 10)


def test_single_long_pair_many_strips_per_round():
    """One 3000 x 2800 pair: 32 strips per round re-swept by 32 waves in parallel."""
    pairs = [synth.protein_pair(1600, 3000, 2800)]
    full, lean = same(pairs, dict(synth.PROTEIN_PARAMS))
    print(f"full: fill {full[4]['fill_ms']:.1f} tb {full[4]['traceback_ms']:.1f} ms | lean: fill {lean[4]['fill_ms']:.1f} "
          f"tb {lean[4]['traceback_ms']:.1f} ms; layers {full[3]['hbm_layer_bytes'] >> 20} -> {lean[3]['hbm_layer_bytes'] >> 20} MiB")


def test_one_pair_of_length_twenty_thousand():
    """124 GB of layers in the default mode; a third of that here with 250 strips re-swept per round (the budget
    allows it), a twelfth with 32: same score, same 40 000-column trace."""
    pairs = [synth.protein_pair(1700, 20000, 19000)]
    full, lean = same(pairs, dict(synth.PROTEIN_PARAMS))
    assert lean[3]["hbm_layer_bytes"] * 3 < full[3]["layer_bytes"]   # (the default mode itself stores packed records: 73 GB)
    small = run(pairs, dict(synth.PROTEIN_PARAMS), lean_trace=True, hbm_budget_bytes=12 << 30)   # a tight budget: fewer strips per round
    assert small[3]["hbm_layer_bytes"] <= 12 << 30
    np.testing.assert_array_equal(small[0], full[0])
    assert small[1] == full[1] and small[2] == full[2]
    print(f"full: fill {full[4]['fill_ms']:.0f} tb {full[4]['traceback_ms']:.0f} ms, {full[3]['hbm_layer_bytes'] / 2**30:.1f} GiB | "
          f"lean: fill {lean[4]['fill_ms']:.0f} tb {lean[4]['traceback_ms']:.0f} ms, {lean[3]['hbm_layer_bytes'] / 2**30:.1f} GiB")


@pytest.mark.parametrize("k", ["1", "4"])
def test_dense_mu2_form(k, monkeypatch):
    monkeypatch.setenv("BIALIGN_RESW_K", k)
    rng = np.random.default_rng(9)
    shapes = [(130, 75), (75, 130), (40, 50), (300, 280), (1, 1), (250, 30)]
    pairs = [synth.protein_pair(1800 + t, n, m) for t, (n, m) in enumerate(shapes)]
    tabs = [rng.integers(-300, 900, size=(n, m)).astype(np.int32) for n, m in shapes]
    for s in (0, 1, 2, 3):
        same(pairs, dict(synth.PROTEIN_PARAMS, max_shift=s), mu2_dense=tabs)


def test_async_run_of_lean_and_chunked_batches():
    """run(wait=False) enqueues all chunks and all re-sweep rounds without touching the host in between."""
    from bialign_amd.batch import make_batch
    shapes = [(300, 280), (90, 5), (64, 64), (170, 180), (100, 100), (260, 30)]
    pairs = [synth.protein_pair(2200 + t, n, m) for t, (n, m) in enumerate(shapes)]
    params = dict(synth.PROTEIN_PARAMS)
    full = run(pairs, params)
    for kw in (dict(lean_trace=True), dict(hbm_budget_bytes=60 << 20), dict(lean_trace=True, hbm_budget_bytes=6 << 20)):
        b = make_batch(pairs, params, **kw)
        b.run(wait=False)
        traces, ok = b.traces()          # implicit wait
        np.testing.assert_array_equal(b.scores(), full[0])
        assert [t.tolist() for t in traces] == full[1] and [bool(v) for v in ok] == full[2]
        b.close()
