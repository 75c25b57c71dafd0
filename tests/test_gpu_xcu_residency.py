"""Cross-CU teams need all their one-wave workgroups resident at once (a wave spins on its
predecessor's progress word).  The engine sizes such grids from the runtime's occupancy figure,
runs them one at a time per device, and -- when a hand-off still times out because another tenant
holds wave slots -- repeats the run with in-workgroup teams.  Results against the CPU oracle."""
import numpy as np
import pytest

from bialign_amd import synth

pytestmark = pytest.mark.gpu


def oracle_results(pairs, params):
    from oracle import oracle
    out = []
    for sa, sb, ta, tb in pairs:
        ref = oracle.solve(sa, sb, ta, tb, params)
        out.append((ref["score"], oracle.trace_to_lists(ref["trace"]), ref["complete"]))
    return out


def batch_results(b):
    from bialign_amd.engine import trace_codes_to_columns
    scores = b.scores()
    traces, ok = b.traces()
    return [(int(scores[t]), trace_codes_to_columns(traces[t]), bool(ok[t])) for t in range(b.npairs)]


@pytest.mark.parametrize("noserial", [False, True])
def test_two_engines_cross_cu_batches_at_once(noserial, monkeypatch):
    """Two engines (two HIP streams), one cross-CU batch each, runs enqueued back to back without
    waiting.  Serialised launches (default) keep every team co-resident; with the ordering switched
    off the launches may overlap, a team may time out and the run is repeated in one workgroup --
    either way both batches equal the oracle."""
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import Engine
    monkeypatch.setenv("BIALIGN_TEAM", "x8")
    if noserial:
        monkeypatch.setenv("BIALIGN_XCU_NOSERIAL", "1")
    params = [dict(synth.PROTEIN_PARAMS), dict(synth.PROTEIN_PARAMS, max_shift=2)]
    sets = [[synth.protein_pair(900 + t, 330 + 7 * t, 640 - 5 * t) for t in range(6)],
            [synth.protein_pair(950 + t, 200 + 3 * t, 520) for t in range(5)]]
    engines = [Engine(0), Engine(0)]
    batches = [make_batch(sets[k], params[k], engine=engines[k]) for k in range(2)]
    for rep in range(3):
        for b in batches:
            b.run(wait=False)
        for k, b in enumerate(batches):
            b.wait()
            if rep == 0:
                t = b.timing()
                assert t["cross_cu"] or t["recovered_runs"] > 0
    for k, b in enumerate(batches):
        assert batch_results(b) == oracle_results(sets[k], params[k])
        b.close()
    for e in engines:
        e.close()


@pytest.mark.parametrize("s", [1, 2])
def test_lost_co_residency_is_recovered(s, monkeypatch):
    """A spin limit of zero makes every wave that has to wait for its predecessor give up at once --
    what a team sees when its partners are not scheduled.  The run must come back correct (repeated
    with an in-workgroup team), say so in the timing, and the batch must stay usable."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    monkeypatch.setenv("BIALIGN_TEAM", "x8")
    monkeypatch.setenv("BIALIGN_XCU_SPIN_LIMIT", "0")
    params = dict(synth.PROTEIN_PARAMS, max_shift=s)
    pairs = [synth.protein_pair(970 + t, 250 + 11 * t, 560) for t in range(3)]
    b = make_batch(pairs, params)
    b.run()
    t = b.timing()
    assert t["recovered_runs"] == 1 and not t["cross_cu"]
    want = oracle_results(pairs, params)
    assert batch_results(b) == want
    n, m = len(pairs[1][0]), len(pairs[1][1])
    ref = oracle.solve(*pairs[1], params, want_trace=False)
    for g, e in zip(oracle.band_values(b.dump_layers(1), n, m, s), oracle.band_values(ref["layers"], n, m, s)):
        np.testing.assert_array_equal(g, e)
    b.run()  # the batch stays on in-workgroup teams: no second failure, no second repeat
    t = b.timing()
    assert t["recovered_runs"] == 1 and not t["cross_cu"]
    assert batch_results(b) == want
    b.close()


def test_one_layer_recurrence_recovers_too(monkeypatch):
    """The non-affine sweep's cross-CU teams (a single long pair from the CLI's default parameters): same recovery."""
    from bialign_amd.batch import make_batch
    monkeypatch.setenv("BIALIGN_XCU_SPIN_LIMIT", "0")
    params = dict(synth.PROTEIN_PARAMS, gap_opening_cost=0, gap_cost=-200, shift_cost=-250, max_shift=2)
    pairs = [synth.protein_pair(995, 900, 1000)]
    b = make_batch(pairs, params)
    b.run()
    t = b.timing()
    assert t["recovered_runs"] == 1 and not t["cross_cu"]
    assert batch_results(b) == oracle_results(pairs, params)
    b.close()


def test_dump_layers_recovers_too(monkeypatch):
    """dump_layers launches one pair on its own (a cross-CU team here): same recovery."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    monkeypatch.setenv("BIALIGN_XCU_SPIN_LIMIT", "0")
    params = dict(synth.PROTEIN_PARAMS)
    pair = synth.protein_pair(990, 420, 700)
    b = make_batch([pair], params)   # one long pair: the default policy spreads it over CUs
    monkeypatch.setenv("BIALIGN_TEAM", "x8")
    layers = b.dump_layers(0)
    ref = oracle.solve(*pair, params, want_trace=False)
    for g, e in zip(oracle.band_values(layers, 420, 700, 1), oracle.band_values(ref["layers"], 420, 700, 1)):
        np.testing.assert_array_equal(g, e)
    b.close()


def test_grid_follows_runtime_occupancy(monkeypatch):
    """A request for more waves per pair than the device can hold at once is cut to what fits."""
    from bialign_amd.batch import make_batch
    monkeypatch.setenv("BIALIGN_TEAM", "x32")
    params = dict(synth.PROTEIN_PARAMS, max_shift=2)
    pairs = synth.protein_batch(400, 700, seed0=1200)
    b = make_batch(pairs[:400], params, score_only=True)
    b.run()
    t = b.timing()
    assert t["recovered_runs"] == 0
    # 400 pairs x 32 waves = 12800 one-wave workgroups would exceed any residency the s=2 kernel has
    assert t["waves_per_pair"] * 400 <= 256 * 8
    b.close()
