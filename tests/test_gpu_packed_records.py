"""Packed layer records (affine sweeps at max_shift 1, 2 and 3): interior steps store base + 16-bit
offsets instead of int32 values, the other steps full records; ghost feed, tracebacks and the dump
decode.  Lossless by construction -- and checked here against the oracle cell by cell, against the
golden vectors, with every team shape, and on inputs whose offsets do NOT fit (the sweep notices
and the run is repeated with full records)."""
import numpy as np
import pytest

from conftest import load_golden
from bialign_amd import synth

pytestmark = pytest.mark.gpu


def solve(pair, params, layers=True):
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    b = make_batch([pair], params)
    b.run()
    t = b.timing()
    traces, ok = b.traces()
    out = dict(score=int(b.scores()[0]), trace=trace_codes_to_columns(traces[0]), complete=bool(ok[0]), timing=t)
    if layers:
        out["layers"] = b.dump_layers(0)
    b.close()
    return out


def check(pair, params, expect_packed=True):
    from oracle import oracle
    n, m, s = len(pair[0]), len(pair[1]), params["max_shift"]
    ref = oracle.solve(*pair, params)
    got = solve(pair, params)
    assert got["timing"]["packed_records"] == expect_packed
    assert got["score"] == ref["score"]
    assert got["trace"] == oracle.trace_to_lists(ref["trace"])
    assert got["complete"] == ref["complete"]
    for g, e in zip(oracle.band_values(got["layers"], n, m, s), oracle.band_values(ref["layers"], n, m, s)):
        np.testing.assert_array_equal(g, e)
    return got


@pytest.mark.parametrize("n,m,s,seed", [(150, 160, 1, 1), (70, 300, 1, 2), (300, 170, 1, 3), (400, 400, 1, 10),
                                         (64, 140, 2, 6), (130, 131, 2, 7), (250, 120, 2, 9), (90, 150, 3, 11), (40, 200, 3, 12)])
def test_default_policy_full_layers(n, m, s, seed, monkeypatch):
    """Shapes the engine packs by itself (a fifth of the bytes saved; s=3 only in batches of 128 pairs and more, so
    forced here)."""
    if s == 3:
        monkeypatch.setenv("BIALIGN_PACK", "1")
    check(synth.protein_pair(4000 + seed, n, m), dict(synth.PROTEIN_PARAMS, max_shift=s))


@pytest.mark.parametrize("seed", range(6))
def test_odd_valued_scores_stay_packed(seed, monkeypatch):
    """Scores with all kinds of low bits (the OR of a lane's offsets may then be 0xffff without any offset being out of
    range): packed, no repeat, every cell equal to the oracle."""
    rng = np.random.default_rng(900 + seed)
    s = 1 + seed % 3
    if s == 3:
        monkeypatch.setenv("BIALIGN_PACK", "1")
    params = dict(synth.PROTEIN_PARAMS, max_shift=s, simmatrix=None, sequence_match_similarity=int(rng.integers(50, 999)),
                  sequence_mismatch_similarity=-int(rng.integers(1, 499)), structure_weight=int(rng.integers(1, 1111)),
                  gap_opening_cost=-int(rng.integers(1, 333)), gap_cost=-int(rng.integers(1, 277)),
                  shift_cost=-int(rng.integers(1, 311)))
    got = check(synth.protein_pair(4900 + seed, 170, 190), params)
    assert got["timing"]["recovered_runs"] == 0


@pytest.mark.parametrize("n,m,s", [(21, 120, 1), (300, 70, 1), (12, 100, 2)])
def test_default_policy_leaves_short_sweeps_alone(n, m, s):
    check(synth.protein_pair(4050 + n, n, m), dict(synth.PROTEIN_PARAMS, max_shift=s), expect_packed=False)


@pytest.mark.parametrize("n,m,s,seed", [(60, 50, 1, 1), (25, 47, 1, 2), (90, 46, 1, 3), (70, 40, 2, 4), (30, 36, 2, 5)])
def test_forced_on_short_pairs(n, m, s, seed, monkeypatch):
    """BIALIGN_PACK=1: packed wherever the layout allows at all -- a handful of interior steps per strip."""
    monkeypatch.setenv("BIALIGN_PACK", "1")
    check(synth.protein_pair(4100 + seed, n, m), dict(synth.PROTEIN_PARAMS, max_shift=s))


@pytest.mark.parametrize("team,n,m,s", [("2", 300, 320, 1), ("4", 170, 400, 1), ("8", 330, 650, 1), ("x3", 130, 300, 1),
                                        ("1", 90, 200, 1), ("3", 300, 320, 1), ("6", 330, 650, 1), ("12", 500, 1000, 1),  # (2, 3, 6, 12 at s=1: the three-waves-per-SIMD kernel)
                                        ("x8", 330, 650, 1), ("4", 100, 300, 2), ("8", 200, 470, 2), ("x7", 200, 400, 2),
                                        ("h2", 360, 810, 2), ("4", 80, 300, 3), ("x4", 90, 400, 3)])
def test_team_shapes(team, n, m, s, monkeypatch):
    monkeypatch.setenv("BIALIGN_TEAM", team)
    if s == 3:
        monkeypatch.setenv("BIALIGN_PACK", "1")
    got = check(synth.protein_pair(4200 + n, n, m), dict(synth.PROTEIN_PARAMS, max_shift=s))
    assert got["timing"]["waves_per_pair"] == (8 * int(team[1:]) if team[0] == "h" else int(team.lstrip("x")))


@pytest.mark.parametrize("team,n,m,s", [("x5", 1500, 300, 2), ("1", 700, 300, 2), ("4", 900, 300, 2), ("x2", 800, 300, 2),
                                        ("x4", 900, 330, 3), ("h2", 2000, 830, 2)])
def test_many_strips_over_a_short_period(team, n, m, s, monkeypatch):
    """Packed sweeps at max_shift 2 and 3 whose waves go round their team 10-30 times (many strips, a short column
    period): ghost rows unpacked from packed and from full records in every round, teams on several CUs, in one
    workgroup, and a single wave; every layer cell, trace and score."""
    monkeypatch.setenv("BIALIGN_TEAM", team)
    monkeypatch.setenv("BIALIGN_PACK", "1")
    got = check(synth.protein_pair(4300 + n, n, m), dict(synth.PROTEIN_PARAMS, max_shift=s))
    assert got["timing"]["waves_per_pair"] == (8 * int(team[1:]) if team[0] == "h" else int(team.lstrip("x")))


@pytest.mark.parametrize("team", ["1", "2"])
def test_two_wave_kernels_stay_covered(team, monkeypatch):
    """BIALIGN_SLIM=0: the s=1 packed sweep on fill_affine_kernel (two waves per SIMD, LDS exchange array), which the
    three-waves-per-SIMD kernel otherwise replaces for teams of 1, 2, 3, 6 and 12."""
    monkeypatch.setenv("BIALIGN_SLIM", "0")
    monkeypatch.setenv("BIALIGN_TEAM", team)
    check(synth.protein_pair(4250, 300, 330), dict(synth.PROTEIN_PARAMS))


@pytest.mark.parametrize("team,npairs", [("3", 7), ("2", 13), ("6", 3)])
def test_slim_workgroups_of_ragged_pairs(team, npairs, monkeypatch):
    """fill_affine_slim_kernel: a workgroup holds 12 / team pairs of different lengths, the last workgroup fewer than
    that (its surplus waves leave after the staging barrier): scores, traces and one pair's layers against the oracle."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    monkeypatch.setenv("BIALIGN_TEAM", team)
    big = 560 if team == "6" else 300
    pairs = [synth.protein_pair(4270 + t, big + 23 * t, big + 140 - 11 * t) for t in range(npairs)]
    params = dict(synth.PROTEIN_PARAMS)
    b = make_batch(pairs, params)
    b.run()
    assert b.timing()["packed_records"] and b.timing()["waves_per_pair"] == int(team)
    scores = b.scores()
    traces, ok = b.traces()
    for t in (0, npairs // 2, npairs - 1):
        ref = oracle.solve(*pairs[t], params)
        assert int(scores[t]) == ref["score"] and bool(ok[t]) == ref["complete"]
        assert trace_codes_to_columns(traces[t]) == oracle.trace_to_lists(ref["trace"])
    t = npairs - 1
    n, m = len(pairs[t][0]), len(pairs[t][1])
    for g, e in zip(oracle.band_values(b.dump_layers(t), n, m, 1), oracle.band_values(oracle.solve(*pairs[t], params)["layers"], n, m, 1)):
        np.testing.assert_array_equal(g, e)
    b.close()


def test_slim_and_two_wave_kernels_agree_on_a_batch(monkeypatch):
    """A ragged batch through both s=1 packed sweeps: identical scores, traces and layers."""
    from bialign_amd.batch import make_batch
    pairs = [synth.protein_pair(4260 + t, 150 + 37 * t, 400 - 23 * t) for t in range(7)]
    out = []
    for slim in ("1", "0"):
        monkeypatch.setenv("BIALIGN_SLIM", slim)
        b = make_batch(pairs, dict(synth.PROTEIN_PARAMS))
        b.run()
        assert b.timing()["packed_records"]
        traces, ok = b.traces()
        out.append((b.scores(), traces, ok, b.dump_layers(3), b.timing()["waves_per_pair"]))
        b.close()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][2], out[1][2])
    for x, y in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(out[0][3], out[1][3])


def test_rna_and_golden_cases():
    from test_gpu_parity import check_against_golden
    check(synth.rna_pair(4300, 140, 150), dict(synth.RNA_PARAMS, max_shift=2))
    check(synth.rna_pair(4301, 150, 140), dict(synth.RNA_PARAMS))
    for rec in load_golden("medium_traces.json"):
        check_against_golden(rec)   # whatever form the policy picks for them


@pytest.mark.parametrize("s", [1, 2, 3])
def test_dense_mu2_packed(s, monkeypatch):
    """DENSE mu2 (per-pair int32 tables, the predicted-structure RNA form) with packed records: full layers vs oracle."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    if s == 3:
        monkeypatch.setenv("BIALIGN_PACK", "1")
    rng = np.random.default_rng(50 + s)
    shapes = [(150, 170), (170, 220)]
    pairs = [synth.rna_pair(4700 + t, n, m) for t, (n, m) in enumerate(shapes)]
    tabs = [rng.integers(0, 1200, size=(n, m)).astype(np.int32) for n, m in shapes]
    params = dict(synth.RNA_PARAMS, max_shift=s)
    b = make_batch(pairs, params, mu2_dense=tabs)
    b.run()
    assert b.timing()["packed_records"] and b.timing()["recovered_runs"] == 0
    scores = b.scores()
    traces, ok = b.traces()
    for t, (pair, (n, m)) in enumerate(zip(pairs, shapes)):
        mu1, _ = oracle.mu_tables(*pair, params)
        mu2 = np.zeros((n + 1, m + 1), dtype=np.int32)
        mu2[1:, 1:] = tabs[t]
        ref = oracle.solve_tables(n, m, params, mu1, mu2)
        assert int(scores[t]) == ref["score"]
        assert trace_codes_to_columns(traces[t]) == oracle.trace_to_lists(ref["trace"])
        for g, e in zip(oracle.band_values(b.dump_layers(t), n, m, s), oracle.band_values(ref["layers"], n, m, s)):
            np.testing.assert_array_equal(g, e)
    b.close()


def test_off_switch_and_equality(monkeypatch):
    pair, params = synth.protein_pair(4400, 180, 200), dict(synth.PROTEIN_PARAMS)
    a = solve(pair, params)
    monkeypatch.setenv("BIALIGN_PACK", "0")
    b = solve(pair, params)
    assert a["timing"]["packed_records"] and not b["timing"]["packed_records"]
    assert a["score"] == b["score"] and a["trace"] == b["trace"]
    np.testing.assert_array_equal(a["layers"], b["layers"])


def test_not_packed_where_it_does_not_apply():
    for params in (dict(synth.PROTEIN_PARAMS, max_shift=4), dict(synth.PROTEIN_PARAMS, max_shift=0),
                   dict(synth.PROTEIN_PARAMS, gap_opening_cost=40),                                   # beta > 0
                   dict(synth.PROTEIN_PARAMS, gap_opening_cost=0, gap_cost=-200, shift_cost=-250)):   # one layer
        check(synth.protein_pair(4500, 150, 160), params, expect_packed=False)


@pytest.mark.parametrize("s", [1, 2, 3])
def test_offsets_that_do_not_fit_fall_back(s, monkeypatch):
    """Scores so spread out that neighbouring states differ by more than 16 bits hold: the sweep flags it,
    the run is repeated with full records (once), results equal the oracle, the batch stays unpacked."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    monkeypatch.setenv("BIALIGN_PACK", "1")
    params = dict(synth.PROTEIN_PARAMS, max_shift=s, simmatrix=None, sequence_match_similarity=5000,
                  sequence_mismatch_similarity=-5000, structure_weight=100, gap_opening_cost=-5000, gap_cost=-5000,
                  shift_cost=-5000)
    pairs = [synth.protein_pair(4600 + t, 130 + 5 * t, 150) for t in range(3)]
    b = make_batch(pairs, params)
    b.run()
    t = b.timing()
    scores = b.scores()
    traces, ok = b.traces()
    if t["recovered_runs"]:
        assert not t["packed_records"]
    for k, pair in enumerate(pairs):
        ref = oracle.solve(*pair, params)
        assert int(scores[k]) == ref["score"]
        assert trace_codes_to_columns(traces[k]) == oracle.trace_to_lists(ref["trace"])
        assert bool(ok[k]) == ref["complete"]
    n, m = len(pairs[1][0]), len(pairs[1][1])
    ref = oracle.solve(*pairs[1], params, want_trace=False)
    for g, e in zip(oracle.band_values(b.dump_layers(1), n, m, s), oracle.band_values(ref["layers"], n, m, s)):
        np.testing.assert_array_equal(g, e)
    first = t["recovered_runs"]
    b.run()
    assert b.timing()["recovered_runs"] == first   # no second repeat
    b.close()
    assert first >= 1, "these scores were meant to overflow the 16-bit offsets"


def test_fallback_replans_a_chunked_batch(monkeypatch):
    """Eight pairs in three packed chunks (tight budget), offsets overflow: the batch is cut into chunks again by the pairs'
    full-record sizes inside the buffer it holds, repeated once, and equals the oracle."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    monkeypatch.setenv("BIALIGN_PACK", "1")
    params = dict(synth.PROTEIN_PARAMS, simmatrix=None, sequence_match_similarity=5000, sequence_mismatch_similarity=-5000,
                  structure_weight=100, gap_opening_cost=-5000, gap_cost=-5000, shift_cost=-5000)
    pairs = [synth.protein_pair(4800 + t, 120 + 9 * t, 170 - 3 * t) for t in range(8)]
    probe = make_batch(pairs, params)
    one_chunk = probe.info["hbm_layer_bytes"]
    probe.close()
    b = make_batch(pairs, params, hbm_budget_bytes=int(one_chunk * 0.4))
    chunks_before = b.info["nchunks"]
    assert chunks_before >= 3
    b.run()
    t = b.timing()
    assert t["recovered_runs"] == 1 and not t["packed_records"]
    scores = b.scores()
    traces, ok = b.traces()
    b.close()
    for k, pair in enumerate(pairs):
        ref = oracle.solve(*pair, params)
        assert int(scores[k]) == ref["score"]
        assert trace_codes_to_columns(traces[k]) == oracle.trace_to_lists(ref["trace"])
        assert bool(ok[k]) == ref["complete"]


def test_dump_layers_first_fill_overflows_and_replans(monkeypatch):
    """The layer dump is where a batch's FIRST fill happens (no run() before it), the offsets overflow, and the
    re-plan by full-record sizes under a tight budget re-chunks and re-orders the batch: the dump must refill and
    read the pair at its NEW launch position and layer offset -- every cell of two pairs against the oracle --
    and a run() afterwards must equal the oracle too."""
    from oracle import oracle
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    monkeypatch.setenv("BIALIGN_PACK", "1")
    params = dict(synth.PROTEIN_PARAMS, simmatrix=None, sequence_match_similarity=5000, sequence_mismatch_similarity=-5000,
                  structure_weight=100, gap_opening_cost=-5000, gap_cost=-5000, shift_cost=-5000)
    lens = [(125, 150), (190, 140), (140, 185), (170, 170)]   # launch order (longest sweep first) differs from pair order
    pairs = [synth.protein_pair(4900 + t, n, m) for t, (n, m) in enumerate(lens)]
    probe = make_batch(pairs, params)
    one_chunk = probe.info["hbm_layer_bytes"]
    probe.close()
    b = make_batch(pairs, params, hbm_budget_bytes=int(one_chunk * 0.62))
    chunks_before = b.info["nchunks"]
    assert chunks_before >= 2
    refs = [oracle.solve(*pair, params) for pair in pairs]
    for k in (2, 0):     # before any run(): the first dump overflows, re-plans and repeats; the second finds the batch on full records
        n, m = lens[k]
        got = b.dump_layers(k)
        for g, e in zip(oracle.band_values(got, n, m, 1), oracle.band_values(refs[k]["layers"], n, m, 1)):
            np.testing.assert_array_equal(g, e)
    assert b.current_info()["nchunks"] >= 1     # (re-planned inside the layer buffer the batch holds: often the engine's cached one)
    b.run()
    t = b.timing()
    assert not t["packed_records"] and t["recovered_runs"] >= 1
    scores = b.scores()
    traces, ok = b.traces()
    for k in (3, 1):     # and after a run
        n, m = lens[k]
        for g, e in zip(oracle.band_values(b.dump_layers(k), n, m, 1), oracle.band_values(refs[k]["layers"], n, m, 1)):
            np.testing.assert_array_equal(g, e)
    b.close()
    for k, ref in enumerate(refs):
        assert int(scores[k]) == ref["score"]
        assert trace_codes_to_columns(traces[k]) == oracle.trace_to_lists(ref["trace"])
        assert bool(ok[k]) == ref["complete"]


def test_s3_policy_by_batch_size():
    from bialign_amd.batch import make_batch
    params = dict(synth.PROTEIN_PARAMS, max_shift=3)
    for npairs, expect in ((4, False), (130, True)):
        b = make_batch(synth.protein_batch(npairs, 150, seed0=5000), params)
        b.run()
        assert b.timing()["packed_records"] == expect
        b.close()


def test_batch_of_1024_len_512_properties():
    """BASELINE configs[1] at full size through the packed path: every trace re-scores to its score."""
    from test_gpu_dropin import _full_config_check
    info, _ = _full_config_check(synth.protein_batch(1024, 512), dict(synth.PROTEIN_PARAMS), 8, 0)  # every 8th trace re-scored
    assert info["npairs"] == 1024 and info["nchunks"] == 1 and info["cells"] == 1024 * 1537 * 1537
