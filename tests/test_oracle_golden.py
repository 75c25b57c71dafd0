"""Pins the CPU oracle (oracle/bialign_oracle.c) to vectors produced by the
compiled reference (tests/golden/make_golden.py): scores, traces, the
incomplete-traceback flag and -- for small cases -- every cell of every layer."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle

KNOWN = load_golden("known_answers.json")
SMALL = load_golden("small_layers.json")
MEDIUM = load_golden("medium_traces.json")
WIDE = load_golden("wide_band.json")  # max_shift 6..10 (the anti-diagonal path of the engine)


def check(rec):
    p = rec["params"]
    res = oracle.solve(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"], p)
    assert res["score"] == rec["score"]
    assert oracle.trace_to_lists(res["trace"]) == rec["trace"]
    assert res["complete"] == rec["complete"]
    if "layers" in rec:
        n, m, s = len(rec["seqA"]), len(rec["seqB"]), p["max_shift"]
        got = oracle.band_values(res["layers"], n, m, s)
        assert len(got) == len(rec["layers"])
        for g, e in zip(got, rec["layers"]):
            np.testing.assert_array_equal(g, np.array(e, dtype=np.int64))


@pytest.mark.parametrize("rec", KNOWN, ids=[r["name"] for r in KNOWN])
def test_known_answers(rec):
    check(rec)


def test_readme_scores():
    by = {r["name"]: r for r in KNOWN}
    assert by["readme_rna_toy"]["score"] == 6800      # reference README.md:96
    assert by["readme_protein"]["score"] == 48500     # reference README.md:134


@pytest.mark.parametrize("rec", SMALL, ids=[r["name"] for r in SMALL])
def test_small_full_layers(rec):
    check(rec)


@pytest.mark.parametrize("rec", MEDIUM, ids=[r["name"] for r in MEDIUM])
def test_medium_traces(rec):
    check(rec)


@pytest.mark.parametrize("rec", WIDE, ids=[r["name"] for r in WIDE])
def test_wide_band_cases(rec):
    check(rec)


def test_empty_sequence_rejected():
    # the reference raises IndexError on empty input (pyx:407); n,m >= 1 is a precondition
    mu = np.zeros((1, 3), dtype=np.int32)
    with pytest.raises(ValueError):
        oracle.affine_fill(0, 2, 1, -150, -50, -150, mu, mu)


# ---- real-valued RNA features (the reference's predicted-structure scoring) ------------------
FEATURES = load_golden("fractional_features.json")


def feature_tables(rec):
    """mu1 from the oracle's own table builder, mu2 = the table the reference returned."""
    n, m = len(rec["seqA"]), len(rec["seqB"])
    mu1, _ = oracle.mu_tables(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"], rec["params"])
    mu2 = np.zeros((n + 1, m + 1), dtype=np.int32)
    mu2[1:, 1:] = np.array(rec["mu2"], dtype=np.int32)
    return mu1, mu2


@pytest.mark.parametrize("rec", FEATURES, ids=[r["name"] for r in FEATURES])
def test_fractional_feature_cases(rec):
    n, m, p = len(rec["seqA"]), len(rec["seqB"]), rec["params"]
    mu1, mu2 = feature_tables(rec)
    res = oracle.solve_tables(n, m, p, mu1, mu2)
    assert res["score"] == rec["score"]
    assert oracle.trace_to_lists(res["trace"]) == rec["trace"]
    assert res["complete"] == rec["complete"]
    if "layers" in rec:
        got = oracle.band_values(res["layers"], n, m, p["max_shift"])
        for g, e in zip(got, rec["layers"]):
            np.testing.assert_array_equal(g, np.array(e, dtype=np.int64))
