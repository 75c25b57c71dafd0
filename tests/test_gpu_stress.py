"""Shapes at the edges of the launch logic: very many tiny pairs, extreme aspect ratios, a
molecule of length 1 against a long one, every max_shift; sampled against the oracle."""
import numpy as np
import pytest

from bialign_amd import synth

pytestmark = pytest.mark.gpu


def check_sample(pairs, params, sample, scores, traces, ok):
    from oracle import oracle
    from bialign_amd.engine import trace_codes_to_columns
    for t in sample:
        ref = oracle.solve(*pairs[t], params)
        assert int(scores[t]) == ref["score"], (t, len(pairs[t][0]), len(pairs[t][1]))
        assert trace_codes_to_columns(traces[t]) == oracle.trace_to_lists(ref["trace"]), t
        assert bool(ok[t]) == ref["complete"]


@pytest.mark.parametrize("affine", [True, False])
def test_twenty_thousand_tiny_pairs(affine):
    from bialign_amd.batch import make_batch
    rng = np.random.default_rng(7)
    shapes = rng.integers(1, 48, size=(20000, 2))
    pairs = [synth.protein_pair(1000 + t, int(n), int(m)) for t, (n, m) in enumerate(shapes)]
    params = dict(synth.PROTEIN_PARAMS)
    if not affine:
        params.update(gap_opening_cost=0, gap_cost=-200, shift_cost=-250, max_shift=2)
    b = make_batch(pairs, params)
    b.run()
    scores = b.scores()
    traces, ok = b.traces()
    assert len(scores) == 20000
    check_sample(pairs, params, rng.choice(20000, size=150, replace=False), scores, traces, ok)
    b.close()


@pytest.mark.parametrize("n,m,s", [(1, 3000, 1), (3000, 1, 1), (2, 2500, 2), (2500, 2, 2), (1, 1, 5),
                                    (700, 3, 3), (3, 700, 4), (64, 64, 5), (63, 65, 0), (4000, 40, 1)])
def test_extreme_aspect_ratios(n, m, s):
    from bialign_amd.batch import make_batch
    pairs = [synth.protein_pair(n * 7 + m, n, m)]
    for params in (dict(synth.PROTEIN_PARAMS, max_shift=s),
                   dict(synth.PROTEIN_PARAMS, max_shift=s, gap_opening_cost=0, gap_cost=-200, shift_cost=-250)):
        b = make_batch(pairs, params)
        b.run()
        traces, ok = b.traces()
        check_sample(pairs, params, [0], b.scores(), traces, ok)
        b.close()


def test_mixed_giants_and_dwarfs_in_one_batch():
    """One launch whose pairs differ by three orders of magnitude in size (team shape is per launch)."""
    from bialign_amd.batch import make_batch
    shapes = [(700, 650), (1, 1), (3, 2000), (40, 40), (900, 5), (2, 2), (600, 620), (17, 900)]
    pairs = [synth.protein_pair(50 + t, n, m) for t, (n, m) in enumerate(shapes)]
    params = dict(synth.PROTEIN_PARAMS)
    b = make_batch(pairs, params)
    b.run()
    traces, ok = b.traces()
    check_sample(pairs, params, range(len(pairs)), b.scores(), traces, ok)
    b.close()
