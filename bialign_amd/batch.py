"""Batch front end: many independent (A, B) pairs under one parameter set.

The reference aligns one pair per ``BiAligner`` (reference bialign.py:11); the
engine is batch-first (a single pair is a batch of one).  ``make_batch`` turns
molecule strings into the LOOKUP form the C ABI takes; ``shard`` splits a batch
over the ranks of a one-process-per-GPU job.
"""
import bisect

from .scoring import ScoreModel


def encode_pairs(pairs, params):
    """pairs: iterable of (seqA, seqB, strA, strB) -> (model, mols_a, mols_b)."""
    pairs = list(pairs)
    for sa, sb, ta, tb in pairs:
        if len(sa) != len(ta) or len(sb) != len(tb):
            raise ValueError("Provided structure and sequence must have the same length.")
    model = ScoreModel(params,
                       sequences=[p[0] for p in pairs] + [p[1] for p in pairs],
                       structures=[p[2] for p in pairs] + [p[3] for p in pairs])
    mols_a = [(model.encode_sequence(sa), model.encode_structure(ta)) for sa, _, ta, _ in pairs]
    mols_b = [(model.encode_sequence(sb), model.encode_structure(tb)) for _, sb, _, tb in pairs]
    return model, mols_a, mols_b


def make_batch(pairs, params, engine=None, hbm_budget_bytes=0, recurrence=0, mu2_dense=None,
               score_only=False, lean_trace=False):
    from .engine import Batch, default_engine  # loads the HIP library (no CPU fallback)
    model, mols_a, mols_b = encode_pairs(pairs, params)
    return Batch(engine or default_engine(), mols_a, mols_b, model.s1, model.s2,
                 params["gap_opening_cost"], params["gap_cost"], params["shift_cost"],
                 params["max_shift"], hbm_budget_bytes=hbm_budget_bytes, recurrence=recurrence,
                 mu2_dense=mu2_dense, score_only=score_only, lean_trace=lean_trace)


def shard(npairs, rank, world_size, costs=None):
    """Contiguous block of pair indices owned by ``rank`` (pairs are independent, so sharding
    needs no data-path collective).  Without ``costs`` the blocks hold equal numbers of pairs;
    with ``costs`` (one non-negative number per pair, e.g. lattice cells) block r ends where the
    running cost first reaches (r+1)/world of the total -- every rank computes the same cuts."""
    if costs is None:
        base, extra = divmod(npairs, world_size)
        start = rank * base + min(rank, extra)
        return range(start, start + base + (1 if rank < extra else 0))
    if len(costs) != npairs:
        raise ValueError("costs needs one entry per pair")
    return _cost_cuts(costs, world_size)[rank]


def _cost_cuts(costs, world_size):
    """world_size contiguous ranges; cut r sits where the running cost is nearest to
    (r+1)/world of the total (integer arithmetic, ties to the earlier cut)."""
    running = [0]
    for c in costs:
        if c < 0:
            raise ValueError("costs must be non-negative")
        running.append(running[-1] + int(c))
    total, n = running[-1], len(costs)
    cuts, start = [], 0
    for r in range(world_size):
        if r == world_size - 1:
            stop = n
        else:
            target = total * (r + 1)                      # compare running[x] * world with it
            stop = bisect.bisect_left(running, -(-target // world_size), lo=start)  # first reach
            stop = min(stop, n)
            if stop > start and 2 * target - (running[stop - 1] + running[stop]) * world_size <= 0:
                stop -= 1                                 # the cut before is at least as near
        cuts.append(range(start, stop))
        start = stop
    return cuts


def pair_cost(pair, max_shift):
    """Lattice cells of one (seqA, seqB, ...) pair: K(n,s) * K(m,s)."""
    s = int(max_shift)
    k = lambda x: (x + 1) * (2 * s + 1) - s * (s + 1)
    return k(len(pair[0])) * k(len(pair[1]))
