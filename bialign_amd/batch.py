"""Batch front end: many independent (A, B) pairs under one parameter set.

The reference aligns one pair per ``BiAligner`` (reference bialign.py:11); the
engine is batch-first (a single pair is a batch of one).  ``make_batch`` turns
molecule strings into the LOOKUP form the C ABI takes; ``shard`` splits a batch
over the ranks of a one-process-per-GPU job.
"""
import bisect

import numpy as np

from .scoring import ScoreModel


class FlatBatch:
    """A batch's molecules as the C ABI wants them (include/bialign.h, bialign_pairs): one uint8 code array per
    side and kind, pair p's molecule A at ``seq_a[off_a[p] : off_a[p] + len_a[p]]``."""
    __slots__ = ("len_a", "len_b", "off_a", "off_b", "seq_a", "cls_a", "seq_b", "cls_b")

    def molecules(self, side):
        """[(sequence codes, class codes)] per pair -- views, nothing is copied."""
        ln, off = (self.len_a, self.off_a) if side == "a" else (self.len_b, self.off_b)
        seq, cls = (self.seq_a, self.cls_a) if side == "a" else (self.seq_b, self.cls_b)
        return [(seq[o:o + n], cls[o:o + n]) for o, n in zip(off.tolist(), ln.tolist())]


def _offsets(lens):
    off = np.zeros(len(lens), dtype=np.int64)
    np.cumsum(lens[:-1], out=off[1:])
    return off


def encode_flat(pairs, params):
    """pairs: iterable of (seqA, seqB, strA, strB) -> (ScoreModel, FlatBatch).  The whole batch is encoded in one
    shot: all molecules of a kind joined into one ``bytes``, one table gather, offsets by cumsum -- no per-molecule
    numpy call (1024 pairs x len 1024: 45 ms -> 4 ms of host time in front of a 50 ms sweep)."""
    pairs = pairs if isinstance(pairs, list) else list(pairs)
    sa, sb, ta, tb = zip(*pairs) if pairs else ((), (), (), ())
    fb = FlatBatch()
    fb.len_a = np.fromiter(map(len, sa), dtype=np.int32, count=len(pairs))
    fb.len_b = np.fromiter(map(len, sb), dtype=np.int32, count=len(pairs))
    if not (np.array_equal(fb.len_a, np.fromiter(map(len, ta), dtype=np.int32, count=len(pairs))) and
            np.array_equal(fb.len_b, np.fromiter(map(len, tb), dtype=np.int32, count=len(pairs)))):
        raise ValueError("Provided structure and sequence must have the same length.")
    fb.off_a, fb.off_b = _offsets(fb.len_a), _offsets(fb.len_b)
    na = int(fb.len_a.sum())
    raw_seq = raw_str = None
    try:  # latin-1: one byte per letter, so byte offsets are letter offsets
        raw_seq = ("".join(sa) + "".join(sb)).encode("latin-1")
        raw_str = ("".join(ta) + "".join(tb)).encode("latin-1")
    except UnicodeEncodeError:
        pass
    if raw_seq is None:
        model = ScoreModel(params, sequences=sa + sb, structures=ta + tb)
        seq = cls = None
    else:
        model = ScoreModel(params, raw_sequences=raw_seq, raw_structures=raw_str)
        seq = model.encode_raw(raw_seq, model.seq_index)
        cls = None if model.is_rna else model.encode_raw(raw_str, model.cls_index)
    cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    if seq is None:  # letters beyond latin-1 or an alphabet without a byte table: molecule by molecule
        seq = cat([model.encode_sequence(x) for x in sa + sb])
    if cls is None:  # RNA: classes come from the bracket structure (a stack per molecule); or the slow alphabet path
        cls = cat([model.encode_structure(x) for x in ta + tb])
    fb.seq_a, fb.seq_b = seq[:na], seq[na:]   # (contiguous slices of the two joined arrays)
    fb.cls_a, fb.cls_b = cls[:na], cls[na:]
    return model, fb


def encode_pairs(pairs, params):
    """pairs: iterable of (seqA, seqB, strA, strB) -> (model, mols_a, mols_b), the molecules as
    (sequence codes, class codes) per pair (views into the flat arrays of ``encode_flat``)."""
    model, fb = encode_flat(pairs, params)
    return model, fb.molecules("a"), fb.molecules("b")


def make_batch(pairs, params, engine=None, hbm_budget_bytes=0, recurrence=0, mu2_dense=None,
               score_only=False, lean_trace=False):
    from .engine import Batch, default_engine  # loads the HIP library (no CPU fallback)
    model, fb = encode_flat(pairs, params)
    return Batch(engine or default_engine(), fb, None, model.s1, model.s2,
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
