"""Batch front end: align many pairs in one GPU launch.

    python -m bialign_amd.batch_cli pairs.tsv --type Protein --simmatrix BLOSUM62 \\
        --gap_opening_cost -150 --gap_cost -50 --shift_cost -150 --structure_weight 800 --max_shift 1

``pairs.tsv``: one pair per line, tab separated ``nameA seqA strA nameB seqB strB`` (lines
starting with ``#`` are skipped).  All scoring / output options of ``bialign.py`` apply to every
pair.  For each pair the output is the reference CLI's block (``Input:`` echo, ``SCORE:``, the
decoded alignment in the chosen ``--outmode``) behind a ``>pair`` header line.  Under
``torchrun`` (one process per GPU) the pairs are sharded over the ranks; every rank prints its
own pairs, rank 0 additionally a ``#scores`` line with all gathered scores.
"""
import argparse
import sys

from . import bialignment
from .cli import _OPTIONS

_PER_PAIR = {"seqA", "seqB", "--strA", "--strB", "--nameA", "--nameB", "--fileinput", "--version"}


def read_pairs(path):
    pairs = []
    with open(path) as fh:
        for ln, line in enumerate(fh, start=1):
            line = line.rstrip("\n")
            if not line.strip() or line.startswith("#"):
                continue
            cols = line.split("\t")
            if len(cols) != 6:
                raise ValueError(f"{path}:{ln}: expected 6 tab-separated fields, got {len(cols)}")
            pairs.append(tuple(cols))
    if not pairs:
        raise ValueError(f"{path}: no pairs")
    return pairs


def build_parser():
    p = argparse.ArgumentParser(description="Batch bialignment on the GPU.")
    p.add_argument("pairs", help="TSV file: nameA seqA strA nameB seqB strB")
    for flags, kwargs in _OPTIONS:
        if flags[0] not in _PER_PAIR:
            p.add_argument(*flags, **kwargs)
    p.add_argument("--score_only", action="store_true",
                   help="Print one 'pair nameA nameB score' line per pair and skip the alignments "
                        "(the GPU then keeps a fraction of the DP layers: faster, far less memory).")
    return p


def pair_block(idx, rec, params, score, trace, complete, verbose):
    """Output lines of one pair, framed like the reference CLI (bialign.py:109-124)."""
    name_a, seq_a, str_a, name_b, seq_b, str_b = rec
    yield f">pair {idx}\t{name_a}\t{name_b}"
    yield "Input:"
    yield "seqA\t " + seq_a
    yield "seqB\t " + seq_b
    yield "strA\t " + str_a
    yield "strB\t " + str_b
    yield "SCORE: " + str(score)
    yield ""
    aligner = bialignment.BiAligner(seq_a, seq_b, str_a, str_b, **dict(params, nameA=name_a, nameB=name_b))
    if not complete:
        yield "WARNING: incomplete traceback. Alignment could be garbage."
    yield from aligner.decode_trace(trace)
    if verbose and aligner._affine:
        yield from aligner.eval_trace(trace)


def main(argv=None):
    """Engine refusals (e.g. --score_only with --max_shift above 5, scores outside the int32 window) end like the
    reference's own input errors (pyx:207-210): ``ERROR: <text>`` and exit status 255, not a Python traceback."""
    from ._lib import BialignError
    try:
        return _main(argv)
    except BialignError as e:
        print("ERROR: " + e.message)
        sys.exit(-1)


def _main(argv=None):
    args = build_parser().parse_args(argv)
    params = {k: v for k, v in vars(args).items() if k not in ("pairs", "verbose", "score_only")}
    records = read_pairs(args.pairs)
    from .batch import make_batch, pair_cost, shard
    from .distributed import gather_scores, init_from_env
    from .engine import default_engine, trace_codes_to_columns
    rank, local_rank, world = init_from_env()
    costs = [pair_cost((r[1], r[4]), params["max_shift"]) for r in records]  # shards balanced by lattice cells
    mine = shard(len(records), rank, world, costs)
    scores = []
    if len(mine):
        batch = make_batch([(r[1], r[4], r[2], r[5]) for r in (records[p] for p in mine)], params,
                           engine=default_engine(local_rank), score_only=args.score_only)
        batch.run()
        scores = batch.scores()
        if args.score_only:
            for t, p in enumerate(mine):
                print(f"pair {p}\t{records[p][0]}\t{records[p][3]}\t{int(scores[t])}")
        else:
            traces, complete = batch.traces()
            for t, p in enumerate(mine):
                trace = trace_codes_to_columns(traces[t], as_tuples=not batch.affine)
                for line in pair_block(p, records[p], params, int(scores[t]), trace, bool(complete[t]), args.verbose):
                    print(line)
        batch.close()
    if world > 1:
        allscores = gather_scores(scores, len(records), costs)
        if rank == 0:
            print("#scores\t" + "\t".join(str(int(s)) for s in allscores))
    return 0


if __name__ == "__main__":
    sys.exit(main())
