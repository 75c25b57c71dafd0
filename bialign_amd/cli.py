"""Command line front end with the reference's options and output framing
(reference src/bialign.py): positional seqA seqB plus the flags below, printing
the input echo, ``SCORE:``, the decoded alignment and -- with ``-v`` -- the
per-column evaluation.  Argument abbreviations (``--structure``, ``--filein``)
work as in the reference because argparse prefix matching stays enabled."""
import argparse
import sys

from . import bialignment

VERSION_STRING = f"BiAlign {bialignment.__version__}"

# (flags, kwargs) in the reference's order (bialign.py:25-96)
_OPTIONS = [
    (("seqA",), dict(help="sequence A")),
    (("seqB",), dict(help="sequence B")),
    (("--strA",), dict(default=None, help="structure A")),
    (("--strB",), dict(default=None, help="structure B")),
    (("--nameA",), dict(default="A", help="name A")),
    (("--nameB",), dict(default="B", help="name B")),
    (("-v", "--verbose"), dict(action="store_true", help="Verbose")),
    (("--type",), dict(default="RNA", type=str, help="Type of molecule: RNA or Protein")),
    (("--nodescription",), dict(action="store_true",
                                help="Don't prefix the strings in output alignment with descriptions")),
    (("--outmode",), dict(default="default",
                          help="Output mode [call --outmode help for a list of options]")),
    (("--sequence_match_similarity",), dict(type=int, default=100, help="Similarity of matching nucleotides")),
    (("--sequence_mismatch_similarity",), dict(type=int, default=0,
                                               help="Similarity of mismatching nucleotides")),
    (("--structure_weight",), dict(type=int, default=400, help="Weighting factor for structure similarity")),
    (("--gap_opening_cost",), dict(type=int, default=0,
                                   help="Similarity of opening a gap (turns on affine gap cost if not 0)")),
    (("--gap_cost",), dict(type=int, default=-200, help="Similarity of a single gap position")),
    (("--shift_cost",), dict(type=int, default=-250,
                             help="Similarity of shifting the two scores against each other")),
    (("--max_shift",), dict(type=int, default=2,
                            help="Maximal number of shifts away from the diagonal in either direction")),
    (("--fileinput",), dict(action="store_true", help="Read sequence and structure input from file")),
    (("--version",), dict(action="version", version=VERSION_STRING)),
    (("--simmatrix",), dict(type=str, default=None, help="Similarity matrix")),
]


def add_bialign_parameters(parser):
    for flags, kwargs in _OPTIONS:
        parser.add_argument(*flags, **kwargs)


def bialign(seqA, seqB, strA, strB, verbose, **args):
    """Generator of output lines for one pair (reference bialign.py:10-22)."""
    aligner = bialignment.BiAligner(seqA, seqB, strA, strB, **args)
    yield "SCORE: " + str(aligner.optimize())
    yield ""
    yield from aligner.decode_trace()
    if verbose:
        yield from aligner.eval_trace()


def main(argv=None):
    parser = argparse.ArgumentParser(description="Bialignment.")
    add_bialign_parameters(parser)
    args = parser.parse_args(argv)
    if args.fileinput:
        args.seqA, args.strA = bialignment.read_molecule_from_file(args.seqA, args.type)
        args.seqB, args.strB = bialignment.read_molecule_from_file(args.seqB, args.type)
    echo = ["Input:", "seqA\t " + args.seqA, "seqB\t " + args.seqB]
    echo += [f"{key}\t " + val for key, val in (("strA", args.strA), ("strB", args.strB)) if val is not None]
    print("\n".join(echo))
    if args.outmode == "help":
        print()
        print("Available modes: " + ", ".join(bialignment.BiAligner.outmodes.keys()))
        print()
        sys.exit()
    from ._lib import BialignError
    try:
        for line in bialign(**vars(args)):
            print(line)
    except BialignError as e:  # an engine refusal (scores outside the int32 window, molecules beyond the LDS staging):
        print("ERROR: " + e.message)  # reported like the reference's own input errors (pyx:207-210), not as a traceback
        sys.exit(-1)


if __name__ == "__main__":
    main()
