"""Input helpers of the BiAlign API that sit in front of the DP: similarity
matrices, CFSSP files, alignment line utilities (reference
bialignment_nonpyx.py:1-141).  Plotting (nonpyx:144-367) is outside the
accelerated path and not provided; ``plot_alignment`` says so when called.
"""
import sys

from .scoring import load_matrix_table, read_simmatrix  # noqa: F401  (re-exported)

__version__ = "0.3"  # API level of the reference this package is a drop-in for (nonpyx:3)


def _blosum62_text():
    """BLOSUM62 in the reference's text layout ('-' corner, 3-wide columns)."""
    keys, mat = load_matrix_table("BLOSUM62")
    head = "-" + "".join(f"{k:>3}" for k in keys)
    body = [f"{a}" + "".join(f"{int(v):>3}" for v in row) for a, row in zip(keys, mat)]
    return "\n".join([head] + body) + "\n"


blosum62 = _blosum62_text()


def read_molecule(content, type):
    """Sequence and structure from the text of a CFSSP (Chou-Fasman server) report:
    the third field of every 4-field 'Query' / 'Struc' line, concatenated
    (nonpyx:61-83)."""
    if type != "Protein":
        raise IOError(f"Cannot read files of type {type}")
    parts = {"Query": [], "Struc": []}
    for raw in content.split("\n"):
        fields = raw.split()
        if fields and fields[0] in parts:
            if len(fields) != 4:
                raise IOError("Cannot parse")
            parts[fields[0]].append(fields[2])
    seq, struc = "".join(parts["Query"]), "".join(parts["Struc"])
    if len(seq) != len(struc):
        raise IOError("Sequence and structure of unequal length.")
    if not seq:
        raise IOError("Input does not contain input sequence and structure.")
    return [seq, struc]


def read_molecule_from_file(filename, type):
    """read_molecule on a file; prints the reference's messages and exits -1 on
    failure (nonpyx:86-98; the reference forgets to import sys there)."""
    try:
        with open(filename, "r") as fh:
            text = fh.read()
        return read_molecule(text, type)
    except FileNotFoundError as e:
        print("Input file not found.")
        print(e)
        sys.exit(-1)
    except IOError as e:
        print(f"Cannot read input file {filename}.")
        print(e)
        sys.exit(-1)


def breaklines(alilines, width):
    """Cut (name, string) alignment lines into blocks of ``width`` columns."""
    total = len(alilines[0][1])
    return [[(name, text[start:start + width]) for name, text in alilines]
            for start in range(0, total, width)]


def runs(s):
    """Maximal runs of equal characters as (char, start, end) triples."""
    start = 0
    for pos in range(1, len(s) + 1):
        if pos == len(s) or s[pos] != s[start]:
            if pos > start:
                yield (s[start], start, pos)
            start = pos


helix_yadd_a = []
helix_yadd_b = []


def fourway_from_full(alilines):
    """The six 'default' lines out of decode_trace_full()'s fourteen."""
    return [alilines[t] for t in (1, 3, 6, 8, 12, 13)]


def plot_alignment(alilines, width, **kwargs):
    raise NotImplementedError(
        "plot_alignment (matplotlib rendering, reference bialignment_nonpyx.py:144-367) is outside "
        "the accelerated DP path and is not part of this package; feed decode_trace_full() output "
        "to the reference's plotting module instead.")
