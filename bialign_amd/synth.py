"""Seeded synthetic molecule pairs for benchmarks, parity tests and golden fixtures.

The generators are the ones SURVEY.md section 8(d) specifies for the BASELINE
configs: one ``random.Random(seed)`` stream per pair, draw order seqA, seqB,
strA, strB.  They are deliberately pure Python + ``random`` so that the very
same inputs can be regenerated in the dev container (where the compiled
reference produces golden vectors) and on the GPU box.
"""
import random

PROTEIN_ALPHABET = "ARNDCQEGHILKMFPSTWYV"
PROTEIN_SS_ALPHABET = "HECT"
RNA_ALPHABET = "ACGU"

#: README protein parameters (reference README.md:118-122) = BASELINE configs 2, 3, 5
PROTEIN_PARAMS = dict(
    type="Protein", simmatrix="BLOSUM62", structure_weight=800,
    gap_opening_cost=-150, gap_cost=-50, shift_cost=-150, max_shift=1,
    sequence_match_similarity=100, sequence_mismatch_similarity=0,
)
#: README RNA toy parameters (reference README.md:82-87) = BASELINE configs 1, 4
RNA_PARAMS = dict(
    type="RNA", simmatrix=None, structure_weight=400,
    gap_opening_cost=-200, gap_cost=-50, shift_cost=-150, max_shift=1,
    sequence_match_similarity=100, sequence_mismatch_similarity=0,
)


def _draw(rng, alphabet, length):
    return "".join(rng.choice(alphabet) for _ in range(length))


def protein_pair(seed, n, m=None):
    """(seqA, seqB, strA, strB) with i.i.d. uniform residues / HECT letters."""
    m = n if m is None else m
    rng = random.Random(seed)
    seq_a = _draw(rng, PROTEIN_ALPHABET, n)
    seq_b = _draw(rng, PROTEIN_ALPHABET, m)
    str_a = _draw(rng, PROTEIN_SS_ALPHABET, n)
    str_b = _draw(rng, PROTEIN_SS_ALPHABET, m)
    return seq_a, seq_b, str_a, str_b


def dotbracket(rng, length):
    """Balanced dot-bracket string: a chain of hairpins with stems of 2-6 bp
    and loops of 3-7 nt separated by 0-3 unpaired bases (never ``()``)."""
    out = []
    left = length
    while left > 0:
        gap = min(left, rng.randint(0, 3))
        out.append("." * gap)
        left -= gap
        stem = rng.randint(2, 6)
        loop = rng.randint(3, 7)
        need = 2 * stem + loop
        if need > left:
            out.append("." * left)
            left = 0
            break
        out.append("(" * stem + "." * loop + ")" * stem)
        left -= need
    return "".join(out)


def rna_pair(seed, n, m=None):
    m = n if m is None else m
    rng = random.Random(seed)
    seq_a = _draw(rng, RNA_ALPHABET, n)
    seq_b = _draw(rng, RNA_ALPHABET, m)
    str_a = dotbracket(rng, n)
    str_b = dotbracket(rng, m)
    return seq_a, seq_b, str_a, str_b


def protein_batch(npairs, n, m=None, seed0=1000):
    """BASELINE configs 2 and 5: pair p uses seed ``seed0 + p``."""
    return [protein_pair(seed0 + p, n, m) for p in range(npairs)]


def rna_batch(npairs, n, m=None, seed0=2000):
    """BASELINE config 4: pair p uses seed ``seed0 + p``."""
    return [rna_pair(seed0 + p, n, m) for p in range(npairs)]


def cells_per_pair(n, m, s):
    """Exact number of in-band lattice points:
    ``K(n,s) * K(m,s)`` with ``K(n,s) = sum_i #{k in [0,n]: |k-i| <= s}``."""
    def K(x):
        return sum(min(x, i + s) - max(0, i - s) + 1 for i in range(x + 1))
    return K(n) * K(m)
