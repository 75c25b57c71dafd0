"""Host-side score inputs: sequences/structures/parameters -> LOOKUP form.

The DP kernels see two integer score functions

    mu1(i,j) = S1[seqcode_A[i-1]][seqcode_B[j-1]]
    mu2(k,l) = S2[class_A[k-1]][class_B[l-1]]

This module produces the uint8 codes and the two int32 tables so that they
reproduce, value for value, what the reference computes per cell through
``BiAligner.mu1/mu2`` (reference bialignment.pyx:404-440) from its
pre-processing (pyx:340-392) and ``read_simmatrix`` (bialignment_nonpyx.py:33-58).
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# RNA structure classes for a fixed dot-bracket string.  The reference derives
# three 0/1 features per position (pyx:366-374): "down" (pairs to the right),
# "up" (pairs to the left, partner < i-1 because range(1, i-1) skips i-1) and
# "unp" = 1 - up - down, then scores int(sw*(sqrt(upA*upB)+sqrt(dnA*dnB)+
# sqrt(unpA*unpB))) (pyx:416-423) -- i.e. sw iff the classes agree.
RNA_UNP, RNA_DOWN, RNA_UP = 0, 1, 2


def load_matrix_table(name):
    """Unscaled substitution table -> (alphabet list, int matrix)."""
    if name == "BLOSUM62":
        path = os.path.join(_DATA, "BLOSUM62.tsv")
        with open(path) as fh:
            rows = [ln.rstrip("\n").split("\t") for ln in fh if ln.strip() and not ln.startswith("#")]
        keys = rows[0][1:]
        mat = np.array([[int(v) for v in r[1:]] for r in rows[1:]], dtype=np.int64)
        if [r[0] for r in rows[1:]] != keys:
            raise ValueError("BLOSUM62.tsv: row and column labels differ")
        return keys, mat
    # user file in the reference's format: a header line starting with "-"
    # followed by one row per key (nonpyx:33-58)
    keys, row_keys, vals = None, [], []
    with open(name, "r") as fh:
        for idx, line in enumerate(fh):
            if keys and idx > len(keys):
                break
            tok = line.split()
            if tok[0] == "-":
                keys = tok[1:]
            else:
                row_keys.append(tok[0])
                vals.append([int(v) for v in tok[1:1 + len(keys)]])
    if keys != row_keys:
        print("ERROR while reading simmatrix {filename}.")  # message kept verbatim (nonpyx:57)
    return row_keys, np.array(vals, dtype=np.int64)


def read_simmatrix(filename, scale=100):
    """Dict-of-dicts view with entries multiplied by ``scale`` (nonpyx:33-58)."""
    keys, mat = load_matrix_table(filename)
    return {a: {b: int(scale * mat[x, y]) for y, b in enumerate(keys)} for x, a in enumerate(keys)}


def rna_classes(structure):
    """uint8 class per position of a fixed dot-bracket string (pyx:366-392)."""
    n = len(structure)
    cls = np.full(n, RNA_UNP, dtype=np.uint8)
    stack = []
    for pos, ch in enumerate(structure):
        if ch == "(":
            stack.append(pos)
        elif ch == ")":
            partner = stack.pop()  # IndexError on unbalanced ")" like the reference (pyx:387)
            cls[partner] = RNA_DOWN
            # "up" needs the partner strictly left of i-1 in 1-based terms
            cls[pos] = RNA_UP if partner < pos - 1 else RNA_UNP
    return cls


def _alphabet(texts=(), raw=None):
    """Sorted distinct letters: of the molecules' joined latin-1 bytes when the caller has them (C-speed passes:
    the letters of the first 4 KiB are deleted from the rest until nothing is left), else of the strings."""
    if raw is not None:
        letters, rest = set(), raw
        while rest:
            new = set(rest[:4096])
            letters |= new
            rest = rest.translate(None, bytes(new))
        return [chr(c) for c in sorted(letters)]
    return sorted(set("".join(texts)))


class ScoreModel:
    """Alphabets + tables for one parameter set; encodes molecules to codes."""

    def __init__(self, params, sequences=(), structures=(), raw_sequences=None, raw_structures=None):
        self.is_rna = params["type"] == "RNA"
        sw = int(params["structure_weight"])
        if params.get("simmatrix"):
            keys, mat = load_matrix_table(params["simmatrix"])
            self.seq_keys = list(keys)
            self.s1 = (100 * mat).astype(np.int32)
        else:
            letters = _alphabet(sequences, raw_sequences)
            if not letters:
                letters = ["N"]
            self.seq_keys = letters
            k = len(letters)
            self.s1 = np.full((k, k), int(params["sequence_mismatch_similarity"]), dtype=np.int32)
            np.fill_diagonal(self.s1, int(params["sequence_match_similarity"]))
        if len(self.seq_keys) > 256:
            raise ValueError("sequence alphabet larger than 256 symbols")
        self.seq_index = {c: x for x, c in enumerate(self.seq_keys)}
        if self.is_rna:
            self.cls_keys = ["unp", "down", "up"]
            k2 = 3
        else:
            letters = _alphabet(structures, raw_structures) or ["C"]
            self.cls_keys = letters
            k2 = len(letters)
            if k2 > 256:
                raise ValueError("structure alphabet larger than 256 symbols")
            self.cls_index = {c: x for x, c in enumerate(letters)}
        self.s2 = np.zeros((k2, k2), dtype=np.int32)
        np.fill_diagonal(self.s2, sw)  # pyx:425-428 / 416-423 with 0/1 features

    def _lut(self, index):
        """256-entry letter -> code table (255 = not in the alphabet), or False for an exotic alphabet."""
        luts = self.__dict__.setdefault("_luts", {})
        lut = luts.get(id(index))
        if lut is None:
            lut = False  # exotic alphabet: per-letter path
            if len(index) < 255 and all(len(c) == 1 and ord(c) < 256 for c in index):
                lut = np.full(256, 255, dtype=np.uint8)
                for c, x in index.items():
                    lut[ord(c)] = x
            luts[id(index)] = lut
        return lut

    def encode_raw(self, raw, index):
        """The whole batch at once: the molecules' joined latin-1 ``bytes`` -> uint8 codes by ONE ``bytes.translate``
        (None if this alphabet has no byte table).  A letter outside the alphabet raises KeyError like the reference."""
        lut = self._lut(index)
        if lut is False:
            return None
        codes = raw.translate(lut.tobytes())
        bad = codes.find(b"\xff")
        if bad >= 0:
            raise KeyError(chr(raw[bad]))
        return np.frombuffer(codes, dtype=np.uint8)

    def _encode(self, text, index):
        """Letters -> uint8 codes through a 256-entry table (one numpy gather per molecule); the
        first letter outside the alphabet raises KeyError like the reference's dict look-up."""
        lut = self._lut(index)
        if lut is not False:
            try:
                raw = np.frombuffer(text.encode("latin-1"), dtype=np.uint8)
            except UnicodeEncodeError:
                raw = None
            if raw is not None:
                codes = lut[raw]
                if codes.size and codes.max() == 255:
                    raise KeyError(text[int(np.argmax(codes == 255))])
                return codes
        return np.fromiter((index[c] for c in text), dtype=np.uint8, count=len(text))

    def encode_sequence(self, seq):
        # an unknown residue raises KeyError as the reference does at its first mu1 look-up (pyx:407)
        return self._encode(seq, self.seq_index)

    def encode_structure(self, structure):
        if self.is_rna:
            return rna_classes(structure)
        return self._encode(structure, self.cls_index)

    def mu1(self, code_a, code_b):
        return int(self.s1[code_a, code_b])

    def mu2(self, cls_a, cls_b):
        return int(self.s2[cls_a, cls_b])


def dense_mu2_from_features(mol_a, mol_b, structure_weight):
    """int32 (n, m) table of the reference's RNA structure similarity for real-valued features
    (predicted structures): int(sw * (sqrt(upA upB) + sqrt(dnA dnB) + sqrt(unpA unpB))), entry
    [k-1, l-1] for k, l 1-based (pyx:416-423).  Same IEEE double operations in the same order as
    the reference's per-cell Python expression; ``int()`` truncation = astype toward zero."""
    terms = []
    for key in ("up", "down", "unp"):
        prod = np.multiply.outer(np.asarray(mol_a[key][1:], dtype=np.float64),
                                 np.asarray(mol_b[key][1:], dtype=np.float64))
        if (prod < 0).any():
            raise ValueError("math domain error")  # math.sqrt of a negative product (pyx:419-421)
        terms.append(np.sqrt(prod))
    total = (terms[0] + terms[1]) + terms[2]
    return np.trunc(structure_weight * total).astype(np.int32)
