"""Drop-in ``bialignment`` API over the MI355X engine.

Mirrors the Python-level surface of the reference's Cython module
(reference src/bialignment.pyx): ``BiAligner`` with
``optimize / traceback / decode_trace(_full) / eval_trace`` and the helper
functions, same names, argument meaning, printed messages and return shapes.
The DP fill and the traceback run on the GPU through libbialign_hip.so
(C ABI: include/bialign.h); everything in this file is host-side glue and
O(columns) string work.  There is no CPU implementation of the DP here.
"""
import itertools
import os
import sys
from math import sqrt

import numpy as np

from . import molecule_io
from .molecule_io import *  # noqa: F401,F403  (reference does `from bialignment_nonpyx import *`)
from .molecule_io import read_simmatrix
from .presentation import (consensus_sbpp, consensus_sequence, highlight_sequence_identity,  # noqa: F401
                           highlight_structure_identity, highlight_structure_similarity, mea,
                           parse_dotbracket)
from .scoring import ScoreModel

__version__ = molecule_io.__version__

NEG_INF = -1 << 30


class SparseMatrix4D:
    """Banded 4-D integer table: (i,j,k,l) with |k-i|, |l-j| <= max_shift, stored
    as [i][j][k-i+s][l-j+s] (pyx:12-41)."""

    def __init__(self, n, m, max_shift, data=None):
        self._n, self._m, self.max_shift = n, m, max_shift
        w = 2 * max_shift + 1
        self._M = np.zeros((n + 1, m + 1, w, w), dtype=int) if data is None else data

    def _slot(self, key):
        i, j, k, l = key
        return i, j, k - i + self.max_shift, l - j + self.max_shift

    def __getitem__(self, key):
        return self._M[self._slot(key)]

    def __setitem__(self, key, value):
        self._M[self._slot(key)] = value


class AffineDPMatrices:
    """Nine SparseMatrix4D layers keyed by the 4-bit gap state (pyx:56-81)."""

    def __init__(self, n, m, max_shift, data=None):
        self._states = [st for st in itertools.product(range(2), repeat=4)
                        if (st[0] or st[1]) and (st[2] or st[3])]
        self._Ms = [None] * 16
        for pos, st in enumerate(self._states):
            layer = None if data is None else data[pos]
            self._Ms[self._slot(st)] = SparseMatrix4D(n, m, max_shift, layer)

    @staticmethod
    def _slot(key):
        a, b, c, d = key
        return ((a * 2 + b) * 2 + c) * 2 + d

    def __getitem__(self, key):
        return self._Ms[self._slot(key)]

    @property
    def states(self):
        return self._states


def affine_score(source_state, x, mu1, mu2, beta, gamma, Delta):
    """Score of column ``x`` appended after a prefix in ``source_state`` (pyx:84-131)."""
    score = Delta * (abs(x[0] - x[2]) + abs(x[1] - x[3]))
    for lo, mu in ((0, mu1), (2, mu2)):
        col = (x[lo], x[lo + 1])
        if col == (1, 1):
            score += mu
        elif col != (0, 0):
            score += gamma
            if (source_state[lo], source_state[lo + 1]) != col:
                score += beta
    return score


def guard_case(o, x, max_shift):
    """Is x - o a lattice point of the band? (pyx:133-148)"""
    p = [x[t] - o[t] for t in range(4)]
    return min(p) >= 0 and abs(p[2] - p[0]) <= max_shift and abs(p[3] - p[1]) <= max_shift


def argmin(xs):
    """Index of the first minimum (pyx:151-152)."""
    xs = list(xs)
    return xs.index(min(xs))


_HALVES = ((1, 1), (1, 0), (0, 1))  # enumeration order of pyx:282
_LINEAR_OFFSETS = ((1, 1, 1, 1), (1, 0, 1, 0), (0, 1, 0, 1), (1, 1, 0, 0), (0, 0, 1, 1),
                   (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1),
                   (1, 0, 1, 1), (0, 1, 1, 1), (1, 1, 1, 0), (1, 1, 0, 1))  # pyx:233-248


def _sequential_row_sums(mat):
    """Row sums accumulated strictly left to right (ufunc.accumulate), i.e. the doubles Python's
    sum() produces on this interpreter; numpy's .sum() adds pairwise and rounds differently."""
    mat = np.asarray(mat, dtype=float)
    if mat.shape[1] == 0:
        return np.zeros(mat.shape[0])
    return np.add.accumulate(mat, axis=1)[:, -1]


def _binary_features(mol):
    """True when up/down/unp hold only 0.0 / 1.0, i.e. come from a fixed structure string."""
    return all(v in (0.0, 1.0) for key in ("up", "down", "unp") for v in mol[key][1:])


class BiAligner:
    """Bi-alignment of two molecules; DP on the GPU (pyx:155)."""

    nl = 14
    outmodes = {
        "default": [1, 3, 6, 8, 12, 13],
        "sorted": [0, 1, 5, 3, 2, 4, nl] + [7, 6, 10, 8, 9, 11, nl] + [12, 13],
        "sorted_sym": [0, 1, 3, 2, 5, 4, nl] + [6, 7, 9, 8, 11, 10, nl] + [12, 13],
        "sorted_terse": [1, 5, 3, 4, nl] + [6, 10, 8, 11, nl] + [12, 13],
        "raw": [1, 3, 7, 9],
        "raw_struct": list(range(4)) + list(range(6, 10)),
        "full": range(nl),
    }  # data table of pyx:168-177

    def __init__(self, seqA, seqB, strA, strB, **params):
        self._params = params
        # bppA / bppB: optional externally computed base-pair probabilities (see _preprocess_seq)
        bpp_a, bpp_b = params.get("bppA"), params.get("bppB")
        self.molA = self._preprocess_seq(seqA, strA) if bpp_a is None else self._preprocess_seq(seqA, strA, bpp_a)
        self.molB = self._preprocess_seq(seqB, strB) if bpp_b is None else self._preprocess_seq(seqB, strB, bpp_b)
        self.gamma = self._params["gap_cost"]
        self.beta = self._params["gap_opening_cost"]
        self.max_shift = self._params["max_shift"]
        self._simmatrix = read_simmatrix(self._params["simmatrix"]) if self._params["simmatrix"] else None
        self._M = None          # DP layers (fetched lazily from HBM)
        self._batch = None      # engine batch holding this pair
        self._score = None
        self._trace_codes = None
        self._trace_complete = True
        self._ran_affine = None
        self.states = None

    # ------------------------------------------------------------------ basics
    @property
    def _is_rna(self):
        return self._params["type"] == "RNA"

    @property
    def _affine(self):
        return self.beta != 0

    @staticmethod
    def error(text):
        print("ERROR:", text)
        sys.exit(-1)

    # ------------------------------------------------------- input preparation
    @staticmethod
    def _symmetrize_bpps(bpp):
        """Upper-triangular pair probabilities -> symmetric matrix with unpaired
        probabilities on the diagonal; 1-based (pyx:326-338)."""
        n = len(bpp) - 1
        upper = np.triu(np.asarray(bpp, dtype=float)[: n + 1, : n + 1], k=1)
        upper[0, :] = 0.0
        sym = upper + upper.T
        # diagonal: 1.0 - (left-to-right sum of the row), the reference's Python sum() (pyx:335-336)
        sym[np.arange(1, n + 1), np.arange(1, n + 1)] = 1.0 - _sequential_row_sums(sym[:, 1:])[1:]
        return sym

    @staticmethod
    def _bp_matrix_from_fixed_structure(structure):
        """0/1 pair matrix of a dot-bracket string, unpaired positions on the
        diagonal; 1-based (pyx:378-392)."""
        n = len(structure)
        bpm = np.zeros((n + 1, n + 1), dtype="float")
        pending = []
        for pos, ch in enumerate(structure, start=1):
            if ch == "(":
                pending.append(pos)
            elif ch == ")":
                mate = pending.pop()
                bpm[pos, mate] = bpm[mate, pos] = 1.0
            else:
                bpm[pos, pos] = 1.0
        return bpm

    @staticmethod
    def _expected_pairing(mol):
        n, sbpp = mol["len"], mol["sbpp"]
        dist = np.arange(n + 1)
        return [0] + [float(np.sum(sbpp[i, 1:] * (dist[1:] - i))) for i in range(1, n + 1)]

    def _preprocess_seq(self, sequence, structure, bpp=None):
        """pyx:340-376.  ``bpp`` (extension): base-pair probabilities
        computed elsewhere, in the layout of ViennaRNA's ``fold_compound.bpp()`` ((n+1) x (n+1),
        1-based, upper triangle) -- the predicted-structure mode without the ViennaRNA dependency."""
        mol = {"seq": str(sequence)}
        mol["len"] = len(mol["seq"])
        if bpp is not None and self._is_rna:
            if len(bpp) != mol["len"] + 1:
                self.error("Provided base pair probabilities and sequence must have matching size.")
            mol["sbpp"] = BiAligner._symmetrize_bpps(bpp)
            mol["mea"] = mea(mol["sbpp"])
            if structure is None:
                structure = mol["mea"][0]
            elif len(structure) != len(sequence):
                self.error("Provided structure and sequence must have the same length.")
            mol["structure"] = structure
            mol["predicted"] = True
        elif structure is None:
            if not self._is_rna:
                self.error("Structures have to be provided when aligning proteins")
            import RNA  # ViennaRNA, exactly as the reference requires (pyx:347)
            fc = RNA.fold_compound(str(sequence))
            mol["mfe"] = fc.mfe()
            mol["pf"] = fc.pf()
            mol["sbpp"] = BiAligner._symmetrize_bpps(fc.bpp())
            mol["mea"] = mea(mol["sbpp"])
            mol["structure"] = mol["pf"][0]
            mol["predicted"] = True
        else:
            if len(structure) != len(sequence):
                self.error("Provided structure and sequence must have the same length.")
            mol["structure"] = structure
            if self._is_rna:
                mol["sbpp"] = BiAligner._bp_matrix_from_fixed_structure(structure)
        if self._is_rna:
            # features, 1-based with an ignored entry 0 (pyx:366-374): "up" sums the
            # partners j <= i-2, "down" the partners j > i
            # Sums run left to right like the reference's sum(): with real-valued probabilities
            # the rounding order reaches mu2 through int() (pyx:416-423).
            n, sbpp = mol["len"], np.asarray(mol["sbpp"], dtype=float)
            col = np.arange(n + 1)
            low = (col[None, :] >= 1) & (col[None, :] <= col[:, None] - 2)
            mol["up"] = [float(v) for v in _sequential_row_sums(np.where(low, sbpp, 0.0))]
            mol["down"] = [float(v) for v in _sequential_row_sums(np.where(col[None, :] > col[:, None], sbpp, 0.0))]
            mol["unp"] = [1.0 - u - d for u, d in zip(mol["up"], mol["down"])]
        return mol

    # ------------------------------------------------------------ score inputs
    def _sequence_similarity(self, i, j):
        a, b = self.molA["seq"][i - 1], self.molB["seq"][j - 1]
        if self._simmatrix:
            return self._simmatrix[a][b]
        key = "sequence_match_similarity" if a == b else "sequence_mismatch_similarity"
        return self._params[key]

    def _structure_similarity(self, i, j):
        sw = self._params["structure_weight"]
        if self._is_rna:
            A, B = self.molA, self.molB
            return int(sw * (sqrt(A["up"][i] * B["up"][j]) + sqrt(A["down"][i] * B["down"][j])
                             + sqrt(A["unp"][i] * B["unp"][j])))
        return sw if self.molA["structure"][i - 1] == self.molB["structure"][j - 1] else 0

    def mu1(self, i, j):
        return self._sequence_similarity(i, j)

    def mu2(self, i, j):
        return self._structure_similarity(i, j)

    # ------------------------------------------------------ recursion (host view)
    def recursion_cases(self, idx):
        """The 13 (offset, score) cases of the non-affine recurrence (pyx:225-252)."""
        i, j, k, l = idx
        m1, m2 = self.mu1(i, j), self.mu2(k, l)
        g, D = self.gamma, self._params["shift_cost"]
        scores = (m1 + m2, g + g, g + g, m1 + D, m2 + D, g + D, g + D, g + D, g + D,
                  g + m2 + D, g + m2 + D, g + m1 + D, g + m1 + D)
        yield from zip(_LINEAR_OFFSETS, scores)

    def affine_recursion_cases(self, state, idx):
        """(source_state, offset, score) triples of the affine recurrence: the
        full-offset group, then the structure-only and sequence-only groups
        (pyx:255-296)."""
        i, j, k, l = idx
        D, beta, gamma = self._params["shift_cost"], self.beta, self.gamma
        m1, m2 = self.mu1(i, j), self.mu2(k, l)
        state = list(state)
        st_list = self.states if self.states is not None else AffineDPMatrices(0, 0, 0).states
        if guard_case(state, idx, self.max_shift):
            for src in st_list:
                yield (list(src), state, affine_score(src, state, m1, m2, beta, gamma, D))
        for off, make in (([0, 0, state[2], state[3]], lambda h: [state[0], state[1], h[0], h[1]]),
                          ([state[0], state[1], 0, 0], lambda h: [h[0], h[1], state[2], state[3]])):
            if guard_case(off, idx, self.max_shift):
                for h in _HALVES:
                    src = make(h)
                    yield (src, off, affine_score(src, off, m1, m2, beta, gamma, D))

    def plus(self, xs):
        xs = list(xs)
        return max(xs) if xs else NEG_INF

    def eval_case(self, x, idx):
        i, j, k, l = idx
        io, jo, ko, lo = x[0]
        return self._layers()[i - io, j - jo, k - ko, l - lo] + x[1]

    # --------------------------------------------------------------- GPU calls
    def _run_engine(self, recurrence):
        from .engine import Batch, default_engine
        A, B = self.molA, self.molB
        model = ScoreModel(self._params, sequences=[A["seq"], B["seq"]],
                           structures=[A["structure"], B["structure"]])
        if self._batch is not None:
            self._batch.close()
        device = int(self._params.get("device", os.environ.get("BIALIGN_DEVICE", 0)) or 0)
        dense = None
        if self._is_rna and any(m.get("predicted") or not _binary_features(m) for m in (A, B)):
            # predicted structures: the features are real numbers, mu2 goes to the engine as a table
            from .scoring import dense_mu2_from_features
            dense = [dense_mu2_from_features(A, B, self._params["structure_weight"])]
            cls_a = np.zeros(A["len"], dtype=np.uint8)
            cls_b = np.zeros(B["len"], dtype=np.uint8)
        else:
            cls_a, cls_b = model.encode_structure(A["structure"]), model.encode_structure(B["structure"])
        self._batch = Batch(default_engine(device),
                            [(model.encode_sequence(A["seq"]), cls_a)],
                            [(model.encode_sequence(B["seq"]), cls_b)],
                            model.s1, model.s2, self.beta, self.gamma, self._params["shift_cost"],
                            self.max_shift, recurrence=recurrence, mu2_dense=dense)
        self._batch.run()
        self._ran_affine = self._batch.affine
        self._score = np.int64(self._batch.scores()[0])
        traces, ok = self._batch.traces()
        self._trace_codes, self._trace_complete = traces[0], bool(ok[0])
        self._M = None
        if self._ran_affine:
            self.states = AffineDPMatrices(0, 0, 0).states
        return self._score

    def _layers(self):
        """DP layers as the reference's container types, copied from HBM on demand."""
        if self._batch is None:
            raise TypeError("'NoneType' object is not subscriptable")  # optimize() not called
        if self._M is None:
            raw = self._batch.dump_layers(0).astype(int)
            n, m = self.molA["len"], self.molB["len"]
            if self._ran_affine:
                self._M = AffineDPMatrices(n, m, self.max_shift, raw)
            else:
                self._M = SparseMatrix4D(n, m, self.max_shift, raw[0])
        return self._M

    def optimize(self):
        """Fill the DP table(s); returns the optimal score (pyx:443-471)."""
        from ._lib import REC_AFFINE, REC_LINEAR
        return self._run_engine(REC_AFFINE if self._affine else REC_LINEAR)

    def affine_optimize(self):
        """pyx:474-509"""
        from ._lib import REC_AFFINE
        return self._run_engine(REC_AFFINE)

    def traceback(self):
        """List of trace arrows start->end (pyx:513-531); tuples here, lists for
        the affine recurrence."""
        if self._affine:
            return self.affine_traceback()
        if self._trace_codes is None or self._ran_affine:
            if self._batch is None:
                raise TypeError("'NoneType' object is not subscriptable")
            self._run_engine(2)
        from .engine import trace_codes_to_columns
        return trace_codes_to_columns(self._trace_codes, as_tuples=True)

    def affine_traceback(self):
        """pyx:535-586"""
        if self._trace_codes is None or not self._ran_affine:
            if self._batch is None:
                raise TypeError("'NoneType' object is not subscriptable")
            self._run_engine(1)
        from .engine import trace_codes_to_columns
        if not self._trace_complete:
            print("WARNING: incomplete traceback. Alignment could be garbage.")
        return trace_codes_to_columns(self._trace_codes)

    # ----------------------------------------------------------- trace -> text
    @staticmethod
    def _transfer_gaps(alistr, seqstr):
        """Project the gap pattern of an alignment row onto another string (pyx:589-599)."""
        src = iter(seqstr)
        return "".join("-" if c == "-" else next(src) for c in alistr)

    @staticmethod
    def _shift_string(ali, idx):
        """'.' where both copies of molecule ``idx`` agree on gap/residue, '>' where
        only the sequence half has the gap, '<' where only the structure half
        has it (pyx:601-621)."""
        first, second = ali[idx], ali[idx + 2]
        marks = {(True, True): ".", (False, False): ".", (True, False): ">", (False, True): "<"}
        return "".join(marks[(first[t] == "-", second[t] == "-")] for t in range(len(ali[0])))

    @staticmethod
    def auto_complete(x, xs):
        """First key (sorted) starting with ``x``, else ``x`` itself (pyx:623-630)."""
        return next((y for y in sorted(xs) if y.startswith(x)), x)

    def decode_trace_full(self, trace=None):
        """Fourteen (name, string) lines of the bi-alignment (pyx:633-707)."""
        if trace is None:
            trace = self.traceback()
        mols = (self.molA, self.molB, self.molA, self.molB)
        rows = []
        for r, mol in enumerate(mols):
            residues = iter(mol["seq"])
            rows.append("".join(next(residues) if y[r] == 1 else "-" for y in trace if y[r] in (0, 1)))
        ss_rows = [self._transfer_gaps(rows[r], mols[r]["structure"]) for r in range(4)]

        def consensus_ss(a, b):
            if self._is_rna:
                return mea(consensus_sbpp(alistrA=a, alistrB=b, sbppA=self.molA["sbpp"],
                                          sbppB=self.molB["sbpp"]), brackets="[]")[0]
            return consensus_sequence(a, b)

        cons_ss = [consensus_ss(ss_rows[0], ss_rows[1]), consensus_ss(ss_rows[2], ss_rows[3])]
        cons = [consensus_sequence(rows[0], rows[1]), consensus_sequence(rows[2], rows[3])]
        shifts = [self._shift_string(rows, 0), self._shift_string(rows, 1)]
        nameA, nameB = self._params["nameA"], self._params["nameB"]
        half_names = [nameA + " ss", nameA, nameB + " ss", nameB, "consensus ss", "consensus"]
        lines = []
        for h in range(2):
            lines += [ss_rows[2 * h], rows[2 * h], ss_rows[2 * h + 1], rows[2 * h + 1], cons_ss[h], cons[h]]
        names = half_names + half_names + [nameA + " shifts", nameB + " shifts"]
        return list(zip(names, lines + shifts))

    def decode_trace(self, trace=None):
        """Formatted text lines in the order of the selected output mode (pyx:709-743)."""
        full = self.decode_trace_full(trace)
        width = max(len(name) for name, _ in full) + 4
        if self._params.get("nodescription"):
            lines = [text for _, text in full]
        else:
            lines = [f"{name:{width}}{text}" for name, text in full]
        lines.append("")
        self._params.setdefault("outmode", "default")
        mode = self.auto_complete(self._params["outmode"], self.outmodes.keys())
        if mode in self.outmodes:
            order = self.outmodes[mode]
        else:
            print("WARNING: unknown output mode. Expect one of " + str(list(self.outmodes.keys())))
            order = self.outmodes["sorted"]
        return [lines[t] for t in order]

    # ------------------------------------------------------------- trace check
    def eval_affine_trace(self, trace=None):
        """Re-score a trace column by column (pyx:745-800); yields one line per column."""
        if trace is None:
            trace = self.traceback()
        beta, gamma, D = self.beta, self.gamma, self._params["shift_cost"]
        state, idx, total = [1, 1, 1, 1], [0, 0, 0, 0], 0
        for y in trace:
            idx = [p + q for p, q in zip(idx, y)]
            score = affine_score(state, y, self.mu1(idx[0], idx[1]), self.mu2(idx[2], idx[3]),
                                 beta, gamma, D)
            total += score
            # a half keeps its previous type while the column leaves it empty (pyx:752-760)
            state = [state[t] if (y[t - t % 2], y[t - t % 2 + 1]) == (0, 0) else y[t] for t in range(4)]
            yield " ".join(str(v) for v in (idx, y, score, "-->", total))

    def eval_trace(self, trace=None):
        """pyx:803-832"""
        if self._affine:
            yield from self.eval_affine_trace(trace)
            return
        if trace is None:
            trace = self.traceback()
        idx = [0, 0, 0, 0]
        for y in trace:
            idx = [p + q for p, q in zip(idx, y)]
            for off, score in self.recursion_cases(idx):
                if off == y:
                    yield " ".join(str(v) for v in (idx, y, score, "-->", self.eval_case((off, score), idx)))
                    break
