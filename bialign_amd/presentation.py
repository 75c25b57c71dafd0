"""Host-side presentation helpers of the BiAlign API: consensus lines, the
maximum-expected-accuracy (MEA) fold used for RNA consensus structures, and the
highlighting utilities.  O(columns) / O(L^2) string and matrix work on the
trace the GPU returns -- outside the accelerated path but
needed so that ``decode_trace`` output is byte-identical to the reference
(bialignment.pyx:835-990; checked against tests/golden/*.json).
"""
import ctypes
import os

import numpy as np

_HOST_LIB = None


def _host_lib():
    """libbialign_host.so (bialign_amd/csrc/bialign_host.c, built by bialign_amd.build)."""
    global _HOST_LIB
    if _HOST_LIB is None:
        path = os.environ.get("BIALIGN_HOST_LIB_OVERRIDE") or \
            os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbialign_host.so")  # override: sanitizer build (tests)
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: run python -m bialign_amd.build")
        lib = ctypes.CDLL(path)
        lib.bialign_host_mea.restype = ctypes.c_int
        lib.bialign_host_mea.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_double,
                                         ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_double)]
        _HOST_LIB = lib
    return _HOST_LIB


def consensus_sequence(alistrA, alistrB):
    """Column-wise: the (upper-cased) symbol where both rows agree, '.' elsewhere
    (pyx:901-908)."""
    up_a, up_b = alistrA.upper(), alistrB.upper()
    return "".join(p if p == q else "." for p, q in zip(up_a, up_b))


def highlight_sequence_identity(alistrA, alistrB):
    """Lower-case both rows, upper-case identical columns (pyx:890-898)."""
    low_a, low_b = alistrA.lower(), alistrB.lower()
    n = min(len(low_a), len(low_b))
    out_a = "".join(low_a[t].upper() if low_a[t] == low_b[t] else low_a[t] for t in range(n))
    out_b = "".join(low_a[t].upper() if low_a[t] == low_b[t] else low_b[t] for t in range(n))
    return [out_a, out_b]


def parse_dotbracket(dbstr):
    """Partner index per position (-1 = unpaired) of a ()-string (pyx:911-922)."""
    partner = [-1] * len(dbstr)
    open_pos = []
    for pos, sym in enumerate(dbstr):
        if sym == "(":
            open_pos.append(pos)
        elif sym == ")":
            mate = open_pos.pop()
            partner[pos], partner[mate] = mate, pos
    return partner


def _ungapped_index(alistr):
    """1-based position in the ungapped molecule per alignment column, 0 at gaps."""
    idx = np.zeros(len(alistr), dtype=np.int64)
    pos = 0
    for col, ch in enumerate(alistr):
        if ch != "-":
            pos += 1
            idx[col] = pos
    return idx


def consensus_sbpp(alistrA, sbppA, alistrB, sbppB):
    """Consensus pair 'probabilities' of two aligned molecules: for columns
    (c0, c1) sqrt(pA * pB) of the projected entries, 0 where either molecule has a
    gap in either column (pyx:926-950).  1-based, row/column 0 stay 0."""
    cols = len(alistrA)
    out = np.zeros((cols + 1, len(alistrB) + 1), dtype=float)
    ia, ib = _ungapped_index(alistrA), _ungapped_index(alistrB)
    sbppA, sbppB = np.asarray(sbppA, dtype=float), np.asarray(sbppB, dtype=float)
    pa = np.where((ia[:, None] > 0) & (ia[None, :] > 0), sbppA[np.ix_(ia, ia)], 0.0)
    pb = np.where((ib[:, None] > 0) & (ib[None, :] > 0), sbppB[np.ix_(ib, ib)], 0.0)
    out[1:, 1:] = np.sqrt(pa * pb)
    return out


def mea(sbpp, gamma=3, *, brackets="()"):
    """Maximum expected accuracy structure of a symmetric pair-probability matrix
    whose diagonal holds the unpaired probabilities (pyx:836-886); returns
    (structure string, accuracy).  Runs the recursion of ``mea_python`` in native
    code (same doubles, same order, same tie-breaks): at ~2000 alignment columns
    the Python loop costs tens of seconds per call, far more than the GPU DP."""
    mat = np.ascontiguousarray(sbpp, dtype=np.float64)
    n = len(mat) - 1
    if mat.ndim != 2 or mat.shape[0] != mat.shape[1] or n < 1:
        return mea_python(sbpp, gamma, brackets=brackets)
    marks = np.zeros(n, dtype=np.uint8)
    score = ctypes.c_double()
    rc = _host_lib().bialign_host_mea(mat.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n, float(gamma),
                                      marks.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(score))
    if rc:
        raise MemoryError("bialign_host_mea: allocation failed")
    table = np.array([ord("."), ord(brackets[0]), ord(brackets[1])], dtype=np.uint8)
    return (table[marks].tobytes().decode("ascii"), np.float64(score.value))


def mea_python(sbpp, gamma=3, *, brackets="()"):
    """Reference-order MEA recursion in Python (pyx:836-886); kept as the readable
    statement of what bialign_host_mea computes and as its test partner.

    Sparse Nussinov-style recursion: F[i][j] = best accuracy of i..j; per right
    end j a candidate list of (k, C) = "j pairs with / is closed from k".  Ties
    resolve exactly as in the reference (strict improvements only, candidates in
    order of discovery) because the choice is visible in the output string.
    """
    n = len(sbpp) - 1
    best = np.zeros((n + 2, n + 2), dtype=float)   # best[i, j], best[i, i-1] = 0
    split = np.zeros((n + 2, n + 2), dtype=np.int64)
    cands = [[] for _ in range(n + 1)]
    for i in range(n, 0, -1):
        cands[i].append((i, sbpp[i, i]))
        row = best[i]
        for j in range(i, n + 1):
            val, arg = row[j], split[i, j]
            for k, gain in cands[j]:
                alt = row[k - 1] + gain
                if val < alt:
                    val, arg = alt, k
            row[j], split[i, j] = val, arg
            if i + 3 >= j:
                continue
            closed = best[i + 1, j - 1] + 2 * gamma * sbpp[i, j]
            if closed > row[j]:
                cands[j].append((i, closed))
                row[j], split[i, j] = closed, i
    out = ["."] * (n + 1)
    todo = [(1, n)]
    while todo:
        i, j = todo.pop()
        k = split[i, j]
        if i + 3 >= j or k == 0:
            continue
        if k == j:            # j unpaired
            todo.append((i, j - 1))
            continue
        out[k], out[j] = brackets[0], brackets[1]
        if k != i:
            todo.append((i, k - 1))
        todo.append((k + 1, j - 1))
    return ("".join(out[1:]), best[1, n])


def highlight_structure_identity(alistrA, alistrB):
    """Mark base pairs present in both aligned ()-strings with [ ] (pyx:954-971)."""
    pa, pb = parse_dotbracket(alistrA), parse_dotbracket(alistrB)
    rows = [[], []]
    for col, (x, y) in enumerate(zip(alistrA.lower(), alistrB.lower())):
        if pa[col] >= 0 and pa[col] == pb[col]:
            x = y = "[" if pa[col] > col else "]"
        rows[0].append(x)
        rows[1].append(y)
    return ["".join(r) for r in rows]


def highlight_structure_similarity(alistrA, alistrB, *, sbppA, sbppB):
    """Mark the MEA consensus pairs with < > in both rows (pyx:975-990)."""
    partner = parse_dotbracket(mea(consensus_sbpp(alistrA, sbppA, alistrB, sbppB))[0])
    rows = [list(alistrA), list(alistrB)]
    for left, right in enumerate(partner):
        if right > left:
            for r in rows:
                r[left], r[right] = "<", ">"
    return ["".join(r) for r in rows]
