"""Python face of the C ABI: Engine (device + stream) and Batch (pairs in HBM).

Thin by design -- argument marshalling only.  All DP work happens in
libbialign_hip.so on the GPU; nothing here computes alignments.
"""
import ctypes
import weakref

import numpy as np

from . import _lib
from ._lib import lib, check

STATES = [(0, 1, 0, 1), (0, 1, 1, 0), (0, 1, 1, 1), (1, 0, 0, 1), (1, 0, 1, 0),
          (1, 0, 1, 1), (1, 1, 0, 1), (1, 1, 1, 0), (1, 1, 1, 1)]  # pyx:61-65


def device_count():
    n = lib.bialign_device_count()
    if n < 0:
        check(n)
    return n


def _ptr(arr, ctype):
    return arr.ctypes.data_as(ctypes.POINTER(ctype))


class Engine:
    """One HIP device + stream.  Single-owner, not thread-safe."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        self._batches = weakref.WeakSet()  # live batches: closed before the engine goes away
        check(lib.bialign_engine_create(int(device), ctypes.byref(self._h)))
        self.device = int(device)

    def reserve(self, nbytes, tries=4):
        """Pre-allocate the layer buffer kept between batches and pick the best-placed of up to
        ``tries`` candidate allocations (bialign_engine_reserve); returns its probe rate in GB/s."""
        rate = ctypes.c_double()
        check(lib.bialign_engine_reserve(self._h, int(nbytes), int(tries), ctypes.byref(rate)))
        return rate.value

    def trim(self):
        """Release the layer buffer the engine keeps between batches (tens of GB after a large batch)."""
        check(lib.bialign_engine_trim(self._h))

    def close(self):
        if getattr(self, "_h", None):
            for batch in list(self._batches):
                batch.close()
            lib.bialign_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


_default_engines = {}


def default_engine(device=0):
    if device not in _default_engines:
        _default_engines[device] = Engine(device)
    return _default_engines[device]


class Batch:
    """A set of independent pairs sharing one parameter set, resident in HBM.

    ``mols_a`` / ``mols_b``: lists of ``(seq_codes uint8[n], class_codes uint8[n])`` -- or ``mols_a`` a
    ``batch.FlatBatch`` (the whole batch's code arrays, as ``batch.encode_flat`` makes them) and ``mols_b`` None.
    ``s1`` / ``s2``: int32 score tables (k1 x k1, k2 x k2).
    ``mu2_dense``: optional list of int32 arrays, pair p's of shape (n_p, m_p) with entry
    [k-1, l-1] = mu2(k, l); replaces the class codes / ``s2`` (DENSE form of include/bialign.h).
    ``score_only``: the batch will never be traced back (BIALIGN_BATCH_SCORE_ONLY): the sweep keeps
    only the rows the next strip needs; ``traces()`` / ``dump_layers()`` raise.
    ``lean_trace``: scores AND traces from that reduced storage (BIALIGN_BATCH_LEAN_TRACE): the
    traceback re-sweeps one strip at a time; for pairs whose full layers would not fit in HBM.
    """

    def __init__(self, engine, mols_a, mols_b, s1, s2, gap_opening_cost, gap_cost, shift_cost,
                 max_shift, hbm_budget_bytes=0, recurrence=0, mu2_dense=None, score_only=False,
                 lean_trace=False):
        if mols_b is None and hasattr(mols_a, "seq_a"):  # a batch.FlatBatch: the arrays are the ABI's already
            fb = mols_a
            if not len(fb.len_a):
                raise ValueError("need the same, non-zero number of A and B molecules")
            self.len_a, self.len_b, off_a, off_b = fb.len_a, fb.len_b, fb.off_a, fb.off_b
            seq_a, cls_a, seq_b, cls_b = fb.seq_a, fb.cls_a, fb.seq_b, fb.cls_b
        else:
            if len(mols_a) != len(mols_b) or not mols_a:
                raise ValueError("need the same, non-zero number of A and B molecules")
            self.len_a = np.array([len(x[0]) for x in mols_a], dtype=np.int32)
            self.len_b = np.array([len(x[0]) for x in mols_b], dtype=np.int32)
            off_a = np.zeros(len(mols_a), dtype=np.int64)
            off_b = np.zeros(len(mols_a), dtype=np.int64)
            off_a[1:] = np.cumsum(self.len_a[:-1])
            off_b[1:] = np.cumsum(self.len_b[:-1])
            seq_a = np.ascontiguousarray(np.concatenate([np.asarray(x[0], dtype=np.uint8) for x in mols_a]))
            cls_a = np.ascontiguousarray(np.concatenate([np.asarray(x[1], dtype=np.uint8) for x in mols_a]))
            seq_b = np.ascontiguousarray(np.concatenate([np.asarray(x[0], dtype=np.uint8) for x in mols_b]))
            cls_b = np.ascontiguousarray(np.concatenate([np.asarray(x[1], dtype=np.uint8) for x in mols_b]))
        self.engine = engine
        self.npairs = len(self.len_a)
        self.max_shift = int(max_shift)
        if len(seq_a) != len(cls_a) or len(seq_b) != len(cls_b):
            raise ValueError("sequence and structure codes must have equal length")
        s1 = np.ascontiguousarray(s1, dtype=np.int32)
        s2 = np.ascontiguousarray(s2, dtype=np.int32)
        if seq_a.size and (seq_a.max() >= s1.shape[0] or seq_b.max() >= s1.shape[0]):
            raise ValueError("sequence code outside the S1 table")
        if cls_a.size and (cls_a.max() >= s2.shape[0] or cls_b.max() >= s2.shape[0]):
            raise ValueError("structure class outside the S2 table")
        mu2_ptr, mu2_off_ptr = None, None
        if mu2_dense is not None:
            if len(mu2_dense) != self.npairs:
                raise ValueError("mu2_dense needs one table per pair")
            flat = []
            for p, tab in enumerate(mu2_dense):
                tab = np.ascontiguousarray(tab, dtype=np.int32)
                if tab.shape != (int(self.len_a[p]), int(self.len_b[p])):
                    raise ValueError(f"mu2_dense[{p}] must have shape (len A, len B)")
                flat.append(tab.ravel())
            mu2_off = np.zeros(self.npairs, dtype=np.int64)
            mu2_off[1:] = np.cumsum([f.size for f in flat[:-1]])
            mu2_flat = np.ascontiguousarray(np.concatenate(flat))
            mu2_ptr, mu2_off_ptr = _ptr(mu2_flat, ctypes.c_int32), _ptr(mu2_off, ctypes.c_int64)
        prm = _lib.Params(int(gap_opening_cost), int(gap_cost), int(shift_cost), int(max_shift),
                          int(recurrence), (_lib.BATCH_SCORE_ONLY if score_only else 0) |
                          (_lib.BATCH_LEAN_TRACE if lean_trace else 0))
        sc = _lib.Scoring(s1.shape[0], _ptr(s1, ctypes.c_int32), s2.shape[0], _ptr(s2, ctypes.c_int32))
        pr = _lib.Pairs(self.npairs, _ptr(self.len_a, ctypes.c_int32), _ptr(self.len_b, ctypes.c_int32),
                        _ptr(off_a, ctypes.c_int64), _ptr(off_b, ctypes.c_int64),
                        _ptr(seq_a, ctypes.c_uint8), _ptr(cls_a, ctypes.c_uint8),
                        _ptr(seq_b, ctypes.c_uint8), _ptr(cls_b, ctypes.c_uint8), mu2_ptr, mu2_off_ptr)
        self._h = ctypes.c_void_p()
        check(lib.bialign_batch_create(engine._h, ctypes.byref(prm), ctypes.byref(sc), ctypes.byref(pr),
                                       int(hbm_budget_bytes), ctypes.byref(self._h)))
        engine._batches.add(self)
        self.info = self.current_info()
        self.affine = bool(self.info["affine"])

    def current_info(self):
        """bialign_batch_get_info now (``info`` is the answer at creation; a fallback to full records may re-chunk)."""
        info = _lib.BatchInfo()
        check(lib.bialign_batch_get_info(self._h, ctypes.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_}

    def close(self):
        if getattr(self, "_h", None):
            lib.bialign_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def run(self, fill_only=False, wait=True):
        """Fill (+ traceback).  ``wait=False`` only enqueues the kernels (BIALIGN_RUN_ASYNC): the host
        may encode / create the next batch meanwhile; ``wait()`` or any result getter completes the run."""
        check(lib.bialign_batch_run(self._h, (_lib.RUN_FILL_ONLY if fill_only else 0) | (0 if wait else _lib.RUN_ASYNC)))

    def wait(self):
        check(lib.bialign_batch_wait(self._h))

    def timing(self):
        t = _lib.Timing()
        check(lib.bialign_batch_get_timing(self._h, ctypes.byref(t)))
        return dict(fill_ms=t.fill_ms, traceback_ms=t.traceback_ms,
                    fill_launches=t.fill_launches, traceback_launches=t.traceback_launches,
                    waves_per_pair=t.waves_per_pair, cross_cu=bool(t.cross_cu), recovered_runs=t.recovered_runs,
                    packed_records=bool(t.packed_records))

    def scores(self):
        out = np.empty(self.npairs, dtype=np.int32)
        check(lib.bialign_batch_get_scores(self._h, _ptr(out, ctypes.c_int32)))
        return out

    def traces(self):
        """-> (list of uint8 arrays of column codes start->end, complete flags)."""
        buf = np.empty(max(1, self.info["trace_bytes"]), dtype=np.uint8)
        off = np.empty(self.npairs, dtype=np.int64)
        ln = np.empty(self.npairs, dtype=np.int32)
        ok = np.empty(self.npairs, dtype=np.int32)
        check(lib.bialign_batch_get_traces(self._h, _ptr(buf, ctypes.c_uint8), _ptr(off, ctypes.c_int64),
                                           _ptr(ln, ctypes.c_int32), _ptr(ok, ctypes.c_int32)))
        return [buf[off[p]:off[p] + ln[p]].copy() for p in range(self.npairs)], ok.astype(bool)

    def dump_layers(self, pair):
        """Layers of one pair in the reference layout [layer][i][j][k-i+s][l-j+s]."""
        n, m, w = int(self.len_a[pair]), int(self.len_b[pair]), 2 * self.max_shift + 1
        nl = 9 if self.affine else 1
        out = np.empty((nl, n + 1, m + 1, w, w), dtype=np.int32)
        check(lib.bialign_batch_dump_layers(self._h, int(pair), _ptr(out, ctypes.c_int32)))
        return out


def trace_codes_to_columns(codes, as_tuples=False):
    """Column codes -> the reference's trace entries (lists for the affine
    recurrence, pyx:568; tuples for the non-affine one, pyx:526)."""
    cols = [[(c >> 3) & 1, (c >> 2) & 1, (c >> 1) & 1, c & 1] for c in codes.tolist()]
    return [tuple(c) for c in cols] if as_tuples else cols
