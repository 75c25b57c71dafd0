"""The engine keeps the layer buffer of a finished batch for the next one (hipMalloc of tens of
GB costs far more than the sweep).  Results must not depend on whose buffer a batch runs in."""
import os

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from bialign_amd import synth

pytestmark = pytest.mark.gpu


def solve(engine, pairs, params):
    from bialign_amd.batch import make_batch
    b = make_batch(pairs, params, engine=engine)
    b.run()
    out = [int(v) for v in b.scores()], [t.tolist() for t in b.traces()[0]]
    return b, out


def test_buffer_reuse_across_batches_and_trim():
    from oracle import oracle
    from bialign_amd.engine import Engine
    eng = Engine(0)
    params = dict(synth.PROTEIN_PARAMS)
    big = [synth.protein_pair(500 + t, 150, 140) for t in range(6)]
    small = [synth.protein_pair(600 + t, 20 + t, 31) for t in range(3)]
    want_big = [oracle.solve(*p, params)["score"] for p in big]
    want_small = [oracle.solve(*p, params)["score"] for p in small]

    b1, r1 = solve(eng, big, params)
    assert r1[0] == want_big
    b1.close()                                  # buffer goes to the engine
    b2, r2 = solve(eng, small, params)          # runs inside the (larger, dirty) cached buffer
    assert r2[0] == want_small
    assert r2[1] == [oracle.solve(*p, params)["trace"].tolist() for p in small]
    b3, r3 = solve(eng, big, params)            # cache is taken: allocates its own
    assert r3 == r1
    b2.close()
    b3.close()
    b4, r4 = solve(eng, big, dict(params, max_shift=2))   # larger than anything cached: reallocates
    assert r4[0] == [oracle.solve(*p, dict(params, max_shift=2))["score"] for p in big]
    b4.close()
    eng.trim()
    b5, r5 = solve(eng, small, params)          # after trim: fresh allocation
    assert r5 == r2
    b5.close()
    eng.close()


def test_engine_close_closes_its_batches():
    from bialign_amd.engine import Engine
    eng = Engine(0)
    b, _ = solve(eng, [synth.protein_pair(1, 30, 30)], dict(synth.PROTEIN_PARAMS))
    eng.close()
    assert b._h is None
    b.close()  # idempotent


def test_reserve_picks_a_buffer_and_batches_use_it():
    from bialign_amd.engine import Engine
    eng = Engine(0)
    rate = eng.reserve(96 << 20, tries=3)
    assert rate > 100.0                      # GB/s of the kept candidate's probe
    pairs = [synth.protein_pair(70 + t, 120, 110) for t in range(4)]
    b, r = solve(eng, pairs, dict(synth.PROTEIN_PARAMS))
    b2, r2 = solve(eng, pairs, dict(synth.PROTEIN_PARAMS))   # second live batch: own allocation
    assert r == r2
    b.close(); b2.close()
    assert eng.reserve(16 << 20, tries=1) > 0   # smaller request: the cached buffer is the candidate
    eng.close()


def test_async_run_two_batches_in_flight():
    """run(wait=False) only enqueues; getters complete the run; a second batch can be created and
    enqueued meanwhile (its uploads use their own stream); both buffers return to the engine."""
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import Engine
    eng = Engine(0)
    params = dict(synth.PROTEIN_PARAMS)
    pa = [synth.protein_pair(2000 + t, 200, 190) for t in range(8)]
    pb = [synth.protein_pair(2100 + t, 150, 210) for t in range(8)]
    ref_a = solve(eng, pa, params); ref_a[0].close()
    ref_b = solve(eng, pb, params); ref_b[0].close()
    a = make_batch(pa, params, engine=eng); a.run(wait=False)
    b = make_batch(pb, params, engine=eng); b.run(wait=False)      # enqueued behind a
    a.wait(); a.wait()                                               # idempotent
    got_b = [int(v) for v in b.scores()], [t.tolist() for t in b.traces()[0]]   # implicit wait
    got_a = [int(v) for v in a.scores()], [t.tolist() for t in a.traces()[0]]
    assert got_a == ref_a[1] and got_b == ref_b[1]
    assert a.timing()["fill_ms"] > 0 and b.timing()["fill_launches"] == 1
    a.run(wait=False); a.run()                                       # a new run first completes the pending one
    assert [int(v) for v in a.scores()] == ref_a[1][0]
    a.close(); b.close()
    c, got_c = solve(eng, pa, params)                                # takes one of the two cached buffers
    assert got_c == ref_a[1]
    c.close(); eng.close()


def test_destroy_order_is_free():
    """C ABI: an engine destroyed before its batches lives on until the last batch is gone (garbage
    collectors finalise cycles in any order); and a script that leaves everything to interpreter exit ends cleanly."""
    import subprocess
    import sys
    from bialign_amd._lib import lib
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import Engine
    eng = Engine(0)
    b = make_batch([synth.protein_pair(3, 60, 50)], dict(synth.PROTEIN_PARAMS), engine=eng)
    b.run()
    want = int(b.scores()[0])
    h, eng._h = eng._h, None                  # bypass Engine.close(), which would close the batch first
    lib.bialign_engine_destroy(h)             # deferred: one live batch
    assert int(b.scores()[0]) == want         # the batch still works
    b.run()
    assert int(b.scores()[0]) == want
    b.close()                                 # ... and takes the engine with it
    script = "\n".join([
        "import sys; sys.path.insert(0, %r)" % REPO,
        "from bialign_amd import synth",
        "from bialign_amd.batch import make_batch",
        "from bialign_amd.engine import Engine",
        "e1, e2 = Engine(0), Engine(0)",
        "q = [make_batch([synth.protein_pair(k, 40, 40)], dict(synth.PROTEIN_PARAMS), engine=(e1, e2)[k % 2]) for k in range(4)]",
        "def f():",                      # a function: module globals become a reference cycle
        "    return [x.run() for x in q]",
        "f(); print('done')"])
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "done" in out.stdout, out.stderr[-500:]


def test_smaller_chunks_when_the_layer_buffer_cannot_be_allocated(monkeypatch):
    """If hipMalloc refuses the planned chunk (fragmentation, another tenant), the batch is re-planned into smaller
    chunks instead of failing -- same results."""
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import Engine
    pairs = [synth.protein_pair(2600 + t, 150 + 3 * t, 170) for t in range(8)]
    params = dict(synth.PROTEIN_PARAMS)
    eng = Engine(0)
    ref = make_batch(pairs, params, engine=eng)
    ref.run()
    want = (ref.scores().tolist(), [t.tolist() for t in ref.traces()[0]])
    assert ref.info["nchunks"] == 1
    ref.close()
    eng.trim()
    monkeypatch.setenv("BIALIGN_TEST_FAIL_ALLOC", "1")
    b = make_batch(pairs, params, engine=eng)
    assert b.info["nchunks"] >= 2
    b.run()
    assert (b.scores().tolist(), [t.tolist() for t in b.traces()[0]]) == want
    b.close()
    eng.close()
