"""Two engines (= two HIP streams) on one device, one resident batch each, runs enqueued alternately:
does batch B's sweep overlap batch A's traceback?  Compares with the same runs on one engine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
from bialign_amd.engine import Engine
E = lambda k, d: int(os.environ.get(k, d))
pairs = synth.protein_batch(E("AB_PAIRS", 1024), E("AB_LEN", 512))
params = dict(synth.PROTEIN_PARAMS)
e1, e2 = Engine(0), Engine(0)
for engines, name in (((e1, e1), "one stream "), ((e1, e2), "two streams")):
    a = make_batch(pairs, params, engine=engines[0]); b = make_batch(pairs, params, engine=engines[1])
    for x in (a, b): x.run(); x.run()
    n = E("AB_ROUNDS", 8)
    t0 = time.perf_counter()
    a.run(wait=False)
    for _ in range(n):
        b.run(wait=False); a.wait(); a.run(wait=False); b.wait()
    a.wait()
    dt = time.perf_counter() - t0
    print(f"{name}: {1e3 * dt / (2 * n + 1):.2f} ms per run   (kernels: fill {a.timing()['fill_ms']:.2f} + traceback {a.timing()['traceback_ms']:.2f})", flush=True)
    a.close(); b.close()
