"""Would a double-buffered chunk pipeline pay for batches larger than HBM?  1024 pairs x len 1024 (348 GB of layers):
(a) one batch, two chunks back to back; (b) the same pairs as four quarter batches on two engines (two buffers, two
streams), enqueued alternately so that chunk k+1's sweep overlaps chunk k's tail and traceback."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
from bialign_amd.engine import Engine
pairs = synth.protein_batch(1024, 1024)
params = dict(synth.PROTEIN_PARAMS)
e1, e2 = Engine(0), Engine(0)
b = make_batch(pairs, params, engine=e1)
b.run()
t0 = time.perf_counter(); b.run(); ta = time.perf_counter() - t0
t = b.timing(); print(f"(a) one batch, {b.info['nchunks']} chunks: {1e3 * ta:.1f} ms  (fill {t['fill_ms']:.1f} + traceback {t['traceback_ms']:.1f})", flush=True)
b.close(); e1.trim()
q = [make_batch(pairs[k * 256:(k + 1) * 256], params, engine=(e1, e2)[k % 2]) for k in range(2)]
for x in q: x.run()
def sweep_all():
    # quarters 0..3 reuse the two resident batches (same sizes): A, B, A, B
    q[0].run(wait=False); q[1].run(wait=False)
    q[0].wait(); q[0].run(wait=False)
    q[1].wait(); q[1].run(wait=False)
    q[0].wait(); q[1].wait()
sweep_all()
t0 = time.perf_counter(); sweep_all(); tb = time.perf_counter() - t0
print(f"(b) four quarter batches, two buffers / streams: {1e3 * tb:.1f} ms", flush=True)
