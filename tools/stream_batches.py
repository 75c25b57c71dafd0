"""A stream of batches that arrive as strings: encode -> create -> run -> results, back to back, versus
the same with the next batch prepared while the current one sweeps (Batch.run(wait=False)).
    AB_PAIRS=1024 AB_LEN=512 AB_BATCHES=12 python tools/stream_batches.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
E = lambda k, d: int(os.environ.get(k, d))
npairs, length, nb = E("AB_PAIRS", 1024), E("AB_LEN", 512), E("AB_BATCHES", 12)
params = dict(synth.PROTEIN_PARAMS)
streams = [synth.protein_batch(npairs, length, seed0=5000 + 7919 * k) for k in range(3)]  # three distinct inputs, reused
make_batch(streams[0], params).run()  # warm-up (code objects, first allocation)

def results(b):
    s = b.scores(); t = b.traces(); b.close(); return int(s.sum())

t0 = time.perf_counter(); chk1 = 0
for k in range(nb):
    b = make_batch(streams[k % 3], params); b.run(); chk1 += results(b)
t_sync = time.perf_counter() - t0

t0 = time.perf_counter(); chk2 = 0
cur = make_batch(streams[0], params); cur.run(wait=False)
for k in range(1, nb + 1):
    nxt = None
    if k < nb:
        nxt = make_batch(streams[k % 3], params)   # encoded and uploaded while `cur` sweeps
    chk2 += results(cur)                           # waits for cur
    if nxt is not None:
        nxt.run(wait=False)
    cur = nxt
t_pipe = time.perf_counter() - t0
assert chk1 == chk2
print(f"{nb} batches of {npairs} x {length}: back to back {1e3 * t_sync / nb:.1f} ms/batch, pipelined {1e3 * t_pipe / nb:.1f} ms/batch "
      f"({t_sync / t_pipe:.2f}x)")
