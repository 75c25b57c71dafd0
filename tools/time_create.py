"""Where does a one-shot batch spend its wall time?  encode / create (upload + hipMalloc) / run / results / destroy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
pairs = synth.protein_batch(int(os.environ.get("AB_PAIRS", 1024)), int(os.environ.get("AB_LEN", 512)))
for rep in range(4):
    t0 = time.perf_counter(); b = make_batch(pairs, dict(synth.PROTEIN_PARAMS))
    t1 = time.perf_counter(); b.run()
    t2 = time.perf_counter(); sc = b.scores(); tr = b.traces()
    t3 = time.perf_counter(); b.close()
    t4 = time.perf_counter()
    print(f"rep {rep}: create {1e3*(t1-t0):.1f} ms  run {1e3*(t2-t1):.1f}  results {1e3*(t3-t2):.1f}  destroy {1e3*(t4-t3):.1f}   "
          f"(kernels {b.timing() if False else ''})", flush=True)
