"""Where does a one-shot batch spend its wall time?  encode / create (upload + allocations) / run / results / destroy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import encode_pairs
from bialign_amd.engine import Batch, default_engine
pairs = synth.protein_batch(int(os.environ.get("AB_PAIRS", 1024)), int(os.environ.get("AB_LEN", 512)))
params = dict(synth.PROTEIN_PARAMS)
for rep in range(5):
    t0 = time.perf_counter(); model, ma, mb = encode_pairs(pairs, params)
    t1 = time.perf_counter()
    b = Batch(default_engine(), ma, mb, model.s1, model.s2, params["gap_opening_cost"], params["gap_cost"],
              params["shift_cost"], params["max_shift"])
    t2 = time.perf_counter(); b.run()
    t3 = time.perf_counter(); sc = b.scores(); tr = b.traces()
    t4 = time.perf_counter(); b.close()
    t5 = time.perf_counter()
    print(f"rep {rep}: encode {1e3*(t1-t0):.1f} ms  create {1e3*(t2-t1):.1f}  run {1e3*(t3-t2):.1f}  "
          f"results {1e3*(t4-t3):.1f}  destroy {1e3*(t5-t4):.1f}", flush=True)
