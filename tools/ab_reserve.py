"""Does Engine.reserve() land the layer buffer on the fast placement level?  Several cycles of
trim -> reserve(tries) -> create/run batch; prints probe rate, reserve wall time and fill time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
from bialign_amd.engine import default_engine
E = lambda k, d: int(os.environ.get(k, d))
pairs = synth.protein_batch(E("AB_PAIRS", 1024), E("AB_LEN", 512))
params = dict(synth.PROTEIN_PARAMS)
eng = default_engine()
b = make_batch(pairs, params); need = b.info["hbm_layer_bytes"] + 64; b.close()
for cycle in range(E("AB_CYCLES", 5)):
    eng.trim()
    t0 = time.perf_counter(); rate = eng.reserve(need, tries=E("AB_TRIES", 4)); t1 = time.perf_counter()
    b = make_batch(pairs, params)
    ts = []
    for _ in range(4):
        b.run(); ts.append(b.timing()["fill_ms"])
    b.close()
    print(f"cycle {cycle}: reserve {t1 - t0:.2f} s, probe {rate:.0f} GB/s, fill ms " + " ".join(f"{x:.2f}" for x in ts), flush=True)
