import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bialign_amd import synth
from bialign_amd.batch import make_batch
rng = np.random.default_rng(51)
for s in (1, 2):
    for kind in ("random", "constant", "lookup-like"):
        shapes = [(150, 170), (170, 220)]
        pairs = [synth.rna_pair(4700 + t, n, m) for t, (n, m) in enumerate(shapes)]
        if kind == "random":
            tabs = [rng.integers(0, 1200, size=(n, m)).astype(np.int32) for n, m in shapes]
        elif kind == "constant":
            tabs = [np.full((n, m), 300, dtype=np.int32) for n, m in shapes]
        else:
            tabs = [(rng.integers(0, 3, size=(n, 1)) == rng.integers(0, 3, size=(1, m))).astype(np.int32) * 400 for n, m in shapes]
        params = dict(synth.RNA_PARAMS, max_shift=s)
        b = make_batch(pairs, params, mu2_dense=tabs)
        b.run()
        print(s, kind, b.timing(), b.info["storage"], flush=True)
        b.close()
