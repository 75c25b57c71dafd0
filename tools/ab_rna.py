"""Few long pairs (BASELINE config 4 shape): fill time over allocations.  AB_PAIRS x AB_LEN RNA, s=2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
from bialign_amd.engine import default_engine
E = lambda k, d: int(os.environ.get(k, d))
pairs = synth.rna_batch(E("AB_PAIRS", 64), E("AB_LEN", 2000))
params = dict(synth.RNA_PARAMS, max_shift=E("AB_S", 2))
if E("AB_LINEAR", 0):  # the one-layer recurrence with the CLI's default costs
    params.update(gap_opening_cost=0, gap_cost=-200, shift_cost=-250)
tabs = None
if E("AB_DENSE", 0):  # dense mu2 (the predicted-structure form): one random int32 table per pair
    import numpy as np
    rng = np.random.default_rng(5)
    tabs = [rng.integers(0, 400, size=(len(p[0]), len(p[1]))).astype(np.int32) for p in pairs]
for cycle in range(E("AB_CYCLES", 3)):
    b = make_batch(pairs, params, score_only=bool(E("AB_LEAN", 0)), lean_trace=bool(E("AB_LEANTRACE", 0)), mu2_dense=tabs)
    ts = []
    for _ in range(3):
        b.run(); ts.append(b.timing()["fill_ms"])
    t = b.timing()
    b_scores = b.scores()
    print(f"cycle {cycle}: fill ms " + " ".join(f"{x:.2f}" for x in ts) +
          f"   team {t['waves_per_pair']}{'x' if t['cross_cu'] else ''} chunks {b.info['nchunks']}  "
          f"{b.info['cells'] * (36 if b.info['affine'] else 4) / min(ts) / 1e9:.2f} TB/s  {b.info['cells'] / min(ts) / 1e6:.1f} Gcells/s", flush=True)
    b.close()
    default_engine().trim()
    if cycle == 0:
        import hashlib
        print("   scores sha", hashlib.sha1(b_scores.tobytes()).hexdigest()[:12], f"traceback {t['traceback_ms']:.1f} ms", flush=True)
