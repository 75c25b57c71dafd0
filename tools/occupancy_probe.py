"""Throughput of a sweep against waves per SIMD: batches of 1024..6144 pairs swept by ONE wave each (BIALIGN_TEAM=1), i.e.
1, 2, 3, 4, 6 waves per SIMD where registers and LDS allow.  AB_AFFINE=0 (default): the one-layer recurrence (few
registers: every occupancy fits); AB_SCORE_ONLY=1: no layer stores.  Prints fill ms and ms per 1024 pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BIALIGN_TEAM", "1")
from bialign_amd import synth
from bialign_amd.batch import make_batch
length = int(os.environ.get("AB_LEN", 512))
affine = os.environ.get("AB_AFFINE", "0") == "1"
params = dict(synth.PROTEIN_PARAMS, max_shift=int(os.environ.get("AB_S", 1)))
if not affine:
    params.update(gap_opening_cost=0, gap_cost=-200)
for mult in (1, 2, 3, 4, 6):
    pairs = synth.protein_batch(1024 * mult, length)
    b = make_batch(pairs, params, score_only=os.environ.get("AB_SCORE_ONLY") == "1")
    ts = []
    for _ in range(6):
        b.run(fill_only=True); ts.append(b.timing()["fill_ms"])
    t = b.timing()
    print(f"{1024*mult:5d} pairs ({mult} waves/SIMD if resident): fill {min(ts[2:]):7.2f} ms = {min(ts[2:])/mult:6.2f} ms per 1024 pairs"
          f"   waves/pair {t['waves_per_pair']} chunks {b.info['nchunks']}", flush=True)
    b.close()
