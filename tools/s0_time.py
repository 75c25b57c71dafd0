"""The affine sweep at max_shift 0 (a lane row per lattice row, one point per lane and step) on the matrix's shapes:
fill time and Gcells/s, full storage and score-only.  S0_S overrides the max_shift."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
s = int(os.environ.get("S0_S", 0))
for npairs, length in ((8192, 128), (8192, 256), (4238, 512), (1059, 1024), (264, 2048)):
    pairs = synth.protein_batch(npairs, length)
    for so in (False, True):
        b = make_batch(pairs, dict(synth.PROTEIN_PARAMS, max_shift=s), score_only=so)
        ts = []
        for _ in range(4):
            b.run(fill_only=True); ts.append(b.timing()["fill_ms"])
        t = b.timing()
        print(f"{npairs:5d} x {length:4d} s={s} {'score-only' if so else 'full      '}: fill {min(ts):7.2f} ms  {b.info['cells']/min(ts)/1e6:6.1f} Gcells/s"
              f"  team {t['waves_per_pair']}{'x' if t['cross_cu'] else ''}", flush=True)
        b.close()
