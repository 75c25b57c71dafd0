"""Fill time of one workload over several ALLOCATIONS of the layer buffer (one process).  The
fill time repeats to +-0.1 ms inside an allocation and jumps between levels from one allocation to
the next (profiles/r01e_placement), so kernel variants are compared level by level:
    AB_PAIRS=1024 AB_LEN=512 AB_S=1 AB_AFFINE=1 AB_LEAN=0 AB_CYCLES=6 python tools/ab_alloc.py
The engine's buffer cache is trimmed between cycles to force a new allocation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
from bialign_amd.engine import default_engine
E = lambda k, d: int(os.environ.get(k, d))
pairs = synth.protein_batch(E("AB_PAIRS", 1024), E("AB_LEN", 512))
params = dict(synth.PROTEIN_PARAMS, max_shift=E("AB_S", 1))
if not E("AB_AFFINE", 1):
    params.update(gap_opening_cost=0, gap_cost=-200, shift_cost=-250)
best = []
for cycle in range(E("AB_CYCLES", 6)):
    b = make_batch(pairs, params, score_only=bool(E("AB_LEAN", 0)))
    ts = []
    for _ in range(E("AB_RUNS", 6)):
        b.run(); ts.append(b.timing()["fill_ms"])
    t = b.timing()
    print(f"cycle {cycle}: fill ms " + " ".join(f"{x:.2f}" for x in ts) +
          f"   team {t['waves_per_pair']}{'x' if t['cross_cu'] else ''} chunks {b.info['nchunks']}", flush=True)
    best.append(min(ts[2:] or ts))
    b.close()
    default_engine().trim()
print(f"levels: " + " ".join(f"{x:.2f}" for x in sorted(best)))
