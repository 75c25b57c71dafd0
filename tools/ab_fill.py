"""Interleaved A/B of fill-kernel builds inside ONE process-per-variant loop is impossible (one .so
per process), so: per variant one process, many steps, print min / median / max of the fill kernel."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
pairs = synth.protein_batch(int(os.environ.get("AB_PAIRS", 1024)), int(os.environ.get("AB_LEN", 512)))
b = make_batch(pairs, dict(synth.PROTEIN_PARAMS))
ts = []
for _ in range(int(os.environ.get("AB_STEPS", 12))):
    b.run(); ts.append(b.timing()["fill_ms"])
ts2 = ts[2:]
print(f"fill ms: min {min(ts2):.2f} med {statistics.median(ts2):.2f} max {max(ts2):.2f}   all: " + " ".join(f"{t:.1f}" for t in ts))
