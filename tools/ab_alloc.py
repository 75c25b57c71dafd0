"""Does the fill time depend on where the layer buffer landed?  One process, several
create/run/destroy cycles of the same batch; prints the fill time of each allocation."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
pairs = synth.protein_batch(int(os.environ.get("AB_PAIRS", 1024)), int(os.environ.get("AB_LEN", 512)))
keep = []
for cycle in range(int(os.environ.get("AB_CYCLES", 6))):
    b = make_batch(pairs, dict(synth.PROTEIN_PARAMS))
    ts = []
    for _ in range(6):
        b.run(); ts.append(b.timing()["fill_ms"])
    print(f"cycle {cycle}: fill ms " + " ".join(f"{t:.2f}" for t in ts), flush=True)
    if os.environ.get("AB_KEEP") and cycle < 2:
        keep.append(b)      # hold the first allocations so that the next ones land elsewhere
    else:
        b.close()
