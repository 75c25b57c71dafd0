"""Wide-band path (max_shift > 5) timings: one 300x300 and one 1000x1000 RNA pair at s=6, 16 pairs x len 200 at s=8
(the shapes DESIGN.md section 3b quotes), full storage and score-only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
for name, pairs, s in (("1 x 300x300 s=6", [synth.rna_pair(1, 300, 300)], 6), ("1 x 1000x1000 s=6", [synth.rna_pair(2, 1000, 1000)], 6),
                       ("16 x len 200 s=8", synth.rna_batch(16, 200), 8)):
    for so in (False, True):
        b = make_batch(pairs, dict(synth.RNA_PARAMS, max_shift=s), score_only=so)
        ts = []
        for _ in range(3):
            b.run(); t = b.timing(); ts.append(t["fill_ms"])
        print(f"{name:20s} {'score-only' if so else 'full      '}: fill {min(ts):8.2f} ms  tb {t['traceback_ms']:6.2f} ms  cells {b.info['cells']/1e6:8.1f} M  "
              f"{b.info['cells']/min(ts)/1e6:6.2f} Gcells/s  waves/pair {t['waves_per_pair']}", flush=True)
        b.close()
# the one-layer recurrence (gap_opening_cost = 0, pyx:443-471) on the same path, full storage (it has no score-only form here)
for name, pairs, s in (("1 x 1000x1000 s=6 1-layer", [synth.rna_pair(2, 1000, 1000)], 6), ("16 x len 200 s=8 1-layer", synth.rna_batch(16, 200), 8)):
    b = make_batch(pairs, dict(synth.RNA_PARAMS, max_shift=s, gap_opening_cost=0))
    ts = []
    for _ in range(3):
        b.run(); t = b.timing(); ts.append(t["fill_ms"])
    print(f"{name:26s}: fill {min(ts):8.2f} ms  tb {t['traceback_ms']:6.2f} ms  cells {b.info['cells']/1e6:8.1f} M  {b.info['cells']/min(ts)/1e6:6.2f} Gcells/s  waves/pair {t['waves_per_pair']}", flush=True)
    b.close()
