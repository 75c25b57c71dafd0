"""One chunk of BASELINE config 4 (RNA pairs x len 2000, max_shift=2; the full 256-pair batch launches two chunks of 128
pairs with packed records, four of 64 with full ones: CFG4_PAIRS, BIALIGN_PACK=0) -- the launch the s=2 counters under
profiles/ are collected on.  CFG4_PAIRS / CFG4_LEN / CFG4_S / CFG4_RUNS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
E = lambda k, d: int(os.environ.get(k, d))
pairs = synth.rna_batch(E("CFG4_PAIRS", 128), E("CFG4_LEN", 2000))
b = make_batch(pairs, dict(synth.RNA_PARAMS, max_shift=E("CFG4_S", 2)))
for _ in range(E("CFG4_RUNS", 3)):
    b.run()
    t, info = b.timing(), b.info
    print(f"cells {info['cells']/1e9:.3f} G chunks {info['nchunks']} team {t['waves_per_pair']}{'x' if t['cross_cu'] else ''} "
          f"fill {t['fill_ms']:.2f} ms tb {t['traceback_ms']:.2f} ms  fill {info['cells']*36/t['fill_ms']/1e9:.2f} TB/s", flush=True)
b.close()
