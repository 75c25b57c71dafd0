"""Throughput over (length, max_shift, recurrence, storage): Gcells/s for fill + traceback (or score only),
with the engine's own team choice; pairs sized to ~20-60 GB of layers.  Markdown table on stdout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch

def run(pairs, params, **kw):
    b = make_batch(pairs, params, **kw)
    b.run(); b.run(); b.run()
    t, info = b.timing(), dict(b.info)
    b.close()
    ms = t["fill_ms"] + t["traceback_ms"]
    return info["cells"] / ms / 1e6, t, info

print("| len | s | recurrence | pairs | team | chunks | full Gcells/s (fill ms + tb ms) | score-only Gcells/s |")
print("|---|---|---|---|---|---|---|---|")
for length in (128, 256, 512, 1024, 2048):
    for s in (0, 1, 2, 3):
        for affine in (True, False):
            W = 2 * s + 1
            per_pair = (length * W) ** 2 * (36 if affine else 4)
            npairs = max(8, min(8192, int(40e9 // per_pair)))
            pairs = synth.protein_batch(npairs, length)
            params = dict(synth.PROTEIN_PARAMS, max_shift=s)
            if not affine:
                params.update(gap_opening_cost=0, gap_cost=-200, shift_cost=-250)
            g, t, info = run(pairs, params)
            g2, t2, _ = run(pairs, params, score_only=True)
            print(f"| {length} | {s} | {'affine' if affine else 'one-layer'} | {npairs} | {t['waves_per_pair']}{'x' if t['cross_cu'] else ''} | "
                  f"{info['nchunks']} | {g:.0f} ({t['fill_ms']:.1f} + {t['traceback_ms']:.1f}) | {g2:.0f} |", flush=True)
