"""Throughput of the other BASELINE shapes / kernel variants (not the bench line)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch

CASES = [
    ("cfg5 shape: 256 protein x1024 s=1 affine", synth.protein_batch(256, 1024), dict(synth.PROTEIN_PARAMS)),
    ("cfg4 full: 256 RNA x2000 s=2 affine", synth.rna_batch(256, 2000), dict(synth.RNA_PARAMS, max_shift=2)),
    ("cfg5 one-GPU share: 1024 protein x1024 s=1", synth.protein_batch(1024, 1024), dict(synth.PROTEIN_PARAMS)),
    ("protein 512 x512 s=2 affine", synth.protein_batch(512, 512), dict(synth.PROTEIN_PARAMS, max_shift=2)),
    ("protein 512 x512 s=3 affine", synth.protein_batch(512, 512), dict(synth.PROTEIN_PARAMS, max_shift=3)),
    ("protein 1024 x512 s=0 affine", synth.protein_batch(1024, 512), dict(synth.PROTEIN_PARAMS, max_shift=0)),
    ("protein 1024 x512 s=2 non-affine (CLI defaults)", synth.protein_batch(1024, 512),
     dict(synth.PROTEIN_PARAMS, gap_opening_cost=0, gap_cost=-200, shift_cost=-250, max_shift=2)),
    ("protein 1024 x512 s=1 non-affine", synth.protein_batch(1024, 512),
     dict(synth.PROTEIN_PARAMS, gap_opening_cost=0, gap_cost=-200, shift_cost=-250, max_shift=1)),
]
EXTRA = [  # reduced-storage sweeps of the same engine
    ("cfg2 shape, score-only", synth.protein_batch(1024, 512), dict(synth.PROTEIN_PARAMS), dict(score_only=True)),
    ("cfg4 full, score-only", synth.rna_batch(256, 2000), dict(synth.RNA_PARAMS, max_shift=2), dict(score_only=True)),
    ("256 protein x512 s=1, lean traceback", synth.protein_batch(256, 512), dict(synth.PROTEIN_PARAMS), dict(lean_trace=True)),
]
if os.environ.get("PERF_ONLY"):  # e.g. PERF_ONLY="s=3,s=0,non-affine,score-only,lean" under rocprofv3 --pmc
    keys = os.environ["PERF_ONLY"].split(",")
    CASES = [c for c in CASES if any(k in c[0] for k in keys)]
    EXTRA = [c for c in EXTRA if any(k in c[0] for k in keys)]
for name, pairs, params, *kw in CASES + EXTRA:
    b = make_batch(pairs, params, **(kw[0] if kw else {}))
    b.run(); b.run()
    t = b.timing(); info = b.info
    nl = 36 if info["affine"] else 4
    print(f"{name:52s} cells {info['cells']/1e9:7.3f} G  chunks {info['nchunks']}  fill {t['fill_ms']:8.2f} ms  "
          f"tb {t['traceback_ms']:6.2f} ms  {info['cells']/ (t['fill_ms']+t['traceback_ms'])/1e6:7.1f} Gcells/s  "
          f"fill {info['cells']*nl/t['fill_ms']/1e9:6.2f} TB/s", flush=True)
    b.close()
