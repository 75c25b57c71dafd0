#!/usr/bin/env python3
"""Golden vectors for RNA alignments whose structure features are REAL numbers
(the reference's predicted-structure mode, bialignment.pyx:343-374, 415-423).

ViennaRNA (``import RNA``) is not installed, so the reference cannot predict
structures here.  What this script pins instead is everything downstream of the
features: a Python subclass of the compiled reference's ``BiAligner`` overrides
``_preprocess_seq`` (a plain ``def`` method the constructor looks up
dynamically) and replaces the 0/1 features of the fixed structure by seeded
fractional ones; the reference's own ``_structure_similarity`` (sqrt / float
sum / int truncation), fill and traceback then run unchanged.  Recorded: the
features, parameters, score, trace, completeness and (small cases) all layers.

    python tests/golden/make_golden_features.py     # dev container only
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from bialign_amd import synth  # noqa: E402


def fractional_features(seed, n):
    """(up, down, unp) per position, 1-based lists with an ignored entry 0; each triple is a
    probability split like ViennaRNA's (some entries exactly 0 or 1)."""
    rng = np.random.default_rng(seed)
    raw = rng.dirichlet([0.6, 0.6, 0.9], size=n + 1)
    kind = rng.integers(0, 6, size=n + 1)
    up, down = raw[:, 0].copy(), raw[:, 1].copy()
    up[kind == 0] = 0.0
    down[kind == 1] = 0.0
    both = kind == 2
    up[both], down[both] = 0.0, 0.0
    up = [float(v) for v in up]
    down = [float(v) for v in down]
    unp = [1.0 - u - d for u, d in zip(up, down)]
    return up, down, unp


def main():
    ba = mg.build_reference()
    feats = {}

    class FeatureAligner(ba.BiAligner):
        def _preprocess_seq(self, sequence, structure):
            mol = super()._preprocess_seq(sequence, structure)
            mol["up"], mol["down"], mol["unp"] = feats[str(sequence)]
            return mol

    out = []
    grid = [  # (seed, n, m, overrides, layers)
        (51, 10, 9, dict(max_shift=1), True),
        (52, 9, 12, dict(max_shift=2), True),
        (53, 11, 11, dict(max_shift=2, gap_opening_cost=0, gap_cost=-200, shift_cost=-250), True),
        (54, 8, 8, dict(max_shift=0, structure_weight=333), True),
        (55, 60, 50, dict(max_shift=1), False),
        (56, 45, 64, dict(max_shift=2, structure_weight=777), False),
        (57, 70, 70, dict(max_shift=2, gap_opening_cost=0, gap_cost=-200, shift_cost=-250), False),
        (58, 40, 33, dict(max_shift=3, structure_weight=1000), False),
    ]
    for seed, n, m, ov, layers in grid:
        sa, sb, ta, tb = synth.rna_pair(seed, n, m)
        if sa == sb:
            sb = sb[::-1]
        params = dict(synth.RNA_PARAMS, **ov, nameA="A", nameB="B")
        fa, fb = fractional_features(seed * 2, n), fractional_features(seed * 2 + 1, m)
        feats.clear()
        feats[sa], feats[sb] = fa, fb
        b = FeatureAligner(sa, sb, ta, tb, **params)
        rec = dict(name=f"rna_feat_s{seed}_{n}x{m}", seqA=sa, seqB=sb, strA=ta, strB=tb,
                   params=params, featuresA=dict(up=fa[0], down=fa[1], unp=fa[2]),
                   featuresB=dict(up=fb[0], down=fb[1], unp=fb[2]))
        rec["score"] = int(b.optimize())
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            trace = b.traceback()
        rec["trace"] = [[int(v) for v in col] for col in trace]
        rec["complete"] = "WARNING" not in buf.getvalue()
        s = params["max_shift"]
        rec["mu2"] = [[int(b.mu2(k, l)) for l in range(1, m + 1)] for k in range(1, n + 1)]
        if layers:
            rec["layers"] = mg.dump_layers(b, n, m, s, params["gap_opening_cost"] != 0)
        out.append(rec)
        print(rec["name"], rec["score"], len(rec["trace"]), rec["complete"])
    with open(os.path.join(HERE, "fractional_features.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))


if __name__ == "__main__":
    main()
