#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the *compiled reference*.

Runs ONLY in the dev container (needs /root/reference, Cython and gcc); the
fixtures it writes are committed, the reference build is not: it lives under
/tmp and never enters this repository or the GPU box.

    python tests/golden/make_golden.py            # everything except DNA-Pol-I
    python tests/golden/make_golden.py --dnapol   # + the 7-minute config-3 run

What is recorded per case: inputs, parameters, optimal score, the trace
(start->end list of 0/1 4-tuples), whether the reference printed its
"incomplete traceback" warning, decode_trace() for every output mode,
eval_trace() lines and -- for small cases -- every in-band cell of every DP
layer in lexicographic (i,j,k,l) order.
"""
import argparse
import contextlib
import io
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
BUILD = "/tmp/bialign_ref_build"

sys.path.insert(0, REPO)
from bialign_amd import synth  # noqa: E402


def build_reference():
    """cythonize the reference .pyx where it lies; all outputs under /tmp."""
    os.makedirs(BUILD, exist_ok=True)
    if not any(f.startswith("bialignment.") and f.endswith(".so") for f in os.listdir(BUILD)):
        script = f"""
from setuptools import setup, Extension
from Cython.Build import cythonize
exts = cythonize([Extension("bialignment", ["{REF}/src/bialignment.pyx"])],
                 build_dir="{BUILD}/c", annotate=False,
                 compiler_directives={{"boundscheck": False, "language_level": 3}})
setup(name="bialign_ref", ext_modules=exts,
      script_args=["-q", "build_ext", "--build-lib", "{BUILD}", "--build-temp", "{BUILD}/tmp"])
"""
        subprocess.run([sys.executable, "-c", script], check=True, cwd=BUILD)
    sys.path.insert(0, BUILD)
    sys.path.insert(1, f"{REF}/src")  # bialignment_nonpyx, bialign (CLI)
    import bialignment
    return bialignment


STATES = [(0, 1, 0, 1), (0, 1, 1, 0), (0, 1, 1, 1), (1, 0, 0, 1), (1, 0, 1, 0),
          (1, 0, 1, 1), (1, 1, 0, 1), (1, 1, 1, 0), (1, 1, 1, 1)]


class _Cap:
    """``x + _Cap()`` returns x: lets eval_case hand back a whole layer."""
    def __radd__(self, other):
        return other


def band_cells(n, m, s):
    for i in range(n + 1):
        for j in range(m + 1):
            for k in range(max(0, i - s), min(n, i + s) + 1):
                for l in range(max(0, j - s), min(m, j + s) + 1):
                    yield (i, j, k, l)


def dump_layers(b, n, m, s, affine):
    if affine:
        layers = [b.eval_case(((0, 0, 0, 0), _Cap()), st) for st in STATES]
        return [[int(layer[c]) for c in band_cells(n, m, s)] for layer in layers]
    return [[int(b.eval_case(((0, 0, 0, 0), 0), c)) for c in band_cells(n, m, s)]]


def run_case(ba, name, seqA, seqB, strA, strB, params, layers=False, decode=True,
             modes=("default",)):
    params = dict(params)
    params.setdefault("nameA", "A")
    params.setdefault("nameB", "B")
    rec = dict(name=name, seqA=seqA, seqB=seqB, strA=strA, strB=strB, params=dict(params))
    b = ba.BiAligner(seqA, seqB, strA, strB, **params)
    score = b.optimize()
    rec["score"] = int(score)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        trace = b.traceback()
    rec["trace"] = [[int(v) for v in col] for col in trace]
    rec["complete"] = "WARNING" not in buf.getvalue()
    n, m, s = len(seqA), len(seqB), params["max_shift"]
    affine = params["gap_opening_cost"] != 0
    if layers:
        rec["layers"] = dump_layers(b, n, m, s, affine)
    if decode:
        rec["decode_full"] = [[nm, st] for nm, st in b.decode_trace_full(trace)]
        rec["decode"] = {}
        for mode in modes:
            p2 = dict(params, outmode=mode)
            b2 = ba.BiAligner(seqA, seqB, strA, strB, **p2)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                lines = b2.decode_trace(trace)
            rec["decode"][mode] = dict(lines=list(lines), stdout=buf.getvalue())
        p3 = dict(params, nodescription=True)
        b3 = ba.BiAligner(seqA, seqB, strA, strB, **p3)
        rec["decode_nodescription"] = list(b3.decode_trace(trace))
        rec["eval_trace"] = list(b.eval_trace(trace))
    return rec


ALL_MODES = ("default", "sorted", "sorted_sym", "sorted_terse", "raw", "raw_struct",
             "full", "so", "sorted_t", "nonsense")


def run_cli(args):
    env = dict(os.environ, PYTHONPATH=f"{BUILD}:{REF}/src")
    out = subprocess.run([sys.executable, f"{REF}/src/bialign.py"] + args, env=env,
                         capture_output=True, text=True, check=True)
    return out.stdout


def wide_band(ba):
    """max_shift beyond the tiled kernels (6, 7, 8, 10): full layer dumps on small inputs (also
    bands wider than the molecules), traces on medium ones, both recurrences, RNA and protein."""
    pp = synth.PROTEIN_PARAMS
    lin = dict(gap_opening_cost=0, gap_cost=-200, shift_cost=-250)
    out = []
    for seed, n, m, ov, layers in [
            (61, 9, 8, dict(max_shift=6), True), (62, 10, 12, dict(max_shift=8), True),
            (63, 3, 5, dict(max_shift=7), True), (64, 14, 6, dict(max_shift=6, gap_opening_cost=80), True),
            (65, 8, 9, dict(max_shift=6, **lin), True), (66, 12, 10, dict(max_shift=10, **lin), True),
            (67, 1, 1, dict(max_shift=8), True), (68, 26, 30, dict(max_shift=6), False),
            (69, 24, 20, dict(max_shift=8, shift_cost=-20), False), (70, 30, 28, dict(max_shift=7, **lin), False)]:
        sa, sb, ta, tb = synth.protein_pair(seed, n, m)
        out.append(run_case(ba, f"protein_s{seed}_{n}x{m}", sa, sb, ta, tb, dict(pp, **ov), layers=layers,
                            decode=False))
    for seed, n, m, ov, layers in [(71, 10, 9, dict(max_shift=6), True), (72, 22, 24, dict(max_shift=8), False)]:
        sa, sb, ta, tb = synth.rna_pair(seed, n, m)
        out.append(run_case(ba, f"rna_s{seed}_{n}x{m}", sa, sb, ta, tb, dict(synth.RNA_PARAMS, **ov),
                            layers=layers, decode=True, modes=("default", "sorted")))
    with open(os.path.join(HERE, "wide_band.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dnapol", action="store_true")
    ap.add_argument("--only-wide", action="store_true", help="only wide_band.json (max_shift 6..10)")
    opts = ap.parse_args()
    ba = build_reference()
    wide_band(ba)
    if opts.only_wide:
        return

    RNA_A, RNA_B = "GCGGGGGAUAUCCCCAUCG", "GGGGAUAUCCCCAUCG"
    RNA_SA, RNA_SB = "...(((.....))).....", ".(((.....)))...."
    PRO_A = "RAKLPLKEKKLTATANYHPGIRYIMTGYSAKYIYSSTYARFR"
    PRO_B = "KAKLPLKEKKLTRTANYHPGIRYIMTGYSAKRIYSSTYAYFR"
    PRO_SA = "CHHHHHHHHHHHHHCCCCTCEEEEEEECCTCEEEEEEEECCC"
    PRO_SB = "HHHHHHHHHHHHCCCCCCTCEEEEEEECCCCCEEEEEEEECC"
    rna_lin = dict(synth.RNA_PARAMS, gap_opening_cost=0, gap_cost=-200, shift_cost=-250,
                   max_shift=2)  # CLI defaults (reference bialign.py:49-86)

    known = []
    known.append(run_case(ba, "readme_rna_toy", RNA_A, RNA_B, RNA_SA, RNA_SB,
                          synth.RNA_PARAMS, layers=True, modes=ALL_MODES))
    known.append(run_case(ba, "readme_protein", PRO_A, PRO_B, PRO_SA, PRO_SB,
                          synth.PROTEIN_PARAMS, layers=False, modes=ALL_MODES))
    known.append(run_case(ba, "rna_toy_linear_defaults", RNA_A, RNA_B, RNA_SA, RNA_SB,
                          rna_lin, layers=True, modes=ALL_MODES))
    known.append(run_case(ba, "protein_2mer", "AR", "RA", "HC", "CH",
                          synth.PROTEIN_PARAMS, layers=True))
    known.append(run_case(ba, "protein_1mer", "A", "A", "H", "H",
                          synth.PROTEIN_PARAMS, layers=True))
    known.append(run_case(ba, "protein_1x3", "W", "AWK", "E", "HEC",
                          dict(synth.PROTEIN_PARAMS, max_shift=2), layers=True))
    with open(os.path.join(HERE, "known_answers.json"), "w") as fh:
        json.dump(known, fh, separators=(",", ":"))

    cli = {}
    cli["readme_rna_toy"] = dict(
        args=[RNA_A, RNA_B, "--strA", RNA_SA, "--strB", RNA_SB, "--structure", "400",
              "--gap_opening_cost", "-200", "--gap_cost", "-50", "--max_shift", "1",
              "--shift_cost", "-150"])
    cli["readme_protein_sorted_v"] = dict(
        args=[PRO_A, PRO_B, "--strA", PRO_SA, "--strB", PRO_SB, "--type", "Protein",
              "--shift_cost", "-150", "--structure_weight", "800", "--simmatrix", "BLOSUM62",
              "--gap_opening_cost", "-150", "--gap_cost", "-50", "--max_shift", "1",
              "--outmode", "sorted", "-v"])
    cli["rna_toy_defaults_v"] = dict(
        args=[RNA_A, RNA_B, "--strA", RNA_SA, "--strB", RNA_SB, "-v"])
    cli["outmode_help"] = dict(args=[RNA_A, RNA_B, "--strA", RNA_SA, "--strB", RNA_SB,
                                     "--outmode", "help"])
    cli["rna_nodescription_raw"] = dict(
        args=[RNA_A, RNA_B, "--strA", RNA_SA, "--strB", RNA_SB, "--nodescription",
              "--outmode", "raw_struct", "--nameA", "first", "--nameB", "second",
              "--max_shift", "1"])
    for rec in cli.values():
        rec["stdout"] = run_cli(rec["args"])
    with open(os.path.join(HERE, "cli_outputs.json"), "w") as fh:
        json.dump(cli, fh, indent=1)

    # Random small cases: full layer dumps, every max_shift, both recurrences,
    # ragged lengths, positive gap opening, RNA with brackets.
    small = []
    pp = synth.PROTEIN_PARAMS
    grid = [  # (seed, n, m, overrides)
        (11, 5, 5, dict(max_shift=0)), (12, 6, 4, dict(max_shift=1)),
        (13, 4, 7, dict(max_shift=2)), (14, 7, 7, dict(max_shift=3)),
        (15, 3, 9, dict(max_shift=4)), (16, 9, 2, dict(max_shift=1)),
        (17, 8, 8, dict(max_shift=1, gap_opening_cost=100)),
        (18, 6, 6, dict(max_shift=2, gap_opening_cost=-1, gap_cost=0)),
        (19, 6, 7, dict(max_shift=1, gap_opening_cost=0, gap_cost=-200, shift_cost=-250)),
        (20, 7, 5, dict(max_shift=2, gap_opening_cost=0, gap_cost=-50, shift_cost=-100)),
        (21, 5, 5, dict(max_shift=0, gap_opening_cost=0, gap_cost=-200, shift_cost=-250)),
        (22, 8, 8, dict(max_shift=3, gap_opening_cost=0, gap_cost=-100, shift_cost=-50)),
        (23, 10, 10, dict(max_shift=1, shift_cost=0)),
        (24, 10, 9, dict(max_shift=2, shift_cost=-1000, structure_weight=0)),
        (25, 2, 2, dict(max_shift=5)),
        (26, 12, 12, dict(max_shift=1, simmatrix=None, sequence_match_similarity=300,
                          sequence_mismatch_similarity=-100)),
    ]
    for seed, n, m, ov in grid:
        sa, sb, ta, tb = synth.protein_pair(seed, n, m)
        small.append(run_case(ba, f"protein_s{seed}_{n}x{m}", sa, sb, ta, tb,
                              dict(pp, **ov), layers=True, decode=True))
    rgrid = [(31, 12, 10, dict(max_shift=1)), (32, 14, 14, dict(max_shift=2)),
             (33, 11, 13, dict(max_shift=2, gap_opening_cost=0, gap_cost=-200, shift_cost=-250)),
             (34, 16, 16, dict(max_shift=0))]
    for seed, n, m, ov in rgrid:
        sa, sb, ta, tb = synth.rna_pair(seed, n, m)
        small.append(run_case(ba, f"rna_s{seed}_{n}x{m}", sa, sb, ta, tb,
                              dict(synth.RNA_PARAMS, **ov), layers=True, decode=True,
                              modes=("default", "sorted")))
    with open(os.path.join(HERE, "small_layers.json"), "w") as fh:
        json.dump(small, fh, separators=(",", ":"))

    # Medium cases: score + trace only (strip changes, ties, several strips).
    medium = []
    mgrid = [(0, 32, 32, dict(max_shift=1)), (0, 64, 64, dict(max_shift=1)),
             (41, 50, 70, dict(max_shift=1)), (42, 45, 45, dict(max_shift=2)),
             (43, 30, 41, dict(max_shift=3)), (44, 100, 90, dict(max_shift=0)),
             (45, 48, 48, dict(max_shift=1, gap_opening_cost=0, gap_cost=-200, shift_cost=-250)),
             (46, 40, 44, dict(max_shift=2, gap_opening_cost=0, gap_cost=-200, shift_cost=-250)),
             (47, 90, 25, dict(max_shift=1)), (48, 25, 90, dict(max_shift=1)),
             (1000, 96, 96, dict(max_shift=1))]
    for seed, n, m, ov in mgrid:
        sa, sb, ta, tb = synth.protein_pair(seed, n, m)
        medium.append(run_case(ba, f"protein_s{seed}_{n}x{m}", sa, sb, ta, tb,
                               dict(pp, **ov), layers=False, decode=False))
    for seed, n, m, ov in [(2000, 60, 60, dict(max_shift=2)), (2001, 80, 64, dict(max_shift=1))]:
        sa, sb, ta, tb = synth.rna_pair(seed, n, m)
        medium.append(run_case(ba, f"rna_s{seed}_{n}x{m}", sa, sb, ta, tb,
                               dict(synth.RNA_PARAMS, **ov), layers=False, decode=True,
                               modes=("default", "sorted")))
    with open(os.path.join(HERE, "medium_traces.json"), "w") as fh:
        json.dump(medium, fh, separators=(",", ":"))

    if opts.dnapol:
        out = run_cli(["--filein", f"{REF}/Examples/DNAPolymerase1_Escherichia.cfssp",
                       f"{REF}/Examples/DNAPolymerase1_Xanthomonas.cfssp", "--type", "Protein",
                       "--shift_cost", "-150", "--structure_weight", "800", "--simmatrix",
                       "BLOSUM62", "--gap_opening_cost", "-150", "--gap_cost", "-50",
                       "--max_shift", "1"])
        with open(os.path.join(HERE, "dnapol_cli_stdout.txt"), "w") as fh:
            fh.write(out)


if __name__ == "__main__":
    main()
