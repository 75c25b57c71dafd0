"""GPU: the drop-in API (BiAligner, CLI) end to end against the compiled
reference's outputs, plus size-independent properties at BASELINE sizes."""
import contextlib
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from bialign_amd import synth

pytestmark = pytest.mark.gpu

KNOWN = load_golden("known_answers.json")
SMALL = load_golden("small_layers.json")
CLI = load_golden("cli_outputs.json")


def supported(rec):
    return rec["params"]["max_shift"] <= 5


@pytest.mark.parametrize("rec", [r for r in KNOWN + SMALL if supported(r)], ids=lambda r: r["name"])
def test_bialigner_end_to_end(rec):
    from bialign_amd import bialignment as ba
    b = ba.BiAligner(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"], **rec["params"])
    score = b.optimize()
    assert isinstance(score, np.int64) and int(score) == rec["score"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        trace = b.traceback()
    affine = rec["params"]["gap_opening_cost"] != 0
    assert trace == [list(c) if affine else tuple(c) for c in rec["trace"]]
    assert ("WARNING: incomplete traceback" in buf.getvalue()) == (not rec["complete"])
    assert [[n, s] for n, s in b.decode_trace_full()] == rec["decode_full"]
    assert b.decode_trace() == rec["decode"]["default"]["lines"]
    assert list(b.eval_trace()) == rec["eval_trace"]   # non-affine: reads layers back through eval_case


def test_layer_containers_match_reference_types():
    from bialign_amd import bialignment as ba
    rec = next(r for r in KNOWN if r["name"] == "protein_2mer")
    b = ba.BiAligner(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"], **rec["params"])
    b.optimize()
    # SURVEY.md appendix A worked example: target (1,0,1,1) at (1,1,1,2)
    cases = list(b.affine_recursion_cases((1, 0, 1, 1), (1, 1, 1, 2)))
    assert len(cases) == 15
    assert [c[2] for c in cases[9:12]] == [500, 500, 500]
    assert [c[2] for c in cases[12:]] == [-350, -200, -350]
    layer = b._layers()[(1, 1, 1, 1)]
    assert layer[0, 0, 0, 0] == 0 and b._layers()[(1, 0, 1, 0)][0, 0, 0, 0] == -(1 << 30)


@pytest.mark.parametrize("name", list(CLI))
def test_cli_stdout(name, capsys):
    from bialign_amd import cli
    rec = CLI[name]
    if any(a == "--max_shift" for a in rec["args"]) is False and "--outmode" in rec["args"] and "help" in rec["args"]:
        pass
    try:
        cli.main(rec["args"])
    except SystemExit:
        pass
    assert capsys.readouterr().out == rec["stdout"]


def test_cli_dnapol_config3(capsys):
    """BASELINE config 3: DNA-Pol-I E. coli (928 aa) vs Xanthomonas (933 aa); the
    fixture is the reference CLI's stdout (score 761500, 1022 columns)."""
    from bialign_amd import cli
    with open(os.path.join(GOLDEN, "dnapol_cli_stdout.txt")) as fh:
        want = fh.read()
    lines = want.split("\n")
    seq_a, seq_b, str_a, str_b = (lines[t].split("\t ")[1] for t in (1, 2, 3, 4))
    assert (len(seq_a), len(seq_b)) == (928, 933) and "SCORE: 761500" in want
    cli.main([seq_a, seq_b, "--strA", str_a, "--strB", str_b, "--type", "Protein", "--shift_cost", "-150",
              "--structure_weight", "800", "--simmatrix", "BLOSUM62", "--gap_opening_cost", "-150",
              "--gap_cost", "-50", "--max_shift", "1"])
    assert capsys.readouterr().out == want


def _property_check(pairs, params):
    from bialign_amd.batch import encode_pairs
    from bialign_amd.engine import Batch, default_engine
    from bialign_amd.verify import rescore_trace
    model, mols_a, mols_b = encode_pairs(pairs, params)
    b = Batch(default_engine(), mols_a, mols_b, model.s1, model.s2, params["gap_opening_cost"],
              params["gap_cost"], params["shift_cost"], params["max_shift"])
    b.run()
    scores = b.scores()
    traces, ok = b.traces()
    affine = params["gap_opening_cost"] != 0
    for p, (sa, sb, _, _) in enumerate(pairs):
        total, consumed, drift = rescore_trace(traces[p], mols_a[p][0], mols_a[p][1], mols_b[p][0], mols_b[p][1],
                                               model.s1, model.s2, params["gap_opening_cost"],
                                               params["gap_cost"], params["shift_cost"], affine)
        assert ok[p]
        assert total == int(scores[p])                       # trace re-scores to the optimum
        assert consumed == (len(sa), len(sb), len(sa), len(sb))  # consumes exactly (n,m,n,m)
        assert drift <= params["max_shift"]                   # never leaves the band
    info = b.info
    b.close()
    return info


def test_properties_config2_shape():
    """BASELINE config 2 shape (len 512, s=1), 96 of the 1024 seeded pairs."""
    pairs = synth.protein_batch(96, 512)
    info = _property_check(pairs, dict(synth.PROTEIN_PARAMS))
    assert info["cells"] == 96 * 1537 * 1537


def test_properties_config5_shape():
    """BASELINE config 5 shape (len 1024, s=1), 24 pairs."""
    _property_check(synth.protein_batch(24, 1024), dict(synth.PROTEIN_PARAMS))


def test_properties_config4_shape():
    """BASELINE config 4 shape: RNA, len 2000, dot-bracket structures, s=2 (3.6 GB of layers per pair)."""
    params = dict(synth.RNA_PARAMS, max_shift=2)
    info = _property_check(synth.rna_batch(6, 2000), params)
    assert info["cells"] == 6 * 9999 * 9999


def test_properties_linear_and_wide_band():
    _property_check(synth.protein_batch(16, 300, 257),
                    dict(synth.PROTEIN_PARAMS, gap_opening_cost=0, gap_cost=-200, shift_cost=-250, max_shift=2))
    _property_check(synth.protein_batch(8, 200, 333), dict(synth.PROTEIN_PARAMS, max_shift=3))
    _property_check(synth.rna_batch(8, 400), dict(synth.RNA_PARAMS, max_shift=0))


def test_full_config2_team_vs_single_wave_under_load(monkeypatch):
    """BASELINE config 2 at full size (1024 pairs, 95 GB of layers, every SIMD busy): the
    two-waves-per-pair sweep and the one-wave sweep agree on every score and trace, and every
    trace re-scores to its optimum -- the hand-off between waves is timing dependent, so it is
    checked under the real load, not only on small cases."""
    from bialign_amd.batch import encode_pairs
    from bialign_amd.engine import Batch, default_engine
    from bialign_amd.verify import rescore_trace
    params = dict(synth.PROTEIN_PARAMS)
    pairs = synth.protein_batch(1024, 512)
    model, mols_a, mols_b = encode_pairs(pairs, params)
    got = {}
    for team in ("2", "1"):
        monkeypatch.setenv("BIALIGN_TEAM", team)
        b = Batch(default_engine(), mols_a, mols_b, model.s1, model.s2, params["gap_opening_cost"],
                  params["gap_cost"], params["shift_cost"], params["max_shift"])
        b.run()
        traces, ok = b.traces()
        got[team] = (b.scores().copy(), traces, ok.copy())
        b.close()
    np.testing.assert_array_equal(got["1"][0], got["2"][0])
    assert all(np.array_equal(x, y) for x, y in zip(got["1"][1], got["2"][1]))
    scores, traces, ok = got["2"]
    assert ok.all()
    for p in range(0, 1024, 3):
        total, consumed, drift = rescore_trace(traces[p], mols_a[p][0], mols_a[p][1], mols_b[p][0], mols_b[p][1],
                                               model.s1, model.s2, -150, -50, -150, True)
        assert total == int(scores[p]) and consumed == (512, 512, 512, 512) and drift <= 1


def test_batch_cli_equals_single_pair_cli(tmp_path, capsys):
    """The batch front end prints, per pair, what the single-pair CLI prints."""
    from bialign_amd import batch_cli, cli
    rows = []
    for t in range(4):
        sa, sb, ta, tb = synth.protein_pair(300 + t, 20 + 3 * t, 25 - t)
        rows.append(("a%d" % t, sa, ta, "b%d" % t, sb, tb))
    f = tmp_path / "pairs.tsv"
    f.write_text("".join("\t".join(r) + "\n" for r in rows))
    opts = ["--type", "Protein", "--simmatrix", "BLOSUM62", "--gap_opening_cost", "-150", "--gap_cost", "-50",
            "--shift_cost", "-150", "--structure_weight", "800", "--max_shift", "1", "--outmode", "sorted", "-v"]
    batch_cli.main([str(f)] + opts)
    blocks = capsys.readouterr().out.split(">pair ")[1:]
    assert len(blocks) == 4
    for t, (na, sa, ta, nb, sb, tb) in enumerate(rows):
        cli.main([sa, sb, "--strA", ta, "--strB", tb, "--nameA", na, "--nameB", nb] + opts)
        single = capsys.readouterr().out
        head, body = blocks[t].split("\n", 1)
        assert head == f"{t}\t{na}\t{nb}"
        assert body == single
    # --score_only: one line per pair with the same score, no alignment
    batch_cli.main([str(f)] + opts + ["--score_only"])
    lines = capsys.readouterr().out.strip().split("\n")
    assert len(lines) == 4
    for t, line in enumerate(lines):
        cols = line.split("\t")
        assert cols[:3] == [f"pair {t}", rows[t][0], rows[t][3]]
        assert ("SCORE: " + cols[3]) in blocks[t]


def test_cli_reports_engine_refusals_like_input_errors(tmp_path, capsys):
    """--score_only of the non-affine recurrence beyond the tiled band and scores outside the int32 window: "ERROR: ..."
    and exit, no traceback."""
    from bialign_amd import batch_cli, cli
    tsv = tmp_path / "pairs.tsv"
    tsv.write_text("a\tARND\tHHEE\tb\tARNE\tHHEC\n")
    with pytest.raises(SystemExit) as e:
        batch_cli.main([str(tsv), "--type", "Protein", "--gap_opening_cost", "0", "--max_shift", "8", "--score_only"])
    assert e.value.code == -1 and capsys.readouterr().out.startswith("ERROR: ")
    with pytest.raises(SystemExit) as e:
        cli.main(["ARND", "ARNE", "--strA", "HHEE", "--strB", "HHEC", "--type", "Protein", "--gap_opening_cost", "-150",
                  "--structure_weight", str(1 << 27)])
    assert e.value.code == -1 and "ERROR: " in capsys.readouterr().out


def test_c_abi_error_paths():
    """Bad arguments come back as negative codes with a message; nothing falls back to a CPU path."""
    import ctypes
    from bialign_amd import _lib
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import Engine
    pairs = [synth.protein_pair(1, 12, 9)]
    with pytest.raises(_lib.BialignError) as e:   # wide bands run (test_gpu_wide_band.py), score-only too, but not the lean traceback
        make_batch(pairs, dict(synth.PROTEIN_PARAMS, max_shift=6), lean_trace=True)
    assert e.value.code == _lib.E_UNSUPPORTED and "max_shift" in e.value.message
    with pytest.raises(_lib.BialignError) as e:
        make_batch(pairs, dict(synth.PROTEIN_PARAMS, max_shift=-1))
    assert e.value.code == _lib.E_INVALID
    with pytest.raises(_lib.BialignError) as e:
        make_batch(pairs, dict(synth.PROTEIN_PARAMS, structure_weight=1 << 27))
    assert e.value.code == _lib.E_RANGE
    with pytest.raises(_lib.BialignError) as e:
        make_batch([synth.protein_pair(2, 400, 400)], dict(synth.PROTEIN_PARAMS), hbm_budget_bytes=1 << 20)
    assert e.value.code == _lib.E_NOMEM
    with pytest.raises(_lib.BialignError) as e:
        Engine(10 ** 6)
    assert e.value.code == _lib.E_INVALID
    b = make_batch(pairs, dict(synth.PROTEIN_PARAMS))
    out = np.empty(1, dtype=np.int32)
    rc = _lib.lib.bialign_batch_get_scores(b._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    assert rc == _lib.E_INVALID and b"bialign_batch_run" in _lib.lib.bialign_last_error()   # not run yet
    b.run(fill_only=True)
    assert int(b.scores()[0]) != 0
    with pytest.raises(_lib.BialignError):
        b.traces()                                  # fill-only run has no trace
    b.close()


def test_unknown_residue_and_predicted_structure():
    from bialign_amd import bialignment as ba
    b = ba.BiAligner("AJA", "AAA", "HHH", "HHH", **dict(synth.PROTEIN_PARAMS, nameA="A", nameB="B"))
    with pytest.raises(KeyError):
        b.optimize()        # reference: KeyError from the similarity matrix look-up (pyx:407)
    with pytest.raises(ImportError):
        ba.BiAligner("ACGU", "ACGU", None, None, **dict(synth.RNA_PARAMS, nameA="A", nameB="B"))  # needs ViennaRNA


def test_properties_long_single_pair():
    """One long pair (5000 x 4000, 6.9 GB of layers, team of 8 waves): trace re-scores to the optimum."""
    _property_check([synth.protein_pair(77, 5000, 4000)], dict(synth.PROTEIN_PARAMS))
    _property_check([synth.rna_pair(78, 3000, 3500)], dict(synth.RNA_PARAMS, max_shift=2))


def _full_config_check(pairs, params, sample_every, expect_chunks_over, hbm_budget_bytes=0):
    from bialign_amd.batch import encode_pairs
    from bialign_amd.engine import Batch, default_engine
    from bialign_amd.verify import rescore_trace
    model, mols_a, mols_b = encode_pairs(pairs, params)
    b = Batch(default_engine(), mols_a, mols_b, model.s1, model.s2, params["gap_opening_cost"],
              params["gap_cost"], params["shift_cost"], params["max_shift"], hbm_budget_bytes=hbm_budget_bytes)
    assert b.info["nchunks"] > expect_chunks_over          # does not fit HBM at once: chunked
    b.run()
    scores = b.scores()
    traces, ok = b.traces()
    info = b.info
    b.close()
    assert ok.all()
    for p in range(0, len(pairs), sample_every):
        total, consumed, drift = rescore_trace(traces[p], mols_a[p][0], mols_a[p][1], mols_b[p][0], mols_b[p][1],
                                               model.s1, model.s2, params["gap_opening_cost"], params["gap_cost"],
                                               params["shift_cost"], True)
        n, m = len(pairs[p][0]), len(pairs[p][1])
        assert total == int(scores[p]) and consumed == (n, m, n, m) and drift <= params["max_shift"]
    return info, scores


def test_full_config4():
    """BASELINE config 4 at full size: 256 RNA pairs, len 2000, dot-bracket structures, max_shift=2
    (921 GB of int32 layers; 2 HBM-budgeted chunks with packed records, 4 without)."""
    info, _ = _full_config_check(synth.rna_batch(256, 2000), dict(synth.RNA_PARAMS, max_shift=2), 8, 1)
    assert info["cells"] == 256 * 9999 * 9999


def _oracle_job(job):
    """Worker (spawned: no GPU state): the CPU oracle on one synthetic protein pair -> score, trace columns, flag."""
    seed, length = job
    from bialign_amd import synth as sy
    from oracle import oracle
    ref = oracle.solve(*sy.protein_pair(seed, length), dict(sy.PROTEIN_PARAMS))
    return ref["score"], oracle.trace_to_lists(ref["trace"]), ref["complete"]


def _oracle_many(jobs):
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(min(len(jobs), max(1, min(16, len(os.sched_getaffinity(0)))))) as pool:
        return pool.map(_oracle_job, jobs, chunksize=1)


def test_full_config5_one_gpu_share():
    """BASELINE config 5, one rank's share (1024 of the 8192 protein pairs, len 1024): 348 GB of int32
    layers -- one launch with packed records (233 GB), chunked under a 150 GB budget; rank r of 8 would
    use seeds 1000 + r*1024 + p.  This is bench.py's own launch shape (packed records, teams of two, every
    SIMD busy): besides the properties, eight pairs spread over the launch order -- first, last, middle --
    are compared with the oracle, scores AND traces (the tie-breaks of pyx:535-586 under load)."""
    from bialign_amd.batch import make_batch
    from bialign_amd.engine import trace_codes_to_columns
    pairs = synth.protein_batch(1024, 1024)
    info, scores = _full_config_check(pairs, dict(synth.PROTEIN_PARAMS), 16, 0)
    assert info["cells"] == 1024 * 3073 * 3073
    info2, scores2 = _full_config_check(pairs, dict(synth.PROTEIN_PARAMS), 64, 1, hbm_budget_bytes=150 * 10 ** 9)
    np.testing.assert_array_equal(scores, scores2)
    sample = [0, 1, 255, 511, 512, 777, 1022, 1023]
    want = _oracle_many([(1000 + p, 1024) for p in sample])
    b = make_batch(pairs, dict(synth.PROTEIN_PARAMS))
    b.run()
    assert b.timing()["packed_records"] == 1 and b.info["nchunks"] == 1 and b.timing()["waves_per_pair"] in (2, 3)
    traces, ok = b.traces()
    got = b.scores()
    b.close()
    np.testing.assert_array_equal(got, scores)
    for p, (score, cols, complete) in zip(sample, want):
        assert int(got[p]) == score and bool(ok[p]) == complete, p
        assert trace_codes_to_columns(traces[p]) == cols, p


def test_config5_all_eight_shards_on_one_gpu():
    """BASELINE config 5 in full -- 8192 protein pairs, len 1024, max_shift 1 -- as its eight rank shards
    (rank r: seeds 1000 + r*1024 + p, what bench.py gives rank r) run one after the other on this one GPU; the
    shards' scores go through the all_gather's block layout (distributed.assemble_scores: padded per-rank parts
    -> global pair order) and two pairs per shard are compared with the oracle."""
    from bialign_amd.batch import make_batch, shard
    from bialign_amd.distributed import assemble_scores, block_layout
    from bialign_amd.engine import default_engine
    params = dict(synth.PROTEIN_PARAMS)
    world, per = 8, 1024
    blocks, width = block_layout(world * per, world)
    assert width == per and [b.start for b in blocks] == [r * per for r in range(world)]
    sample = [(r, p) for r in range(world) for p in (37 * r % per, per - 1 - 101 * r)]
    jobs = [(1000 + r * per + p, 1024) for r, p in sample]
    import multiprocessing as mp
    pool = mp.get_context("spawn").Pool(min(16, len(os.sched_getaffinity(0))))
    pending = pool.map_async(_oracle_job, jobs, chunksize=1)   # the host cores work while the GPU sweeps
    engine = default_engine()   # (its cached layer buffer serves all eight shards: a second engine would find HBM taken)
    parts = []
    for r in range(world):
        assert shard(world * per, r, world) == range(r * per, (r + 1) * per)
        b = make_batch(synth.protein_batch(per, 1024, seed0=1000 + r * per), params, engine=engine)
        b.run()
        assert b.timing()["packed_records"] == 1 and b.info["nchunks"] == 1
        parts.append(b.scores())
        b.close()
    allscores = assemble_scores(parts, world * per)
    assert len(allscores) == 8192
    for r in range(world):   # every block holds its own shard's scores (the shards are different pairs)
        np.testing.assert_array_equal(allscores[r * per:(r + 1) * per], parts[r])
        assert r == 0 or not np.array_equal(parts[r], parts[0])
    want = pending.get(timeout=900)
    pool.close()
    for (r, p), (score, _, _) in zip(sample, want):
        assert int(allscores[r * per + p]) == score, (r, p)


def test_long_molecules_large_lds():
    """30 000 x 2 000 at max_shift 0: sequence staging above the default 64 KB of dynamic LDS."""
    _property_check([synth.protein_pair(91, 30000, 2000)], dict(synth.PROTEIN_PARAMS, max_shift=0))


def test_cli_config3_from_files_readme_command(capsys):
    """The README's own command line (reference README.md:159-161, with the `--filein` abbreviation)
    on the two example CFSSP files: stdout identical to the reference's."""
    from bialign_amd import cli
    with open(os.path.join(GOLDEN, "dnapol_cli_stdout.txt")) as fh:
        want = fh.read()
    cli.main(["--filein", os.path.join(GOLDEN, "DNAPolymerase1_Escherichia.cfssp"),
              os.path.join(GOLDEN, "DNAPolymerase1_Xanthomonas.cfssp"), "--type", "Protein", "--shift_cost", "-150",
              "--structure_weight", "800", "--simmatrix", "BLOSUM62", "--gap_opening_cost", "-150",
              "--gap_cost", "-50", "--max_shift", "1"])
    assert capsys.readouterr().out == want


def test_c_abi_from_plain_c(tmp_path):
    """examples/align_one.c: the boundary used from C alone (gcc, no Python in the process)."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "align_one")
    subprocess.run(["gcc", "-std=c11", "-I" + os.path.join(repo, "include"), os.path.join(repo, "examples", "align_one.c"),
                    "-o", exe, "-L" + os.path.join(repo, "bialign_amd"), "-lbialign_hip",
                    "-Wl,-rpath," + os.path.join(repo, "bialign_amd")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    assert "SCORE: 6800" in out
    rec = [r for r in load_golden("known_answers.json") if r["name"] == "readme_rna_toy"][0]
    want = "".join("%x" % (c[0] * 8 + c[1] * 4 + c[2] * 2 + c[3]) for c in rec["trace"])
    assert f"TRACE ({len(rec['trace'])} columns, complete): {want}" in out
