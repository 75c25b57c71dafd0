"""Host-side mirror of the reference API (no GPU needed): score inputs, trace
decoding in every output mode, trace evaluation -- against the golden vectors
of the compiled reference."""
import contextlib
import io

import numpy as np
import pytest

from conftest import load_golden
from bialign_amd import bialignment as ba
from bialign_amd import scoring, synth
from bialign_amd.batch import encode_pairs, shard
from oracle import oracle

KNOWN = load_golden("known_answers.json")
SMALL = load_golden("small_layers.json")
MEDIUM = load_golden("medium_traces.json")
DECODED = [r for r in KNOWN + SMALL + MEDIUM if "decode_full" in r]


def aligner(rec, **extra):
    return ba.BiAligner(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"], **dict(rec["params"], **extra))


def as_trace(rec):
    affine = rec["params"]["gap_opening_cost"] != 0
    return [list(c) if affine else tuple(c) for c in rec["trace"]]


@pytest.mark.parametrize("rec", DECODED, ids=[r["name"] for r in DECODED])
def test_decode_trace_full(rec):
    got = aligner(rec).decode_trace_full(as_trace(rec))
    assert [[n, s] for n, s in got] == rec["decode_full"]


@pytest.mark.parametrize("rec", DECODED, ids=[r["name"] for r in DECODED])
def test_decode_trace_modes(rec):
    for mode, want in rec["decode"].items():
        b = aligner(rec, outmode=mode)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            lines = b.decode_trace(as_trace(rec))
        assert lines == want["lines"], mode
        assert buf.getvalue() == want["stdout"], mode
    assert aligner(rec, nodescription=True).decode_trace(as_trace(rec)) == rec["decode_nodescription"]


@pytest.mark.parametrize("rec", [r for r in DECODED if r["params"]["gap_opening_cost"] != 0],
                         ids=lambda r: r["name"])
def test_eval_affine_trace(rec):
    assert list(aligner(rec).eval_trace(as_trace(rec))) == rec["eval_trace"]
    # free self-check of any trace: the column scores add up to the optimum
    last = rec["eval_trace"][-1].split()[-1]
    assert int(last) == rec["score"]


@pytest.mark.parametrize("rec", KNOWN + SMALL[:8] + SMALL[-4:], ids=lambda r: r["name"])
def test_lookup_tables_match_reference_mu(rec):
    """codes + tables reproduce mu1/mu2 of the plain restatement for every (i,j)."""
    p = rec["params"]
    model, mols_a, mols_b = encode_pairs([(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"])], p)
    mu1, mu2 = oracle.mu_tables(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"], p)
    (ca, ka), (cb, kb) = mols_a[0], mols_b[0]
    np.testing.assert_array_equal(model.s1[ca][:, cb], mu1[1:, 1:])
    np.testing.assert_array_equal(model.s2[ka][:, kb], mu2[1:, 1:])
    b = aligner(rec)
    n, m = len(rec["seqA"]), len(rec["seqB"])
    for i, j in [(1, 1), (n, m), (1, m), (n, 1), ((n + 1) // 2, (m + 1) // 2)]:
        assert b.mu1(i, j) == mu1[i, j]
        assert b.mu2(i, j) == mu2[i, j]


def test_rna_classes_edge_cases():
    # "()" : the closing bracket's partner is i-1, which range(1, i-1) skips (pyx:367-369)
    assert list(scoring.rna_classes("()")) == [scoring.RNA_DOWN, scoring.RNA_UNP]
    assert list(scoring.rna_classes("(.)")) == [scoring.RNA_DOWN, scoring.RNA_UNP, scoring.RNA_UP]
    assert list(scoring.rna_classes("x[.")) == [scoring.RNA_UNP] * 3
    with pytest.raises(IndexError):
        scoring.rna_classes("())")


def test_simmatrix_and_unknown_residue():
    sim = ba.read_simmatrix("BLOSUM62")
    assert sim["W"]["W"] == 1100 and sim["A"]["R"] == -100 and sim["*"]["*"] == 100
    assert ba.blosum62.splitlines()[0].split() == ["-"] + list("ARNDCQEGHILKMFPSTWYVBZX*")
    model = scoring.ScoreModel(dict(synth.PROTEIN_PARAMS))
    with pytest.raises(KeyError):
        model.encode_sequence("AJA")  # J is not a BLOSUM62 key (reference: KeyError in mu1)


def test_simmatrix_user_file(tmp_path):
    f = tmp_path / "m.txt"
    f.write_text("-  A  C\nA  2 -1\nC -1  3\n")
    assert ba.read_simmatrix(str(f)) == {"A": {"A": 200, "C": -100}, "C": {"A": -100, "C": 300}}


def test_constructor_errors(capsys):
    with pytest.raises(SystemExit):
        ba.BiAligner("ACGU", "ACG", "....", "..", **synth.RNA_PARAMS)
    assert "ERROR: Provided structure and sequence must have the same length." in capsys.readouterr().out
    with pytest.raises(SystemExit):
        ba.BiAligner("ACD", "ACD", None, None, **synth.PROTEIN_PARAMS)
    assert "ERROR: Structures have to be provided when aligning proteins" in capsys.readouterr().out


def test_module_surface():
    for name in ("SparseMatrix4D", "AffineDPMatrices", "guard_case", "argmin", "BiAligner", "mea",
                 "consensus_sequence", "consensus_sbpp", "parse_dotbracket", "highlight_sequence_identity",
                 "highlight_structure_identity", "highlight_structure_similarity", "blosum62",
                 "read_simmatrix", "read_molecule", "read_molecule_from_file", "breaklines", "runs",
                 "fourway_from_full", "plot_alignment", "helix_yadd_a", "helix_yadd_b", "__version__"):
        assert hasattr(ba, name), name
    assert ba.BiAligner.nl == 14 and list(ba.BiAligner.outmodes) == [
        "default", "sorted", "sorted_sym", "sorted_terse", "raw", "raw_struct", "full"]
    assert ba.guard_case((1, 1, 1, 1), (1, 1, 1, 1), 0) and not ba.guard_case((1, 0, 0, 0), (1, 1, 2, 1), 1)
    assert ba.argmin([[2, 1], [1, 5], [1, 5]]) == 1
    m = ba.AffineDPMatrices(2, 3, 1)
    assert len(m.states) == 9 and m.states[0] == (0, 1, 0, 1) and m.states[-1] == (1, 1, 1, 1)
    m[(1, 0, 1, 1)][1, 2, 2, 3] = 7
    assert m[(1, 0, 1, 1)][1, 2, 2, 3] == 7


def test_read_molecule():
    text = "Query 1 MVQ 3\nStruc 1 HHC 3\n\nQuery 4 IP 5\nStruc 4 EE 5\nTotal Residues: H: 2\n"
    assert ba.read_molecule(text, "Protein") == ["MVQIP", "HHCEE"]
    with pytest.raises(IOError):
        ba.read_molecule(text, "RNA")
    with pytest.raises(IOError):
        ba.read_molecule("Query 1 MV 2\nStruc 1 H 1\n", "Protein")
    assert list(ba.runs("HHHCC-")) == [("H", 0, 3), ("C", 3, 5), ("-", 5, 6)]
    assert ba.breaklines([("a", "abcdef"), ("b", "ABCDEF")], 4) == [
        [("a", "abcd"), ("b", "ABCD")], [("a", "ef"), ("b", "EF")]]


def test_shard_partition():
    for n, w in [(8192, 8), (10, 4), (3, 8)]:
        got = [list(shard(n, r, w)) for r in range(w)]
        assert sum(got, []) == list(range(n))
        assert max(map(len, got)) - min(map(len, got)) <= 1


def test_native_mea_equals_python_recursion():
    """bialign_host_mea (C) against the readable recursion, on matrices full of exact ties."""
    from bialign_amd import presentation as pr
    rng = np.random.default_rng(5)
    for n in (1, 2, 4, 9, 40, 90):
        for _ in range(12):
            m = np.zeros((n + 1, n + 1))
            u = np.triu(rng.integers(0, 4, size=(n, n)) / 4.0 * (rng.random((n, n)) < 0.3), 1)
            m[1:, 1:] = u + u.T
            np.fill_diagonal(m, rng.integers(0, 3, size=n + 1) / 2.0)
            fast, slow = pr.mea(m, brackets="[]"), pr.mea_python(m, brackets="[]")
            assert fast[0] == slow[0] and fast[1] == slow[1]


def test_batch_cli_input(tmp_path):
    from bialign_amd import batch_cli
    f = tmp_path / "p.tsv"
    f.write_text("# comment\nA1\tACD\tHHC\tB1\tAD\tHC\n\nA2\tW\tE\tB2\tWK\tEC\n")
    assert batch_cli.read_pairs(str(f)) == [("A1", "ACD", "HHC", "B1", "AD", "HC"), ("A2", "W", "E", "B2", "WK", "EC")]
    (tmp_path / "bad.tsv").write_text("A\tB\n")
    with pytest.raises(ValueError):
        batch_cli.read_pairs(str(tmp_path / "bad.tsv"))
    ns = batch_cli.build_parser().parse_args([str(f), "--type", "Protein", "--max_shift", "1", "--outmode", "raw"])
    assert ns.type == "Protein" and ns.max_shift == 1 and ns.gap_cost == -200 and not hasattr(ns, "seqA")


def test_cfssp_example_files_parse_to_config3_inputs():
    """The reference's two example inputs (BASELINE config 3; data files copied as fixtures) parse to
    exactly the sequences / structures the reference CLI echoed."""
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "dnapol_cli_stdout.txt")) as fh:
        lines = fh.read().split("\n")
    want = [lines[t].split("\t ")[1] for t in (1, 2, 3, 4)]
    a = ba.read_molecule_from_file(os.path.join(GOLDEN, "DNAPolymerase1_Escherichia.cfssp"), "Protein")
    b = ba.read_molecule_from_file(os.path.join(GOLDEN, "DNAPolymerase1_Xanthomonas.cfssp"), "Protein")
    assert [a[0], b[0], a[1], b[1]] == want and (len(a[0]), len(b[0])) == (928, 933)


def test_missing_input_file_exits_like_the_intended_reference(capsys):
    with pytest.raises(SystemExit):
        ba.read_molecule_from_file("/nonexistent/file.cfssp", "Protein")
    assert "Input file not found." in capsys.readouterr().out


# ---- real-valued RNA features -> dense mu2 table ---------------------------------------------
FEATURES = load_golden("fractional_features.json")


def feature_mol(feat):
    return {k: feat[k] for k in ("up", "down", "unp")}


@pytest.mark.parametrize("rec", FEATURES, ids=[r["name"] for r in FEATURES])
def test_dense_mu2_from_features_equals_reference_mu2(rec):
    """The vectorised sqrt / sum / truncation reproduces the reference's per-cell int(...) exactly."""
    tab = scoring.dense_mu2_from_features(feature_mol(rec["featuresA"]), feature_mol(rec["featuresB"]),
                                          rec["params"]["structure_weight"])
    assert tab.dtype == np.int32
    np.testing.assert_array_equal(tab, np.array(rec["mu2"], dtype=np.int32))


def test_dense_mu2_negative_product_raises_like_math_sqrt():
    a = dict(up=[0, 0.5], down=[0, 0.6], unp=[0, 1.0 - 0.5 - 0.6])
    b = dict(up=[0, 0.2], down=[0, 0.2], unp=[0, 0.6])
    with pytest.raises(ValueError):
        scoring.dense_mu2_from_features(a, b, 400)


# ---- base-pair probabilities -> features: same doubles as the reference's Python sums -----------
def literal_symmetrize(bpp):
    """pyx:326-338, word for word in plain Python."""
    n = len(bpp) - 1
    sb = np.zeros((n + 1, n + 1), dtype="float")
    for i in range(1, n + 1):
        for j in range(i + 1, n + 1):
            sb[i, j] = bpp[i][j]
            sb[j, i] = bpp[i][j]
    for i in range(1, n + 1):
        sb[i, i] = 1.0 - sum(sb[i, j] for j in range(1, n + 1))
    return sb


def literal_features(sbpp, n):
    """pyx:366-374 in plain Python."""
    up = [sum(sbpp[i][j] for j in range(1, i - 1)) for i in range(0, n + 1)]
    down = [sum(sbpp[i][j] for j in range(i + 1, n + 1)) for i in range(0, n + 1)]
    unp = [1.0 - up[i] - down[i] for i in range(0, n + 1)]
    return up, down, unp


def random_bpp(seed, n):
    rng = np.random.default_rng(seed)
    bpp = np.triu(rng.random((n + 1, n + 1)) ** 6, k=1)   # mostly tiny, a few sizeable entries
    bpp[0, :] = 0.0
    bpp *= 0.9 / max(1.0, (bpp + bpp.T).sum(axis=1).max())  # every row's pairing mass below 1
    return bpp


@pytest.mark.parametrize("seed,n", [(1, 1), (2, 2), (3, 3), (4, 17), (5, 60), (6, 131)])
def test_bpp_to_features_bit_equal_to_literal_python(seed, n):
    bpp = random_bpp(seed, n)
    sym = ba.BiAligner._symmetrize_bpps(bpp)
    want = literal_symmetrize(bpp)
    assert sym.tobytes() == want.tobytes()
    seq = "".join("ACGU"[(seed + t) % 4] for t in range(n))
    b = ba.BiAligner(seq, seq, None, None, bppA=bpp, bppB=bpp, **dict(synth.RNA_PARAMS))
    up, down, unp = literal_features(want, n)
    for got, ref in ((b.molA["up"], up), (b.molA["down"], down), (b.molA["unp"], unp)):
        assert np.array(got, dtype=float).tobytes() == np.array(ref, dtype=float).tobytes()
    assert b.molA["predicted"] and len(b.molA["structure"]) == n


def test_fixed_structure_features_unchanged_by_the_sequential_sums():
    for rec in load_golden("small_layers.json"):
        if rec["params"]["type"] != "RNA":
            continue
        b = ba.BiAligner(rec["seqA"], rec["seqB"], rec["strA"], rec["strB"], **rec["params"])
        up, down, unp = oracle.rna_features(rec["strA"])
        assert b.molA["up"] == up and b.molA["down"] == down and b.molA["unp"] == unp


def test_cost_balanced_shards_partition_and_balance():
    from bialign_amd.batch import pair_cost
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        for trial in range(20):
            costs = [int(c) for c in rng.integers(0, 1000, size=int(rng.integers(0, 40)))]
            blocks = [shard(len(costs), r, world, costs) for r in range(world)]
            assert [i for b in blocks for i in b] == list(range(len(costs)))       # ordered partition
            loads = [sum(costs[i] for i in b) for b in blocks]
            if costs:
                assert max(loads) <= sum(costs) / world + max(costs)                # never worse than one pair off
    assert pair_cost(("A" * 512, "C" * 512), 1) == 1537 * 1537
    with pytest.raises(ValueError):
        shard(3, 0, 2, [1, 2])


def test_one_shot_batch_encoding_equals_per_molecule_encoding():
    """batch.encode_flat joins a whole batch into one bytes object per kind and encodes it with one translate
    (round 3: 45 -> 3 ms for 1024 x len 1024); the codes, offsets and alphabets must be those of the per-molecule
    encoders -- proteins with a similarity matrix, proteins without (alphabet from the batch), RNA (classes from the
    bracket structure), ragged lengths, letters beyond latin-1 (per-molecule fallback), an unknown residue (KeyError
    like the reference's dict look-up), unequal sequence / structure lengths (the reference's ValueError text)."""
    from bialign_amd.batch import encode_flat
    from bialign_amd.scoring import ScoreModel
    cases = [
        ([synth.protein_pair(10 + t, 5 + 3 * t, 9 - t) for t in range(6)], dict(synth.PROTEIN_PARAMS)),
        ([synth.protein_pair(20 + t, 7, 4 + t) for t in range(4)], dict(synth.PROTEIN_PARAMS, simmatrix=None)),
        ([synth.rna_pair(30 + t, 25 + t, 31 - t) for t in range(5)], dict(synth.RNA_PARAMS)),
        ([("AΩA", "ΩΩ", "HHE", "EC"), ("A", "ΩA", "C", "HH")], dict(synth.PROTEIN_PARAMS, simmatrix=None)),
    ]
    for pairs, params in cases:
        model, fb = encode_flat(pairs, params)
        ref = ScoreModel(params, sequences=[p[0] for p in pairs] + [p[1] for p in pairs],
                         structures=[p[2] for p in pairs] + [p[3] for p in pairs])
        assert model.seq_keys == ref.seq_keys and model.cls_keys == ref.cls_keys
        np.testing.assert_array_equal(model.s1, ref.s1)
        np.testing.assert_array_equal(model.s2, ref.s2)
        assert fb.len_a.tolist() == [len(p[0]) for p in pairs] and fb.len_b.tolist() == [len(p[1]) for p in pairs]
        for t, (sa, sb, ta, tb) in enumerate(pairs):
            (ca, xa), (cb, xb) = fb.molecules("a")[t], fb.molecules("b")[t]
            np.testing.assert_array_equal(ca, ref.encode_sequence(sa))
            np.testing.assert_array_equal(cb, ref.encode_sequence(sb))
            np.testing.assert_array_equal(xa, ref.encode_structure(ta))
            np.testing.assert_array_equal(xb, ref.encode_structure(tb))
            assert fb.off_a[t] == sum(len(p[0]) for p in pairs[:t]) and fb.off_b[t] == sum(len(p[1]) for p in pairs[:t])
    with pytest.raises(KeyError):
        encode_flat([("ARND", "AJA", "HHEE", "HHE")], dict(synth.PROTEIN_PARAMS))
    with pytest.raises(ValueError, match="Provided structure and sequence must have the same length."):
        encode_flat([("ARND", "ARN", "HHE", "HHE")], dict(synth.PROTEIN_PARAMS))
    model, fb = encode_flat([], dict(synth.PROTEIN_PARAMS))
    assert len(fb.len_a) == 0 and len(fb.seq_a) == 0
