"""Wall time of BASELINE config 3 (DNA-Pol-I pair) through the drop-in CLI path."""
import io, os, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bialign_amd import cli
lines = open(os.path.join(ROOT, "tests", "golden", "dnapol_cli_stdout.txt")).read().split("\n")
sa, sb, ta, tb = (lines[t].split("\t ")[1] for t in (1, 2, 3, 4))
args = [sa, sb, "--strA", ta, "--strB", tb, "--type", "Protein", "--shift_cost", "-150", "--structure_weight", "800",
        "--simmatrix", "BLOSUM62", "--gap_opening_cost", "-150", "--gap_cost", "-50", "--max_shift", "1"]
for rep in range(3):
    buf = io.StringIO(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        cli.main(args)
    dt = time.perf_counter() - t0
    print(f"run {rep}: {dt*1e3:.1f} ms, identical to reference stdout: {buf.getvalue() == chr(10).join(lines)}")
from bialign_amd import bialignment as ba
b = ba.BiAligner(sa, sb, ta, tb, type="Protein", shift_cost=-150, structure_weight=800, simmatrix="BLOSUM62",
                 gap_opening_cost=-150, gap_cost=-50, max_shift=1, nameA="A", nameB="B")
t0 = time.perf_counter(); s = b.optimize(); dt = time.perf_counter() - t0
print("optimize():", s, f"{dt*1e3:.1f} ms", b._batch.timing())
