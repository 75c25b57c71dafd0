"""One pair (BASELINE config 3 shape by default): fill time per team shape.  AB_N x AB_M, AB_S."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bialign_amd import synth
from bialign_amd.batch import make_batch
E = lambda k, d: int(os.environ.get(k, d))
pairs = [synth.protein_pair(5, E("AB_N", 928), E("AB_M", 933))]
params = dict(synth.PROTEIN_PARAMS, max_shift=E("AB_S", 1))
for team in os.environ.get("AB_TEAMS", "1 2 4 8 x2 x4 x8 x16").split():
    os.environ["BIALIGN_TEAM"] = team
    b = make_batch(pairs, params)
    ts = []
    for _ in range(5):
        b.run(); ts.append(b.timing()["fill_ms"])
    t = b.timing()
    print(f"team {team:>4}: fill ms " + " ".join(f"{x:.2f}" for x in ts) + f"   ran as {t['waves_per_pair']}{'x' if t['cross_cu'] else ''}  tb {t['traceback_ms']:.2f}", flush=True)
    b.close()
