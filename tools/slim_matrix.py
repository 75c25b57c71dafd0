"""fill time of the s=1 packed affine sweep per (pairs, team, kernel): tools/slim_matrix.py  [AB_LEN=1024] [SLIM_MATRIX=rounds]"""
import os, sys, subprocess
length = os.environ.get("AB_LEN", "1024")
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
from bialign_amd import synth
from bialign_amd.batch import make_batch
pairs = synth.protein_batch(int(os.environ["AB_PAIRS"]), int(os.environ["AB_LEN"]))
b = make_batch(pairs, dict(synth.PROTEIN_PARAMS))
ts = []
for _ in range(7):
    b.run(fill_only=True); ts.append(b.timing()["fill_ms"])
t = b.timing()
n = int(os.environ["AB_PAIRS"])
print(f"pairs {n:5d} team {os.environ.get('BIALIGN_TEAM','auto'):>4s} slim {os.environ.get('BIALIGN_SLIM','1')}: fill {min(ts[2:]):7.2f} ms = {min(ts[2:])*1024/n:6.2f} per 1024 pairs  waves/pair {t['waves_per_pair']} packed {t['packed_records']} chunks {b.info['nchunks']}", flush=True)
b.close()
'''
cases = [(3072, "1", "1"), (1536, "2", "1"), (1024, "3", "1"), (1024, "6", "1"), (512, "6", "1"), (256, "12", "1"), (2048, "3", "1"),
         (2048, "1", "0"), (1024, "2", "0"), (3072, "1", "0")]
if os.environ.get("SLIM_MATRIX") == "rounds":  # pair counts beyond one round of twelve-wave workgroups (256 CUs): which team, which kernel
    cases = [(n, t, sl) for n in (1280, 1536, 2048, 3072, 4096) for t, sl in (("2", "1"), ("3", "1"), ("6", "1"), ("1", "0"), ("2", "0"), ("", "1"))]
for pairs, team, slim in cases:
    env = dict(os.environ, AB_PAIRS=str(pairs), AB_LEN=length, BIALIGN_TEAM=team, BIALIGN_SLIM=slim)
    subprocess.run([sys.executable, "-c", code], env=env, timeout=300)
