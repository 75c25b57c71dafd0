"""Does importing torch / initialising its CUDA context change the fill kernel's time?"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode in ("torch", "torch_sync"):
    import torch
    torch.cuda.set_device(0)
    if mode == "torch_sync":
        torch.cuda.synchronize()
from bialign_amd import synth
from bialign_amd.batch import make_batch
pairs = synth.protein_batch(1024, 512)
b = make_batch(pairs, dict(synth.PROTEIN_PARAMS))
ts = []
for _ in range(10):
    b.run(); ts.append(b.timing()["fill_ms"])
    if mode == "torch_sync":
        torch.cuda.synchronize()
print(f"{mode:10s} fill ms: med {statistics.median(ts[2:]):.2f}   all: " + " ".join(f"{t:.1f}" for t in ts))
