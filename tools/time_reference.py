"""Throughput of the *reference* (Cython) fill on this container's cores.
Dev container only: needs the out-of-tree reference build of tests/golden/make_golden.py."""
import multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/tmp/bialign_ref_build"); sys.path.insert(1, "/root/reference/src")
from bialign_amd import synth


def one(args):
    seed, n, params = args
    import bialignment
    sa, sb, ta, tb = synth.protein_pair(seed, n)
    b = bialignment.BiAligner(sa, sb, ta, tb, **dict(params, nameA="A", nameB="B"))
    t0 = time.perf_counter(); b.optimize(); dt = time.perf_counter() - t0
    return synth.cells_per_pair(n, n, params["max_shift"]) / dt


if __name__ == "__main__":
    p = dict(synth.PROTEIN_PARAMS)
    for n in (128, 256):
        r = [one((1000 + t, n, p)) for t in range(3)]
        print(f"single core, len {n}, s=1: " + ", ".join(f"{x/1e3:.1f}" for x in r) + " kcells/s", flush=True)
    with mp.Pool(8) as pool:
        t0 = time.perf_counter(); r = pool.map(one, [(1000 + t, 128, p) for t in range(8)]); dt = time.perf_counter() - t0
    print(f"8 processes, len 128: {8*synth.cells_per_pair(128,128,1)/dt/1e3:.1f} kcells/s aggregate ({dt:.1f} s wall)")
