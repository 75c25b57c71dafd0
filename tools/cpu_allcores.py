"""The CPU oracle (compiled port of the reference recurrence) on all host cores of the GPU box:
one process per core, config-2-shaped pairs."""
import multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bialign_amd import synth


def one(seed):
    from oracle import oracle
    sa, sb, ta, tb = synth.protein_pair(seed, 512)
    p = dict(synth.PROTEIN_PARAMS)
    mu1, mu2 = oracle.mu_tables(sa, sb, ta, tb, p)
    t0 = time.perf_counter()
    _, lay = oracle.affine_fill(512, 512, 1, -150, -50, -150, mu1, mu2)
    oracle.affine_traceback(512, 512, 1, -150, -50, -150, mu1, mu2, lay)
    return time.perf_counter() - t0


if __name__ == "__main__":
    from oracle import oracle
    oracle.build()
    # the GPU box exposes all host CPUs but grants a 16-CPU share per GPU
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("CPU_SHARE", 16)))
    cells = synth.cells_per_pair(512, 512, 1)
    t1 = one(1000)
    print(f"1 core: {cells / t1 / 1e6:.2f} Mcells/s")
    with mp.Pool(cores) as pool:
        t0 = time.perf_counter(); pool.map(one, [1000 + t for t in range(2 * cores)]); dt = time.perf_counter() - t0
    print(f"{cores} cores: {2 * cores * cells / dt / 1e6:.2f} Mcells/s aggregate ({dt:.1f} s wall, {2 * cores} pairs)")
