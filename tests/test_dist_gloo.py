"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): contiguous pair
shards, no data-path collective, one all_gather of int32 scores.  The per-rank
DP is played by the CPU oracle here (test infrastructure) because the engine is
GPU only; on the GPU box the same gather runs over RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, npairs, q, balanced=False):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from bialign_amd import synth
    from bialign_amd.batch import pair_cost, shard
    from bialign_amd.distributed import gather_scores, init_from_env
    from oracle import oracle
    r, _, w = init_from_env("gloo")
    assert (r, w) == (rank, world)
    params = dict(synth.PROTEIN_PARAMS)
    pairs = [synth.protein_pair(1000 + p, 12 + p % 5, 10 + p % 3) for p in range(npairs)]
    if balanced:  # one pair far larger than the rest: shards of very different pair counts
        pairs[0] = synth.protein_pair(999, 60, 50)
    costs = [pair_cost(p, params["max_shift"]) for p in pairs] if balanced else None
    mine = shard(npairs, rank, world, costs)
    if balanced:
        assert len(mine) == (1 if rank == 0 else npairs - 1)
    local = [oracle.solve(*pairs[p], params, want_trace=False)["score"] for p in mine]
    allscores = gather_scores(np.array(local, dtype=np.int32), npairs, costs)
    dist.barrier()
    q.put((rank, allscores.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("npairs,balanced", [(6, False), (7, False), (7, True)])
def test_two_rank_score_gather(npairs, balanced):
    import torch.multiprocessing as mp
    sys.path.insert(0, REPO)
    from bialign_amd import synth
    from oracle import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, npairs, q, balanced)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params = dict(synth.PROTEIN_PARAMS)
    pairs = [synth.protein_pair(1000 + p, 12 + p % 5, 10 + p % 3) for p in range(npairs)]
    if balanced:
        pairs[0] = synth.protein_pair(999, 60, 50)
    want = [oracle.solve(*p, params, want_trace=False)["score"] for p in pairs]
    assert got[0] == want and got[1] == want
