"""One-process-per-GPU sharding of a batch.

Pairs are independent (the reference builds one BiAligner per pair,
bialign.py:11), so ranks own contiguous blocks of pairs and never exchange DP
data.  The single collective is the final gather of int32 scores (4 B/pair):
``torch.distributed`` all_gather -- RCCL over xGMI with the "nccl" backend on
the GPUs, gloo on CPU for the tests.
"""
import numpy as np

from .batch import shard


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun)."""
    import os
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kwargs)
    return rank, local_rank, world


def block_layout(npairs_total, world, costs=None):
    """(blocks, width): the contiguous pair range of every rank and the common padded width of the
    all_gather's per-rank parts (every shard is padded to the widest)."""
    blocks = [shard(npairs_total, r, world, costs) for r in range(world)]
    return blocks, max(1, max(len(b) for b in blocks))


def assemble_scores(parts, npairs_total, costs=None):
    """The gathered per-rank parts (rank order, each padded to the common width) -> the int32 score
    vector in global pair order.  Split out of ``gather_scores`` so that the layout can be checked
    without N ranks (tests/test_gpu_dropin.py runs config 5's eight shards on one GPU through it)."""
    blocks, width = block_layout(npairs_total, len(parts), costs)
    out = np.empty(npairs_total, dtype=np.int32)
    for blk, part in zip(blocks, parts):
        part = np.asarray(part, dtype=np.int32)
        assert len(part) == width
        out[blk.start:blk.stop] = part[:len(blk)]
    return out


def gather_scores(local_scores, npairs_total, costs=None):
    """All ranks -> the full int32 score vector in global pair order.

    ``local_scores`` are this rank's scores for ``shard(npairs_total, rank, world, costs)``.
    """
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local_scores, dtype=np.int32)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert len(local) == npairs_total
        return local.copy()
    world, rank = dist.get_world_size(), dist.get_rank()
    blocks, width = block_layout(npairs_total, world, costs)
    assert len(local) == len(blocks[rank])
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.zeros(width, dtype=torch.int32, device=dev)
    mine[:len(local)] = torch.from_numpy(local).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return assemble_scores([p.cpu().numpy() for p in parts], npairs_total, costs)


def align_sharded(pairs, params, device=None, hbm_budget_bytes=0, balance=True):
    """Align this rank's shard of ``pairs`` on its GPU; every rank returns the scores of ALL
    pairs (gathered) and the traces of its own shard.  ``balance``: cut the shards by lattice
    cells ("cost-balance by n*m when lengths vary") instead of pair count."""
    import torch.distributed as dist
    from .batch import make_batch, pair_cost
    from .engine import default_engine
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    costs = [pair_cost(p, params["max_shift"]) for p in pairs] if balance else None
    mine = shard(len(pairs), rank, world, costs)
    if device is None:
        import torch
        device = torch.cuda.current_device()
    local_scores, local = np.zeros(0, dtype=np.int32), {}
    if len(mine):  # a rank may own nothing (fewer pairs than ranks, one giant pair)
        batch = make_batch([pairs[p] for p in mine], params, engine=default_engine(device),
                           hbm_budget_bytes=hbm_budget_bytes)
        batch.run()
        traces, complete = batch.traces()
        local_scores = batch.scores()
        local = {p: (traces[t], bool(complete[t])) for t, p in enumerate(mine)}
        batch.close()
    return gather_scores(local_scores, len(pairs), costs), local
