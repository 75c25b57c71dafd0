#!/usr/bin/env python3
"""Headline benchmark: giga-DP-cells/s of the bi-alignment hot path on MI355X.

One *step* = one pass of the hot path (affine DP fill + traceback, scores
gathered) over one rank's batch of synthetic pairs, inputs resident in HBM.

Workload = what BASELINE.json's metric is quoted on: **1024 synthetic protein
pairs per GPU, n = m = 1024, BLOSUM62, affine gaps, max_shift = 1** -- config 5's
per-GPU share (rank r owns pairs [r*1024, (r+1)*1024), pair p
is drawn from seed 1000 + p).  Its nine int32 layers are 348 GB; with packed
layer records (233 GB in HBM) one step is ONE fill launch + one traceback launch.
Pairs are independent: ranks share nothing on the data path, the only collective
is the final all_gather of int32 scores over RCCL (weak scaling).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

``python bench.py --gpus N`` (N > 1) outside torchrun starts its own N rank
processes: a child ``python -m torch.distributed.run --nproc-per-node N bench.py ...``
launched before this process has touched torch or the GPU; the parent relays
rank 0's JSON line and the child's exit code.

Prints ONE JSON line on rank 0.  At N=1 the line also carries
`cpu_baseline` (the C oracle on one host core and on all host cores of this box,
timed BEFORE the GPU is initialised) and `extra.config2` (BASELINE configs[1],
1024 pairs x len 512, with the default allocation and with a placement-probed
buffer), both outside the timed region.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def _oracle_pair(job):
    """Worker: the CPU oracle on one synthetic protein pair -> (seconds, cells)."""
    seed, length, s = job
    from bialign_amd import synth
    from oracle import oracle
    sa, sb, ta, tb = synth.protein_pair(seed, length)
    p = dict(synth.PROTEIN_PARAMS, max_shift=s)
    mu1, mu2 = oracle.mu_tables(sa, sb, ta, tb, p)  # input preparation, not timed
    t0 = time.perf_counter()
    score, layers = oracle.affine_fill(length, length, s, p["gap_opening_cost"], p["gap_cost"], p["shift_cost"], mu1, mu2)
    oracle.affine_traceback(length, length, s, p["gap_opening_cost"], p["gap_cost"], p["shift_cost"], mu1, mu2, layers)
    return time.perf_counter() - t0, synth.cells_per_pair(length, length, s), int(score)


def cpu_baseline(length, s, seed0, npairs, budget_s=12.0):
    """oracle/bialign_oracle.c (a literal C port of the reference recurrence, kind "port") on
    the first pairs of the same workload: one host core, then one process per host core of
    this box's share.  Runs before anything touches the GPU (the pool forks)."""
    import multiprocessing as mp
    from oracle import oracle
    oracle.build()
    spent, cells, n1, scores = 0.0, 0, 0, []
    while spent < budget_s and n1 < min(8, npairs):
        dt, c, sc = _oracle_pair((seed0 + n1, length, s))
        spent += dt
        cells += c
        scores.append(sc)  # kept: the GPU's scores of the same pairs are compared with them after the timed region
        n1 += 1
    one = cells / spent / 1e9
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("CPU_SHARE", 16)))
    per_core = max(1, min(n1, 4))
    jobs = [(seed0 + t, length, s) for t in range(cores * per_core)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_oracle_pair, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    return {"value": one, "unit": "Gcells/s", "cores": 1, "kind": "port", "_scores": scores,
            "sample": f"first {n1} pairs of the workload (fill+traceback, {cells} cells, {spent:.1f} s on one host core)",
            "all_cores": {"value": sum(r[1] for r in res) / wall / 1e9, "unit": "Gcells/s", "cores": cores,
                          "sample": f"first {len(jobs)} pairs, one process per core, {wall:.1f} s wall"},
            # the Cython reference itself cannot travel to this box; its rate was measured in the dev container
            "reference_cython": {"value": 21e-6, "unit": "Gcells/s", "cores": 1, "measured_live": False,
                                 "source": "BASELINE.md section 4.1 (tools/time_reference.py, len 128/256 pairs of this family)"}}


def run_steps(batch, count, gather, npairs_total):
    """count steps; -> (fill ms, traceback ms, fill launches) summed over them."""
    acc = [0.0, 0.0, 0]
    for _ in range(count):
        batch.run()                      # fill + traceback, all chunks; returns when the device is done
        gather(batch.scores(), npairs_total)  # the one collective (no-op at N=1)
        t = batch.timing()               # HIP-event times recorded on the engine's stream
        acc[0] += t["fill_ms"]; acc[1] += t["traceback_ms"]; acc[2] += t["fill_launches"]
    return acc


VALU_CYCLES = 4  # SIMD cycles one wave64 VALU instruction of this sweep occupies: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU is
# exactly one quad-cycle in every pass kept under profiles/, and the sweep's time follows its instruction count, not its
# occupancy (2 or 3 waves per SIMD) nor its bytes (tools/valu_rate.hip has the per-instruction rates)


def roofline(info, fill_ms, launches, key):
    """`achieved` = ALGORITHMIC bytes (36 B per DP cell) of one fill launch / its average
    duration (HIP events on the engine's stream).  Beside it, from the committed rocprofv3 counter passes of the same
    launch shape (profiles/hbm_traffic.json names the directory): `traffic` = FETCH_SIZE + WRITE_SIZE bytes per launch,
    `traffic_frac` = that / launch time / 8 TB/s (what the memory system really moves), `issue` = VALU
    wave-instructions x 4 cycles / (1024 SIMDs x GUI-active cycles of the profiled launch) -- the fraction of the SIMDs'
    vector-issue capacity the sweep uses (`issue_2cyc`: the same at the 2 cycles a wave64 instruction would take if two
    waves' instructions overlapped perfectly, which this instruction mix does not do); `bound_detail` names the largest."""
    fill_avg_ms = fill_ms / max(launches, 1)                  # average fill-kernel launch
    bytes_per_launch = info["layer_bytes"] / info["nchunks"]  # 36 B x cells of one launch
    achieved = bytes_per_launch / (fill_avg_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": "fill_affine_slim_kernel / fill_affine_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS,
           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "traffic_source": None,
           "bytes_per_launch": bytes_per_launch, "avg_launch_ms": fill_avg_ms}
    tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):  # PMC counters cannot be read from inside the run: per-launch values of the same
        with open(tpath) as fh:  # launch shape from the committed rocprofv3 passes, with their directory
            entry = json.load(fh).get(key)
        if entry:
            out["traffic"], out["traffic_source"] = entry["bytes_per_launch"], entry["profile"]
            out["traffic_frac"] = entry["bytes_per_launch"] / (fill_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
            fr = {"hbm (measured traffic)": out["traffic_frac"]}  # physical limits only; `frac` is the section 8d yardstick
            if entry.get("valu_wave_insts_per_launch") and entry.get("gui_active_cycles_per_launch"):
                simd_cycles = 1024 * entry["gui_active_cycles_per_launch"]
                out["issue"] = entry["valu_wave_insts_per_launch"] * VALU_CYCLES / simd_cycles
                out["issue_2cyc"] = entry["valu_wave_insts_per_launch"] * 2 / simd_cycles
                out["issue_source"] = (f"SQ_INSTS_VALU {entry['valu_wave_insts_per_launch']:.4g} x {VALU_CYCLES} cycles / (1024 SIMDs x "
                                       f"GRBM_GUI_ACTIVE/8 = {entry['gui_active_cycles_per_launch']:.4g} cycles), both of the profiled launch")
                fr["valu issue"] = out["issue"]
            out["bound_detail"] = max(fr, key=fr.get)
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` outside torchrun: start the N rank processes as a CHILD (never an exec, and before
    this process has imported torch or touched the GPU), relay rank 0's JSON line and the child's exit code."""
    import subprocess
    # --standalone: the launcher binds its rendezvous store to a free port itself (no pick-then-bind race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:  # the ranks' stderr passes through; of stdout only the JSON line is ours
        try:
            if "metric" in json.loads(out):
                line = out.strip()
                continue
        except ValueError:
            pass
        sys.stderr.write(out)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    raise SystemExit(rc if rc else (0 if line else 1))


class DryBatch:
    """BIALIGN_BENCH_REHEARSE=dry: the launch / shard / gather / timing plumbing with NO engine behind it (a box
    without a GPU: the non-GPU test of the N > 1 flow).  Its "scores" are the global pair indices, so the gather's
    block layout can be checked; nothing is measured."""
    def __init__(self, first, count, length, s):
        from bialign_amd import synth
        import numpy as np
        self._scores = np.arange(first, first + count, dtype=np.int32)
        cells = synth.cells_per_pair(length, length, s) * count
        self.info = {"cells": cells, "npairs": count, "nchunks": 1, "layer_bytes": 36 * cells, "hbm_layer_bytes": 0}

    def run(self):
        pass

    def scores(self):
        return self._scores

    def timing(self):
        return {"fill_ms": 1.0, "traceback_ms": 0.0, "fill_launches": 1, "waves_per_pair": 0, "packed_records": 0}

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1024, help="pairs per GPU")
    ap.add_argument("--len", type=int, default=1024, dest="length")
    ap.add_argument("--max_shift", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed config-2 block")
    ap.add_argument("--reserve-tries", type=int, default=1,
                    help="headline batch: candidate placements of the layer buffer probed in setup (Engine.reserve); "
                         "needs twice the buffer in free HBM, i.e. not possible at the default workload")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # BIALIGN_BENCH_REHEARSE: rehearse the N>1 flow on a box with fewer GPUs than ranks -- for testing the script,
    # never for numbers.  "1": ranks share the GPU(s), collectives over gloo;  "dry": no GPU at all (DryBatch).
    rehearse = os.environ.get("BIALIGN_BENCH_REHEARSE", "") in ("1", "dry")
    dry = os.environ.get("BIALIGN_BENCH_REHEARSE") == "dry"

    cpu = None
    if world == 1 and not args.no_cpu_baseline and not dry:  # host cores only, before the GPU is initialised
        cpu = cpu_baseline(args.length, args.max_shift, 1000, args.pairs)

    import numpy as np
    import torch
    import torch.distributed as dist
    device = 0
    if not dry:
        device = local_rank % torch.cuda.device_count() if rehearse else local_rank
        torch.cuda.set_device(device)
    cdev = torch.device("cpu") if rehearse else torch.device("cuda", device)  # where collectives run
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))

    from bialign_amd import synth
    from bialign_amd.distributed import gather_scores

    params = dict(synth.PROTEIN_PARAMS, max_shift=args.max_shift)
    seed0 = 1000 + rank * args.pairs
    placement = {"reserve_tries": 1, "probe_gbps": None}
    engine = None
    if dry:
        batch = DryBatch(rank * args.pairs, args.pairs, args.length, args.max_shift)
    else:
        from bialign_amd.batch import make_batch
        from bialign_amd.engine import Engine
        pairs = synth.protein_batch(args.pairs, args.length, seed0=seed0)
        engine = Engine(device)
        budget = 0
        if rehearse:  # ranks share one GPU's memory
            budget = int(torch.cuda.mem_get_info(device)[0] * 0.8 / world)
        if args.reserve_tries > 1 and not rehearse:
            probe = make_batch(pairs, params, engine=engine)
            need = probe.info["hbm_layer_bytes"] + 64
            probe.close()
            try:
                placement = {"reserve_tries": args.reserve_tries, "probe_gbps": engine.reserve(need, tries=args.reserve_tries)}
            except Exception as e:  # placement is an optimisation: never fail the run over it
                placement["reserve_error"] = str(e)[:200]
        batch = make_batch(pairs, params, engine=engine, hbm_budget_bytes=budget)  # inputs now resident in HBM
    info = batch.info
    for _ in range(2):  # engine warm-up, not steps: the first launches load the code objects, touch the buffer's pages
        batch.run()     # for the first time and ramp the clocks

    def barrier():
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    run_steps(batch, args.warmup, gather_scores, args.pairs * world)
    barrier()
    t0 = time.perf_counter()
    fill_ms, tb_ms, launches = run_steps(batch, args.steps, gather_scores, args.pairs * world)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([info["cells"], info["npairs"]], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        total_cells, total_pairs = int(tot[0].item()), int(tot[1].item())
    else:
        total_cells, total_pairs = info["cells"], info["npairs"]
    timing = batch.timing()
    # ---- outside the timed region: what was timed is checked.  Every rank's gathered score vector must hold its own
    # scores in its block (the gather's layout); rank 0's first pairs are the ones the CPU oracle solved for
    # `cpu_baseline` -- the GPU's scores of the same pairs must equal them.
    gathered = gather_scores(batch.scores(), args.pairs * world)
    own_ok = bool(np.array_equal(gathered[rank * args.pairs:(rank + 1) * args.pairs], batch.scores()))
    checked = {"gather_layout_ok": own_ok, "gathered_pairs": int(len(gathered))}
    if dry:
        checked["gather_layout_ok"] = own_ok and bool(np.array_equal(gathered, np.arange(args.pairs * world)))
    if cpu is not None:
        want = cpu.pop("_scores")
        got = [int(x) for x in batch.scores()[:len(want)]]
        checked.update({"pairs": len(want), "scores_equal": got == want, "oracle_scores": want, "gpu_scores": got,
                        "against": "oracle/bialign_oracle.c on the cpu_baseline pairs (seeds 1000..)"})
    batch.close()

    line = None
    if rank == 0:
        value = total_cells * args.steps / elapsed / 1e9
        is_cfg5 = (args.pairs, args.length, args.max_shift) == (1024, 1024, 1)
        line = {
            "metric": "giga-DP-cells/sec", "value": None if dry else value,
            "unit": "Gcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            **({"rehearsal": ("no engine, no GPU (DryBatch): launch/shard/gather plumbing only" if dry else
                              "ranks share GPUs, gloo collectives") + ": NOT a measurement"} if rehearse else {}),
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "pairs_per_s": total_pairs * args.steps / elapsed,
            # the layer buffer of the headline is whatever hipMalloc returned (no placement probing)
            **({"value_default_alloc": value} if placement["probe_gbps"] is None and not dry else {}),
            "config": {"baseline_config": "BASELINE.json configs[4] per-GPU share (the metric's 1k x 1k, max_shift=1 workload)"
                                          if is_cfg5 else "custom (--pairs/--len/--max_shift)",
                       "workload": f"{args.pairs} synthetic protein pairs per GPU, len {args.length}, "
                                   f"BLOSUM62, affine gaps (beta=-150, gamma=-50, Delta=-150, sw=800), "
                                   f"max_shift={args.max_shift}; fill + traceback + score gather",
                       "pairs_per_gpu": args.pairs, "len": args.length, "max_shift": args.max_shift,
                       "cells_per_gpu": info["cells"], "chunks_per_step": info["nchunks"],
                       "layer_bytes_per_gpu": info["layer_bytes"], "hbm_layer_buffer_bytes": info["hbm_layer_bytes"],
                       "waves_per_pair": timing["waves_per_pair"],
                       # interior steps store base + 16-bit offsets (lossless, decoded by ghost feed and traceback)
                       "layer_records": "packed" if timing.get("packed_records") else "full",
                       "sharding": f"rank r owns pairs [r*{args.pairs}, (r+1)*{args.pairs}) of {args.pairs * world}; "
                                   f"no data-path collective, one all_gather of int32 scores",
                       "layer_buffer_placement": placement},
            "kernel_ms": {"fill": fill_ms / args.steps, "traceback": tb_ms / args.steps},
            "roofline": None if dry else roofline(info, fill_ms, launches, f"protein_{args.pairs}x{args.length}_s{args.max_shift}"),
            "checked": checked,
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu  # measured at N=1 only (None otherwise)

    if rank == 0 and world == 1 and not args.no_extra and not rehearse:
        from bialign_amd.batch import make_batch
        # ---- untimed extra: BASELINE configs[1] (1024 pairs x len 512), first on the buffer the engine
        # already holds (default allocation), then on a placement-probed one (profiles/r01e_placement)
        c2pairs = synth.protein_batch(1024, 512, seed0=1000)
        c2params = dict(synth.PROTEIN_PARAMS)
        extra = {"workload": "BASELINE.json configs[1]: 1024 synthetic protein pairs, len 512, max_shift=1, affine"}
        engine.trim()
        for label, tries in (("default_alloc", 1), ("reserved", 4)):
            b2 = make_batch(c2pairs, c2params, engine=engine)
            i2 = b2.info
            if tries > 1:
                b2.close()
                try:
                    rate = engine.reserve(i2["hbm_layer_bytes"] + 64, tries=tries)
                except Exception as e:
                    extra[label] = {"error": str(e)[:200]}
                    continue
                b2 = make_batch(c2pairs, c2params, engine=engine)
            for _ in range(3):
                b2.run()
            t1 = time.perf_counter()
            f2, tb2, l2 = run_steps(b2, 5, gather_scores, 1024)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            extra[label] = {"value": i2["cells"] * 5 / dt / 1e9, "unit": "Gcells/s", "ms_per_step": dt / 5 * 1e3,
                            "kernel_ms": {"fill": f2 / 5, "traceback": tb2 / 5},
                            "roofline": roofline(i2, f2, l2, "protein_1024x512_s1"),
                            **({"probe_gbps": rate, "reserve_tries": tries} if tries > 1 else {})}
            b2.close()
        line["extra"] = {"config2": extra}

    if rank == 0:
        print(json.dumps(line), flush=True)
    if engine is not None:
        engine.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and not checked.get("scores_equal", True):
        raise SystemExit("bench: GPU scores differ from the oracle's on the checked pairs")
    if not checked["gather_layout_ok"]:
        raise SystemExit("bench: gathered scores do not match this rank's own block")


if __name__ == "__main__":
    main()
