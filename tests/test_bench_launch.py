"""bench.py's launch plumbing without a GPU (BIALIGN_BENCH_REHEARSE=dry: no engine, the "scores" are
the global pair indices): a plain `python bench.py --gpus 2` must start its own two ranks -- a child
`torch.distributed.run`, never an exec -- and print ONE JSON line with n_gpus 2; the torchrun form
the driver uses must keep working.  The same flow with real kernels: tests/test_gpu_bench.py."""
import json
import os
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--pairs", "3", "--len", "40", "--steps", "2", "--warmup", "1"]


def _run(cmd, self_launched=False):
    env = dict(os.environ, BIALIGN_BENCH_REHEARSE="dry")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd() if callable(cmd) else cmd, env=env, capture_output=True, text=True, timeout=280, cwd=REPO)
    if callable(cmd) and r.returncode != 0 and any(k in r.stderr for k in ("EADDRINUSE", "ddress already in use", "DistNetworkError")):
        # the port picked for --master-port was taken between the pick and the launcher's bind: once more, with another
        r = subprocess.run(cmd(), env=env, capture_output=True, text=True, timeout=280, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # (gloo prints its connection notes to stdout)
    assert len(lines) == 1, r.stdout
    if self_launched:
        assert r.stdout.strip() == lines[0]  # the parent relays rank 0's line and nothing else
    return json.loads(lines[0])


def _check(line, n):
    assert line["n_gpus"] == n and line["steps"] == 2 and line["warmup"] == 1
    assert line["value"] is None and "NOT a measurement" in line["rehearsal"]
    assert line["scaling"] == "weak" and line["metric"] == "giga-DP-cells/sec"
    assert line["checked"] == {"gather_layout_ok": True, "gathered_pairs": 3 * n}
    assert line["config"]["pairs_per_gpu"] == 3


def test_plain_invocation_starts_its_own_ranks():
    _check(_run([sys.executable, "bench.py", "--gpus", "2"] + ARGS, self_launched=True), 2)


def test_torchrun_invocation_still_works():
    def cmd():  # the driver's form, on a port that is free right now
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2"] + ARGS
    _check(_run(cmd), 2)


def test_single_rank_dry():
    _check(_run([sys.executable, "bench.py"] + ARGS), 1)


def test_wrong_world_size_is_refused():
    env = dict(os.environ, BIALIGN_BENCH_REHEARSE="dry", WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + ARGS, env=env, capture_output=True, text=True,
                       timeout=120, cwd=REPO)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
