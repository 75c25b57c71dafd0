"""bench.py end to end on the GPU box: the self-launched two-rank flow with real kernels (ranks share
the one GPU, gloo collectives: BIALIGN_BENCH_REHEARSE=1) and the N=1 line's `checked` block (GPU
scores of the cpu_baseline pairs against the oracle's).  Small shapes: script tests, not numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, **env_extra):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py"] + args, env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_plain_two_rank_invocation_with_kernels():
    line = _bench(["--gpus", "2", "--pairs", "3", "--len", "60", "--steps", "2", "--warmup", "1"], BIALIGN_BENCH_REHEARSE="1")
    assert line["n_gpus"] == 2 and "NOT a measurement" in line["rehearsal"]
    assert line["checked"]["gather_layout_ok"] and line["checked"]["gathered_pairs"] == 6
    assert line["value"] > 0 and line["config"]["cells_per_gpu"] == 3 * (61 * 3 - 2) ** 2


def test_single_gpu_line_is_checked_against_the_oracle():
    line = _bench(["--pairs", "6", "--len", "150", "--steps", "2", "--warmup", "1", "--no-extra"])
    c = line["checked"]
    assert c["scores_equal"] and c["pairs"] >= 3 and c["oracle_scores"] == c["gpu_scores"] and c["gather_layout_ok"]
    assert line["cpu_baseline"]["kind"] == "port" and "_scores" not in line["cpu_baseline"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
