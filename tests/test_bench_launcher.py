"""`python bench.py --gpus N` starts its own ranks: what can be checked of that without a GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_starts_the_ranks_and_reports_their_failure():
    """Without a GPU every rank ends with "needs an MI355X": the parent, which opens nothing itself, starts the two
    ranks with the rendezvous variables set, passes their messages on and exits with a non-zero status and no line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("on a GPU box tests/test_gpu_multirank_rehearsal.py runs the launcher for real")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=170)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
    assert p.stderr.count("bench.py needs an MI355X") == 2


def test_a_rank_with_the_wrong_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=170)
    assert p.returncode != 0 and "WORLD_SIZE=3 does not match --gpus 2" in p.stderr
