"""bench.py's launch contract on CPU: `python bench.py --gpus N` run bare spawns N ranks itself (one per GPU under
torch.distributed.run, rendezvous on 127.0.0.1) and refuses a WORLD_SIZE / --gpus mismatch loudly."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    return env


@pytest.mark.timeout(600)
def test_bare_launch_spawns_one_rank_per_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run"], capture_output=True, text=True, env=_env(), timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 3 and line["dry_run"] is True


def test_world_size_mismatch_fails_loudly():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_more_gpus_than_the_node_has_fails_loudly():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=_env(), timeout=300)
    assert r.returncode != 0 and "exposes" in (r.stderr + r.stdout)


def test_every_tool_script_compiles():
    """tools/*.py run on the GPU box only; a syntax error there costs a gpurun call -- compile them all here."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py")))
    assert len(files) >= 20
    for f in files:
        compile(open(f).read(), f, "exec")
