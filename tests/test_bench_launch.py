"""CPU: `python bench.py --gpus N` starts its own N ranks (child process, before any GPU call) and never reports a
line for a different number of GPUs than it was asked for."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    try:
        import ctypes as C
        from aether_primitives_amd import _lib
        n = C.c_int(0)
        return _lib.load().aeth_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.mark.timeout(300)
def test_gpus_2_spawns_two_ranks_and_fails_loudly_without_gpus():
    if _have_gpu():
        pytest.skip("a GPU is visible: covered by the gpu-marked test")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=280)
    assert p.returncode != 0                               # no GPU here: both ranks die, the parent reports it
    assert "No HIP GPUs" in p.stderr or "no GPU" in p.stderr.lower() or "HIP" in p.stderr
    # two ranks were started (torchrun names each failed rank) and no 1-GPU line was printed instead
    assert "local_rank: 1" in p.stderr or "rank      : 1" in p.stderr or p.stderr.count("No HIP GPUs") >= 2
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            assert json.loads(line).get("n_gpus") == 2


def test_rank_count_mismatch_is_an_error():
    """WORLD_SIZE from a launcher that disagrees with --gpus must not turn into a normal-looking line."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
    assert not any(l.startswith("{") for l in p.stdout.splitlines())
