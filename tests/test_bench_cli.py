"""bench.py must never report a smaller job than the one asked for (cli/lightning.py:325-331 is the
reference's 4-GPU launch; the driver runs `bench.py --gpus N` at N = 1, 2, 4, 8)."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env,
                          capture_output=True, text=True, timeout=300)


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def test_more_gpus_than_devices_is_an_error_not_a_one_gpu_result():
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(max(2, n + 1)), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert not _json_lines(r.stdout), r.stdout
    assert "requested" in (r.stderr + r.stdout)


def test_launcher_world_size_must_match_gpus_flag():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"],
             env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0
    assert not _json_lines(r.stdout)
    assert "WORLD_SIZE" in (r.stderr + r.stdout)
