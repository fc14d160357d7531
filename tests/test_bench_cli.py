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


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_world2_body_of_bench_runs_on_one_gpu_over_gloo():
    """Every line of bench.py's `world > 1` body (process-group init, parameter broadcast, the bucketed
    exchange leaving from inside the backward pass, barrier + max-over-ranks timing, the rank/device census,
    the second timed pass with the exchange after the backward pass, the `exchange` record) executes here
    before an 8-GPU driver runs it for the first time: two ranks, both on device 0
    (EDGEDICT_BENCH_SHARE_DEVICE=1), gloo transport instead of RCCL (a one-GPU box cannot host a 2-rank RCCL
    communicator), launched exactly as the driver launches N > 1 - through torch.distributed.run."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(EDGEDICT_BENCH_BACKEND="gloo", EDGEDICT_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--seconds", "5", "--labels", "20"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=840)
    if r.returncode != 0:                                  # keep the ranks' own words (pytest truncates the assert)
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, "bench_world2_failure.txt"), "w") as fh:
                fh.write(r.stdout[-20000:] + "\n=== stderr ===\n" + r.stderr[-40000:])
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]              # rank 0 only
    out = lines[0]
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["rank_devices"] == [0, 0]
    assert out["config"]["global_batch"] == 32 and out["scaling"] == "weak"
    ex = out["exchange"]
    assert ex["backend"] == "gloo" and ex["buckets"] >= 8 and ex["bytes"] > 2e8
    # the joint's and the encoder layers' buckets left from inside the backward pass (grads_final path)
    assert ex["left_during_backward"] >= 7, ex
    assert ex["ms_per_step_overlap"] > 0 and ex["ms_per_step_after_backward"] > 0
    # per-rank figures (the first real multi-GPU run must be diagnosable from this one line): the reported time is
    # the slowest rank's, every rank sent the same buckets early
    assert len(ex["rank_ms_per_step"]) == 2 and abs(max(ex["rank_ms_per_step"]) - out["ms_per_step"]) < 1e-3
    assert ex["rank_ms_per_step_spread"] >= 0 and len(ex["rank_host_enqueue_ms_per_step"]) == 2
    assert ex["rank_left_during_backward"] == [ex["left_during_backward"]] * 2
    # `value` is the DEFAULT exchange mode (what TrainEngine ships: overlapped), never the faster of the two
    assert ex["overlap_default"] is True and ex["mode_reported"] == "overlap"
    assert abs(out["ms_per_step"] - ex["ms_per_step_overlap"]) < 1e-6
    assert abs(out["value"] - 32 * 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
    assert out["value"] > 0 and out["roofline"] is not None
