"""CPU, world_size 2, gloo: the bucketed gradient all-reduce (edgedict_amd.dp) turns per-rank
gradients of sharded batches into the gradient of the global batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(12, 33), torch.nn.Tanh(), torch.nn.Linear(33, 7),
                               torch.nn.Tanh(), torch.nn.Linear(7, 1))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edgedict_amd.dp import BucketedAllReduce, shard_batch
    from edgedict_amd.optim import FlatParams
    model = _model()
    flat = FlatParams(model)
    red = BucketedAllReduce(flat, bucket_bytes=4 * 100)     # several small buckets
    assert len(red.bounds) >= 2
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 12, generator=g)
    y = torch.randn(8, 1, generator=g)
    xs, ys = shard_batch([x, y], rank, world)
    results = []
    for it in range(2):                      # second iteration checks the counters re-arm
        flat.zero_grad()
        # two accumulation sub-batches: only the last one triggers the exchange
        red.armed = False
        loss = ((model(xs[:2]) - ys[:2]) ** 2).mean() / 2
        loss.backward()
        red.armed = True
        loss = ((model(xs[2:]) - ys[2:]) ** 2).mean() / 2
        loss.backward()
        scale = red.finish()
        results.append((flat.grad * scale).clone())
    q.put((rank, [r.numpy() for r in results]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_allreduce_world2_matches_global_batch():
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # single-process reference: mean over ranks of per-rank (mean of two sub-batch means)
    sys.path.insert(0, ROOT)
    from edgedict_amd.optim import FlatParams
    model = _model()
    flat = FlatParams(model)
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 12, generator=g)
    y = torch.randn(8, 1, generator=g)
    flat.zero_grad()
    total = 0
    for r in range(world):
        xs, ys = x[r * 4:(r + 1) * 4], y[r * 4:(r + 1) * 4]
        total = total + (((model(xs[:2]) - ys[:2]) ** 2).mean() / 2 +
                         ((model(xs[2:]) - ys[2:]) ** 2).mean() / 2) / world
    total.backward()
    ref = flat.grad.numpy()
    for r in range(world):
        for it in range(2):
            assert abs(out[r][it] - ref).max() < 1e-6
    assert abs(out[0][0] - out[1][0]).max() == 0     # ranks agree bit-for-bit


def test_shard_batch_contiguous_equal_chunks():
    sys.path.insert(0, ROOT)
    from edgedict_amd.dp import shard_batch
    x = torch.arange(10)
    assert shard_batch([x], 0, 4)[0].tolist() == [0, 1, 2]
    assert shard_batch([x], 3, 4)[0].tolist() == [9]


def test_flagfile_reader_and_presets(tmp_path):
    sys.path.insert(0, ROOT)
    from edgedict_amd.flags import make_flags, model_kwargs, read_flagfile
    f = make_flags("E6D2")
    kw = model_kwargs(f)
    assert kw["input_size"] == 240 and kw["enc_hidden_size"] == 1024 and kw["enc_layers"] == 6
    assert kw["dec_layers"] == 2 and kw["joint_size"] == 640 and kw["vocab_size"] == 2048
    p = tmp_path / "ff.txt"
    p.write_text("--enc_hidden_size=256\n--enc_layers=4\n--nodelta\n--apex\n--lr=0.0005\n"
                 "--feature=logfbank\n--bpe_size=2048\n--hop_length=160\n")
    g = read_flagfile(str(p))
    assert g.enc_hidden_size == 256 and g.enc_layers == 4 and g.delta is False
    assert g.apex is True and abs(g.lr - 5e-4) < 1e-12 and g.feature == "logfbank"
    large = make_flags("E6D2_LARGE_Batch")
    assert (large.dec_hidden_size, large.dec_proj_size, large.hop_length) == (512, 640, 320)
