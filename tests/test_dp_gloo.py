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


def _worker_ready(rank, world, port, q):
    """Parameters accumulated behind autograd's back report through ready(): their buckets leave
    before finish(), cut at the boundaries passed in, and the reduced gradient is still exact."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edgedict_amd.dp import BucketedAllReduce
    from edgedict_amd.optim import FlatParams
    model = _model()
    flat = FlatParams(model)
    params = list(model.parameters())
    red = BucketedAllReduce(flat, bucket_bytes=1 << 30, boundaries=[params[0], params[2], params[4]], min_bytes=4)
    assert len(red.bounds) == 3                       # one bucket per Linear, cut at its first tensor
    g = torch.Generator().manual_seed(5 + rank)
    flat.zero_grad()
    for p in params:                                   # "kernels" wrote the gradients in place
        p.grad.copy_(torch.randn(p.shape, generator=g))
    mine = flat.grad.clone()
    red.ready(params[4:6])                             # last layer first, as backward finishes things
    red.ready(params[2:3])                             # half a bucket: must not leave yet
    early = red.issued_early
    red.ready(params[3:4])
    early2 = red.issued_early
    scale = red.finish()                               # flushes the first Linear's bucket
    q.put((rank, mine.numpy(), flat.grad.clone().numpy(), early, early2, red.last_issued_early, scale))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_in_place_gradients_report_ready_and_leave_early():
    world, port = 2, 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_ready, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        r = q.get(timeout=60)
        out[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    import numpy as np
    total = out[0][0] + out[1][0]
    for r in range(world):
        np.testing.assert_allclose(out[r][1], total, rtol=1e-6, atol=1e-6)
        assert out[r][2:] == (1, 2, 2, 0.5)


def _worker_order(rank, world, port, q):
    """Ranks complete their buckets in DIFFERENT orders (rank 1 the other way round, as a rank on the
    per-layer autograd path would) and one bucket holds a `late` parameter: every rank must still issue
    the same collectives in the same order - a bucket that completes early is held back - and the sum
    must be exact (a mismatched order pairs buffers of different sizes: wrong sums or a hang)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edgedict_amd.dp import BucketedAllReduce
    from edgedict_amd.optim import FlatParams
    model = _model()
    flat = FlatParams(model)
    params = list(model.parameters())
    red = BucketedAllReduce(flat, bucket_bytes=1 << 30, boundaries=[params[0], params[2], params[4]],
                            min_bytes=4, late=[params[3]])
    assert len(red.bounds) == 3
    # bucket 0 = last Linear, 1 = middle (late: moved to the end), 2 = first
    assert red.issue_order == [0, 2, 1]
    issued = []
    orig = red._issue

    def spy(b):
        issued.append(b)
        orig(b)
    red._issue = spy
    g = torch.Generator().manual_seed(9 + rank)
    flat.zero_grad()
    for p in params:
        p.grad.copy_(torch.randn(p.shape, generator=g))
    mine = flat.grad.clone()
    groups = [params[4:6], params[2:4], params[0:2]]
    if rank == 1:
        groups = groups[::-1]
    held = []
    for grp in groups:
        red.ready(grp)
        held.append(list(issued))
    scale = red.finish()
    q.put((rank, mine.numpy(), flat.grad.clone().numpy(), issued, held, scale))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_issue_order_is_rank_invariant():
    world, port = 2, 33500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_order, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        r = q.get(timeout=60)
        out[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    import numpy as np
    total = out[0][0] + out[1][0]
    for r in range(world):
        np.testing.assert_allclose(out[r][1], total, rtol=1e-6, atol=1e-6)
        assert out[r][2] == [0, 2, 1]                 # the same order on both ranks
    # rank 0 completes 0, 1 (late: held), 2 -> [0], [0], [0, 2, 1];  rank 1 completes 2 (held behind 0), 1, 0
    assert out[0][3] == [[0], [0], [0, 2, 1]]
    assert out[1][3] == [[], [], [0, 2, 1]]


def _worker_ragged(rank, world, port, q):
    """Per-rank ragged T: every rank slices its shard to ITS longest utterance (rnnt/models.py:229-230), so the encoder
    stack of rank 0 (T0 = 401) and of rank 1 (T0 = 283) run different launch schedules and report their layers'
    weight gradients final (edgedict_stack_desc_t.grads_final -> BucketedAllReduce.ready) at different launches.  The
    NATIVE scheduler is driven in its dry-run mode (no device) with the real callback wiring; the joint's bucket is
    reported first, the remaining parameters complete through hooks at the end, as in a training step."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from edgedict_amd import encoder_stack as es
    from edgedict_amd.dp import BucketedAllReduce
    from edgedict_amd.optim import FlatParams
    L = 6
    torch.manual_seed(0)
    model = torch.nn.Module()
    model.layers = torch.nn.ModuleList([torch.nn.ParameterList(          # w_ih, w_hh, b_ih, b_hh of a layer, in this order
        [torch.nn.Parameter(torch.zeros(n)) for n in (40, 48, 8, 8)]) for _ in range(L)])
    model.joint = torch.nn.Linear(9, 11)
    model.rest = torch.nn.Linear(5, 3)
    flat = FlatParams(model)
    layer_params = [list(m) for m in model.layers]
    red = BucketedAllReduce(flat, bucket_bytes=1 << 30, min_bytes=4,
                            boundaries=[lp[0] for lp in layer_params] + [model.joint.weight, model.rest.weight],
                            late=list(model.rest.parameters()))
    assert len(red.bounds) == L + 2 and red.issue_order == list(range(1, L + 2)) + [0]
    issued = []
    orig = red._issue

    def spy(b):
        issued.append(b)
        orig(b)
    red._issue = spy
    g = torch.Generator().manual_seed(77 + rank)
    flat.zero_grad()
    for p in flat.params:
        p.grad.copy_(torch.randn(p.shape, generator=g))
    mine = flat.grad.clone()
    T0 = (401, 283)[rank]
    red.ready(list(model.joint.parameters()))                  # the joint's backward node reports first
    reported = []

    def on_final(layer):
        reported.append(layer)
        red.ready(layer_params[layer])
    _, _, n_launches, _ = es.schedule(T0, 240, 1024, [1, 2, 1, 1, 1, 1], backward=True, grads_final=on_final)
    early = list(issued)
    scale = red.finish()
    q.put((rank, mine.numpy(), flat.grad.clone().numpy(), issued, early, reported, n_launches, scale))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_ranks_with_different_launch_counts_issue_the_same_collectives():
    world, port = 2, 35500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        r = q.get(timeout=90)
        out[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    import numpy as np
    total = out[0][0] + out[1][0]
    for r in range(world):
        np.testing.assert_allclose(out[r][1], total, rtol=1e-6, atol=1e-6)
        assert out[r][4] == [5, 4, 3, 2, 1, 0]            # layers become final top-down on every rank
    assert out[0][5] != out[1][5] and out[0][5] == 36       # different launch counts (tests/test_stack_schedule.py pins 36)
    assert out[0][2] == out[1][2]                           # ... the same collectives in the same order
    assert out[0][3] == out[1][3] and len(out[0][3]) == 7   # joint + six layers left before finish(), `rest` from it
