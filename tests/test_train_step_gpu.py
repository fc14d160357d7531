"""GPU: one full training step of the engine (features -> Transducer -> loss -> backward -> Adam,
the loop of cli/baseline.py:214-248) must not depend on HOW the work is scheduled: weight
gradients accumulated on the auxiliary stream (side.py) vs returned through autograd, and the
layer-pipelined bf16 encoder stack vs the per-layer path."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _flags():
    return types.SimpleNamespace(
        downsample=3, win_length=320, hop_length=160, n_fft=512, feature_size=80, dither=0.0,
        sample_rate=16000, lr=1e-3, gradclip=None, sub_batch_size=None, bpe_size=40,
        vocab_embed_size=8, enc_hidden_size=64, enc_layers=3, enc_dropout=0.0, enc_proj_size=24,
        dec_hidden_size=32, dec_layers=2, dec_dropout=0.0, dec_proj_size=16, joint_size=32,
        enc_time_reductions=[1], delta=False)


def _step(defer, stack, dtype="bf16", seed=0):
    from edgedict_amd import config
    from edgedict_amd.trainer import TrainEngine
    old = (config.DEFER_WEIGHT_GRADS, config.USE_ENCODER_STACK)
    config.DEFER_WEIGHT_GRADS, config.USE_ENCODER_STACK = defer, stack
    try:
        torch.manual_seed(seed)
        eng = TrainEngine(_flags(), vocab_size=40, device="cuda", compute_dtype=dtype)
        g = torch.Generator(device="cpu").manual_seed(seed + 1)
        wave = (0.1 * torch.randn(5, 16000, generator=g)).cuda()
        wlen = torch.tensor([16000, 15000, 12000, 16000, 9000], dtype=torch.int32).cuda()
        ys = torch.randint(4, 40, (5, 7), generator=g, dtype=torch.int32).cuda()
        ylen = torch.tensor([7, 5, 6, 3, 7], dtype=torch.int32).cuda()
        loss = eng.train_step(wave, wlen, ys, ylen)
        torch.cuda.synchronize()
        return loss.item(), eng.flat.grad.clone(), eng.flat.data.clone()
    finally:
        config.DEFER_WEIGHT_GRADS, config.USE_ENCODER_STACK = old


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_deferred_weight_gradients_equal_autograd_path(hip_lib, dtype):
    la, ga, pa = _step(True, True, dtype)
    lb, gb, pb = _step(False, True, dtype)
    assert la == lb
    scale = gb.abs().max().item()
    assert (ga - gb).abs().max().item() <= 2e-5 * scale      # fp32 atomics reorder sums only
    assert (pa - pb).abs().max().item() <= 1e-6 + 1e-4 * 1e-3  # one Adam step of lr 1e-3


def test_stack_and_per_layer_steps_agree_in_bf16(hip_lib):
    la, ga, _ = _step(True, True)
    lb, gb, _ = _step(True, False)
    assert abs(la - lb) / abs(lb) < 2e-2
    cos = torch.nn.functional.cosine_similarity(ga.double(), gb.double(), dim=0).item()
    assert cos > 0.99, cos


def test_host_side_lengths_give_the_same_step(hip_lib):
    """Lengths handed over on the host (as a DataLoader does, rnnt/dataset.py:225-240) must give
    the same step as device-resident lengths; they only remove the xlen.max() device sync."""
    from edgedict_amd.trainer import TrainEngine
    res = []
    for host in (False, True):
        torch.manual_seed(0)
        eng = TrainEngine(_flags(), vocab_size=40, device="cuda", compute_dtype="fp32")
        g = torch.Generator(device="cpu").manual_seed(1)
        wave = (0.1 * torch.randn(4, 16000, generator=g)).cuda()
        wlen = torch.tensor([16000, 15000, 12000, 9000], dtype=torch.int32)
        ys = torch.randint(4, 40, (4, 7), generator=g, dtype=torch.int32).cuda()
        ylen = torch.tensor([7, 5, 6, 3], dtype=torch.int32)
        if not host:
            wlen, ylen = wlen.cuda(), ylen.cuda()
        loss = eng.train_step(wave, wlen, ys, ylen)
        torch.cuda.synchronize()
        res.append((loss.item(), eng.flat.grad.clone()))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1]) or \
        (res[0][1] - res[1][1]).abs().max().item() <= 2e-5 * res[0][1].abs().max().item()


def test_checkpoint_round_trip_in_the_reference_layout(hip_lib, tmp_path):
    """save() writes {'optim', 'model', 'sched'} as cli/train.py:321-336 does; a fresh engine that
    loads it continues bit-identically, and the 'optim' entry loads into torch.optim.Adam."""
    from edgedict_amd.trainer import TrainEngine
    fl = _flags()
    fl.sched, fl.warmup_step = True, 4

    def batch(seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        wave = (0.1 * torch.randn(3, 12000, generator=g)).cuda()
        ys = torch.randint(4, 40, (3, 5), generator=g, dtype=torch.int32).cuda()
        return wave, torch.tensor([12000, 9000, 11000], dtype=torch.int32), ys, \
            torch.tensor([5, 3, 4], dtype=torch.int32)

    torch.manual_seed(0)
    a = TrainEngine(fl, vocab_size=40, device="cuda", compute_dtype="fp32")
    for s in (1, 2):
        a.train_step(*batch(s))
    a.validation_end(3.0)
    path = str(tmp_path / "2.pt")
    a.save(path)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) >= {"optim", "model", "sched"} and "encoder.lstm.lstms.0.weight_ih_l0" in ck["model"]
    ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros_like(p)) for p in a.flat.params], lr=1.0)
    ref.load_state_dict(ck["optim"])
    torch.manual_seed(123)                                   # different initial weights on purpose
    b = TrainEngine(fl, vocab_size=40, device="cuda", compute_dtype="fp32")
    b.load(path)
    assert b.step_count == 2 and b.optim.step_count == 2 and b.sched.best == 3.0
    la, lb = a.train_step(*batch(3)), b.train_step(*batch(3))
    torch.cuda.synchronize()
    assert la.item() == lb.item()
    assert (a.flat.data - b.flat.data).abs().max().item() <= 1e-7
    assert abs(a.optim.param_groups[0]["lr"] - fl.lr * 3 / 4) < 1e-12   # warm-up: step 3 of 4
