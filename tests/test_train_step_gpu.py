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


def test_front_end_prefetch_gives_the_same_loss_sequence(hip_lib):
    """train_step(..., next_batch=...) computes the next batch's front-end (dither -> log-mel -> stacking -> SpecAugment) on
    the auxiliary stream under this step's encoder forward (the reference's DataLoader workers do the same on the CPU,
    rnnt/dataset.py:102-103): the losses of 4 steps over 3 different batches - dither and SpecAugment ON, so the features
    depend on the per-call seeds and mask draws - are bit-identical to the serial order; a next_batch that is NOT the
    batch of the next call is discarded, and the parameters after the steps agree."""
    from edgedict_amd.trainer import TrainEngine

    def run(prefetch):
        fl = _flags()
        fl.dither = 1e-3
        fl.T_mask, fl.T_num_mask, fl.F_mask, fl.F_num_mask = 5, 2, 6, 2
        torch.manual_seed(3)
        eng = TrainEngine(fl, vocab_size=40, device="cuda", compute_dtype="bf16")
        g = torch.Generator(device="cpu").manual_seed(11)
        batches = []
        for i in range(3):
            wave = (0.1 * torch.randn(5, 16000, generator=g)).cuda()
            wlen = torch.tensor([16000, 15000 - 500 * i, 12000, 16000, 9000 + 300 * i], dtype=torch.int32)
            ys = torch.randint(4, 40, (5, 7), generator=g, dtype=torch.int32).cuda()
            ylen = torch.tensor([7, 5, 6, 3, 7], dtype=torch.int32)
            batches.append((wave, wlen, ys, ylen))
        order = [0, 1, 2, 1]
        losses = []
        import random
        random.seed(5)
        torch.manual_seed(17)          # the SpecAugment draws
        for i, b in enumerate(order):
            nxt = None
            if prefetch and i + 1 < len(order):
                # the third call announces the WRONG batch: its prefetched features must be thrown away
                nb = batches[order[i + 1]] if i != 2 else batches[0]
                nxt = (nb[0], nb[1])
            losses.append(eng.train_step(*batches[b], next_batch=nxt))
        torch.cuda.synchronize()
        eng.check()
        out = [x.item() for x in losses], eng.flat.data.clone()
        eng.close()
        return out

    serial, p_serial = run(False)
    pre, p_pre = run(True)
    assert all(x == x and x > 0 for x in serial)
    # steps 0-2: every batch's front-end ran once, in batch order -> identical features -> identical losses
    assert pre[:3] == serial[:3], (pre, serial)
    # step 3: the discarded prefetch consumed one dither seed / one mask draw more - same batch, another augmentation
    assert abs(pre[3] - serial[3]) / serial[3] < 0.2
    assert torch.isfinite(p_pre).all()
