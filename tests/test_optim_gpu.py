"""GPU parity of the optimiser half of the training step (cli/train.py:262-268 / cli/baseline.py:
the same lines: ``clip_grad_norm_(parameters, FLAGS.gradclip)`` then ``optim.Adam.step()``):

  * ``adam_kernel`` (csrc/elementwise.hip) vs ``torch.optim.Adam`` on the CPU, 3 steps, fp32;
  * ``grad_clip_coef`` vs ``torch.nn.utils.clip_grad_norm_``;
  * ONE ``TrainEngine.train_step`` with ``gradclip`` set vs the CPU oracle end to end:
    restated log-mel -> float64 oracle model/loss gradients -> clip_grad_norm_ -> torch Adam.
"""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _flat_module(sizes, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    m = torch.nn.Module()
    for i, n in enumerate(sizes):
        m.register_parameter("p%d" % i, torch.nn.Parameter(torch.randn(n, generator=g)))
    return m


@pytest.mark.parametrize("wd,scale", [(0.0, 1.0), (0.01, 0.5)])
def test_adam_kernel_matches_torch_adam_three_steps(hip_lib, wd, scale):
    from edgedict_amd.optim import FusedAdam
    sizes = [(1000003,), (17, 33), (64,), (5, 7, 3)]
    ref_m = _flat_module(sizes, 0)
    dev_m = _flat_module(sizes, 0).cuda()
    ref = torch.optim.Adam(ref_m.parameters(), lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    opt = FusedAdam(dev_m, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    g = torch.Generator(device="cpu").manual_seed(1)
    for step in range(3):
        opt.zero_grad()
        for pr, pd in zip(ref_m.parameters(), dev_m.parameters()):
            gr = torch.randn(pr.shape, generator=g) * (10.0 ** (step - 1))
            pr.grad = gr * scale              # torch sees the already-scaled gradient
            pd.grad.copy_(gr)                 # the kernel applies grad_scale itself
        ref.step()
        opt.step(grad_scale=scale)
        for (n, pr), pd in zip(ref_m.named_parameters(), dev_m.parameters()):
            a, b = pd.detach().cpu().double(), pr.detach().double()
            err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
            assert err < 1e-6, (step, n, err)
    sd = opt.state_dict()["state"]
    for i, pr in enumerate(ref_m.parameters()):
        st = ref.state[pr]
        # m = b1 m + (1-b1) g cancels for some elements: bound relative to the tensor's scale
        for key in ("exp_avg", "exp_avg_sq"):
            want = st[key].numpy()
            got = sd[i][key].cpu().numpy().reshape(want.shape)
            err = np.abs(got - want).max() / np.abs(want).max()
            assert err < 1e-5, (i, key, err)


@pytest.mark.parametrize("max_norm,gscale", [(10.0, 1.0), (0.5, 1.0), (3.0, 0.25), (1e9, 1.0)])
def test_grad_clip_coef_matches_clip_grad_norm(hip_lib, max_norm, gscale):
    from edgedict_amd.optim import FusedAdam
    sizes = [(300001,), (129, 65), (7,)]
    ref_m = _flat_module(sizes, 3)
    dev_m = _flat_module(sizes, 3).cuda()
    opt = FusedAdam(dev_m, lr=1e-3, max_grad_norm=max_norm)
    g = torch.Generator(device="cpu").manual_seed(4)
    for pr, pd in zip(ref_m.parameters(), dev_m.parameters()):
        gr = torch.randn(pr.shape, generator=g) * 0.05
        pr.grad = gr * gscale
        pd.grad.copy_(gr)
    total = torch.nn.utils.clip_grad_norm_(ref_m.parameters(), max_norm)
    ref = torch.optim.Adam(ref_m.parameters(), lr=1e-3)
    ref.step()
    opt.step(grad_scale=gscale)
    torch.cuda.synchronize()
    assert abs(opt.grad_norm.item() - total.item()) / total.item() < 1e-5
    want = min(1.0, max_norm / (total.item() + 1e-6))
    assert abs(opt._coef.item() - want) / want < 1e-5
    for pr, pd in zip(ref_m.parameters(), dev_m.parameters()):
        assert (pd.detach().cpu() - pr.detach()).abs().max().item() < 2e-6   # lr-sized updates


def _flags(gradclip):
    return types.SimpleNamespace(
        downsample=3, win_length=320, hop_length=160, n_fft=512, feature_size=80, dither=0.0,
        sample_rate=16000, lr=2e-3, gradclip=gradclip, sub_batch_size=None, bpe_size=40,
        vocab_embed_size=8, enc_hidden_size=32, enc_layers=3, enc_dropout=0.0, enc_proj_size=24,
        dec_hidden_size=16, dec_layers=2, dec_dropout=0.0, dec_proj_size=16, joint_size=32,
        enc_time_reductions=[1], delta=False, T_mask=0, T_num_mask=0, F_mask=0, F_num_mask=0)


@pytest.mark.parametrize("gradclip,sub", [(0.05, None), (None, None), (0.05, 2)])
def test_train_step_matches_oracle_gradients_plus_torch_adam(hip_lib, gradclip, sub):
    """cli/baseline.py:214-248: sub-batch loop (loss / n_sub), backward, clip, Adam - against the CPU
    oracle (float64 gradients) followed by clip_grad_norm_ + torch.optim.Adam in fp32."""
    from edgedict_amd.trainer import TrainEngine
    from oracle import features_ref as Fr
    from oracle import models_ref as M
    from oracle import rnnt_loss_ref as R
    fl = _flags(gradclip)
    fl.sub_batch_size = sub
    torch.manual_seed(0)
    eng = TrainEngine(fl, vocab_size=40, device="cuda", compute_dtype="fp32")
    sd0 = {k: v.detach().cpu().clone() for k, v in eng.model.state_dict().items()}
    g = torch.Generator(device="cpu").manual_seed(5)
    B, N = 4, 9600                      # equal lengths: the oracle front-end has no length input
    wave = 0.1 * torch.randn(B, N, generator=g)
    ys = torch.randint(4, 40, (B, 6), generator=g, dtype=torch.int32)
    ylen = torch.tensor([6, 4, 5, 6], dtype=torch.int32)
    loss = eng.train_step(wave.cuda(), None, ys.cuda(), ylen)
    torch.cuda.synchronize()
    # ---- oracle side
    params = {k: torch.nn.Parameter(v.double()) for k, v in sd0.items()}
    starts = list(range(0, B, sub or B))
    total = 0.0
    for s in starts:
        e = min(B, s + (sub or B))
        xs = Fr.stacked_features(wave[s:e], 3, True, win_length=320, hop_length=160, n_fft=512, n_filt=80)
        xlen = torch.full((e - s,), xs.shape[1], dtype=torch.int32)
        logits, act = M.transducer_logits(params, xs.double(), ys[s:e], xlen, ylen[s:e])
        yl = ylen[s:e]
        costs, dl = R.rnnt_loss_torch_fast(logits.detach(), ys[s:e, :int(yl.max())], act, yl)
        logits.backward(dl / (e - s) / len(starts))
        total += float(costs.mean()) / len(starts)
    assert abs(loss.item() - total) / total < 1e-5
    ref_params = [torch.nn.Parameter(sd0[k].clone()) for k in sd0]
    for p, k in zip(ref_params, sd0):
        p.grad = params[k].grad.float()
    if gradclip is not None:
        norm = torch.nn.utils.clip_grad_norm_(ref_params, gradclip)
        assert norm.item() > gradclip        # the clip is active in this case
        assert abs(eng.optim.grad_norm.item() - norm.item()) / norm.item() < 1e-3
    opt = torch.optim.Adam(ref_params, lr=fl.lr)
    opt.step()
    new = eng.model.state_dict()
    for p, k in zip(ref_params, sd0):
        got = new[k].detach().cpu()
        # the first Adam step moves an element by lr * g / (|g| + eps): sign-like, so the comparison
        # is well conditioned only where |g| is not tiny against the engine's gradient error
        # (<= 2e-3 of the tensor's max, tests/test_models_gpu.py); elsewhere bound it by 2 lr
        du_ref = p.detach() - sd0[k]
        du = got - sd0[k]
        big = p.grad.abs() > 1e-2 * max(p.grad.abs().max().item(), 1e-30)
        assert ((du - du_ref).abs()[big] <= 0.02 * fl.lr).all(), (k, (du - du_ref).abs()[big].max().item())
        assert (du - du_ref).abs().max().item() <= 2.0 * fl.lr * (1 + 1e-3), k
        zero = p.grad == 0                      # e.g. the PAD embedding row: no update at all
        assert (du[zero] == 0).all(), k


def test_adam_step_is_skipped_on_the_device_when_a_give_up_word_is_set(hip_lib):
    """Advisor r2: a step kernel of the encoder stack that gives up its bounded flag wait computes on a chunk
    product that does not exist yet, and the trainer applied those gradients before anything raised.  The
    trainer now hands the device-visible give-up words to the Adam kernel (edgedict_adam_step_guarded): with
    a word set the update is skipped on the device (no host sync), the host raises at its next check and the
    word is cleared by that read."""
    import ctypes
    from edgedict_amd import encoder_stack
    from edgedict_amd.optim import FusedAdam
    m = _flat_module([(4099,), (33, 7)], 3).cuda()
    opt = FusedAdam(m, lr=1e-2)
    dev_words = hip_lib.edgedict_stack_error_words(0)
    host_words = hip_lib.edgedict_stack_error_words(1)
    assert dev_words and host_words
    encoder_stack.check_wsr_error()              # clear whatever an earlier test left
    words = (ctypes.c_uint * 3).from_address(host_words)
    opt.zero_grad()
    for p in m.parameters():
        p.grad.fill_(0.5)
    before = opt.flat.data.clone()
    words[2] = 503                               # "forward pass, launch slot 3 gave up"
    opt.step(guard=dev_words)
    torch.cuda.synchronize()
    assert torch.equal(opt.flat.data, before)
    assert float(opt.m.abs().max()) == 0.0 and float(opt.v.abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="gave up"):
        encoder_stack.check_wsr_error()
    assert words[2] == 0
    opt.step(guard=dev_words)                    # the word is clear again: the update happens
    torch.cuda.synchronize()
    assert not torch.equal(opt.flat.data, before)
