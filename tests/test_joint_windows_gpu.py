"""GPU: the joint network's backward pass in TIME WINDOWS (models._JointLossWinFn, window-major packed lattice, the
encoder stack reading a gradient that arrives window by window) against the one-pass node (models._JointLossFn).

Reference: rnnt/models.py:135 (encoder.proj), 169-179 (Joint.forward), 221,238 (the loss call) and their autograd."""
import ctypes
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------- kernel level
def _lattice(B, T, U1, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    al = torch.randint(1, T + 1, (B,), generator=g)
    ll = torch.randint(0, U1, (B,), generator=g)
    al[0], ll[0] = T, U1 - 1
    return al, ll


def _offsets(al, ll, bounds):
    nw = len(bounds) - 1
    lo = torch.tensor(bounds[:-1])[:, None]
    hi = torch.tensor(bounds[1:])[:, None]
    rows = (torch.minimum(al[None, :], hi) - lo).clamp_(min=0) * (ll[None, :] + 1)
    off = torch.zeros(nw * len(al), dtype=torch.int64)
    off[1:] = torch.cumsum(rows.reshape(-1), 0)[:-1]
    return off.view(nw, len(al)), int(rows.sum())


def _perm(al, ll, bounds):
    """perm[r_window_major] = r_utterance_major for every cell of the packed lattice."""
    off1, M = _offsets(al, ll, [0, bounds[-1]])
    offw, Mw = _offsets(al, ll, bounds)
    assert M == Mw
    perm = torch.empty(M, dtype=torch.int64)
    for w in range(len(bounds) - 1):
        for b in range(len(al)):
            W = int(ll[b]) + 1
            for t in range(bounds[w], min(int(al[b]), bounds[w + 1])):
                dst = int(offw[w, b]) + (t - bounds[w]) * W
                src = int(off1[0, b]) + t * W
                perm[dst:dst + W] = torch.arange(src, src + W)
    return perm, off1[0], offw, M


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_window_major_kernels_equal_the_one_pass_kernels_row_by_row(hip_lib, dtype):
    """joint_hidden fwd / bwd and the loss gradient, window by window on the window-major lattice, against the one-pass
    entry points on the utterance-major lattice: the same cells, bit for bit (dD1: atomics, to rounding)."""
    from edgedict_amd import _lib
    lib = _lib.load()
    B, T, U1, J, V = 5, 37, 7, 64, 256
    bounds = [0, 8, 24, 37]
    al, ll = _lattice(B, T, U1, 3)
    perm, off1, offw, M = _perm(al, ll, bounds)
    code = _lib.dtype_code(dtype)
    g = torch.Generator(device="cpu").manual_seed(4)
    E1 = torch.randn(B, T, J, generator=g).to(dtype).cuda()
    D1 = torch.randn(B, U1, J, generator=g).to(dtype).cuda()
    al_d, ll_d = al.int().cuda(), ll.int().cuda()
    off1_d, offw_d = off1.cuda(), offw.cuda()
    perm_d = perm.cuda()
    # ---- tanh(E1 + D1)
    hid1 = torch.zeros(M, J, dtype=dtype, device="cuda")
    hidw = torch.zeros(M, J, dtype=dtype, device="cuda")
    _lib.call("joint_hidden_fwd_packed", code, E1, D1, hid1, al_d, ll_d, off1_d, B, T, U1, J)
    for w in range(len(bounds) - 1):
        _lib.call("joint_hidden_fwd_packed_win", code, E1, D1, hidw, al_d, ll_d, offw_d[w], B, T, U1, J,
                  bounds[w], bounds[w + 1])
    assert torch.equal(hidw, hid1[perm_d])
    # ---- loss forward (separate-pass entry on the utterance-major rows) + gradient, one pass vs windows
    logits1 = torch.randn(M, V, generator=g).to(dtype).cuda()
    logitsw = logits1[perm_d].contiguous()
    labels = torch.randint(1, V, (B, U1 - 1), generator=g, dtype=torch.int32).cuda()
    ws = torch.zeros(lib.edgedict_rnnt_workspace_bytes(B, T, U1), dtype=torch.uint8, device="cuda")
    costs, red = torch.empty(B, device="cuda"), torch.empty(1, device="cuda")
    _lib.call("rnnt_loss_forward_packed", logits1, code, labels, al_d, ll_d, off1_d, B, T, U1, V, 0, costs, red,
              1.0 / B, ws)
    dl1 = torch.zeros_like(logits1)
    _lib.call("rnnt_loss_backward_packed", logits1, code, dl1, labels, al_d, ll_d, off1_d, B, T, U1, V, 0, ws,
              1.0 / B, None, 0)
    dlw = torch.full_like(logitsw, 7.0)
    for w in range(len(bounds) - 1, -1, -1):
        _lib.call("rnnt_loss_backward_packed_win", logitsw, code, dlw, labels, al_d, ll_d, offw_d[w], B, T, U1, V, 0,
                  ws, 1.0 / B, None, 0, bounds[w], bounds[w + 1])
    assert torch.equal(dlw, dl1[perm_d])
    # ---- tanh backward: dE1 (f32 batch-first + the time-major copy in the lattice's dtype), dD1 accumulated
    dhid1 = torch.randn(M, J, generator=g).to(dtype).cuda()
    dhidw = dhid1[perm_d].contiguous()
    dE1a = torch.empty(B, T, J, device="cuda")
    dD1a = torch.empty(B, U1, J, device="cuda")
    _lib.call("joint_hidden_bwd_packed", code, dhid1, hid1, dE1a, dD1a, al_d, ll_d, off1_d, B, T, U1, J)
    dE1b = torch.full((B, T, J), 7.0, device="cuda")
    dD1b = torch.zeros(B, U1, J, device="cuda")
    tm = torch.full((T, B, J), 7.0, dtype=torch.bfloat16, device="cuda")
    for w in range(len(bounds) - 1, -1, -1):
        _lib.call("joint_hidden_bwd_packed_win", code, dhidw, hidw, dE1b, dD1b, tm, al_d, ll_d, offw_d[w], B, T, U1, J,
                  bounds[w], bounds[w + 1], 0)
    assert torch.equal(dE1b, dE1a)
    assert torch.equal(tm, dE1a.to(torch.bfloat16).transpose(0, 1).contiguous())
    assert (dD1b - dD1a).abs().max().item() <= 1e-5 * max(1.0, dD1a.abs().max().item())


def test_window_major_loss_forward_from_partials_equals_the_one_pass_loss(hip_lib):
    """gemm_nt_lse on window-major rows + edgedict_rnnt_loss_forward_packed_parts_win: costs bit-identical to the
    utterance-major one-pass loss (row order in memory is the only difference)."""
    from edgedict_amd import _lib
    lib = _lib.load()
    B, T, U1, J, V = 4, 50, 9, 128, 512
    bounds = [0, 16, 32, 50]
    al, ll = _lattice(B, T, U1, 8)
    perm, off1, offw, M = _perm(al, ll, bounds)
    g = torch.Generator(device="cpu").manual_seed(9)
    hid1 = torch.tanh(torch.randn(M, J, generator=g)).to(torch.bfloat16).cuda()
    W2 = (0.2 * torch.randn(V, J, generator=g)).to(torch.bfloat16).cuda()
    b2 = (0.1 * torch.randn(V, generator=g)).cuda()
    labels = torch.randint(1, V, (B, U1 - 1), generator=g, dtype=torch.int32).cuda()
    al_d, ll_d = al.int().cuda(), ll.int().cuda()
    slots = (V + 63) // 64
    out = []
    for hid, offs, win in ((hid1, off1.cuda(), None), (hid1[perm.cuda()].contiguous(), offw.cuda(), bounds)):
        logits = torch.empty(M, V, dtype=torch.bfloat16, device="cuda")
        parts = torch.empty(M, slots, 2, device="cuda")
        _lib.call("gemm_nt_lse", hid, ctypes.c_longlong(J), W2, ctypes.c_longlong(J), logits, ctypes.c_longlong(V), M, V, J,
                  b2, parts)
        ws = torch.zeros(lib.edgedict_rnnt_workspace_bytes(B, T, U1), dtype=torch.uint8, device="cuda")
        costs, red = torch.empty(B, device="cuda"), torch.empty(1, device="cuda")
        if win is None:
            _lib.call("rnnt_loss_forward_packed_parts", logits, labels, al_d, ll_d, offs, B, T, U1, V, 0, costs, red,
                      1.0 / B, ws, parts, slots)
        else:
            t0s = (ctypes.c_int * len(win))(*win)
            _lib.call("rnnt_loss_forward_packed_parts_win", logits, labels, al_d, ll_d, offs, t0s, len(win) - 1, B, T, U1,
                      V, 0, costs, red, 1.0 / B, ws, parts, slots)
        out.append((costs.cpu(), red.cpu(), logits))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[1][2], out[0][2][perm.cuda()])
    with pytest.raises(RuntimeError):           # windows that do not cover [0, T)
        t0s = (ctypes.c_int * 3)(0, 16, 32)
        _lib.call("rnnt_loss_forward_packed_parts_win", logits, labels, al_d, ll_d, offw.cuda(), t0s, 2, B, T, U1, V, 0,
                  costs, red, 1.0 / B, ws, parts, slots)


# ---------------------------------------------------------------------------------------------- model level
def _flags(H=128, L=3):
    return types.SimpleNamespace(
        downsample=3, win_length=320, hop_length=160, n_fft=512, feature_size=80, dither=0.0,
        sample_rate=16000, lr=1e-3, gradclip=None, sub_batch_size=None, bpe_size=256,
        vocab_embed_size=16, enc_hidden_size=H, enc_layers=L, enc_dropout=0.0, enc_proj_size=96,
        dec_hidden_size=64, dec_layers=2, dec_dropout=0.0, dec_proj_size=64, joint_size=128,
        enc_time_reductions=[1], delta=False)


def _step(windows, seed=0, serial=False, defer=True):
    from edgedict_amd import config, encoder_stack, ops
    from edgedict_amd.trainer import TrainEngine
    old = (config.JOINT_BWD_WINDOWS, config.DEFER_WEIGHT_GRADS, encoder_stack.FLAGS)
    config.JOINT_BWD_WINDOWS, config.DEFER_WEIGHT_GRADS = windows, defer
    if serial:
        encoder_stack.FLAGS = encoder_stack.FLAGS | encoder_stack.SERIAL
    try:
        torch.manual_seed(seed)
        eng = TrainEngine(_flags(), vocab_size=256, device="cuda", compute_dtype="bf16")
        g = torch.Generator(device="cpu").manual_seed(seed + 1)
        N = 105600                      # 6.6 s -> 661 frames -> 221 stacked -> 111 encoder frames
        wave = (0.1 * torch.randn(6, N, generator=g)).cuda()
        wlen = torch.tensor([N, N - 9000, N - 30000, N, N - 52000, N - 70000], dtype=torch.int32)
        ys = torch.randint(4, 256, (6, 12), generator=g, dtype=torch.int32).cuda()
        ylen = torch.tensor([12, 9, 12, 3, 7, 10], dtype=torch.int32)
        ops.LAST.pop("joint_windows", None)
        loss = eng.train_step(wave, wlen, ys, ylen)
        torch.cuda.synchronize()
        encoder_stack.check_wsr_error()
        return loss.item(), eng.flat.grad.clone(), ops.LAST.get("joint_windows"), eng
    finally:
        config.JOINT_BWD_WINDOWS, config.DEFER_WEIGHT_GRADS, encoder_stack.FLAGS = old


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def test_windowed_joint_backward_equals_the_one_pass_step(hip_lib):
    """One full training step (TrainEngine, bf16, wavefront encoder stack): time windows vs one pass.  The loss is
    bit-identical (same per-cell arithmetic); every parameter gradient agrees to the K-order of the products."""
    l1, g1, w1, eng = _step("")
    lw, gw, ww, _ = _step("0.12,0.4,0.7")
    assert w1 is None and ww is not None and len(ww) >= 4 and ww[0] == 0, (w1, ww)
    assert l1 == lw
    assert _rel(gw, g1) < 2e-3, _rel(gw, g1)
    # per parameter: nothing may be left out (a window whose event was never waited for would leave its frames' share
    # of the encoder gradients wrong, which a flat-norm bound over 1e6 numbers could hide)
    for p, off in zip(eng.flat.params, eng.flat.offsets):
        a, b = gw[off:off + p.numel()], g1[off:off + p.numel()]
        if b.norm().item() > 0:
            assert _rel(a, b) < 2e-2, (tuple(p.shape), _rel(a, b))


def test_windowed_joint_backward_is_schedule_independent(hip_lib):
    """Two windows, five windows, the serial debug schedule of the stack, weight gradients through autograd instead of
    the auxiliary stream: all the same step."""
    l1, g1, _, _ = _step("")
    for spec, kw in (("0.5", {}), ("0.1,0.25,0.5,0.75", {}), ("0.12,0.4,0.7", dict(serial=True)),
                     ("0.12,0.4,0.7", dict(defer=False))):
        lw, gw, ww, _ = _step(spec, **kw)
        assert ww is not None, spec
        assert l1 == lw, (spec, kw)
        assert _rel(gw, g1) < 2e-3, (spec, kw, _rel(gw, g1))
