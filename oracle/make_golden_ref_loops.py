#!/usr/bin/env python
"""Generate tests/golden/ref_loops.npz by EXECUTING the reference's own driver loops (oracle/ref_lift.py) over the
REFERENCE's modules on the CPU:

  * ``Trainer.train_step`` of cli/baseline.py:214-248 (the log-mel trainer: sub-batch loop, loss / n_sub, backward,
    ``clip_grad_norm_``, ``optim.step()``) - 3 optimiser steps, ``torch.optim.Adam`` as cli/baseline.py:140-142 builds it;
  * ``Trainer.train_step`` of cli/train.py:223-271 (the FrontEnd trainer: conv front-end, length rescaling,
    ``enc_time_reductions=[]``), Adam over model + front-end parameters as cli/train.py:136-138 builds it;
  * ``stream_decode`` of cli/openvino_wav_inference.py:29-46 (the chunk loop) over the reference's
    ``PytorchStreamDecoder`` (rnnt/stream.py:78-120, as oracle/make_golden_stream.py instantiates it);
  * the microphone ``callback`` of stream.py:71-99 (two-block buffer, /65536 scaling, reset after 35 blank chunks).

What the loops import and this container lacks is supplied as stand-ins, named here so the GPU-side test
(tests/test_reference_loops_gpu.py) can use the SAME ones: ``FLAGS`` (a namespace with the fields the loops read),
``amp`` (never reached: FLAGS.apex is False), ``warprnnt_pytorch.RNNTLoss`` for the CPU side only (the reference's loss
op is not installable: oracle.rnnt_loss_ref.rnnt_loss_torch in float64, 'mean' reduction, shape (1,)).

Stored: the loss every train_step returned, a checksum of the parameters after the last step, the streamed texts.

    python oracle/ref_lift.py && python oracle/make_golden_ref_loops.py        # needs /root/reference
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import models_ref as M                       # noqa: E402
from oracle import ref_lift                              # noqa: E402

# ---------------------------------------------------------------- scenarios (shared with the GPU test)
TRAIN_CFG = dict(vocab_embed_size=16, vocab_size=64, input_size=240, enc_hidden_size=64, enc_layers=3,
                 enc_proj_size=48, dec_hidden_size=32, dec_layers=2, dec_proj_size=32, joint_size=64)
TRAIN = dict(cfg=TRAIN_CFG, wseed=21, xseed=22, B=4, sub_B=2, T0=37, U=6, steps=3, lr=1e-3, gradclip=2.0)

FRONT_PARAMS = [(4, 2, 8), (3, 2, 24), (2, 1, 16)]
FRONT_CFG = dict(vocab_embed_size=16, vocab_size=64, input_size=16, enc_hidden_size=64, enc_layers=2,
                 enc_proj_size=48, dec_hidden_size=32, dec_layers=1, dec_proj_size=32, joint_size=64,
                 enc_time_reductions=[])
# cli/train.py:233-235 permutes the front-end's [B, T, C] output to [B, C, T] before the model (whose LayerNorm is over
# C): as written the loop only runs when the front-end yields exactly T == C frames - N = 70 samples -> 16 frames of 16
# channels here - and then treats the CHANNEL axis as time (xlen is rescaled by max(xlen) / C).  Parity means
# reproducing what the loop computes, so the scenario is built to be runnable, not to be sensible.
FRONT = dict(cfg=FRONT_CFG, params=FRONT_PARAMS, wseed=31, fseed=32, xseed=33, B=4, sub_B=4, N=70, U=5, steps=3,
             lr=1e-3, gradclip=None)

STREAM_FLAGS = dict(win_length=320, hop_length=200, downsample=3, step_n_frame=2)
MIC_BLOCKS, MIC_BLOCK = 48, 660          # stream.py:128: two buffered blocks = one 1320-sample window


def train_flags(case):
    return types.SimpleNamespace(batch_size=case["B"], sub_batch_size=case["sub_B"], multi_gpu=False, apex=False,
                                 gradclip=case["gradclip"])


def train_batch(case):
    return M.make_batch(case["cfg"], case["xseed"], case["B"], case["T0"], case["U"])


def front_batch(case):
    g = torch.Generator(device="cpu").manual_seed(case["xseed"])
    B, N, U = case["B"], case["N"], case["U"]
    wave = (0.1 * torch.randn(B, N, generator=g)).clamp(-1, 1)
    ys = torch.randint(4, case["cfg"]["vocab_size"], (B, U), generator=g, dtype=torch.int32)
    xlen = torch.tensor([N, N - 4, N - 20, N - 32][:B], dtype=torch.int32)
    ylen = torch.tensor([U, U - 1, U - 2, U][:B], dtype=torch.int32)
    for b in range(B):
        wave[b, xlen[b]:] = 0
        ys[b, ylen[b]:] = M.PAD
    return wave, ys, xlen, ylen


def frontend_state_dict(params, seed):
    """Reference key names / shapes of FrontEnd(frontend_params, bias=True) (rnnt/models.py:341-359), seeded."""
    g = torch.Generator().manual_seed(seed)
    ks, ch = [p[0] for p in params], [p[2] for p in params]
    sd = {"conv1.weight": torch.randn(ch[0], 1, ks[0], generator=g) * (2.0 / ks[0]) ** 0.5,
          "conv1.bias": 0.1 * torch.randn(ch[0], generator=g)}
    for i in range(1, len(params)):
        p = "encode.%d." % (i - 1)
        sd[p + "conv.weight"] = torch.randn(ch[i], ch[i - 1], ks[i], generator=g) * (2.0 / (ch[i - 1] * ks[i])) ** 0.5
        sd[p + "conv.bias"] = 0.1 * torch.randn(ch[i], generator=g)
        sd[p + "gn.weight"] = 1.0 + 0.1 * torch.randn(ch[i - 1], generator=g)
        sd[p + "gn.bias"] = 0.1 * torch.randn(ch[i - 1], generator=g)
    sd["layer_norm.weight"] = 1.0 + 0.1 * torch.randn(ch[-1], generator=g)
    sd["layer_norm.bias"] = 0.1 * torch.randn(ch[-1], generator=g)
    return sd


def mic_blocks():
    """int16-range float blocks [MIC_BLOCK, 1] as sounddevice hands them to stream.py's callback; the second
    half is silence, so that more than 35 consecutive chunks decode to '' and the callback resets the decoder."""
    g = torch.Generator(device="cpu").manual_seed(41)
    x = (0.1 * torch.randn(MIC_BLOCKS, MIC_BLOCK, 1, generator=g) * 65536.0).numpy().astype(np.float32)
    x[MIC_BLOCKS // 4:] = 0.0
    return x


def checksum(params):
    """Order-dependent fingerprint of a parameter list: (sum |p|, sum p * ramp) per tensor, float64."""
    out = []
    for p in params:
        q = p.detach().double().cpu().flatten()
        ramp = torch.linspace(-1.0, 1.0, q.numel(), dtype=torch.float64)
        out.append([q.abs().sum().item(), (q * ramp).sum().item()])
    return np.array(out)


def run_train_steps(piece, namespace, trainer, batches):
    """Bind the lifted ``train_step`` to ``trainer`` and run it over ``batches``; returns the losses."""
    ns = ref_lift.load(piece, namespace)
    step = types.MethodType(ns["train_step"], trainer)
    return [float(step(b)) for b in batches]


class StubTextTokenizer:
    """Stand-in for rnnt.tokenizer's decode_plus (rnnt/tokenizer.py: ids <= 3 dropped, the rest joined): the text IS the
    id list, so WER compares token ids."""

    def decode_plus(self, seqs):
        return [" ".join("t%d" % int(t) for t in seq if int(t) > 3) for seq in seqs]


def stub_jiwer():
    """jiwer.wer(truth, hypothesis) on lists of strings: word-level edit distance / reference words (jiwer's default)."""
    def wer(truth, hyp):
        errs = words = 0
        for t, h in zip(truth, hyp):
            a, b = t.split(), h.split()
            d = list(range(len(b) + 1))
            for i in range(1, len(a) + 1):
                prev, d[0] = d[0], i
                for j in range(1, len(b) + 1):
                    cur = min(d[j] + 1, d[j - 1] + 1, prev + (a[i - 1] != b[j - 1]))
                    prev, d[j] = d[j], cur
            errs += d[len(b)]
            words += len(a)
        return errs / max(1, words)
    return types.SimpleNamespace(wer=wer)


def run_evaluate_step(namespace, trainer, batch):
    """cli/baseline.py's Trainer.evaluate_step (loss in eval mode, greedy_decode, decode_plus, jiwer.wer) bound to
    ``trainer``; returns (loss, wer, pred_seq, true_seq)."""
    ns = ref_lift.load("baseline_eval", namespace)
    trainer.model.eval()
    with torch.no_grad():
        return types.MethodType(ns["evaluate_step"], trainer)(batch)


LIGHTNING = dict(cfg=TRAIN_CFG, wseed=51, xseed=52, B=3, T0=33, U=5, steps=3, lr=2e-3, clip=10.0, warmup_step=2)


def lightning_module(model, loss_fn, optimizer, tokenizer):
    """What ParallelTraining.__init__ (cli/lightning.py:29-70) and Lightning's trainer leave on the module, without
    pytorch_lightning: the attributes its training_step / validation_step read."""
    log = types.SimpleNamespace(experiment=types.SimpleNamespace(add_scalar=lambda *a, **k: None,
                                                                 add_text=lambda *a, **k: None))
    return types.SimpleNamespace(model=model, loss_fn=loss_fn, optimizer=optimizer, tokenizer=tokenizer, logger=log,
                                 steps=0, epoch=0)


def run_lightning(namespace, module, batch, steps, clip):
    """cli/lightning.py's training_step `steps` times, each followed by what Lightning's trainer does around it
    (cli/lightning.py:325-331: backward, gradient_clip_val=10, optimizer step, zero_grad), then one validation_step.
    Returns (losses, validation dict before the steps, validation dict after them)."""
    ns = ref_lift.load("lightning_steps", namespace)
    module.warmup_optimizer_step = types.MethodType(ns["warmup_optimizer_step"], module)
    train = types.MethodType(ns["training_step"], module)
    val = types.MethodType(ns["validation_step"], module)
    module.model.eval()
    with torch.no_grad():
        v0 = val(batch, 0)            # on the seeded weights: token-exact on both sides
    losses = []
    for i in range(steps):
        module.model.train()
        out = train(batch, i)
        out["loss"].sum().backward()
        torch.nn.utils.clip_grad_norm_(module.model.parameters(), clip)
        module.optimizer.step()
        module.optimizer.zero_grad()
        losses.append(float(out["log"]["loss"]))
    module.model.eval()
    with torch.no_grad():
        v = val(batch, 0)             # after the optimiser steps: weights equal to rounding only
    return losses, v0, v


def stub_jiwer_measures():
    j = stub_jiwer()
    j.compute_measures = lambda truth, hyp: {"wer": j.wer(truth, hyp)}
    return j


def run_mic(namespace, decoder, blocks):
    """stream.py's callback over ``blocks``; returns what it printed."""
    namespace.update(buffer=[], blank_counter=0, stream_decoder=decoder, np=np, torch=torch)
    ns = ref_lift.load("mic_callback", namespace)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        for blk in blocks:
            ns["callback"](blk, None, len(blk), None, None)
    return out.getvalue()


# ---------------------------------------------------------------- reference side (CPU)
class _OracleRNNTLoss:
    """Stand-in for warprnnt_pytorch.RNNTLoss on the CPU side (float64 restatement, 'mean', shape (1,))."""

    def __init__(self, blank=0):
        self.blank = blank

    def __call__(self, acts, labels, act_lens, label_lens):
        from oracle.rnnt_loss_ref import rnnt_loss_torch
        costs = rnnt_loss_torch(acts.double(), labels, act_lens, label_lens, self.blank)
        return costs.mean().reshape(1).to(acts.dtype)


def _reference_modules():
    from oracle.make_golden import reference_model
    stub = types.ModuleType("warprnnt_pytorch")
    stub.RNNTLoss = _OracleRNNTLoss
    sys.modules["warprnnt_pytorch"] = stub           # rnnt/models.py:8-11 imports it inside try/except
    try:
        reference_model(dict(vocab_embed_size=8, vocab_size=40, input_size=24, enc_hidden_size=32, enc_layers=1,
                             enc_proj_size=24, dec_hidden_size=16, dec_layers=1, dec_proj_size=16, joint_size=32), None)
    except Exception:
        pass                                         # only for its side effect: sys.modules["rnnt.models"]
    ref = sys.modules["rnnt.models"]
    assert ref.__file__.startswith(ref_lift.REF) and ref.RNNTLoss is _OracleRNNTLoss
    return ref


def main():
    assert ref_lift.available(), "run `python oracle/ref_lift.py` first"
    torch.set_num_threads(8)
    ref = _reference_modules()
    out = {}

    # ---- cli/baseline.py train_step
    c = TRAIN
    model = ref.Transducer(enc_dropout=0.0, dec_dropout=0.0, output_loss=True, **c["cfg"])
    model.load_state_dict(M.make_state_dict(c["cfg"], c["wseed"]), strict=True)
    model.train()
    tr = types.SimpleNamespace(model=model, optim=torch.optim.Adam(model.parameters(), lr=c["lr"]))
    ns = dict(FLAGS=train_flags(c), device=torch.device("cpu"), torch=torch, amp=None)
    out["baseline_losses"] = np.array(run_train_steps("baseline_train_step", ns, tr, [train_batch(c)] * c["steps"]))
    out["baseline_checksum"] = checksum(model.parameters())
    print("cli/baseline.py train_step losses:", out["baseline_losses"])

    # ---- cli/baseline.py evaluate_step on the seeded weights (a fresh model: independent of the optimiser path)
    model = ref.Transducer(enc_dropout=0.0, dec_dropout=0.0, output_loss=True, **c["cfg"])
    model.load_state_dict(M.make_state_dict(c["cfg"], c["wseed"]), strict=True)
    tr = types.SimpleNamespace(model=model, tokenizer=StubTextTokenizer())
    ns = dict(FLAGS=train_flags(c), device=torch.device("cpu"), torch=torch, np=np, jiwer=stub_jiwer())
    loss, wer, pred, true = run_evaluate_step(ns, tr, train_batch(c))
    out["eval_loss"], out["eval_wer"] = np.float64(loss), np.float64(wer)
    out["eval_pred"], out["eval_true"] = np.array(pred), np.array(true)
    print("cli/baseline.py evaluate_step: loss %.5f wer %.4f pred[0] %r" % (loss, wer, pred[0]))

    # ---- cli/lightning.py training_step / validation_step: the EXTERNAL loss call (Transducer(output_loss=False))
    c = LIGHTNING
    model = ref.Transducer(enc_dropout=0.0, dec_dropout=0.0, output_loss=False, **c["cfg"])
    model.load_state_dict(M.make_state_dict(c["cfg"], c["wseed"]), strict=True)
    mod = lightning_module(model, _OracleRNNTLoss(blank=M.NUL), torch.optim.Adam(model.parameters(), lr=c["lr"]),
                           StubTextTokenizer())
    ns = dict(FLAGS=types.SimpleNamespace(warmup_step=c["warmup_step"], lr=c["lr"]), torch=torch, np=np,
              jiwer=stub_jiwer_measures())
    batch = M.make_batch(c["cfg"], c["xseed"], c["B"], c["T0"], c["U"])
    losses, v0, v = run_lightning(ns, mod, batch, c["steps"], c["clip"])
    out["lightning_losses"] = np.array(losses)
    out["lightning_val0_loss"], out["lightning_val0_wer"] = np.float64(v0["val_loss"]), np.float64(v0["wer"])
    out["lightning_val0_hypothesis"] = np.array(v0["hypothesis"])
    out["lightning_val_loss"] = np.float64(v["val_loss"])
    print("cli/lightning.py training_step losses:", losses, "validation before:", v0["val_loss"], v0["wer"],
          repr(v0["hypothesis"])[:70], "after:", v["val_loss"])

    # ---- cli/train.py train_step (FrontEnd)
    c = FRONT
    model = ref.Transducer(enc_dropout=0.0, dec_dropout=0.0, output_loss=True, **c["cfg"])
    model.load_state_dict(M.make_state_dict(c["cfg"], c["wseed"]), strict=True)
    front = ref.FrontEnd(frontend_params=c["params"], bias=True)
    front.load_state_dict(frontend_state_dict(c["params"], c["fseed"]), strict=True)
    model.train(), front.train()
    tr = types.SimpleNamespace(model=model, frontend=front, optim=torch.optim.Adam(
        list(model.parameters()) + list(front.parameters()), lr=c["lr"]))
    ns = dict(FLAGS=train_flags(c), device=torch.device("cpu"), torch=torch, amp=None)
    out["frontend_losses"] = np.array(run_train_steps("frontend_train_step", ns, tr, [front_batch(c)] * c["steps"]))
    out["frontend_checksum"] = checksum(list(model.parameters()) + list(front.parameters()))
    print("cli/train.py train_step losses:", out["frontend_losses"])

    # ---- cli/openvino_wav_inference.py stream_decode + stream.py callback over the reference's decoder
    from oracle import make_golden_features as GF
    from oracle import make_golden_stream as GS
    Decoder = GS.reference_decoder_class()
    RF, DS = GF.reference_rnnt_features(), GF.reference_downsample()
    cfg, wseed, xseed, S, n_chunks, resets, bias = GS.CASES["small"]
    sd = GS.state_dict(cfg, wseed, bias)
    g = torch.Generator(device="cpu").manual_seed(xseed)
    wave = 0.1 * torch.randn(1, GS.WIN + n_chunks * GS.HOP, generator=g)
    dec = GS.reference_stream(cfg, sd, Decoder, RF, DS)
    ns = ref_lift.load("stream_decode", dict(FLAGS=types.SimpleNamespace(**STREAM_FLAGS)))
    text, frames = ns["stream_decode"](dec, wave)
    out["stream_decode_text"] = np.array(text)
    out["stream_decode_frames"] = np.int64(frames)
    print("stream_decode:", repr(text), frames)
    dec = GS.reference_stream(cfg, sd, Decoder, RF, DS)
    printed = run_mic({}, dec, mic_blocks())
    assert " [Background]" in printed, "the mic scenario never reached the 35-blank reset"
    out["mic_printed"] = np.array(printed)
    print("callback printed:", repr(printed))

    path = os.path.join(ROOT, "tests", "golden", "ref_loops.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
