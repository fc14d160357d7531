"""Streaming RNN-T decoding on the GPU with the reference's ``rnnt/stream.py`` API.

``PytorchStreamDecoder(FLAGS)`` keeps ``reset()`` / ``decode(frame) -> str`` / ``reset_profile()``
and the ``encoder_elapsed`` / ``joint_elapsed`` / ``decoder_elapsed`` lists
(rnnt/stream.py:15-120).  ``BatchedStreamDecoder`` is the MI355X generalisation the reference
does not have: S concurrent streams advance in lock-step on one GPU, state tensors are
``[L, S, H]`` and ``reset`` takes a per-stream mask.  Both drive the same kernels: fused
log-mel on the chunk, stateful encoder, then the on-device search loop of csrc/decode.hip in
stream mode (raw-logit arg-max, the ``<unk>`` rule, prediction net advanced on non-blank).
"""
import os
import time

import torch


from . import encoder_stack
from .decode import init_search_state, run_search
from .features import StackedLogFbank
from .flags import model_kwargs
from .models import Transducer, convert_lightning2normal
from .tokenizer import NUL, UNK


class StreamTransducerDecoder:
    def reset_profile(self):
        self.encoder_elapsed = []
        self.decoder_elapsed = []
        self.joint_elapsed = []

    def reset(self):
        raise NotImplementedError()

    def decode(self, frame):
        raise NotImplementedError()


def chunk_geometry(flags, step_n_frame=2):
    """(win_size, hop_size) in samples of one streaming chunk (stream.py:80-88 of the reference;
    cli/openvino_wav_inference.py:30-34)."""
    win = flags.win_length + flags.hop_length * (flags.downsample * step_n_frame - 1)
    hop = flags.hop_length * flags.downsample * step_n_frame
    return win, hop


class BatchedStreamDecoder(StreamTransducerDecoder):
    """S independent streams decoded concurrently on one GPU."""

    def __init__(self, transducer, flags, n_streams, unk_id=UNK, dither=None):
        self.model = transducer.eval()
        self.flags = flags
        self.S = n_streams
        self.unk_id = unk_id
        dev = transducer.decoder.embed.weight.device
        self.device = dev
        self.transform = StackedLogFbank(
            n_frame=flags.downsample, pad_to_divisible=False, win_length=flags.win_length,
            hop_length=flags.hop_length, n_fft=flags.n_fft, n_filt=flags.feature_size,
            dither=1e-5 if dither is None else dither).to(dev)
        self.reset_profile()
        self.reset()

    @torch.no_grad()
    def reset(self, mask=None):
        """Reset every stream, or only those where ``mask[s]`` is true."""
        enc = self.model.encoder.lstm
        L, H = len(enc.lstms), enc.hidden_size
        fresh = init_search_state(self.model, self.S)
        if mask is None or not hasattr(self, "state"):
            self.enc_h = torch.zeros(L, self.S, H, device=self.device)
            self.enc_c = torch.zeros(L, self.S, H, device=self.device)
            self.state = fresh
            return
        m = mask.to(self.device).bool()
        self.enc_h[:, m] = 0
        self.enc_c[:, m] = 0
        self.state.dec_out[m] = fresh.dec_out[m]
        self.state.h[:, m] = fresh.h[:, m]
        self.state.c[:, m] = fresh.c[:, m]

    @torch.no_grad()
    def decode(self, frames):
        """frames: float32 [S, win_size] on the device -> int32 [S, k] token ids (0 = blank),
        k = encoder frames produced by this chunk."""
        t0 = time.time()
        encoder_stack.check_wsr_error()      # a bounded in-kernel wait of an EARLIER chunk that gave up (host word, no sync)
        xs, _ = self.transform(frames)
        enc_out, (self.enc_h, self.enc_c) = self.model.encoder(xs, (self.enc_h, self.enc_c))
        self.encoder_elapsed.append(time.time() - t0)
        t0 = time.time()
        tokens, _ = run_search(self.model, enc_out.contiguous(), self.state, unk=self.unk_id,
                               want_score=False)
        self.joint_elapsed.append(time.time() - t0)
        return tokens


class PytorchStreamDecoder(StreamTransducerDecoder):
    """Single-stream decoder with the reference's constructor: ``PytorchStreamDecoder(FLAGS)``
    loads ``logs/<FLAGS.name>/models/<FLAGS.model_name>`` (or ``logs/<name>/<model_name>``) and
    the cached BPE vocabulary ``BPE-<bpe_size>/`` exactly as rnnt/stream.py:29-76 does.  For
    tests and embedding in other programs a ready ``transducer`` and ``tokenizer`` (anything with
    ``id_to_token(int) -> str``) can be injected instead."""

    def __init__(self, FLAGS, transducer=None, tokenizer=None, device="cuda", dither=None):
        self.FLAGS = FLAGS
        if tokenizer is None:
            tokenizer = _load_bpe_tokenizer(FLAGS)
        self.tokenizer = tokenizer
        if transducer is None:
            transducer = _load_checkpointed_transducer(FLAGS, _vocab_size(tokenizer, FLAGS))
        transducer = transducer.to(device).eval()
        self.encoder = transducer.encoder
        self.decoder = transducer.decoder
        self.joint = transducer.joint
        unk = _find_unk(tokenizer, transducer.joint.joint[2].weight.shape[0])
        self._batched = BatchedStreamDecoder(transducer, FLAGS, 1, unk_id=unk, dither=dither)
        self.transform = self._batched.transform
        self.reset_profile()
        self._batched.encoder_elapsed = self.encoder_elapsed
        self._batched.joint_elapsed = self.joint_elapsed
        self._batched.decoder_elapsed = self.decoder_elapsed

    def reset(self):
        self._batched.reset()

    # state attributes the reference exposes (rnnt/stream.py:80-91)
    @property
    def enc_h(self):
        return self._batched.enc_h

    @property
    def enc_c(self):
        return self._batched.enc_c

    @torch.no_grad()
    def decode(self, frame):
        """frame: float32 [1, n_samples] (any device) -> decoded text of this chunk."""
        frame = frame.to(self._batched.device, torch.float32)
        tokens = self._batched.decode(frame.contiguous())[0].tolist()
        out = []
        for tok in tokens:
            if tok != NUL:
                out.append(_token_text(self.tokenizer, tok).replace('</w>', ' '))
        return "".join(out)


def _token_text(tokenizer, tok):
    inner = getattr(tokenizer, "tokenizer", tokenizer)
    return inner.id_to_token(int(tok))


def _find_unk(tokenizer, vocab):
    """Id whose token string is '<unk>' (the reference compares strings, rnnt/stream.py:106)."""
    try:
        for i in range(min(vocab, 16)):
            if _token_text(tokenizer, i) == '<unk>':
                return i
    except Exception:
        pass
    return UNK


def _vocab_size(tokenizer, FLAGS):
    return getattr(tokenizer, "vocab_size", FLAGS.bpe_size)


def _load_bpe_tokenizer(FLAGS):
    from tokenizers import CharBPETokenizer
    cache = 'BPE-' + str(FLAGS.bpe_size)
    name = "%d-%s" % (FLAGS.bpe_size, None)
    vocab = os.path.join(cache, name + '-vocab.json')
    merges = os.path.join(cache, name + '-merges.txt')
    if not (os.path.exists(vocab) and os.path.exists(merges)):
        raise FileNotFoundError("cached BPE vocabulary %s / %s not found" % (vocab, merges))

    class _Tok:
        pass
    t = _Tok()
    t.tokenizer = CharBPETokenizer(vocab, merges, lowercase=True)
    t.vocab_size = FLAGS.bpe_size
    return t


def _load_checkpointed_transducer(FLAGS, vocab_size):
    logdir = os.path.join('logs', FLAGS.name)
    path = os.path.join(logdir, 'models', FLAGS.model_name)
    if not os.path.exists(path):
        path = os.path.join(logdir, FLAGS.model_name)
    checkpoint = torch.load(path, map_location="cpu")
    model = Transducer(output_loss=False, **model_kwargs(FLAGS, vocab_size=vocab_size))
    model.load_state_dict(convert_lightning2normal(checkpoint)['model'])
    return model
