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


from . import _lib, config, encoder_stack
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


class _ChunkPlan:
    """The native calls of ONE chunk step of ``BatchedStreamDecoder.decode`` with their arguments bound once.

    A chunk step is six native calls - dither, fused log-mel + stacking (csrc/fbank.hip), the encoder step on the
    chunk's frames (``edgedict_stream_encoder_step``), the encoder's output projection, the encoder half of the joint's
    first Linear (two small products), the search frames (``edgedict_greedy_decode``).  Driven through the module path
    (transform -> Encoder.forward -> run_search) their Python glue - parameter look-ups, weight-cache checks, ctypes
    argument conversion, a dozen tensor allocations - costs 0.33 ms per chunk step, more than the ~0.2 ms of kernels
    behind it at 256 streams.  The plan keeps every intermediate buffer and every converted argument; per chunk only the
    caller's frame pointer, the dither seed and the token buffer change.  Same kernels, same arguments, same order:
    bit-identical to the module path (tests/test_stream_gpu.py).  Rebuilt when a parameter changes (version counters,
    the engine's parameter epoch), the frames' shape changes or the model moves."""

    def __init__(self, dec, frames):
        import ctypes
        from . import ops
        from ._lib import dtype_code, ptr
        from .models import WEIGHTS, ResLayerNormLSTM
        m, enc = dec.model, dec.model.encoder
        lstm = enc.lstm
        cd = torch.bfloat16
        fb, tr = dec.transform.fbank, dec.transform
        S, N = frames.shape
        self.ok = False
        self.key = (S, N, frames.stride(0))
        # (module._parameters dict, name, Parameter) of every parameter: a REPLACED Parameter object (load_state_dict(assign=True),
        # `module.weight = ...`, module._apply swapping tensors) fails the identity check, `p.data = ...` the data_ptr check, an
        # in-place update the version check - the plan holds the old tensors alive, so it must never outlive them silently
        self.slots = [(mod._parameters, name, p) for mod in m.modules() for name, p in mod._parameters.items() if p is not None]
        self.params = [p for _, _, p in self.slots]
        self.dec = dec
        self.sig = self._signature()
        k, M = tr.n_frame, fb.n_filt
        T0 = tr.output_frames(N)
        I0 = M * k
        if not (isinstance(lstm, ResLayerNormLSTM) and enc.has_proj and config.STREAM_ENCODER_STEP and config.STREAM_FAST_CHUNK
                and _compute_dtype(enc) == cd and _compute_dtype(m.decoder) == cd and _compute_dtype(m.joint) == cd
                and 0 < T0 < config.STACK_MIN_FRAMES and tr.out_dtype == torch.float32
                and S <= (config.STREAM_STEP_MAX_ROWS_SHORT if T0 <= 2 else config.STREAM_STEP_MAX_ROWS)
                and lstm.hidden_size % 32 == 0 and I0 % 8 == 0
                # (train mode + dropout: the module path applies it.  Encoder.forward decides from the ENCODER's own flags,
                # which differ from the Transducer's after m.eval(); m.encoder.train() or with frozen sub-modules)
                and (getattr(lstm, "dropout", 0) == 0 or not (m.training or enc.training or lstm.training))):
            return
        dev = frames.device
        lib = _lib.load()
        H, L = lstm.hidden_size, len(lstm.lstms)
        ll, ci, vp = ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p
        P = lambda t: vp(0 if t is None else t.data_ptr())          # noqa: E731
        self.keep = []                                              # every tensor a bound pointer refers to

        def hold(t):
            self.keep.append(t)
            return t
        # ---- A. dither + log-mel + stacking (features.FilterbankFeatures._run / StackedLogFbank.forward)
        self.fb = fb
        # (bf16 straight from the log-mel kernel: the module path writes fp32 and the encoder step casts it - the same rounding,
        # one launch more)
        self.xs = hold(torch.empty(S, T0, I0, dtype=cd, device=dev))
        lo, hi = fb._win_support
        self.dither_args = [None, ll(frames.stride(0)), ci(S), ci(N), vp(0), ctypes.c_float(float(fb.dither)), None]
        self.fbank_args = [None, ll(frames.stride(0)), ci(S), ci(N), vp(0), P(fb._window_full), P(fb._twiddle), P(fb.fb),
                           P(fb._fb_range), ci(fb.n_fft), ci(lo), ci(hi), ci(fb.hop_length), ci(fb.n_filt),
                           ctypes.c_float(float(fb.preemph if fb.preemph is not None else 0.0)), ci(int(bool(fb.log))),
                           P(self.xs), ci(dtype_code(self.xs.dtype)), ll(T0 * I0), ll(I0), ll(M), ll(1), ci(k), ci(T0 * k)]
        # ---- B. encoder step (models._stream_encoder_step) on the decoder's own state tensors, in place
        T_out = T0
        for r in lstm.reductions:
            T_out = (T_out + r - 1) // r
        self.T_out = T_out
        w_ih = [hold(WEIGHTS.get(mm.layer(0)[0], cd)) for mm in lstm.lstms]
        w_hh = [hold(WEIGHTS.get(mm.layer(0)[1], cd)) for mm in lstm.lstms]
        b_ih = [hold(mm.layer(0)[2].detach()) for mm in lstm.lstms]
        b_hh = [hold(mm.layer(0)[3].detach()) for mm in lstm.lstms]
        gam = [hold(pp[0].weight.detach()) for pp in lstm.projs]
        bet = [hold(pp[0].bias.detach()) for pp in lstm.projs]
        arr = lambda ts: hold_raw(self, (vp * len(ts))(*[t.data_ptr() for t in ts]))      # noqa: E731
        self.rows = hold(torch.empty(S, T_out, H, dtype=cd, device=dev))
        ws_e = hold(torch.empty(lib.edgedict_stream_encoder_workspace_bytes(S, T0, I0, H, L), dtype=torch.uint8, device=dev))
        red = hold_raw(self, (ci * L)(*[int(r) for r in lstm.reductions]))
        self.t_out = ci(0)
        self.enc_args = [P(self.xs), ci(dtype_code(self.xs.dtype)), ci(S), ci(T0), ci(I0), ci(H), ci(L),
                         P(hold(enc.norm.weight.detach())), P(hold(enc.norm.bias.detach())), arr(w_ih), arr(w_hh), arr(b_ih),
                         arr(b_hh), arr(gam), arr(bet), red, None, None, P(self.rows), ctypes.byref(self.t_out), P(ws_e)]
        # ---- C. output projection (models._LinearFn) and D. the encoder half of the joint's first Linear (decode.run_search)
        l1, l2 = m.joint.joint[0], m.joint.joint[2]
        dcd = m.decoder
        wp_e = hold(WEIGHTS.get(enc.proj.weight, cd))
        Pe = wp_e.shape[0]
        self.enc_out = hold(torch.empty(S * T_out, Pe, dtype=cd, device=dev))
        w1c = hold(WEIGHTS.get(l1.weight, cd))
        w2c = hold(WEIGHTS.get(l2.weight, cd))
        wpc = hold(WEIGHTS.get(dcd.proj.weight, cd))
        J, V = l1.weight.shape[0], l2.weight.shape[0]
        self.E1 = hold(torch.empty(S * T_out, J, dtype=cd, device=dev))
        code = dtype_code(cd)

        def gemm_args(a_ptr, lda, Mr, b, ldb, out, Nc, K, bias):
            return [ci(code), ci(code), a_ptr, ll(lda), ci(1), P(b), ll(ldb), ci(1), P(out), ll(Nc), ci(Mr), ci(Nc), ci(K),
                    P(bias), vp(0), ci(0), ci(1)]
        Mr = S * T_out
        self.proj_args = gemm_args(P(self.rows), H if Mr > 1 else max(H, H), Mr, wp_e, wp_e.stride(0), self.enc_out, Pe, H,
                                   hold(enc.proj.bias.detach()) if enc.proj.bias is not None else None)
        self.e1_args = gemm_args(P(self.enc_out), Pe, Mr, w1c, w1c.stride(0), self.E1, J, Pe, None)
        # ---- E. the search frames (decode.run_search), state in place
        P2 = dcd.proj.weight.shape[0]
        Ld, Hd = dcd.lstm.num_layers, dcd.lstm.hidden_size
        E = dcd.embed.weight.shape[1]
        dw_ih = [hold(WEIGHTS.get(dcd.lstm.layer(i)[0], cd)) for i in range(Ld)]
        dw_hh = [hold(WEIGHTS.get(dcd.lstm.layer(i)[1], cd)) for i in range(Ld)]
        db_ih = [hold(dcd.lstm.layer(i)[2].detach()) for i in range(Ld)]
        db_hh = [hold(dcd.lstm.layer(i)[3].detach()) for i in range(Ld)]
        ws_g = hold(torch.empty(lib.edgedict_greedy_workspace_bytes(code, S, J, V, E, Ld, Hd, P2), dtype=torch.uint8, device=dev))
        w1d = w1c[:, Pe:]
        emb = hold(dcd.embed.weight.detach())
        self.greedy_args = [ci(code), P(self.E1), ll(T_out * J), ll(J), ci(S), ci(T_out), ci(J), vp(w1d.data_ptr()),
                            ll(w1c.stride(0)), P(hold(l1.bias.detach())), ci(P2), P(w2c), P(hold(l2.bias.detach())), ci(V), P(emb),
                            ci(dtype_code(emb.dtype)), ci(E), ci(Ld), arr(dw_ih), arr(dw_hh), arr(db_ih), arr(db_hh), ci(Hd),
                            P(wpc), P(hold(dcd.proj.bias.detach())), None, None, None, ci(int(m.blank)), ci(int(dec.unk_id)),
                            None, None, vp(0), P(ws_g)]
        self.S, self.dev = S, dev
        self.lib = lib
        self.ok = True

    def _signature(self):
        # everything a bound argument list was derived from: parameter identity / storage / version, the engine's parameter
        # epoch, train / eval mode, and the scalars baked into the calls (dither amplitude, <unk> id, blank id)
        dec, fb = self.dec, self.dec.transform.fbank
        return (config.param_epoch(), sum(p._version for p in self.params),
                hash(tuple(p.data_ptr() for p in self.params)),
                all(d.get(n) is p for d, n, p in self.slots),
                (dec.model.training, dec.model.encoder.training, dec.model.encoder.lstm.training),
                float(fb.dither), int(dec.unk_id), int(dec.model.blank))

    def valid(self, frames):
        return (self.key == (frames.shape[0], frames.shape[1], frames.stride(0)) and frames.dtype == torch.float32
                and frames.stride(1) == 1 and frames.is_cuda and self.sig == self._signature())

    def run(self, dec, frames):
        import ctypes
        lib, fb = self.lib, self.fb
        vp = ctypes.c_void_p
        st = _lib.stream_ptr()
        x = vp(frames.data_ptr())
        t0 = time.time()
        if fb.dither > 0:           # in place on the caller's tensor, as the reference does (rnnt/features.py:111-112)
            fb._seed = (fb._seed * 1664525 + 1013904223) & 0xFFFFFFFF
            a = self.dither_args
            a[0], a[6] = x, ctypes.c_uint(fb._seed)
            _lib.check(lib.edgedict_dither(*a, st), "dither")
        a = self.fbank_args
        a[0] = x
        _lib.check(lib.edgedict_fbank_forward(*a, st), "fbank_forward")
        a = self.enc_args
        a[16], a[17] = vp(dec.enc_h.data_ptr()), vp(dec.enc_c.data_ptr())
        _lib.check(lib.edgedict_stream_encoder_step(*a, st), "stream_encoder_step")
        _lib.check(lib.edgedict_gemm(*self.proj_args, st), "gemm")
        dec.encoder_elapsed.append(time.time() - t0)
        t0 = time.time()
        _lib.check(lib.edgedict_gemm(*self.e1_args, st), "gemm")
        tokens = torch.empty(self.S, max(self.T_out, 1), dtype=torch.int32, device=self.dev)
        state = dec.state
        a = self.greedy_args
        a[25], a[26], a[27] = vp(state.h.data_ptr()), vp(state.c.data_ptr()), vp(state.dec_out.data_ptr())
        a[30], a[31] = vp(tokens.data_ptr()), ctypes.c_int(tokens.stride(0))
        _lib.check(lib.edgedict_greedy_decode(*a, st), "greedy_decode")
        dec.joint_elapsed.append(time.time() - t0)
        return tokens[:, :self.T_out]


def hold_raw(plan, obj):
    plan.keep.append(obj)
    return obj


def _compute_dtype(module):
    cd = getattr(module, "compute_dtype", None) or config.get_compute_dtype()
    return torch.bfloat16 if cd in ("bf16", torch.bfloat16) else (torch.float32 if cd in ("fp32", torch.float32) else cd)


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
    def nonfinite_streams(self):
        """bool [S] (device): streams whose carried state - encoder (h, c), prediction network (h, c), its projected
        output - holds a NaN or Inf.  The search frames never emit an out-of-range id: a row of logits without one
        comparable value emits blank, and a row with SOME NaNs picks its best finite logit (torch.argmax, which the
        reference's loop uses - rnnt/stream.py:100-103 -, would return the first NaN's index instead).  A stream that has
        gone non-finite therefore keeps producing plausible-looking tokens from poisoned state; this is the caller's
        signal - one reduction over the states, no host synchronisation until the result is read - e.g. every few
        hundred chunks: ``bad = dec.nonfinite_streams(); dec.reset(bad)``."""
        bad = ~torch.isfinite(self.enc_h).all(dim=2).all(dim=0)
        bad |= ~torch.isfinite(self.enc_c).all(dim=2).all(dim=0)
        bad |= ~torch.isfinite(self.state.h).all(dim=2).all(dim=0)
        bad |= ~torch.isfinite(self.state.c).all(dim=2).all(dim=0)
        bad |= ~torch.isfinite(self.state.dec_out.float()).all(dim=-1).reshape(self.S, -1).all(dim=1)
        return bad

    @torch.no_grad()
    def decode(self, frames):
        """frames: float32 [S, win_size] on the device -> int32 [S, k] token ids (0 = blank),
        k = encoder frames produced by this chunk."""
        plan = getattr(self, "_plan", None)
        if plan is None or not plan.valid(frames):
            plan = self._plan = _ChunkPlan(self, frames) if (frames.is_cuda and frames.dim() == 2 and frames.dtype == torch.float32
                                                             and frames.stride(1) == 1) else None
        if plan is not None and plan.ok and self.enc_h.is_contiguous() and self.enc_c.is_contiguous():
            return plan.run(self, frames)
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

    # state attributes the reference exposes (rnnt/stream.py:80-91).  NOTE (differs from the reference, which rebinds
    # fresh tensors every chunk, rnnt/stream.py:97-98): on the bound-argument fast path (_ChunkPlan) the encoder state is
    # updated IN PLACE - a caller that wants a rollback point must keep `dec.enc_h.clone()`, not the tensor itself.
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
