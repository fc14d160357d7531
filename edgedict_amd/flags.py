"""Hyper-parameter presets and a reader for absl-style flagfiles.

The reference configures everything through ``absl.flags`` (rnnt/args.py:1-92) and ships its
model configurations as flagfiles (flagfiles/E4D1.txt, E6D2.txt, E6D2_LARGE_Batch.txt).  absl is
not available in the build/bench environment, so this module restates the three named presets
and parses ``--name=value`` / ``--[no]name`` files into a plain namespace whose attribute names
match the reference's ``FLAGS`` object (what ``PytorchStreamDecoder(FLAGS)`` reads,
rnnt/stream.py:29-67).
"""
import types

# defaults of the flags the hot path reads (rnnt/args.py)
_DEFAULTS = dict(
    name="rnn-t-v5", model_name="model.pt", mode="train",
    optim="adam", lr=1e-4, batch_size=8, sub_batch_size=8, eval_batch_size=4,
    sched=False, sched_patience=1, sched_factor=0.5, sched_min_lr=1e-6, warmup_step=0,
    enc_type="LSTM", enc_hidden_size=600, enc_layers=4, enc_dropout=0.0, enc_proj_size=600,
    dec_hidden_size=150, dec_layers=2, dec_dropout=0.0, dec_proj_size=150, joint_size=512,
    audio_max_length=14, feature="mfcc", feature_size=80, n_fft=400, win_length=400,
    hop_length=200, sample_rate=16000, delta=False, cmvn=False, downsample=3,
    T_mask=50, T_num_mask=2, F_mask=5, F_num_mask=1,
    tokenizer="char", bpe_size=256, vocab_embed_size=16, gradclip=None,
    apex=True, opt_level="O1", multi_gpu=False,
)

_COMMON = dict(
    optim="adam", enc_dropout=0.0, feature="logfbank", feature_size=80, n_fft=512,
    delta=False, cmvn=False, downsample=3, T_mask=50, T_num_mask=2, F_mask=5, F_num_mask=1,
    tokenizer="bpe", bpe_size=2048, vocab_embed_size=64, apex=True, opt_level="O1",
)

PRESETS = {
    # flagfiles/E4D1.txt
    "E4D1": dict(_COMMON, lr=5e-4, sched=True, sched_patience=1, sched_factor=0.5, sched_min_lr=1e-6,
                 warmup_step=10000, batch_size=32, sub_batch_size=16, eval_batch_size=2,
                 enc_hidden_size=256, enc_layers=4, enc_proj_size=256,
                 dec_hidden_size=256, dec_layers=1, dec_dropout=0.0, dec_proj_size=256,
                 joint_size=256, audio_max_length=16, win_length=320, hop_length=160),
    # flagfiles/E6D2.txt
    "E6D2": dict(_COMMON, lr=5e-4, sched=True, sched_patience=1, sched_factor=0.5, sched_min_lr=1e-6,
                 warmup_step=10000, batch_size=32, sub_batch_size=32, eval_batch_size=4,
                 enc_hidden_size=1024, enc_layers=6, enc_proj_size=640,
                 dec_hidden_size=256, dec_layers=2, dec_dropout=0.0, dec_proj_size=256,
                 joint_size=640, audio_max_length=16, win_length=320, hop_length=200),
    # flagfiles/E6D2_LARGE_Batch.txt
    "E6D2_LARGE_Batch": dict(_COMMON, lr=8e-4, sched=True, sched_patience=1, sched_factor=0.7,
                             sched_min_lr=1e-6, warmup_step=5000, batch_size=128, sub_batch_size=7,
                             eval_batch_size=4, enc_type="LSTM",
                             enc_hidden_size=1024, enc_layers=6, enc_proj_size=640,
                             dec_hidden_size=512, dec_layers=2, dec_dropout=0.1,
                             dec_proj_size=640, joint_size=640, audio_max_length=14,
                             win_length=400, hop_length=320),
}


def _coerce(old, text):
    if isinstance(old, bool):
        return text.lower() in ("1", "true", "yes")
    if isinstance(old, int) and not isinstance(old, bool):
        return int(text)
    if isinstance(old, float):
        return float(text)
    if old is None:
        try:
            return float(text)
        except ValueError:
            return text
    return text


def make_flags(preset=None, **overrides):
    """Namespace with the reference's FLAGS attribute names."""
    d = dict(_DEFAULTS)
    if preset is not None:
        d.update(PRESETS[preset])
    d.update(overrides)
    return types.SimpleNamespace(**d)


def read_flagfile(path, **overrides):
    """Parse an absl flagfile (one ``--flag=value`` / ``--flag`` / ``--noflag`` per line)."""
    d = dict(_DEFAULTS)
    with open(path) as f:
        for raw in f:
            line = raw.strip()
            if not line or line.startswith("#") or not line.startswith("--"):
                continue
            body = line[2:]
            if "=" in body:
                k, v = body.split("=", 1)
                d[k] = _coerce(d.get(k), v)
            elif body.startswith("no") and body[2:] in d and isinstance(d[body[2:]], bool):
                d[body[2:]] = False
            else:
                d[body] = True
    d.update(overrides)
    return types.SimpleNamespace(**d)


def model_kwargs(flags, vocab_size=None, input_size=None):
    """Constructor kwargs of ``Transducer`` as the reference's scripts build them
    (rnnt/stream.py:53-67, cli/train.py:113-126, cli/lightning.py:50-65)."""
    if input_size is None:
        input_size = flags.feature_size * (3 if flags.delta else 1) * max(1, flags.downsample)
    return dict(
        vocab_embed_size=flags.vocab_embed_size,
        vocab_size=vocab_size if vocab_size is not None else flags.bpe_size,
        input_size=input_size,
        enc_hidden_size=flags.enc_hidden_size, enc_layers=flags.enc_layers,
        enc_dropout=flags.enc_dropout, enc_proj_size=flags.enc_proj_size,
        dec_hidden_size=flags.dec_hidden_size, dec_layers=flags.dec_layers,
        dec_dropout=flags.dec_dropout, dec_proj_size=flags.dec_proj_size,
        joint_size=flags.joint_size,
        module_type=getattr(flags, "enc_type", "LSTM"),      # cli/lightning.py:63
    )
