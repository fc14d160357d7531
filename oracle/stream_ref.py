"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement of the reference's single-stream online decoder,
``PytorchStreamDecoder.reset/decode`` (rnnt/stream.py:78-120), on top of
oracle.models_ref / oracle.features_ref.  Token *ids* are returned instead of text so parity can
be checked bit-exactly without a BPE vocabulary.

PARITY STATUS: **pinned**.  ``oracle/make_golden_stream.py`` EXECUTES the reference's own
``PytorchStreamDecoder.reset`` / ``.decode`` (lifted from rnnt/stream.py with ``ast``; its
``__init__`` - flags, checkpoint, vocabulary - bypassed) on the reference ``Transducer``'s
sub-modules and the reference's own feature classes, stores the returned texts in
``tests/golden/stream.npz`` and asserts this restatement emits the same ids, chunk by chunk, for
single streams, resets and independent multi-stream runs; ``tests/test_oracle_stream.py``
re-checks it without the reference.
"""
import torch

from . import features_ref as Fr
from . import models_ref as M


class StreamOracle:
    def __init__(self, sd, flags, unk_id=M.UNK):
        self.sd, self.flags, self.unk = sd, flags, unk_id
        self.L = M.n_enc_layers(sd)
        self.H = sd["encoder.lstm.lstms.0.weight_hh_l0"].shape[1]
        self.reset()

    def reset(self):
        self.enc_h = torch.zeros(self.L, 1, self.H)
        self.enc_c = torch.zeros(self.L, 1, self.H)
        Ld = M.n_dec_layers(self.sd)
        Hd = self.sd["decoder.lstm.weight_hh_l0"].shape[1]
        bos = torch.full((1, 1), M.BOS, dtype=torch.long)
        self.dec_x, (self.dec_h, self.dec_c) = M.decoder_forward(
            self.sd, bos, (torch.zeros(Ld, 1, Hd), torch.zeros(Ld, 1, Hd)))

    @torch.no_grad()
    def decode(self, frame):
        f = self.flags
        xs = Fr.stacked_features(frame, f.downsample, False, win_length=f.win_length,
                                 hop_length=f.hop_length, n_fft=f.n_fft, n_filt=f.feature_size)
        enc, (self.enc_h, self.enc_c) = M.encoder_forward(self.sd, xs, (self.enc_h, self.enc_c))
        out = []
        for k in range(enc.shape[1]):
            logits = M.joint_forward(self.sd, enc[:, k], self.dec_x[:, 0])
            pred = int(logits.argmax(dim=-1))
            if pred == self.unk:
                logits[:, pred] = 0
                pred = int(logits.argmax(dim=-1))
            out.append(pred)
            if pred != M.NUL:
                tok = torch.full((1, 1), pred, dtype=torch.long)
                self.dec_x, (self.dec_h, self.dec_c) = M.decoder_forward(
                    self.sd, tok, (self.dec_h, self.dec_c))
        return out
