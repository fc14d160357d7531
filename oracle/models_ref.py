"""ORACLE (test infrastructure only — never imported by the product path).

CPU restatement, in functional PyTorch, of the reference's RNN-T model arithmetic:

    encoder          rnnt/models.py:119-136 (Encoder), :32-75 (ResLayerNormLSTM), :16-29 (TimeReduction)
    prediction net   rnnt/models.py:139-157 (Decoder)
    joint            rnnt/models.py:160-179 (Joint)
    scale_length     rnnt/models.py:223-226
    forward          rnnt/models.py:228-241 (logits; the loss comes from oracle/rnnt_loss_ref.py)
    greedy decode    rnnt/models.py:243-269
    stream decode    rnnt/stream.py:78-120

All functions take a plain ``state_dict`` with the reference's key names, so the same weights
drive the reference module (when /root/reference is importable), this oracle and the HIP engine.

PARITY STATUS: **pinned** — ``oracle/make_golden.py`` runs the reference's own
``rnnt.models.Transducer`` (imported from /root/reference in the build container) on seeded
inputs, checks this restatement against it and stores the reference's outputs under
``tests/golden/``; ``tests/test_oracle_models.py`` re-checks the restatement against those files.
"""
import math

import torch

NUL, PAD, BOS, UNK = 0, 1, 2, 3


# ---------------------------------------------------------------- deterministic weights / inputs
def param_shapes(cfg):
    """Ordered (name, shape) list of the reference Transducer's state_dict for a config dict with
    keys vocab_embed_size, vocab_size, input_size, enc_hidden_size, enc_layers, enc_proj_size,
    dec_hidden_size, dec_layers, dec_proj_size, joint_size."""
    H, L, I = cfg["enc_hidden_size"], cfg["enc_layers"], cfg["input_size"]
    ng = 3 if cfg.get("module_type", "LSTM") == "GRU" else 4
    out = [("encoder.norm.weight", (I,)), ("encoder.norm.bias", (I,))]
    for i in range(L):
        isz = I if i == 0 else H
        out += [("encoder.lstm.lstms.%d.weight_ih_l0" % i, (ng * H, isz)),
                ("encoder.lstm.lstms.%d.weight_hh_l0" % i, (ng * H, H)),
                ("encoder.lstm.lstms.%d.bias_ih_l0" % i, (ng * H,)),
                ("encoder.lstm.lstms.%d.bias_hh_l0" % i, (ng * H,)),
                ("encoder.lstm.projs.%d.0.weight" % i, (H,)),
                ("encoder.lstm.projs.%d.0.bias" % i, (H,))]
    out += [("encoder.proj.weight", (cfg["enc_proj_size"], H)),
            ("encoder.proj.bias", (cfg["enc_proj_size"],)),
            ("decoder.embed.weight", (cfg["vocab_size"], cfg["vocab_embed_size"]))]
    Hd = cfg["dec_hidden_size"]
    for k in range(cfg["dec_layers"]):
        isz = cfg["vocab_embed_size"] if k == 0 else Hd
        out += [("decoder.lstm.weight_ih_l%d" % k, (4 * Hd, isz)),
                ("decoder.lstm.weight_hh_l%d" % k, (4 * Hd, Hd)),
                ("decoder.lstm.bias_ih_l%d" % k, (4 * Hd,)),
                ("decoder.lstm.bias_hh_l%d" % k, (4 * Hd,))]
    P = cfg["enc_proj_size"] + cfg["dec_proj_size"]
    out += [("decoder.proj.weight", (cfg["dec_proj_size"], Hd)),
            ("decoder.proj.bias", (cfg["dec_proj_size"],)),
            ("joint.joint.0.weight", (cfg["joint_size"], P)),
            ("joint.joint.0.bias", (cfg["joint_size"],)),
            ("joint.joint.2.weight", (cfg["vocab_size"], cfg["joint_size"])),
            ("joint.joint.2.bias", (cfg["vocab_size"],))]
    return out


def make_state_dict(cfg, seed, dtype=torch.float32):
    """Deterministic synthetic weights (CPU generator, so identical on every box with this
    torch build): U(-1/sqrt(fan_in), 1/sqrt(fan_in)) matrices/biases, LayerNorm gains around 1,
    embedding N(0,1) with a zero PAD row.  Not the reference's init — the point is that the
    reference module, the oracle and the engine all load the SAME tensors."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg):
        if name.endswith("norm.weight") or ".projs." in name and name.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("norm.bias") or ".projs." in name and name.endswith("bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif name == "decoder.embed.weight":
            t = torch.randn(shape, generator=g)
            t[PAD] = 0
        else:
            if "lstm" in name:
                fan = cfg["enc_hidden_size"] if name.startswith("encoder") else cfg["dec_hidden_size"]
            else:
                fan = shape[-1] if len(shape) == 2 else None
                if fan is None:  # Linear bias: fan_in of its weight
                    fan = dict(param_shapes(cfg))[name.replace("bias", "weight")][-1]
            k = 1.0 / math.sqrt(fan)
            t = (torch.rand(shape, generator=g) * 2 - 1) * k
        sd[name] = t.to(dtype)
    return sd


def make_batch(cfg, seed, B, T0, U, ragged=True):
    """Synthetic (xs f32[B,T0,I], ys i32[B,U], xlen i32[B], ylen i32[B]) in seq_collate layout
    (rnnt/dataset.py:225-240): ids drawn from [4, V), first row full length."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    xs = torch.randn(B, T0, cfg["input_size"], generator=g)
    ys = torch.randint(4, cfg["vocab_size"], (B, U), generator=g, dtype=torch.int32)
    if ragged:
        xlen = torch.randint(max(1, (3 * T0) // 4), T0 + 1, (B,), generator=g, dtype=torch.int32)
        ylen = torch.randint(max(1, U // 2), U + 1, (B,), generator=g, dtype=torch.int32)
    else:
        xlen = torch.full((B,), T0, dtype=torch.int32)
        ylen = torch.full((B,), U, dtype=torch.int32)
    xlen[0], ylen[0] = T0, U
    for b in range(B):
        xs[b, xlen[b]:] = 0          # seq_collate zero-pads features ...
        ys[b, ylen[b]:] = PAD        # ... and PAD-pads labels
    return xs, ys, xlen, ylen


# ---------------------------------------------------------------- building blocks
def layer_norm(x, w, b, eps=1e-5):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w + b


def time_reduction(x, factor=2):
    """rnnt/models.py:21-29: zero-pad T to a multiple of ``factor`` at the end, mean of groups."""
    B, T, H = x.shape
    pad = (factor - T % factor) % factor
    if pad:
        x = torch.cat([x, x.new_zeros(B, pad, H)], dim=1)
    return x.reshape(B, -1, factor, H).mean(dim=2)


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, h0, c0, explicit=False):
    """One batch_first LSTM layer, PyTorch cell semantics (gate order i,f,g,o).
    ``explicit=True`` spells the recurrence out step by step; otherwise the same arithmetic runs
    through torch's CPU LSTM primitive (what the reference's nn.LSTM executes on CPU)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    if h0 is None:
        h0 = x.new_zeros(B, H)
    if c0 is None:
        c0 = x.new_zeros(B, H)
    if not explicit:
        out, h, c = torch._VF.lstm(x, (h0[None], c0[None]), [w_ih, w_hh, b_ih, b_hh], True, 1,
                                   0.0, False, False, True)
        return out, h[0], c[0]
    h, c = h0, c0
    ys = []
    for t in range(T):
        pre = x[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, g, o = pre.split(H, dim=1)
        i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
        c = f * c + i * g
        h = o * torch.tanh(c)
        ys.append(h)
    return torch.stack(ys, 1), h, c


def gru_layer(x, w_ih, w_hh, b_ih, b_hh, h0):
    """One batch_first GRU layer, PyTorch cell semantics (gate order r,z,n), spelled out step by
    step - what nn.GRU computes for ResLayerNormGRU (rnnt/models.py:87-88,107)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H) if h0 is None else h0
    ys = []
    for t in range(T):
        gi = x[:, t] @ w_ih.t() + b_ih
        gh = h @ w_hh.t() + b_hh
        ir, iz, inn = gi.split(H, dim=1)
        hr, hz, hn = gh.split(H, dim=1)
        r = torch.sigmoid(ir + hr)
        z = torch.sigmoid(iz + hz)
        n = torch.tanh(inn + r * hn)
        h = (1 - z) * n + z * h
        ys.append(h)
    return torch.stack(ys, 1), h


def is_gru(sd):
    """module_type='GRU' checkpoints keep the LSTM variant's key names with 3H-row matrices."""
    w = sd["encoder.lstm.lstms.0.weight_hh_l0"]
    return w.shape[0] == 3 * w.shape[1]


def encoder_forward_gru(sd, xs, hiddens=None, time_reductions=(1,)):
    """rnnt/models.py:131-136 + :99-116 (ResLayerNormGRU).  Returns (ys [B,T',P], hs [L,B,H])."""
    x = layer_norm(xs, sd["encoder.norm.weight"], sd["encoder.norm.bias"])
    hs = []
    for i in range(n_enc_layers(sd)):
        p = "encoder.lstm.lstms.%d." % i
        y, h = gru_layer(x, sd[p + "weight_ih_l0"], sd[p + "weight_hh_l0"], sd[p + "bias_ih_l0"],
                         sd[p + "bias_hh_l0"], None if hiddens is None else hiddens[i])
        x = y if i == 0 else x + y
        x = layer_norm(x, sd["encoder.lstm.projs.%d.0.weight" % i],
                       sd["encoder.lstm.projs.%d.0.bias" % i])
        if i in time_reductions:
            x = time_reduction(x)
        hs.append(h)
    y = x @ sd["encoder.proj.weight"].t() + sd["encoder.proj.bias"]
    return y, torch.stack(hs)


def n_enc_layers(sd):
    n = 0
    while "encoder.lstm.lstms.%d.weight_ih_l0" % n in sd:
        n += 1
    return n


def n_dec_layers(sd):
    n = 0
    while "decoder.lstm.weight_ih_l%d" % n in sd:
        n += 1
    return n


def encoder_forward(sd, xs, hiddens=None, time_reductions=(1,), explicit=False):
    """rnnt/models.py:131-136 + :55-75.  Returns (ys [B,T',P], (hs, cs) [L,B,H]); for a GRU
    state dict (module_type='GRU') the second item is hs alone, as in the reference."""
    if is_gru(sd):
        return encoder_forward_gru(sd, xs, hiddens, time_reductions)
    x = layer_norm(xs, sd["encoder.norm.weight"], sd["encoder.norm.bias"])
    hs, cs = [], []
    for i in range(n_enc_layers(sd)):
        p = "encoder.lstm.lstms.%d." % i
        h0 = c0 = None
        if hiddens is not None:
            h0, c0 = hiddens[0][i], hiddens[1][i]
        y, h, c = lstm_layer(x, sd[p + "weight_ih_l0"], sd[p + "weight_hh_l0"],
                             sd[p + "bias_ih_l0"], sd[p + "bias_hh_l0"], h0, c0, explicit)
        x = y if i == 0 else x + y
        x = layer_norm(x, sd["encoder.lstm.projs.%d.0.weight" % i],
                       sd["encoder.lstm.projs.%d.0.bias" % i])
        if i in time_reductions:
            x = time_reduction(x)
        hs.append(h)
        cs.append(c)
    y = x @ sd["encoder.proj.weight"].t() + sd["encoder.proj.bias"]
    return y, (torch.stack(hs), torch.stack(cs))


def decoder_forward(sd, ys, hidden=None, explicit=False):
    """rnnt/models.py:150-157.  ``hidden is None`` => training mode: BOS is left-padded."""
    ys = ys.long()
    if hidden is None:
        ys = torch.cat([torch.full((ys.shape[0], 1), BOS, dtype=torch.long), ys], dim=1)
    x = sd["decoder.embed.weight"][ys]
    hs, cs = [], []
    for k in range(n_dec_layers(sd)):
        h0 = c0 = None
        if hidden is not None:
            h0, c0 = hidden[0][k], hidden[1][k]
        x, h, c = lstm_layer(x, sd["decoder.lstm.weight_ih_l%d" % k],
                             sd["decoder.lstm.weight_hh_l%d" % k],
                             sd["decoder.lstm.bias_ih_l%d" % k],
                             sd["decoder.lstm.bias_hh_l%d" % k], h0, c0, explicit)
        hs.append(h)
        cs.append(c)
    y = x @ sd["decoder.proj.weight"].t() + sd["decoder.proj.bias"]
    return y, (torch.stack(hs), torch.stack(cs))


def joint_forward(sd, h_enc, h_dec):
    """rnnt/models.py:169-179: concat -> Linear -> Tanh -> Linear, raw logits."""
    if h_enc.dim() == 3 and h_dec.dim() == 3:
        T, U1 = h_enc.shape[1], h_dec.shape[1]
        h_enc = h_enc[:, :, None, :].expand(-1, -1, U1, -1)
        h_dec = h_dec[:, None, :, :].expand(-1, T, -1, -1)
    h = torch.cat([h_enc, h_dec], dim=-1)
    h = torch.tanh(h @ sd["joint.joint.0.weight"].t() + sd["joint.joint.0.bias"])
    return h @ sd["joint.joint.2.weight"].t() + sd["joint.joint.2.bias"]


def scale_length(T_out, xlen):
    """rnnt/models.py:223-226."""
    scale = (xlen.max().float() / T_out).ceil()
    return (xlen / scale).ceil().int()


def transducer_logits(sd, xs, ys, xlen, ylen, time_reductions=(1,), explicit=False):
    """rnnt/models.py:228-241 with output_loss=False; also returns the scaled act_lens."""
    xs = xs[:, :int(xlen.max())]
    ys = ys[:, :int(ylen.max())]
    h_enc, _ = encoder_forward(sd, xs, None, time_reductions, explicit)
    h_dec, _ = decoder_forward(sd, ys, None, explicit)
    logits = joint_forward(sd, h_enc, h_dec)
    return logits, scale_length(logits.shape[1], xlen)


def greedy_decode(sd, xs, xlen, blank=NUL, time_reductions=(1,)):
    """rnnt/models.py:243-269: at most one symbol per encoder frame; blanks stay in the output;
    the prediction network advances for every row and the new state is kept only where the
    emitted symbol is not blank; rows are truncated to the UN-scaled xlen."""
    h_enc, _ = encoder_forward(sd, xs, None, time_reductions)
    B = xs.shape[0]
    h_dec, (h, c) = decoder_forward(sd, torch.zeros(B, 0, dtype=torch.long), None)
    seq, logp = [], []
    for t in range(h_enc.shape[1]):
        logits = joint_forward(sd, h_enc[:, t], h_dec[:, 0])
        lp = torch.log_softmax(logits, dim=1)
        best, pred = lp.max(dim=1)
        seq.append(pred)
        logp.append(best)
        h_dec_new, (h_new, c_new) = decoder_forward(sd, pred[:, None], (h, c))
        keep = pred != blank
        h_dec = torch.where(keep[:, None, None], h_dec_new, h_dec)
        h = torch.where(keep[None, :, None], h_new, h)
        c = torch.where(keep[None, :, None], c_new, c)
    seq = torch.stack(seq, dim=1)
    score = -torch.stack(logp, dim=1).sum(dim=1)
    return [s[:int(n)].numpy() for s, n in zip(seq, xlen)], score
