"""Generate tests/golden/*.npz by running the REFERENCE's own rnnt.models.Transducer.

Run in the build container only (needs /root/reference; the GPU box does not have it):

    python oracle/make_golden.py

For each case the script (1) builds deterministic weights/inputs with oracle.models_ref,
(2) loads them into the reference module imported from /root/reference, (3) records the
reference's logits / scaled lengths / greedy tokens, (4) asserts that the oracle restatement
reproduces them, and (5) adds the float64 RNN-T loss of the reference logits computed by
oracle.rnnt_loss_ref (the reference's own loss op, warprnnt_pytorch, is not installable here).
Only small arrays are stored (full tensors for the tiny case, strided samples for E4D1).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import models_ref as M          # noqa: E402
from oracle import rnnt_loss_ref as R       # noqa: E402

CASES = {
    # name: (cfg, B, T0, U, seed)
    "tiny": (dict(vocab_embed_size=8, vocab_size=40, input_size=24, enc_hidden_size=32,
                  enc_layers=3, enc_proj_size=24, dec_hidden_size=16, dec_layers=2,
                  dec_proj_size=16, joint_size=32), 3, 11, 5, 101),
    # module_type='GRU' encoder variant (ResLayerNormGRU, rnnt/models.py:77-116)
    "gru_tiny": (dict(vocab_embed_size=8, vocab_size=40, input_size=24, enc_hidden_size=32,
                      enc_layers=3, enc_proj_size=24, dec_hidden_size=16, dec_layers=2,
                      dec_proj_size=16, joint_size=32, module_type="GRU"), 3, 11, 5, 202),
    "E4D1": (dict(vocab_embed_size=64, vocab_size=2048, input_size=240, enc_hidden_size=256,
                  enc_layers=4, enc_proj_size=256, dec_hidden_size=256, dec_layers=1,
                  dec_proj_size=256, joint_size=256), 4, 167, 20, 0),
    # the BENCHED configuration (BASELINE config 2; flagfiles/E6D2.txt) at full model size and full
    # 15 s length (T0 = 401 stacked frames -> T' = 201, U = 64), two utterances
    "E6D2": (dict(vocab_embed_size=64, vocab_size=2048, input_size=240, enc_hidden_size=1024,
                  enc_layers=6, enc_proj_size=640, dec_hidden_size=256, dec_layers=2,
                  dec_proj_size=256, joint_size=640), 2, 401, 64, 7),
    # BASELINE config 3's model (flagfiles/E6D2_LARGE_Batch.txt: prediction net 2x512 -> 640,
    # hop 320 -> T0 = 251 at 15 s), eval mode (dec_dropout inactive)
    "E6D2_LARGE": (dict(vocab_embed_size=64, vocab_size=2048, input_size=240, enc_hidden_size=1024,
                        enc_layers=6, enc_proj_size=640, dec_hidden_size=512, dec_layers=2,
                        dec_proj_size=640, joint_size=640), 2, 251, 64, 9),
}


def reference_model(cfg, sd):
    # the repository root holds an `rnnt` shim package of the same name, and the reference's `rnnt`
    # is a namespace package (no __init__.py), which loses against a regular package wherever it
    # sits on sys.path: load the reference's modules from their files under the name `rnnt`
    import importlib.util
    import types
    for k in [k for k in sys.modules if k == "rnnt" or k.startswith("rnnt.")]:
        del sys.modules[k]
    if REF not in sys.path:
        sys.path.append(REF)          # `modules.*` (rnnt/models.py:14) lives at the reference's root
    pkg = types.ModuleType("rnnt")
    pkg.__path__ = [os.path.join(REF, "rnnt")]
    sys.modules["rnnt"] = pkg
    for sub in ("tokenizer", "models"):
        spec = importlib.util.spec_from_file_location("rnnt." + sub, os.path.join(REF, "rnnt", sub + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["rnnt." + sub] = mod
        spec.loader.exec_module(mod)
    ref = sys.modules["rnnt.models"]
    assert ref.__file__.startswith(REF), ref.__file__
    m = ref.Transducer(enc_dropout=0.0, dec_dropout=0.0, output_loss=False, **cfg)
    missing = m.load_state_dict(sd, strict=True)
    m.eval()
    return m


def main(only=None):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for name, (cfg, B, T0, U, seed) in CASES.items():
        if only and name not in only:
            continue
        sd = M.make_state_dict(cfg, seed)
        xs, ys, xlen, ylen = M.make_batch(cfg, seed + 1, B, T0, U)
        ref = reference_model(cfg, sd)
        with torch.no_grad():
            logits = ref(xs, ys, xlen, ylen)
            act_lens = ref.scale_length(logits, xlen)
            tokens, score = ref.greedy_decode(xs, xlen)
            enc, hid = ref.encoder(xs)
            hs, cs = (hid, hid) if cfg.get("module_type") == "GRU" else hid   # GRU: one state tensor
            dec, _ = ref.decoder(ys)
            # oracle restatement must reproduce the reference
            o_logits, o_lens = M.transducer_logits(sd, xs, ys, xlen, ylen)
            o_tokens, o_score = M.greedy_decode(sd, xs, xlen)
        err = (o_logits - logits).abs().max().item()
        assert err < 2e-5, (name, err)
        assert torch.equal(o_lens, act_lens)
        for a, b in zip(tokens, o_tokens):
            assert np.array_equal(a, b), name
        assert (o_score - score).abs().max().item() < 1e-3
        costs, _ = R.rnnt_loss(logits.double().numpy(), ys.numpy(), act_lens.numpy(),
                               ylen.numpy(), want_grads=False)
        out = dict(
            seed=np.int64(seed), B=np.int64(B), T0=np.int64(T0), U=np.int64(U),
            xlen=xlen.numpy(), ylen=ylen.numpy(), act_lens=act_lens.numpy(),
            costs=costs, loss_mean=np.float64(costs.mean()),
            greedy_tokens=np.stack([np.pad(t, (0, logits.shape[1] - len(t)), constant_values=-1)
                                    for t in tokens]).astype(np.int64),
            greedy_score=score.numpy(),
            enc_hN=hs.numpy()[:, :, :8], enc_cN=cs.numpy()[:, :, :8],
        )
        if name in ("tiny", "gru_tiny"):
            out.update(logits=logits.numpy(), enc_out=enc.numpy(), dec_out=dec.numpy())
        else:
            out.update(logits_sample=logits.numpy()[:, ::7, ::3, ::64],
                       enc_out_sample=enc.numpy()[:, ::5, ::16],
                       dec_out_sample=dec.numpy()[:, ::2, ::16])
        path = os.path.join(outdir, "transducer_%s.npz" % name)
        np.savez_compressed(path, **out)
        print("%s: reference vs oracle max |dlogits| = %.2e, loss_mean = %.6f, wrote %s (%d KB)"
              % (name, err, costs.mean(), path, os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main(sys.argv[1:])
