"""CPU: the import surface the reference's scripts rely on resolves to the engine when this
repository precedes the reference on sys.path, without a GPU."""
import importlib

import pytest


@pytest.mark.parametrize("module,names", [
    ("rnnt.models", ["Transducer", "Encoder", "ResLayerNormLSTM", "ResLayerNormGRU", "FrontEnd",
                     "convert_lightning2normal"]),                 # cli/train.py:18, rnnt/stream.py:10, rnnt/wav2vec.py:12
    ("rnnt.stream", ["PytorchStreamDecoder"]),                    # stream.py:17
    ("rnnt.transforms", ["build_transform"]),                     # cli/baseline.py, rnnt/stream.py:11
    ("rnnt.features", ["FilterbankFeatures"]),                    # rnnt/transforms.py:7
    ("rnnt.dataset", ["seq_collate", "zero_pad_concat", "end_pad_concat"]),   # cli/train.py:17
    ("rnnt.tokenizer", ["NUL", "PAD", "BOS"]),                    # rnnt/models.py:13
    ("warprnnt_pytorch", ["RNNTLoss"]),                           # rnnt/models.py:9
])
def test_reference_import_sites_resolve(module, names):
    m = importlib.import_module(module)
    for n in names:
        assert hasattr(m, n), (module, n)
    assert "edgedict_amd" in (getattr(m, names[0]).__module__ if callable(getattr(m, names[0])) else "edgedict_amd")


def test_out_of_path_modules_fail_loudly_not_silently():
    from rnnt.models import CTCEncoder, FrontEnd, ResLayerNormGRU, Transducer
    fe = FrontEnd()          # implemented (csrc/frontend.hip); reference state-dict layout
    assert fe.state_dict()["encode.0.conv.weight"].shape == (32, 16, 8)
    with pytest.raises(NotImplementedError, match="CTC"):
        CTCEncoder(40, 24, 32, 2, 0.0, 24)
    # the GRU variant IS implemented (csrc/gru.hip): same state-dict key names, 3H-row matrices
    t = Transducer(16, 40, 24, 32, 2, 0.0, 24, 32, 1, 0.0, 24, 32, module_type="GRU")
    assert isinstance(t.encoder.lstm, ResLayerNormGRU)
    assert t.state_dict()["encoder.lstm.lstms.1.weight_hh_l0"].shape == (96, 32)
    with pytest.raises(ValueError):
        Transducer(16, 40, 24, 32, 2, 0.0, 24, 32, 1, 0.0, 24, 32, module_type="RNN")   # rnnt/models.py:191-192
