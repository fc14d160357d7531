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
    ("parts.features", ["FilterbankFeatures"]),                   # the Jasper-derived twin, parts/features.py:228
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


def test_unshimmed_names_fall_through_to_a_reference_checkout(tmp_path):
    """The reference's `rnnt` is a namespace package; this repository's regular `rnnt` package would
    shadow it completely.  Modules and names the shim does not provide must still resolve to a
    reference checkout that FOLLOWS this repository on sys.path (a fake one here)."""
    import subprocess
    import sys
    ref = tmp_path / "ref" / "rnnt"
    ref.mkdir(parents=True)
    (ref / "args.py").write_text("FLAGS = 'reference flags'\n")
    (ref / "tokenizer.py").write_text("NUL = 99\nclass HuggingFaceTokenizer:\n    origin = 'reference'\n")
    (ref / "dataset.py").write_text("from rnnt.tokenizer import PAD\nclass Librispeech:\n    pad = PAD\n")
    (ref / "models.py").write_text("class Transducer:\n    origin = 'reference'\n")
    repo = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import rnnt.args, rnnt.models\n"
        "from rnnt.tokenizer import HuggingFaceTokenizer, NUL\n"
        "from rnnt.dataset import seq_collate, Librispeech\n"
        "assert rnnt.args.FLAGS == 'reference flags'\n"
        "assert HuggingFaceTokenizer.origin == 'reference' and NUL == 0\n"          # constants: the engine's
        "assert seq_collate.__module__ == 'edgedict_amd.collate' and Librispeech.pad == 1\n"
        "assert rnnt.models.Transducer.__module__ == 'edgedict_amd.models'\n"        # shimmed modules win
        "print('ok')\n" % (repo, str(tmp_path / "ref")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-800:]


def test_parts_shim_keeps_the_rest_of_the_reference_package_importable(tmp_path):
    """stream.py:12 and modules/tokenizer.py:5 import parts.text.cleaners; the `parts` shim must not
    hide it (nor parts.segment etc.), while parts.features.FilterbankFeatures is the engine's twin
    and other names of that module fall through to the reference file."""
    import os
    import subprocess
    import sys
    ref = tmp_path / "ref" / "parts"
    (ref / "text").mkdir(parents=True)
    (ref / "__init__.py").write_text("")
    (ref / "text" / "__init__.py").write_text("")
    (ref / "text" / "cleaners.py").write_text("def english_cleaners(t):\n    return t.lower()\n")
    (ref / "segment.py").write_text("class AudioSegment:\n    origin = 'reference'\n")
    (ref / "features.py").write_text("class SpectrogramFeatures:\n    origin = 'reference'\n"
                                     "class FilterbankFeatures:\n    origin = 'reference'\n")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "from parts.text.cleaners import english_cleaners\n"
        "from parts.segment import AudioSegment\n"
        "from parts.features import FilterbankFeatures, SpectrogramFeatures\n"
        "assert english_cleaners('AB') == 'ab' and AudioSegment.origin == 'reference'\n"
        "assert FilterbankFeatures.__module__ == 'edgedict_amd.features'\n"
        "assert SpectrogramFeatures.origin == 'reference'\n"
        "print('ok')\n" % (repo, str(tmp_path / "ref")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-800:]
