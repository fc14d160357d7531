"""CPU: hyper-parameter presets and the flagfile reader (edgedict_amd/flags.py) against the
reference's shipped flagfiles (tests/golden/flagfiles.json, oracle/make_golden_flags.py)."""
import json
import os

import pytest

from edgedict_amd import flags as F

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "flagfiles.json")))


def _value(raw, like):
    if raw is True:
        return True
    if isinstance(like, bool):
        return raw.lower() in ("1", "true", "yes")
    if isinstance(like, int):
        return int(raw)
    if isinstance(like, float):
        return float(raw)
    return raw


@pytest.mark.parametrize("name", sorted(GOLD))
def test_presets_restate_the_reference_flagfiles(name):
    preset = F.make_flags(name)
    for k, raw in GOLD[name].items():
        key, val = k, raw
        if raw is True and k.startswith("no") and hasattr(preset, k[2:]):
            key, val = k[2:], False            # absl's --noflag
        if not hasattr(preset, key):
            continue                            # flags the hot path does not read (epochs, save_step, ...)
        got = getattr(preset, key)
        want = val if isinstance(val, bool) else _value(val, got)
        assert got == want, (name, key, got, want)


@pytest.mark.parametrize("name", sorted(GOLD))
def test_flagfile_reader_round_trip(tmp_path, name):
    p = tmp_path / (name + ".txt")
    p.write_text("\n".join(("--%s" % k) if v is True else ("--%s=%s" % (k, v)) for k, v in GOLD[name].items()) + "\n")
    got = F.read_flagfile(str(p))
    preset = F.make_flags(name)
    for k in vars(preset):
        assert getattr(got, k) == getattr(preset, k), (name, k)


def test_model_kwargs_follow_the_reference_constructors():
    f = F.make_flags("E6D2")
    kw = F.model_kwargs(f)
    assert kw["input_size"] == 240 and kw["vocab_size"] == 2048 and kw["enc_layers"] == 6
    assert kw["module_type"] == "LSTM"                       # cli/lightning.py:63 passes FLAGS.enc_type
    assert F.model_kwargs(F.make_flags("E6D2", enc_type="GRU"))["module_type"] == "GRU"
    assert F.model_kwargs(f, vocab_size=1000, input_size=128)["input_size"] == 128
