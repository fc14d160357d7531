"""``from parts.features import FilterbankFeatures`` -> fused HIP front-end with the twin's
signature (reference parts/features.py:228-357: seconds-based ctor, ``forward(x, seq_len)``)."""
import os as _os
import sys as _sys

from edgedict_amd.features import PartsFilterbankFeatures as FilterbankFeatures  # noqa: F401

_state = {"mod": None}


def __getattr__(name):
    """Names the engine does not provide (SpectrogramFeatures, AudioPreprocessing, audio_from_file
    ...) are looked up in the reference's parts/features.py when a checkout is on sys.path."""
    if name.startswith("__"):
        raise AttributeError(name)
    if _state["mod"] is None:
        import importlib.util
        here = _os.path.abspath(__file__)
        for p in _sys.path:
            f = _os.path.join(p or ".", "parts", "features.py")
            if _os.path.isfile(f) and _os.path.abspath(f) != here:
                spec = importlib.util.spec_from_file_location("parts._reference_features", f)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _state["mod"] = mod
                break
        else:
            raise AttributeError("parts.features.%s is not provided by the MI355X engine and no "
                                 "reference checkout is on sys.path" % name)
    return getattr(_state["mod"], name)
