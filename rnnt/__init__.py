"""Drop-in import shim: ``rnnt.models`` / ``rnnt.stream`` / ``rnnt.features`` / ``rnnt.transforms`` / ``rnnt.dataset``
/ ``rnnt.tokenizer`` resolve to the MI355X engine when this repository precedes the reference
checkout on ``sys.path``.

The reference's own ``rnnt`` directory is a namespace package (no ``__init__.py``) and would be
shadowed completely by this regular package; its remaining modules (``rnnt.args``,
``rnnt.data_utils``, ``rnnt.wav2vec``, the corpus readers ...) are kept importable by appending every
other ``rnnt`` directory found on ``sys.path`` to this package's search path - modules present here
win, everything else falls through to the reference.
"""
import os as _os
import sys as _sys

_here = _os.path.abspath(_os.path.dirname(__file__))
for _p in list(_sys.path):
    _d = _os.path.join(_p or ".", "rnnt")
    if _os.path.isdir(_d) and _os.path.abspath(_d) != _here and _d not in __path__:
        __path__.append(_d)


def _reference_fallback(modname, here_file):
    """``__getattr__`` for a shim module: names it does not define are looked up in the reference's
    module of the same name (loaded from its file under a private name), so
    ``from rnnt.dataset import seq_collate, Librispeech`` gets the engine's collate function and the
    reference's corpus reader."""
    state = {"mod": None}

    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        if state["mod"] is None:
            import importlib.util
            here = _os.path.abspath(here_file)
            for p in _sys.path:
                f = _os.path.join(p or ".", "rnnt", modname + ".py")
                if _os.path.isfile(f) and _os.path.abspath(f) != here:
                    spec = importlib.util.spec_from_file_location("rnnt._reference_" + modname, f)
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                    state["mod"] = mod
                    break
            else:
                raise AttributeError("rnnt.%s.%s is not provided by the MI355X engine and no reference "
                                     "checkout is on sys.path" % (modname, name))
        return getattr(state["mod"], name)
    return __getattr__
