"""Drop-in import shim: ``parts.features.FilterbankFeatures`` (the Jasper-derived twin of the
log-mel front-end, reference parts/features.py:228-357) resolves to the MI355X engine when this
repository precedes the reference checkout on ``sys.path``.  Every other module
of the reference's ``parts`` package (``parts.text.cleaners`` - imported by stream.py:12 and
modules/tokenizer.py:5 - ``parts.segment``, ``parts.perturb``, ``parts.manifest``) stays
importable: the other ``parts`` directories on ``sys.path`` are appended to this package's search
path, so modules present here win and everything else falls through to the reference."""
import os as _os
import sys as _sys

_here = _os.path.abspath(_os.path.dirname(__file__))
for _p in list(_sys.path):
    _d = _os.path.join(_p or ".", "parts")
    if _os.path.isdir(_d) and _os.path.abspath(_d) != _here and _d not in __path__:
        __path__.append(_d)
