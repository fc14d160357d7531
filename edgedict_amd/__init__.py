"""MI355X-native RNN-Transducer engine (see README.md / DESIGN.md)."""
import os as _os

# The hot path is a chain of ~2000 dependent kernel launches per training step: kernel arguments in
# device memory shorten every launch (27.2 vs 29.2 ms per step).  PyTorch-ROCm
# enables this itself; set it for hosts that initialise HIP through this package first.  Must happen
# before the HIP runtime initialises, i.e. before the first device call of the process.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

# EDGEDICT_POISON=1 (debug): uninitialised allocations are filled with NaN / 0xA5 (see _poison.py); the module is only
# imported - and torch's factory functions only wrapped - when the switch is on
if _os.environ.get("EDGEDICT_POISON", "0") == "1":
    from . import _poison as _poison  # noqa: E402,F401
