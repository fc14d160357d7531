"""EDGEDICT_POISON=1: debug mode that fills every tensor the host code allocates UNINITIALISED with a poison pattern.

Every output, workspace and scratch buffer of the hot path is a ``torch.empty`` / ``empty_like`` / ``new_empty`` block
that a native call is expected to overwrite (or to initialise itself, where it needs zeros).  In normal operation a
kernel that reads such a block before writing it - or a late writer of a block the caching allocator has already handed
to somebody else - sees whatever the previous owner left there: usually plausible numbers, occasionally not, i.e. a
flake.  With the poison on, those reads see NaN (floating types), 0xA5 bytes (uint8 workspaces) or a large negative
number (integers), so the defect fails EVERY run instead of one in thirty.  The GPU suite is run once this way per
round (tools/gpu_flake.sh poison; DESIGN.md "Stability of the GPU suite").

Not a numerics mode: the fill is an extra eager kernel per allocation; never enable it for measurements.
"""
import os

import torch

ON = os.environ.get("EDGEDICT_POISON", "0") == "1"


def _fill(t):
    if isinstance(t, torch.Tensor) and t.is_cuda and t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(0xA5)
        elif t.dtype == torch.bool:
            t.fill_(True)
        elif not t.is_complex():
            t.fill_(-1515870811)        # 0xA5A5A5A5 as int32; wraps the same way in the narrower types
    return t


def install():
    if getattr(torch, "_edgedict_poisoned", False):
        return
    torch._edgedict_poisoned = True
    empty, empty_like, new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def p_empty(*a, **k):
        # torch.set_default_device works through a function mode that recognises the ORIGINAL factory functions: behind
        # this wrapper it would no longer see `torch.empty`, so the default device is passed on explicitly
        # (only where nothing else already fixes the placement: `out=` carries its own device, pinned staging buffers
        # are host memory by definition; torch < 2.3 has no get_default_device)
        if (k.get("device") is None and k.get("out") is None and not k.get("pin_memory")
                and hasattr(torch, "get_default_device")):          # (nn.Embedding & co. pass device=None explicitly)
            k["device"] = torch.get_default_device()
        return _fill(empty(*a, **k))

    def p_empty_like(*a, **k):
        return _fill(empty_like(*a, **k))

    def p_new_empty(self, *a, **k):
        return _fill(new_empty(self, *a, **k))
    torch.empty, torch.empty_like, torch.Tensor.new_empty = p_empty, p_empty_like, p_new_empty


if ON:
    install()
