"""Off-critical-path work on the library's low-priority stream.

Weight gradients never feed the rest of the backward pass, so they do not have to run where
autograd runs: ``deferred(...)`` executes a block on the library's weight-gradient stream
(``edgedict_aux_stream(2)``, the same one the encoder stack uses — no additional HIP stream is
created, see include/edgedict_hip.h), after everything enqueued so far on the current stream, and
registers ONE autograd end-of-backward callback that makes the current stream wait for it.  The
block must ACCUMULATE into existing ``.grad`` buffers (the autograd node returns ``None`` for
those inputs); tensors it reads are ``record_stream``-ed so the caching allocator keeps them
alive until the side stream is done with them.
"""
import contextlib
import threading

import torch

from . import _lib

_streams = {}
_tls = threading.local()


def stream(device):
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _streams.get(idx)
    if s is None:
        with torch.cuda.device(idx):
            p = _lib.load().edgedict_aux_stream(2)
        if not p:
            raise RuntimeError("edgedict_amd: could not obtain the auxiliary stream")
        s = _streams[idx] = torch.cuda.ExternalStream(p, device=idx)
    return s


def peek(device):
    """The auxiliary stream of ``device`` if it has been created already, else None."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return _streams.get(idx)


def _join(idx):
    def cb():
        torch.cuda.current_stream(idx).wait_stream(_streams[idx])
        _tls.pending.discard(idx)
    return cb


@contextlib.contextmanager
def deferred(device, *reads):
    """Run the body on the auxiliary stream, ordered after the current stream's work so far.
    Only valid inside an autograd backward (the join is an end-of-backward callback)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    side = stream(idx)
    side.wait_stream(torch.cuda.current_stream(idx))
    for t in reads:
        if t is not None:
            t.record_stream(side)
    with torch.cuda.stream(side):
        yield side
    pending = getattr(_tls, "pending", None)
    if pending is None:
        pending = _tls.pending = set()
    if idx not in pending:
        pending.add(idx)
        torch.autograd.Variable._execution_engine.queue_callback(_join(idx))
