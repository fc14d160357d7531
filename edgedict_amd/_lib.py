"""ctypes binding of libedgedict_hip.so (the C ABI declared in include/edgedict_hip.h).

PyTorch tensors are used purely as containers: a tensor crosses this boundary as
``tensor.data_ptr()`` and the current HIP stream as ``torch.cuda.current_stream().cuda_stream``.
There is NO fallback: if the shared library is missing, or a call returns a non-zero status,
a RuntimeError is raised.
"""
import ctypes
import os
import re
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# EDGEDICT_LIB: an alternative build of the library (tuning experiments: tools/build_variant.sh)
LIB_PATH = os.environ.get("EDGEDICT_LIB") or os.path.join(_HERE, "csrc", "libedgedict_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "edgedict_hip.h")

ED_F32 = 0
ED_BF16 = 1

_lock = threading.Lock()
_lib = None


def dtype_code(dtype):
    if dtype == torch.float32:
        return ED_F32
    if dtype == torch.bfloat16:
        return ED_BF16
    raise TypeError("edgedict_amd: unsupported dtype %s (fp32 or bf16 only)" % dtype)


def declared_symbols():
    """Names of every function declared in include/edgedict_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(edgedict_[a-z0-9_]+)\s*\(", text)))


def load():
    """Load (once) and return the ctypes handle. Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "edgedict_amd: %s not found. Build it with `python -m edgedict_amd.build` "
                "(hipcc, gfx950). There is no CPU/PyTorch fallback for the hot path." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        lib.edgedict_last_error.restype = ctypes.c_char_p
        lib.edgedict_rnnt_workspace_bytes.restype = ctypes.c_size_t
        lib.edgedict_rnnt_workspace_view.restype = ctypes.c_void_p
        lib.edgedict_greedy_workspace_bytes.restype = ctypes.c_size_t
        lib.edgedict_beam_workspace_bytes.restype = ctypes.c_size_t
        lib.edgedict_stream_encoder_workspace_bytes.restype = ctypes.c_size_t
        lib.edgedict_blaslt_calls.restype = ctypes.c_longlong
        lib.edgedict_gelu_groupnorm_bwd_workspace_bytes.restype = ctypes.c_size_t
        lib.edgedict_lstm_workspace_bytes.restype = ctypes.c_size_t
        lib.edgedict_stack_workspace_bytes.restype = ctypes.c_size_t
        lib.edgedict_stack_struct_bytes.restype = ctypes.c_size_t
        lib.edgedict_aux_stream.restype = ctypes.c_void_p
        lib.edgedict_stack_error_words.restype = ctypes.c_void_p
        for name in declared_symbols():
            if not hasattr(lib, name):
                raise RuntimeError("edgedict_amd: %s lacks symbol %s declared in the header"
                                   % (LIB_PATH, name))
        if lib.edgedict_abi_version() != 1:
            raise RuntimeError("edgedict_amd: ABI version mismatch")
        _lib = lib
    return _lib


def ptr(t):
    """Device pointer of a tensor (or NULL for None) as a ctypes void*."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """Raw hipStream_t of the current stream.  torch.cuda.current_stream() costs ~0.2 ms per call
    here (it re-reads the environment through is_available() every time); the raw getter is a
    single C call."""
    if device is None:
        idx = torch._C._cuda_getDevice()
    elif isinstance(device, int):
        idx = device
    else:
        idx = torch.device(device).index
        if idx is None:
            idx = torch._C._cuda_getDevice()
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "edgedict_amd: the hot path runs only on an MI355X HIP device; got a %s tensor. "
                "There is deliberately no CPU fallback (the CPU oracle lives under oracle/ and is "
                "test infrastructure only)." % t.device)


def check(status, what):
    if status != 0:
        msg = load().edgedict_last_error()
        raise RuntimeError("edgedict_amd: %s failed (status %d): %s"
                           % (what, status, msg.decode() if msg else "?"))


def _conv(a):
    if isinstance(a, torch.Tensor):
        return ctypes.c_void_p(a.data_ptr())
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, float):
        return ctypes.c_float(a)
    if isinstance(a, bool):
        return ctypes.c_int(int(a))
    if isinstance(a, int):
        return ctypes.c_int(a)
    return a


def call(name, *args):
    """Call ``edgedict_<name>(*args, stream)`` on the current stream and check its status.

    Tensors become device pointers, Python ints become C ints, floats become C floats;
    pass ctypes values explicitly for anything else (e.g. ``ctypes.c_longlong``)."""
    lib = load()
    fn = getattr(lib, "edgedict_" + name)
    cargs = [_conv(a) for a in args]
    cargs.append(stream_ptr())
    check(fn(*cargs), name)
