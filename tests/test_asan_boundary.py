"""The ctypes boundary under AddressSanitizer (host side): `python -m edgedict_amd.build --asan` compiles the same
sources with -fsanitize=address into csrc/asan/libedgedict_hip_asan.so; a child python preloads the sanitizer
runtime, binds that library (EDGEDICT_LIB) and drives every host-only entry point - the dry-run scheduler over many
geometries in all its modes, the size queries, the argument checks and their error strings.  Any heap / stack /
global overflow or use-after-free in the host code aborts the child with an AddressSanitizer report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, itertools, os, sys
sys.path.insert(0, %r)
import numpy as np
from edgedict_amd import _lib, encoder_stack as es
lib = _lib.load()
assert "asan" in os.path.basename(_lib.LIB_PATH if hasattr(_lib, "LIB_PATH") else os.environ["EDGEDICT_LIB"])
n = 0
for lpw, sk in ((0, 0), (1, 1), (1, 0)):
    os.environ["EDGEDICT_STACK_LPW"] = str(lpw)
    os.environ["EDGEDICT_STACK_BWD_SK"] = str(sk)
    for T0, red, chunk, H, B in itertools.product((1, 7, 50, 401), ([1], [1, 2, 1], [2, 1, 2, 1, 1, 1], [1] * 8),
                                                  (1, 4, 12), (64, 1024), (3, 64)):
        for backward in (False, True):
            steps, enq, nl, ms = es.schedule(T0, 64, H, red, B=B, chunk=chunk, backward=backward)
            assert nl >= 1 and all((s >= 0).all() for s in steps)
            n += 1
# size queries and argument checks (every one returns a status, none may touch memory it was not given)
assert lib.edgedict_rnnt_workspace_bytes(64, 201, 65) > 0
lib.edgedict_last_error.restype = ctypes.c_char_p
bad = lib.edgedict_stack_schedule(None, 0, None, None, None, None)
assert bad != 0 and b"null" in lib.edgedict_last_error()
k, st = ctypes.c_int(0), ctypes.c_int(0)
lib.edgedict_stack_last_mode(1, ctypes.byref(k), ctypes.byref(st))      # no device: an error code, not a crash
try:
    es.schedule(10, 64, 48, [1], B=3, chunk=4)          # H not a multiple of 32
except RuntimeError as e:
    assert "H" in str(e)
else:
    raise AssertionError("bad geometry accepted")
print("ASAN_CHILD_OK", n)
''' % ROOT


def test_host_entry_points_under_address_sanitizer():
    from edgedict_amd import build
    rt = build.asan_runtime()
    if rt is None:
        pytest.skip("no AddressSanitizer runtime in this toolchain")
    lib = build.build_asan(verbose=False)
    env = dict(os.environ)
    env.update(EDGEDICT_LIB=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0",
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "ASAN_CHILD_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
