"""In-tree build of libedgedict_hip.so (gfx950 only) with hipcc.

Every ``*.hip`` / ``*.cpp`` file under ``edgedict_amd/csrc`` is compiled to an object file
(in parallel, skipped when up to date) and linked into ``edgedict_amd/csrc/libedgedict_hip.so``.
hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
``.so`` travels to the GPU box with the repo snapshot.

Run ``python -m edgedict_amd.build`` (add ``--force`` to rebuild everything).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, "include")
LIB_NAME = "libedgedict_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
            "-I", INCLUDE, "-I", CSRC]


def _sources():
    out = []
    # the hipBLASLt bridge is quarantined under csrc/optional/: the default library contains the stubs of
    # blaslt_off.cpp instead (EDGEDICT_WITH_BLASLT=1 swaps them)
    with_vendor = os.environ.get("EDGEDICT_WITH_BLASLT", "0") == "1"
    for fn in sorted(os.listdir(CSRC)):
        if fn.endswith(".hip") or fn.endswith(".cpp"):
            if fn == "blaslt_off.cpp" and with_vendor:
                continue
            out.append(os.path.join(CSRC, fn))
    if with_vendor:
        out.append(os.path.join(CSRC, "optional", "blaslt.cpp"))
    return out


def _deps_mtime():
    m = 0.0
    for d in (CSRC, INCLUDE):
        for fn in os.listdir(d):
            if fn.endswith((".hpp", ".h")):
                m = max(m, os.path.getmtime(os.path.join(d, fn)))
    return m


def _compile_one(src, force, hdr_mtime):
    obj = os.path.splitext(src)[0] + ".o"
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_mtime)):
        return obj, None
    cmd = [HIPCC, "--offload-arch=" + ARCH, "-x", "hip"] + CXXFLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr))
    return obj, r.stderr


def build_all(force=False, verbose=True):
    """Compile + link the HIP library. Returns the path of the shared object."""
    srcs = _sources()
    hdr_mtime = _deps_mtime()
    workers = max(1, min(len(srcs), (os.cpu_count() or 2)))
    with ThreadPoolExecutor(workers) as ex:
        results = list(ex.map(lambda s: _compile_one(s, force, hdr_mtime), srcs))
    objs = [o for o, _ in results]
    rebuilt = [o for o, msg in results if msg is not None]
    for o, msg in results:
        if msg and verbose and msg.strip():
            sys.stderr.write(msg)
    need_link = (force or rebuilt or not os.path.exists(LIB_PATH)
                 or any(os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs))
    if need_link:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    if verbose:
        print("[edgedict_amd.build] %s (%d objects, %d recompiled)" %
              (LIB_PATH, len(objs), len(rebuilt)))
    return LIB_PATH


ASAN_DIR = os.path.join(CSRC, "asan")
ASAN_LIB = os.path.join(ASAN_DIR, "libedgedict_hip_asan.so")


def asan_runtime():
    """Path of the AddressSanitizer runtime of the ROCm clang (to LD_PRELOAD into a python that loads the library)."""
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"],
                       capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


def build_asan(verbose=True):
    """HOST-side AddressSanitizer build of the same sources -> csrc/asan/libedgedict_hip_asan.so (the gfx950 code
    objects are compiled as usual: device-side ASAN needs an xnack+ target).  For the ctypes boundary: load it with
    EDGEDICT_LIB=<path>, LD_PRELOAD=<asan_runtime()>, ASAN_OPTIONS=detect_leaks=0 (tests/test_asan_boundary.py runs
    the host-only entry points - dry-run scheduler, size queries, argument checks - that way)."""
    os.makedirs(ASAN_DIR, exist_ok=True)
    srcs = _sources()
    hdr_mtime = _deps_mtime()
    flags = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address", "-shared-libsan", "-Wno-option-ignored"]
    base = [f for f in CXXFLAGS if f != "-O3"]

    def one(src):
        obj = os.path.join(ASAN_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
        if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_mtime):
            return obj
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-x", "hip"] + base + flags + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (asan) failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr))
        return obj

    with ThreadPoolExecutor(max(1, min(len(srcs), (os.cpu_count() or 2)))) as ex:
        objs = list(ex.map(one, srcs))
    if not os.path.exists(ASAN_LIB) or any(os.path.getmtime(o) > os.path.getmtime(ASAN_LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan",
               "-Wno-option-ignored", "-o", ASAN_LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link (asan) failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    if verbose:
        print("[edgedict_amd.build] %s (%d objects)" % (ASAN_LIB, len(objs)))
    return ASAN_LIB


if __name__ == "__main__":
    if "--asan" in sys.argv:
        build_asan()
    else:
        build_all(force="--force" in sys.argv)
