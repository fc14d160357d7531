"""ORACLE (test infrastructure only - never imported by the product path).

The reference's own DRIVER LOOPS for the hot path, compiled from the sources where they lie under
/root/reference into ``oracle/_ref/*.bin`` (marshalled CPython code objects: build outputs, git-ignored, they
travel to the GPU box with the snapshot like the built ``.so`` - nothing of the reference's source is copied
into this repository, and /root/reference is never read at test time):

  piece                  reference lines                               what runs it
  ---------------------  --------------------------------------------  -----------------------------------------
  baseline_train_step    cli/baseline.py  Trainer.train_step           tests/test_reference_loops_gpu.py (engine),
  baseline_eval          cli/baseline.py  Trainer.evaluate_step/save/load  (same two runners)
  lightning_steps        cli/lightning.py ParallelTraining.training_step / validation_step / warmup_optimizer_step
                                          (the EXTERNAL loss call: Transducer(output_loss=False) -> scale_length ->
                                          warprnnt_pytorch.RNNTLoss, cli/lightning.py:40,85-91)
  frontend_train_step    cli/train.py     Trainer.train_step             oracle/make_golden_ref_loops.py (reference
  stream_decode          cli/openvino_wav_inference.py stream_decode     modules on the CPU -> tests/golden/
  stream_classes         rnnt/stream.py   StreamTransducerDecoder,       ref_loops.npz)
                                          PytorchStreamDecoder
  mic_callback           stream.py        callback (the microphone loop: 2-block buffer, reset after 35 blank chunks)

The files these loops import (absl, apex, tensorboardX, jiwer, sounddevice, torchaudio ...) are absent here, so the
loops are lifted with ``ast`` (functions / classes by name, line numbers kept: a traceback cites the reference
file:line) and executed in a namespace the caller fills with what their module would have imported - for the
engine-side tests: ``rnnt.models`` / ``rnnt.stream`` / ``rnnt.transforms`` THROUGH THE ROOT SHIMS.

    python oracle/ref_lift.py          # needs /root/reference; __graft_entry__.build() runs it when present
"""
import ast
import hashlib
import json
import marshal
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "oracle", "_ref")

# piece -> (file under the reference, enclosing class or None, names)
PIECES = {
    "baseline_train_step": ("cli/baseline.py", "Trainer", ["train_step"]),
    "frontend_train_step": ("cli/train.py", "Trainer", ["train_step"]),
    "baseline_eval": ("cli/baseline.py", "Trainer", ["evaluate_step", "save", "load"]),
    "lightning_steps": ("cli/lightning.py", "ParallelTraining", ["warmup_optimizer_step", "training_step", "validation_step"]),
    "stream_decode": ("cli/openvino_wav_inference.py", None, ["stream_decode"]),
    "stream_classes": ("rnnt/stream.py", None, ["StreamTransducerDecoder", "PytorchStreamDecoder"]),
    "mic_callback": ("stream.py", None, ["callback"]),
}


def _lift(path, cls, names):
    tree = ast.parse(open(path).read(), filename=path)
    body = tree.body
    if cls is not None:
        (klass,) = [n for n in body if isinstance(n, ast.ClassDef) and n.name == cls]
        body = klass.body
    keep = [n for n in body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert [n.name for n in keep] == names, (path, [n.name for n in keep])
    lines = (min(n.lineno for n in keep), max(n.end_lineno for n in keep))
    return compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), lines


def build(verbose=True):
    """Compile every piece into oracle/_ref/<piece>.bin + manifest.json.  Returns the manifest."""
    assert os.path.isdir(REF), "%s is not present: oracle/_ref can only be built where the reference is" % REF
    os.makedirs(OUT, exist_ok=True)
    manifest = {"python": list(sys.version_info[:3]), "pieces": {}}
    for name, (rel, cls, names) in PIECES.items():
        path = os.path.join(REF, rel)
        code, lines = _lift(path, cls, names)
        blob = marshal.dumps(code)
        with open(os.path.join(OUT, name + ".bin"), "wb") as f:
            f.write(blob)
        manifest["pieces"][name] = {
            "file": rel, "class": cls, "names": names, "lines": list(lines),
            "sha256_of_reference_file": hashlib.sha256(open(path, "rb").read()).hexdigest(),
            "sha256_of_bin": hashlib.sha256(blob).hexdigest()}
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    if verbose:
        print("[oracle.ref_lift] %d reference loops compiled into %s" % (len(PIECES), OUT))
    return manifest


def available():
    """True when oracle/_ref holds every piece, built by this Python version."""
    try:
        man = json.load(open(os.path.join(OUT, "manifest.json")))
    except OSError:
        return False
    return (man.get("python") == list(sys.version_info[:3])
            and all(os.path.exists(os.path.join(OUT, n + ".bin")) and "sha256_of_bin" in man.get("pieces", {}).get(n, {})
                    for n in PIECES))


def load(name, namespace):
    """Execute piece ``name`` in ``namespace`` (a dict that already holds what the reference module would have
    imported) and return the namespace: the lifted functions / classes are then its entries."""
    with open(os.path.join(OUT, name + ".bin"), "rb") as f:
        blob = f.read()
    # the blobs are untracked build outputs: only what THIS build wrote (manifest digest) is unmarshalled and executed
    man = json.load(open(os.path.join(OUT, "manifest.json")))
    want = man["pieces"][name]["sha256_of_bin"]
    got = hashlib.sha256(blob).hexdigest()
    if got != want:
        raise RuntimeError("oracle/_ref/%s.bin does not match its manifest digest (%s != %s): rebuild with "
                           "python oracle/ref_lift.py" % (name, got[:12], want[:12]))
    exec(marshal.loads(blob), namespace)
    return namespace


if __name__ == "__main__":
    build()
