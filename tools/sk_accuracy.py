"""Gradient accuracy of the two bf16 BPTT kernels against the fp32 parity mode at the E6D2 encoder geometry
(B = 64, T0 = 401, 6 x 1024): norm-relative deviation of every parameter gradient, step kernels vs split-K kernel.
usage: python tools/sk_accuracy.py      (run on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from test_encoder_stack_gpu import _encoder, _run  # noqa: E402


def rel(a, b):
    return (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-12)


case = (64, 401, 240, 1024, 6, [1], 12, 0)
enc, xs = _encoder(case)
ref = _run(enc, xs, torch.float32)
runs = {}
for name, sk in (("step kernels", "0"), ("split-K", "1")):
    os.environ["EDGEDICT_STACK_BWD_SK"] = sk
    runs[name] = _run(enc, xs, torch.bfloat16, chunk=12)
print("%-34s %12s %12s %12s" % ("gradient", "step vs fp32", "splitK vs fp32", "step vs splitK"))
worst = [0.0, 0.0, 0.0]
for n in ref[3]:
    a, b = runs["step kernels"][3][n], runs["split-K"][3][n]
    r = (rel(a, ref[3][n]), rel(b, ref[3][n]), rel(b, a))
    worst = [max(x, y) for x, y in zip(worst, r)]
    if "lstms.0." in n or "lstms.5." in n or n.startswith("norm") or n.startswith("proj"):
        print("%-34s %12.2e %12.2e %12.2e" % (n, *r))
print("%-34s %12.2e %12.2e %12.2e" % ("worst over all parameters", *worst))
