"""Encoder-stack forward / backward time at the E6D2 bench geometry for a list of environment settings
(each in-process: the scheduler re-reads EDGEDICT_STACK_LPW / EDGEDICT_LPW_STEPS / EDGEDICT_LPW_MARGIN per call).
usage: python tools/fwd_time.py "LPW=0" "LPW=1,STEPS=6" "LPW=1,STEPS=4,CHUNK=8" ...   (run on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from edgedict_amd import encoder_stack, ops  # noqa: E402
from edgedict_amd.models import Encoder  # noqa: E402

B, T0 = 64, 401
torch.manual_seed(0)
enc = Encoder(240, 1024, 6, 0.0, 640).cuda()
enc.compute_dtype = torch.bfloat16
xs = torch.randn(B, T0, 240, device="cuda")
for p in enc.parameters():          # gradients accumulate in place (flat-buffer training does the same)
    p.grad = torch.zeros_like(p)
KEYS = {"LPW": "EDGEDICT_STACK_LPW", "STEPS": "EDGEDICT_LPW_STEPS", "MARGIN": "EDGEDICT_LPW_MARGIN", "POLL": "EDGEDICT_LPW_POLL", 
        "SK": "EDGEDICT_STACK_BWD_SK", "SKSTEPS": "EDGEDICT_SK_STEPS", "MARGINB": "EDGEDICT_LPW_MARGIN_B"}
for spec in sys.argv[1:] or ["LPW=0"]:
    kv = dict(x.split("=") for x in spec.split(","))
    for k, v in kv.items():
        if k in KEYS:
            os.environ[KEYS[k]] = v
    encoder_stack.CHUNK = int(kv.get("CHUNK", 16))
    bwd = kv.get("BWD", "0") == "1"
    res = []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if bwd:
            out, _ = enc(xs)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out.float().sum().backward()
        else:
            with torch.no_grad():
                out, _ = enc(xs)
            t1 = None
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append((1e3 * ((t1 or t2) - t0), 1e3 * (t2 - (t1 or t2))))
    encoder_stack.check_wsr_error()
    res = res[2:]
    print("%-40s fwd %.3f ms (min %.3f)%s   checksum %.6f" % (
        spec, sum(r[0] for r in res) / len(res), min(r[0] for r in res),
        ("  bwd %.3f ms" % (sum(r[1] for r in res) / len(res))) if bwd else "", float(out.float().abs().mean())), flush=True)
