#!/usr/bin/env python
"""Where a gemm_nt256 tile's time goes on the joint's logits product [543526 x 2048 x 640] with fused
log-sum-exp partials: EDGEDICT_NT256_DEBUG bits switch parts of the kernel off (1 lse, 2 C store, 4 MFMA,
8 C staging); one process per setting (the variable is read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from edgedict_amd import _lib, ops
    M, V, J = 543526, 2048, 640
    hid = torch.randn(M, J, device="cuda").bfloat16()
    w2 = (0.05 * torch.randn(V, J, device="cuda")).bfloat16()
    b2 = torch.zeros(V, device="cuda")
    if os.environ.get("ABLATE_ZERO"):       # zero operands: the data-dependent part of the power draw is gone
        hid.zero_(); w2.zero_()
    logits = torch.empty(M, V, device="cuda", dtype=torch.bfloat16)
    parts = torch.empty(M, V // 64, 2, device="cuda")
    for K in (J, 2048):
        a = hid if K == J else torch.randn(M // 4, K, device="cuda").bfloat16()
        w = w2 if K == J else (0.05 * torch.randn(V, K, device="cuda")).bfloat16()
        m = a.shape[0]
        for _ in range(2):
            _lib.call("gemm_nt_lse", a, ops._ll(K), w, ops._ll(K), logits, ops._ll(V), m, V, K, b2, parts)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            _lib.call("gemm_nt_lse", a, ops._ll(K), w, ops._ll(K), logits, ops._ll(V), m, V, K, b2, parts)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        tiles = ((m + 255) // 256) * (V // 256)
        mhz = ""
        if int(os.environ.get("EDGEDICT_NT256_DEBUG", "0")) & 64:
            st = parts.view(-1)[:1024].view(torch.int64).cpu().view(256, 2).double()
            mhz = "  shader clock %6.0f MHz (workgroup cycles / 100 MHz ticks)" % (st[:, 0].sum() / st[:, 1].sum() * 100).item()
        print("  dbg %2s  M=%6d K=%4d: %7.3f ms  %6.1f TFLOP/s  %5.1f us per tile-slot (tiles/256 CUs)%s"
              % (os.environ.get("EDGEDICT_NT256_DEBUG", "0"), m, K, ms, 2.0 * m * V * K / ms / 1e9,
                 ms * 1e3 / (tiles / 256.0), mhz), flush=True)
else:
    for dbg in sys.argv[1:] or ["0", "1", "2", "3", "11", "4", "15"]:
        env = dict(os.environ, EDGEDICT_NT256_DEBUG=dbg)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
