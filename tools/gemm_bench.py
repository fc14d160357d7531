"""GEMM throughput probe on the RNN-T shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgedict_amd import ops

def bench(name, M, N, K, ta=False, tb=False, out_dtype=torch.bfloat16, split_k=1, iters=5):
    dt = torch.bfloat16
    a = (torch.randn(K, M, device="cuda").to(dt).t() if ta else torch.randn(M, K, device="cuda").to(dt))
    b = (torch.randn(K, N, device="cuda").to(dt).t() if tb else torch.randn(N, K, device="cuda").to(dt))
    out = torch.zeros(M, N, device="cuda", dtype=out_dtype)
    for _ in range(2):
        ops.gemm(a, b, out=out, split_k=split_k)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm(a, b, out=out, split_k=split_k)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print("%-28s M=%7d N=%5d K=%7d  %8.3f ms  %7.1f TF/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))

M = 64 * 201 * 65
MP = 543526      # packed lattice of the bench batch
bench("joint logits packed (NT)", MP, 2048, 640)
bench("joint dhid packed W2^T (NT)", MP, 640, 2048)
bench("joint logits (NT)", M, 2048, 640)
bench("joint dhid (NN)", M, 640, 2048, tb=True)
bench("joint dW2 (TN, splitk)", 2048, 640, M, ta=True, tb=True, out_dtype=torch.float32, split_k=9)
bench("joint dhid via W2^T copy (NT)", M, 640, 2048)
bench("enc L1 input gemm (NT)", 64 * 401, 4096, 1024)
bench("enc chunk input gemm (NT)", 64 * 32, 4096, 1024, iters=20)
bench("enc chunk dX, Wih^T copy (NT)", 64 * 16, 1024, 4096, iters=20)
bench("enc chunk dX (NN)", 64 * 16, 1024, 4096, tb=True, iters=20)
bench("enc dX (NN)", 64 * 401, 1024, 4096, tb=True)
bench("enc dW_ih (TN)", 4096, 1024, 64 * 401, ta=True, tb=True, out_dtype=torch.float32, split_k=3)
bench("square 4096 (NT)", 4096, 4096, 4096)
bench("square 8192 (NT)", 8192, 8192, 8192, iters=3)
