"""GPU: MFMA GEMM vs a plain torch fp32/fp64 reference of the same product."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, dtype, seed, transposed=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if transposed:
        x = torch.randn(shape[1], shape[0], generator=g).to(dtype).cuda().t()
    else:
        x = torch.randn(*shape, generator=g).to(dtype).cuda()
    return x


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (64, 16, 1024), (300, 200, 72), (1, 7, 8),
                                   (513, 1030, 264), (129, 640, 896)])
@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_layouts(hip_lib, dtype, M, N, K, ta, tb):
    from edgedict_amd import ops
    a = _mk((M, K), dtype, 1, ta)
    b = _mk((N, K), dtype, 2, tb)   # asymmetric operands: catches row/col swaps
    bias = torch.arange(N, dtype=torch.float32).cuda() * 0.01
    out = ops.gemm(a, b, bias=bias, out_dtype=torch.float32)
    ref = (a.double() @ b.double().t() + bias.double()).float()
    tol = 2e-5 * (K ** 0.5) if dtype == torch.float32 else 2e-5 * (K ** 0.5)
    # bf16 products are exact in fp32; only the accumulation order differs in both modes
    assert (out - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item() / 10)


def test_gemm_bf16_output_accumulate_and_splitk(hip_lib):
    from edgedict_amd import ops
    M, N, K = 256, 384, 4096
    a = _mk((M, K), torch.bfloat16, 3)
    b = _mk((N, K), torch.bfloat16, 4)
    ref = a.double() @ b.double().t()
    out = ops.gemm(a, b)  # bf16 out
    assert out.dtype == torch.bfloat16
    assert (out.double() - ref).abs().max().item() < 0.02 * ref.abs().max().item()
    acc = torch.ones(M, N, device="cuda")
    ops.gemm(a, b, out=acc, accumulate=True)
    assert (acc.double() - ref - 1).abs().max().item() < 1e-3 * ref.abs().max().item()
    sk = torch.ones(M, N, device="cuda")
    ops.gemm(a, b, out=sk, accumulate=True, split_k=8)
    assert (sk.double() - ref - 1).abs().max().item() < 1e-3 * ref.abs().max().item()
    sk2 = torch.full((M, N), 7.0, device="cuda")
    ops.gemm(a, b, out=sk2, split_k=5)      # not accumulating: prior contents must not leak
    assert (sk2.double() - ref).abs().max().item() < 1e-3 * ref.abs().max().item()


def test_gemm_strided_views_and_second_bias(hip_lib):
    from edgedict_amd import ops
    w = _mk((640, 896), torch.float32, 5)
    x = _mk((77, 640), torch.float32, 6)
    b1 = torch.randn(640).cuda()
    b2 = torch.randn(640).cuda()
    out = ops.gemm(x, w[:, :640], bias=b1, bias2=b2)       # column slice of W (ld = 896)
    ref = x @ w[:, :640].t() + b1 + b2
    assert (out - ref).abs().max().item() < 1e-3
    out2 = ops.gemm(x[:, :256].contiguous(), w[:, 640:])
    ref2 = x[:, :256] @ w[:, 640:].t()
    assert (out2 - ref2).abs().max().item() < 1e-3


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1, 8, 64), (300, 200, 640), (1029, 2048, 128),
                                   (64 * 33, 4096, 1024), (130, 136, 4096)])
def test_gemm_nt_direct_to_lds_path(hip_lib, M, N, K):
    """bf16 x bf16 -> bf16 with K-contiguous operands and K % 64 == 0 runs gemm_nt.hip
    (global_load_lds double buffering, source-side swizzle): ragged M/N edges, both biases,
    strided operands, in-place accumulation."""
    from edgedict_amd import ops
    a_full = _mk((M, K + 64), torch.bfloat16, 11)
    a = a_full[:, 64:]                                  # lda = K + 64, 128-byte offset
    b = _mk((N, K), torch.bfloat16, 12)
    b1 = torch.randn(N, generator=torch.Generator().manual_seed(1)).cuda()
    b2 = torch.randn(N, generator=torch.Generator().manual_seed(2)).cuda()
    ref = a.double() @ b.double().t() + b1.double() + b2.double()
    out = ops.gemm(a, b, bias=b1, bias2=b2)
    assert out.dtype == torch.bfloat16
    # one bf16 rounding of an fp32-accumulated value
    err = (out.double() - ref).abs()
    assert (err <= 2.0 ** -8 * ref.abs() + 1e-3 * (K ** 0.5)).all()
    base = _mk((M, N), torch.bfloat16, 13)
    acc = base.clone()
    ops.gemm(a, b, out=acc, accumulate=True)
    ref2 = base.double() + (a.double() @ b.double().t())
    err2 = (acc.double() - ref2).abs()
    assert (err2 <= 2.0 ** -7 * ref2.abs() + 2e-3 * (K ** 0.5)).all()


def test_large_short_k_product_vendor_route_matches_own_kernel(hip_lib):
    """M*N >= 2^28 with K <= 1024 (the joint's logits product) may run in hipBLASLt (csrc/blaslt.cpp);
    a row slice of the same operands is small enough to run in gemm_nt.hip: same values up to the
    summation order of an fp32-accumulated, bf16-rounded result, ragged M included."""
    from edgedict_amd import ops
    M, N, K = 131072 + 37, 2048, 640
    a = _mk((M, K), torch.bfloat16, 21)
    b = _mk((N, K), torch.bfloat16, 22)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3)).cuda()
    out = ops.gemm(a, b, bias=bias)
    assert out.dtype == torch.bfloat16 and out.shape == (M, N)
    for r0 in (0, 70001, M - 300):
        own = ops.gemm(a[r0:r0 + 300], b, bias=bias)
        ref = a[r0:r0 + 300].double() @ b.double().t() + bias.double()
        assert ((out[r0:r0 + 300].double() - ref).abs() <= 2.0 ** -8 * ref.abs() + 0.03).all()
        assert ((out[r0:r0 + 300].float() - own.float()).abs() <= 2.0 ** -7 * own.float().abs() + 0.03).all()


@pytest.mark.parametrize("M,N,K", [(256 * 33 + 37, 4096, 128), (256 * 66 + 1, 2048, 640), (256 * 130, 1000, 192)])
def test_gemm_nt256_macro_tile_path(hip_lib, M, N, K):
    """Large bf16 NT products (>= 512 macro-tiles) run gemm_nt256.hip (256 x 256 tiles, half-tile DMA
    pipeline with counted waits): ragged M and N edges, both biases, strided A - against fp64 on row
    slices, and against the 128 x 128 kernel on the same operands (EDGEDICT_GEMM_NT256 is read once per
    process, so the comparison kernel is reached through a row slice that is too small for this path)."""
    from edgedict_amd import ops
    a_full = _mk((M, K + 64), torch.bfloat16, 31)
    a = a_full[:, 64:]
    b = _mk((N, K), torch.bfloat16, 32)
    b1 = torch.randn(N, generator=torch.Generator().manual_seed(4)).cuda()
    b2 = torch.randn(N, generator=torch.Generator().manual_seed(5)).cuda()
    out = ops.gemm(a, b, bias=b1, bias2=b2)
    assert out.dtype == torch.bfloat16 and out.shape == (M, N)
    for r0 in (0, 255, M // 2 + 3, M - 300):
        ref = a[r0:r0 + 300].double() @ b.double().t() + b1.double() + b2.double()
        err = (out[r0:r0 + 300].double() - ref).abs()
        assert (err <= 2.0 ** -8 * ref.abs() + 1e-3 * (K ** 0.5)).all(), r0
        small = ops.gemm(a[r0:r0 + 300], b, bias=b1, bias2=b2)        # 128 x 128 kernel
        assert ((out[r0:r0 + 300].float() - small.float()).abs() <= 2.0 ** -7 * small.float().abs() + 0.03).all()
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("M,N,K,lse,bias", [
    (256 * 40, 2048, 640, True, True),          # 320 tiles: some workgroups walk two tiles, all full
    (256 * 70 + 37, 2048, 640, True, True),     # the logits shape, ragged last row panel, 3 tiles per workgroup
    (256 * 100 + 5, 640, 2048, False, False),   # the dhid shape: 3 column tiles, the last one ragged (N = 640)
    (256 * 36, 4096, 128, False, True),         # K = 128: two K tiles per tile, the ring wraps every tile
    (256 * 80 + 100, 1000, 192, True, True),    # odd number of K tiles (slot parity flips between tiles), ragged N
    (300, 520, 256, False, True),               # fewer tiles than CUs: one tile per workgroup
])
def test_gemm_nt256_ring_kernel_full_output(hip_lib, M, N, K, lse, bias):
    """The persistent ring kernel (gemm_nt256r.hip): EVERY element of C - and every log-sum-exp partial - against
    the one-tile-per-workgroup kernel of round 2-5 on the same operands (bit-identical: same MFMA order over K) and
    against an fp32 product; twice in a row (a stale ring slot or a race would not repeat)."""
    import os
    from edgedict_amd import _lib
    from edgedict_amd.ops import _ll
    a = _mk((M, K), torch.bfloat16, 41)
    b = _mk((N, K), torch.bfloat16, 42)
    bv = torch.randn(N, generator=torch.Generator().manual_seed(6)).cuda() if bias else None
    slots = (N + 63) // 64

    def run():
        c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        parts = torch.full((M, slots, 2), float("nan"), device="cuda") if lse else None
        if lse:
            _lib.call("gemm_nt_lse", a, _ll(K), b, _ll(K), c, _ll(N), M, N, K, bv, parts)
        else:
            # the macro-tile entry behind edgedict_gemm needs >= 512 tiles; the lse entry with a scratch buffer does not
            scratch = torch.empty(M, slots, 2, device="cuda")
            _lib.call("gemm_nt_lse", a, _ll(K), b, _ll(K), c, _ll(N), M, N, K, bv, scratch)
        torch.cuda.synchronize()
        return c, parts

    old_env = os.environ.get("EDGEDICT_GEMM_NT256R")
    try:
        os.environ["EDGEDICT_GEMM_NT256R"] = "1"
        c1, p1 = run()
        c2, p2 = run()
        os.environ["EDGEDICT_GEMM_NT256R"] = "0"
        c0, p0 = run()
    finally:
        if old_env is None:
            os.environ.pop("EDGEDICT_GEMM_NT256R", None)
        else:
            os.environ["EDGEDICT_GEMM_NT256R"] = old_env
    assert torch.isfinite(c1.float()).all()
    assert torch.equal(c1.view(torch.int16), c2.view(torch.int16))
    assert torch.equal(c1.view(torch.int16), c0.view(torch.int16))
    if lse:
        assert torch.isfinite(p1).all()
        assert torch.equal(p1, p2)
        # same values, same reduction tree; the exponentials go through the same instructions
        assert torch.allclose(p1, p0, rtol=1e-6, atol=0)
        mx, sm = p1[..., 0].double(), p1[..., 1].double()
        got = (mx.max(dim=1).values + torch.log((sm * torch.exp(mx - mx.max(dim=1, keepdim=True).values)).sum(1)))
        pad = slots * 64 - N
        cf = c1.double()
        ref_lse = torch.logsumexp(cf, dim=1)
        assert (got - ref_lse).abs().max().item() < 1e-4, pad
    ref = a.float() @ b.float().t()
    if bias:
        ref = ref + bv
    err = (c1.float() - ref).abs()
    assert (err <= 2.0 ** -7 * ref.abs() + 1e-3 * (K ** 0.5)).all()


@pytest.mark.parametrize("M,N,K,split", [(512, 256, 4096, 2), (1024, 240, 5003, 4), (264, 648, 1111, 1),
                                         (2048, 640, 9000, 4)])
def test_gemm_tn256_weight_gradient_path(hip_lib, M, N, K, split):
    """Background weight-gradient products (dW = dY^T X, both operands row-major over the reduction) run
    gemm_tn256.hip: 256 x 128 tiles, transpose reads out of LDS, fp32 K-slice partials written once and summed
    by the reduce pass.  Ragged M / N edges, K not a multiple of the K tile or of the slice count, strided
    operands, accumulate into an existing gradient - against fp64 (bf16 products are exact in fp32, only
    the summation order differs)."""
    from edgedict_amd import ops
    dy_full = _mk((K, M + 8), torch.bfloat16, 41)
    dy = dy_full[:, 8:]                       # [K, M], row stride M + 8
    x = _mk((K, N), torch.bfloat16, 42)
    grad = torch.randn(M, N, generator=torch.Generator().manual_seed(6)).cuda()
    want = grad.double() + dy.double().t() @ x.double()
    ops.gemm(dy.t(), x.t(), out=grad, accumulate=True, split_k=split, max_wg_per_cu=2)
    tol = 3e-5 * (K ** 0.5) * max(1.0, want.abs().max().item() / 10)
    assert (grad.double() - want).abs().max().item() <= tol
    # and without accumulate, fresh output
    out = ops.gemm(dy.t(), x.t(), out_dtype=torch.float32, split_k=split, max_wg_per_cu=1)
    assert (out.double() - dy.double().t() @ x.double()).abs().max().item() <= tol


@pytest.mark.parametrize("M,N,K", [(768, 1024, 4096), (1000, 1000, 1024), (1536, 1024, 4096), (70, 200, 2048)])
def test_gemm_nt_small_long_k_ring_path(hip_lib, M, N, K):
    """Small-M long-K bf16 NT products with in-place accumulation (the encoder stack's per-chunk
    dX = dG x W_ih under the BPTT) run the 64 x 64-tile ring kernel of gemm_nt.hip (four K stages, counted
    waits): ragged edges, accumulate and plain store, against fp64 with bf16 output rounding."""
    from edgedict_amd import ops
    a = _mk((M, K), torch.bfloat16, 51)
    b = _mk((N, K), torch.bfloat16, 52)
    c0 = _mk((M, N), torch.bfloat16, 53)
    prod = a.double() @ b.double().t()
    out = ops.gemm(a, b)
    assert ((out.double() - prod).abs() <= 2.0 ** -7 * prod.abs() + 1e-3 * (K ** 0.5)).all()
    acc = c0.clone()
    ops.gemm(a, b, out=acc, accumulate=True)
    want = c0.double() + out.double()         # the kernel adds its bf16-rounded tile to the bf16 C
    assert ((acc.double() - want).abs() <= 2.0 ** -6 * want.abs() + 1e-2).all()
