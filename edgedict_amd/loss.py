"""RNN-T loss operator backed by the hand-written gfx950 kernels in csrc/rnnt_loss.hip.

Mirrors the call surface of ``warprnnt_pytorch.RNNTLoss`` as the reference uses it
(``rnnt/models.py:221,238``; ``cli/lightning.py:40,91``)::

    loss_fn = RNNTLoss(blank=0)
    loss = loss_fn(acts, labels, act_lens, label_lens)      # Tensor[1], differentiable wrt acts

* ``acts``        float32 / bfloat16 ``[B, T, U+1, V]`` raw logits, contiguous
* ``labels``      int32 ``[B, U]``
* ``act_lens``    int32 ``[B]``, ``max == T``
* ``label_lens``  int32 ``[B]``, ``max == U``

Difference from upstream that is deliberate: upstream computes the full gradient tensor in
``forward`` and rescales it by ``grad_output`` in ``backward`` (a second pass over
``B*T*(U+1)*V`` elements).  Here ``forward`` only fills the small alpha/beta workspace and
``backward`` writes the already-scaled gradient once.
"""
import torch

from . import _lib


def _certify_inputs(acts, labels, act_lens, label_lens, check_lengths):
    # same error classes / wording style as warprnnt_pytorch.certify_inputs
    if acts.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("acts must be float32 or bfloat16, got %s" % acts.dtype)
    for name, t in (("labels", labels), ("act_lens", act_lens), ("label_lens", label_lens)):
        if t.dtype != torch.int32:
            raise TypeError("%s must be int32, got %s" % (name, t.dtype))
    for name, t in (("acts", acts), ("labels", labels), ("act_lens", act_lens),
                    ("label_lens", label_lens)):
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous" % name)
    if acts.dim() != 4:
        raise ValueError("acts must have 4 dimensions [B,T,U+1,V], got %d" % acts.dim())
    if labels.dim() != 2:
        raise ValueError("labels must have 2 dimensions [B,U], got %d" % labels.dim())
    if act_lens.dim() != 1 or label_lens.dim() != 1:
        raise ValueError("act_lens and label_lens must have 1 dimension")
    B, T, U1, _ = acts.shape
    if act_lens.shape[0] != B:
        raise ValueError("must have a length per example (act_lens has %d, batch is %d)"
                         % (act_lens.shape[0], B))
    if label_lens.shape[0] != B or labels.shape[0] != B:
        raise ValueError("must have a label length per example")
    if labels.shape[1] != U1 - 1:
        raise ValueError("Output length mismatch: labels has U=%d but acts has U+1=%d"
                         % (labels.shape[1], U1))
    if check_lengths:  # one host sync, exactly what upstream does
        if int(act_lens.max()) != T:
            raise ValueError("Input length mismatch")
        if int(label_lens.max()) != U1 - 1:
            raise ValueError("Output length mismatch")


class _RNNTLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acts, labels, act_lens, label_lens, blank, reduction):
        _lib.require_cuda(acts, labels, act_lens, label_lens)
        B, T, U1, V = acts.shape
        lib = _lib.load()
        ws = torch.empty(lib.edgedict_rnnt_workspace_bytes(B, T, U1), dtype=torch.uint8,
                         device=acts.device)
        costs = torch.empty(B, dtype=torch.float32, device=acts.device)
        reduced = torch.empty(1, dtype=torch.float32, device=acts.device)
        scale = 1.0 / B if reduction == "mean" else 1.0
        _lib.call("rnnt_loss_forward", acts, _lib.dtype_code(acts.dtype), labels, act_lens,
                  label_lens, B, T, U1, V, int(blank), costs, reduced, float(scale), ws)
        ctx.save_for_backward(acts, labels, act_lens, label_lens, ws)
        ctx.blank = int(blank)
        ctx.reduction = reduction
        ctx.costs = costs
        return costs if reduction == "none" else reduced

    @staticmethod
    def backward(ctx, grad_output):
        acts, labels, act_lens, label_lens, ws = ctx.saved_tensors
        B, T, U1, V = acts.shape
        grads = torch.empty_like(acts)
        go = grad_output.contiguous().float()
        host_scale = 1.0 / B if ctx.reduction == "mean" else 1.0
        stride = 1 if ctx.reduction == "none" else 0
        _lib.call("rnnt_loss_backward", acts, _lib.dtype_code(acts.dtype), grads, labels,
                  act_lens, label_lens, B, T, U1, V, ctx.blank, ws, float(host_scale), go, stride)
        return grads, None, None, None, None, None


class RNNTLoss(torch.nn.Module):
    """Drop-in for ``warprnnt_pytorch.RNNTLoss``.

    ``reduction``: ``'mean'`` (default; ``sum_b cost_b / B`` with shape ``(1,)``), ``'sum'``
    (shape ``(1,)``) or ``'none'`` (shape ``(B,)``).
    """

    def __init__(self, blank=0, reduction="mean", check_lengths=True):
        super().__init__()
        if reduction not in ("mean", "sum", "none"):
            raise ValueError("reduction must be 'mean', 'sum' or 'none'")
        self.blank = blank
        self.reduction = reduction
        self.check_lengths = check_lengths

    def forward(self, acts, labels, act_lens, label_lens):
        _certify_inputs(acts, labels, act_lens, label_lens, self.check_lengths)
        return _RNNTLossFn.apply(acts, labels, act_lens, label_lens, self.blank, self.reduction)


def rnnt_loss_debug(acts, labels, act_lens, label_lens, blank=0):
    """Test hook: run forward and return (costs, denominators, alphas, betas, loglikes[B,2])."""
    _lib.require_cuda(acts)
    B, T, U1, V = acts.shape
    lib = _lib.load()
    nbytes = lib.edgedict_rnnt_workspace_bytes(B, T, U1)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=acts.device)
    costs = torch.empty(B, dtype=torch.float32, device=acts.device)
    _lib.call("rnnt_loss_forward", acts, _lib.dtype_code(acts.dtype), labels, act_lens,
              label_lens, B, T, U1, V, int(blank), costs, None, 1.0, ws)
    base = ws.data_ptr()

    def view(which, shape, dtype):
        p = lib.edgedict_rnnt_workspace_view(_lib.ptr(ws), B, T, U1, which)
        esz = 8 if dtype == torch.float64 else 4
        off = (p - base) // esz
        n = 1
        for s in shape:
            n *= s
        return ws.view(dtype)[off:off + n].view(*shape).clone()

    return (costs, view(0, (B, T, U1), torch.float32), view(1, (B, T, U1), torch.float64),
            view(2, (B, T, U1), torch.float64), view(3, (B, 2), torch.float64))
