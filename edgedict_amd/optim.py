"""Adam on flat fp32 buffers — one kernel launch per optimiser step.

Semantics of ``torch.optim.Adam`` as the reference configures it (cli/train.py:135-146:
``optim.Adam(params, lr)``; step at cli/train.py:268) plus ``clip_grad_norm_``
(cli/train.py:262-267).  All parameters are re-pointed into ONE contiguous fp32 buffer and all
gradients into another, so the update is a single streaming kernel over ~51 M elements instead of
~60 per-tensor launches, and a data-parallel gradient bucket is just a slice of the flat buffer.
"""
import torch

from . import config
from ._lib import call, require_cuda
from .ops import _ll


class FlatParams:
    """Flat views of a module's parameters and gradients (parameter order = module order)."""

    def __init__(self, module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        dev = params[0].device
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            if p.dtype != torch.float32:
                raise TypeError("master parameters must be float32")
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64          # keep every tensor 256-byte aligned
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(params, self.offsets):
            flat = self.data[off:off + p.numel()].view_as(p)
            flat.copy_(p.data)
            p.data = flat
            p.grad = self.grad[off:off + p.numel()].view_as(p)
        config.bump_param_epoch()

    def zero_grad(self):
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):   # re-attach if someone set grads to None
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view_as(p)


class FusedAdam:
    def __init__(self, module_or_flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 max_grad_norm=None):
        self.flat = module_or_flat if isinstance(module_or_flat, FlatParams) else FlatParams(module_or_flat)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.m = torch.zeros_like(self.flat.data)
        self.v = torch.zeros_like(self.flat.data)
        self.step_count = 0
        dev = self.flat.data.device
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._coef = torch.ones(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        # state_dict-compatible handle for schedulers that poke param_groups[0]['lr']
        self.param_groups = [{"lr": lr, "params": self.flat.params}]

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self, grad_scale=1.0):
        """``grad_scale`` multiplies every gradient first (1/world_size for data parallelism,
        1/n_sub for gradient accumulation when the caller did not pre-scale the loss)."""
        self.lr = self.param_groups[0]["lr"]
        self.step_count += 1
        coef = None
        if self.max_grad_norm is not None:
            call("grad_clip_coef", self.flat.grad, _ll(self.flat.numel), float(self.max_grad_norm),
                 float(grad_scale), self._sumsq, self._coef, self.grad_norm)
            coef = self._coef
        call("adam_step", self.flat.data, self.flat.grad, self.m, self.v, _ll(self.flat.numel),
             float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
             int(self.step_count), float(self.weight_decay), float(grad_scale), coef, None)
        config.bump_param_epoch()

    def state_dict(self):
        return {"step": self.step_count, "m": self.m, "v": self.v, "lr": self.lr}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.lr = self.param_groups[0]["lr"] = float(sd["lr"])
