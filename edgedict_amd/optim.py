"""Adam on flat fp32 buffers — one kernel launch per optimiser step.

Semantics of ``torch.optim.Adam`` as the reference configures it (cli/train.py:135-146:
``optim.Adam(params, lr)``; step at cli/train.py:268) plus ``clip_grad_norm_``
(cli/train.py:262-267).  All parameters are re-pointed into ONE contiguous fp32 buffer and all
gradients into another, so the update is a single streaming kernel over ~51 M elements instead of
~60 per-tensor launches, and a data-parallel gradient bucket is just a slice of the flat buffer.
"""
import ctypes

import torch

from . import config
from ._lib import call
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
        self._guard_scratch = torch.zeros(1, dtype=torch.int32, device=dev)   # device copy of the give-up words
        # state_dict-compatible handle for schedulers that poke param_groups[0]['lr']
        self.param_groups = [{"lr": lr, "params": self.flat.params}]

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self, grad_scale=1.0, guard=None):
        """``grad_scale`` multiplies every gradient first (1/world_size for data parallelism,
        1/n_sub for gradient accumulation when the caller did not pre-scale the loss).
        ``guard``: device-visible address of 3 words (``edgedict_stack_error_words(0)``); the step is
        skipped on the device when one of them is non-zero (a bounded in-kernel wait gave up: the
        gradients are garbage) - no host synchronisation."""
        self.lr = self.param_groups[0]["lr"]
        self.step_count += 1
        coef = None
        if self.max_grad_norm is not None:
            call("grad_clip_coef", self.flat.grad, _ll(self.flat.numel), float(self.max_grad_norm),
                 float(grad_scale), self._sumsq, self._coef, self.grad_norm)
            coef = self._coef
        call("adam_step_guarded", self.flat.data, self.flat.grad, self.m, self.v, _ll(self.flat.numel),
             float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
             int(self.step_count), float(self.weight_decay), float(grad_scale), coef, None,
             ctypes.c_void_p(guard or 0), 3 if guard else 0, self._guard_scratch if guard else None)
        config.bump_param_epoch()

    def state_dict(self):
        """``torch.optim.Adam.state_dict()`` layout (what the reference's checkpoints hold under
        ``'optim'``, cli/train.py:321-336): per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``
        (clones of the flat buffers' views) and one param group, so a checkpoint written here
        loads into ``torch.optim.Adam`` over the same parameter list and vice versa."""
        state = {}
        for i, (p, off) in enumerate(zip(self.flat.params, self.flat.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.m[off:off + n].view_as(p).clone(),
                        "exp_avg_sq": self.v[off:off + n].view_as(p).clone()}
        group = {"lr": self.param_groups[0]["lr"], "betas": tuple(self.betas), "eps": self.eps,
                 "weight_decay": self.weight_decay, "amsgrad": False,
                 "params": list(range(len(self.flat.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "param_groups" in sd:           # torch.optim.Adam layout
            g = sd["param_groups"][0]
            self.lr = self.param_groups[0]["lr"] = float(g["lr"])
            self.betas = tuple(g.get("betas", self.betas))
            self.eps = float(g.get("eps", self.eps))
            self.weight_decay = float(g.get("weight_decay", self.weight_decay))
            steps = set()
            for i, (p, off) in enumerate(zip(self.flat.params, self.flat.offsets)):
                st = sd["state"].get(i, sd["state"].get(str(i)))
                n = p.numel()
                if st is None:             # parameter that never received a gradient
                    self.m[off:off + n].zero_()
                    self.v[off:off + n].zero_()
                    continue
                self.m[off:off + n].view_as(p).copy_(st["exp_avg"])
                self.v[off:off + n].view_as(p).copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise ValueError("per-parameter Adam step counts differ (%s): the flat Adam kernel "
                                 "uses one bias correction for all parameters" % sorted(steps))
            self.step_count = steps.pop() if steps else 0
            return
        self.step_count = int(sd["step"])  # round-1 flat layout
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.lr = self.param_groups[0]["lr"] = float(sd["lr"])


class WarmupLR:
    """Linear warm-up of the reference's training loop (cli/train.py:189-191):
    ``lr = base_lr * step / warmup_step`` for ``step <= warmup_step`` (steps count from 1)."""

    def __init__(self, optimizer, base_lr, warmup_step):
        self.optimizer, self.base_lr, self.warmup_step = optimizer, float(base_lr), int(warmup_step)

    def step(self, step):
        if self.warmup_step > 0 and step <= self.warmup_step:
            self.optimizer.param_groups[0]["lr"] = self.base_lr * step / self.warmup_step
        return self.optimizer.param_groups[0]["lr"]


class ReduceLROnPlateau:
    """``torch.optim.lr_scheduler.ReduceLROnPlateau(optim, patience, factor, min_lr)`` as the
    reference configures it (cli/train.py:142-146: mode 'min', relative threshold 1e-4, no
    cooldown, eps 1e-8), for any object with ``param_groups`` (FusedAdam is not a
    ``torch.optim.Optimizer``).  ``state_dict`` uses torch's key names."""

    def __init__(self, optimizer, patience=10, factor=0.1, min_lr=0.0, threshold=1e-4, eps=1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.optimizer = optimizer
        self.patience, self.factor, self.min_lr = patience, factor, min_lr
        self.threshold, self.eps = threshold, eps
        self.best = float("inf")
        self.num_bad_epochs = 0
        self.last_epoch = 0

    def step(self, metric):
        current = float(metric)
        self.last_epoch += 1
        if current < self.best * (1.0 - self.threshold):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            for g in self.optimizer.param_groups:
                old = float(g["lr"])
                new = max(old * self.factor, self.min_lr)
                if old - new > self.eps:
                    g["lr"] = new
            self.num_bad_epochs = 0

    def state_dict(self):
        return {"best": self.best, "num_bad_epochs": self.num_bad_epochs,
                "last_epoch": self.last_epoch, "patience": self.patience, "factor": self.factor,
                "min_lrs": [self.min_lr], "threshold": self.threshold, "eps": self.eps,
                "mode": "min", "threshold_mode": "rel", "cooldown": 0, "cooldown_counter": 0}

    def load_state_dict(self, sd):
        self.best = float(sd["best"])
        self.num_bad_epochs = int(sd["num_bad_epochs"])
        self.last_epoch = int(sd.get("last_epoch", 0))
