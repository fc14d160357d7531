"""Data-parallel gradient exchange: one process per GPU, RCCL (``backend='nccl'``) over xGMI.

The reference scales with ``nn.DataParallel`` (cli/train.py:152-153: broadcast + reduce to GPU0)
or Lightning DDP (cli/lightning.py:325-331).  Here utterances are sharded over ranks with no
data-path collective; the only exchange is the gradient all-reduce, issued per *bucket* — a
contiguous slice of the flat gradient buffer (optim.FlatParams) — as soon as autograd has
finished every parameter in it, so RCCL traffic overlaps the remaining BPTT
(joint -> prediction net -> encoder layer 5 ... 0).  MI355X xGMI is point-to-point
(7 links/GPU): buckets are large (default 64 MiB => 4 collectives for the 203 MB of E6D2
gradients) so each ring step moves MBs per link rather than paying latency many times.
The sum is turned into the mean inside the Adam kernel (``grad_scale = 1/world``).
"""
import torch.distributed as dist


class BucketedAllReduce:
    def __init__(self, flat, process_group=None, bucket_bytes=64 << 20):
        self.flat = flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # buckets are built from the END of the flat buffer: backward reaches those params first
        per = max(1, bucket_bytes // 4)
        bounds = []
        hi = flat.numel
        cur_lo = hi
        acc = 0
        self.param_bucket = {}
        order = list(zip(flat.params, flat.offsets))
        for p, off in reversed(order):
            acc += p.numel()
            cur_lo = off
            self.param_bucket[id(p)] = len(bounds)
            if acc >= per:
                bounds.append((cur_lo, hi))
                hi, acc = cur_lo, 0
        if acc > 0:
            bounds.append((cur_lo, hi))
        self.bounds = bounds
        self.expected = [0] * len(bounds)
        for p in flat.params:
            self.expected[self.param_bucket[id(p)]] += 1
        self.pending = list(self.expected)
        self.handles = []
        self._hooks = []
        self.armed = True   # disarm while accumulating sub-batches; arm for the last backward
        if self.world > 1:
            for p in flat.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p):
        if not self.armed:
            return
        b = self.param_bucket[id(p)]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            lo, hi = self.bounds[b]
            self.handles.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM,
                                                group=self.group, async_op=True))

    def finish(self):
        """Wait for every bucket; flush buckets whose hooks did not all fire (unused params)."""
        if self.world > 1:
            for b, left in enumerate(self.pending):
                if left > 0:
                    lo, hi = self.bounds[b]
                    self.handles.append(dist.all_reduce(self.flat.grad[lo:hi],
                                                        op=dist.ReduceOp.SUM, group=self.group,
                                                        async_op=True))
            for h in self.handles:
                h.wait()
        self.handles = []
        self.pending = list(self.expected)
        return 1.0 / self.world


def shard_batch(tensors, rank, world):
    """Contiguous equal shards along dim 0 (what DataParallel.scatter does)."""
    out = []
    for t in tensors:
        n = t.shape[0]
        per = (n + world - 1) // world
        out.append(t[rank * per:min(n, (rank + 1) * per)])
    return out
