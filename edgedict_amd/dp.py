"""Data-parallel gradient exchange: one process per GPU, RCCL (``backend='nccl'``) over xGMI.

The reference scales with ``nn.DataParallel`` (cli/train.py:152-153: broadcast + reduce to GPU0)
or Lightning DDP (cli/lightning.py:325-331: bucketed all-reduce overlapped with backward).  Here
utterances are sharded over ranks with no data-path collective; the only exchange is the gradient
all-reduce, issued per *bucket* - a contiguous slice of the flat gradient buffer
(optim.FlatParams) - as soon as every parameter in it is final, so RCCL traffic overlaps the rest
of the backward pass.  A parameter becomes final in one of two ways:

  * autograd accumulated it (``register_post_accumulate_grad_hook``): the fp32 parity mode, the
    prediction network, the projections;
  * the engine accumulated it straight into the flat buffer on the auxiliary stream, behind
    autograd's back (bf16 mode: the encoder stack's and the joint network's weight gradients, 99 % of
    the bytes) and reports it through ``ready(params, stream)`` - from the joint's backward node, and
    from INSIDE the encoder stack's native backward call, per layer, the moment that layer's last
    weight-gradient kernel is enqueued (``edgedict_stack_desc_t.grads_final``).  The collective is
    then ordered behind that stream's position: the buckets of encoder layers 5 ... 1 leave while
    the layers below are still in their BPTT.

Buckets are cut at the boundaries the engine passes (one per encoder layer: 34 MB of fp32, the
joint, the rest), never smaller than ``min_bytes``: xGMI is point-to-point (7 links per GPU), so a
ring step should move MBs per link rather than pay latency many times.  The sum is turned into the
mean inside the Adam kernel (``grad_scale = 1/world``).
"""
import torch
import torch.distributed as dist

import os

# set by the training engine when world_size > 1: callable(params, stream) -> None
READY_HOOK = None
# smallest bucket cut at a boundary (bytes); tests shrink it to exercise per-layer buckets on tiny models
MIN_BUCKET_BYTES = int(os.environ.get("EDGEDICT_DP_MIN_BUCKET", str(1 << 20)))


class BucketedAllReduce:
    def __init__(self, flat, process_group=None, bucket_bytes=64 << 20, boundaries=(), min_bytes=None,
                 late=()):
        """``boundaries``: parameters at which a new bucket must start (in parameter order).
        ``late``: parameters that are only final at the end of the backward pass; a bucket holding one is
        issued after every other bucket.

        Buckets leave STRICTLY in ``issue_order`` (buckets from the end of the flat buffer to its start -
        the order backward finishes them in - with the late ones moved to the end): a bucket that completes
        before its predecessors is held back.  Which mechanism completes a bucket (autograd hook, the
        engine's ``ready()`` report, ``finish()``) and when may differ between ranks - e.g. a rank whose
        batch is too short for the wavefront stack takes the per-layer autograd path - but every rank then
        still issues the same collectives in the same order (RCCL matches collectives by issue order)."""
        self.flat = flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # tools/overlap_probe.py: exercise the whole issue path on ONE rank (a 1-rank communicator)
        self.force = os.environ.get("EDGEDICT_DP_FORCE", "0") == "1" and dist.is_initialized()
        # how a bucket is issued: "stream" = async_op=False, which ProcessGroupNCCL enqueues on the CURRENT
        # stream - the one that accumulated the gradients (auxiliary stream for the in-place accumulated ones,
        # the caller's for hooks and finish()); "async" = async_op=True on the process group's own stream,
        # ordered behind the current one by an event.  "async" puts a FIFTH active stream beside the engine's
        # four, and HIP has four hardware queues: measured on one rank (tools/overlap_probe.py nccl) 36-40 ms
        # per step instead of 26, even when every bucket leaves after the backward pass.
        self.early_mode = os.environ.get("EDGEDICT_DP_EARLY", "stream")
        per = max(1, bucket_bytes // 4)
        lo_min = max(1, (MIN_BUCKET_BYTES if min_bytes is None else min_bytes) // 4)
        starts = {id(p) for p in boundaries}
        # buckets are built from the END of the flat buffer: backward reaches those params first
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
            if acc >= per or (id(p) in starts and acc >= lo_min):
                bounds.append((cur_lo, hi))
                hi, acc = cur_lo, 0
        if acc > 0:
            bounds.append((cur_lo, hi))
        self.bounds = bounds
        self.expected = [0] * len(bounds)
        for p in flat.params:
            self.expected[self.param_bucket[id(p)]] += 1
        self.pending = list(self.expected)
        self._seen = set()                  # parameters counted in this step (_done)
        self.issued = [False] * len(bounds)
        late_b = {self.param_bucket[id(p)] for p in late if id(p) in self.param_bucket}
        self.issue_order = [b for b in range(len(bounds)) if b not in late_b] + sorted(late_b)
        self.complete = [None] * len(bounds)     # None, or the streams the bucket's gradients were produced on
        self._next = 0                           # position in issue_order of the next bucket to leave
        self.handles = []
        self.issued_early = 0     # buckets that left before finish() in this step ...
        self.last_issued_early = 0   # ... and in the last finished one (reporting / tests)
        self.early_buckets, self.last_early_buckets = [], []    # their indices, in issue order
        self.early_by, self.last_early_by = [], []              # 'ready' / 'hook': what completed them
        self.ready_calls = 0         # ready() invocations in the last finished step
        self._ready_calls = 0
        self._hooks = []
        self.armed = True   # disarm while accumulating sub-batches; arm for the last backward
        # EDGEDICT_DP_OVERLAP=0: every bucket leaves from finish(), after the backward pass (to weigh the
        # overlap against what RCCL's kernels cost the BPTT beside them, on a multi-GPU box)
        self.overlap = os.environ.get("EDGEDICT_DP_OVERLAP", "1") != "0"
        if self.world > 1 or self.force:
            for p in flat.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _issue(self, b):
        lo, hi = self.bounds[b]
        self.issued[b] = True
        if self.early_mode == "stream":
            # async_op=False: ProcessGroupNCCL enqueues the collective on the CURRENT stream (no stream of its
            # own, no cross-stream wait); nothing to wait for later - consumers are ordered by that stream
            dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=False)
            return
        self.handles.append(dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM,
                                            group=self.group, async_op=True))

    def _exchange_stream(self):
        """Stream mode: every collective issued DURING the backward pass goes onto ONE stream, the engine's
        auxiliary stream.  A communicator orders its collectives globally - one issued on the caller's stream
        would wait there for the earlier ones still queued behind weight-gradient products on the auxiliary
        stream, and the caller's stream is where the backward pass itself is enqueued."""
        if not self.flat.grad.is_cuda:
            return None
        from . import side
        return side.peek(self.flat.grad.device)

    def _done(self, p, stream=None):
        b = self.param_bucket.get(id(p))
        if b is None or self.issued[b] or self.complete[b] is not None:
            return
        # once per parameter and step: a parameter whose gradient was accumulated in place is reported through ready()
        # AND (torch 2.10: the post-accumulate hook also fires for a None gradient) by its autograd hook - counted
        # twice, a bucket would leave before its other parameters are final (found by tests/test_dp_gpu.py when the
        # per-layer LSTM blocks and nn.Linear started to report, round 6)
        if id(p) in self._seen:
            return
        self._seen.add(id(p))
        self.pending[b] -= 1
        if self.pending[b] > 0:
            return
        dev = self.flat.grad.device
        srcs = []
        if self.flat.grad.is_cuda:
            # the bucket may mix hook-accumulated parameters (produced on the caller's stream) and in-place
            # accumulated ones (on `stream`): the collective waits for both
            srcs = [torch.cuda.current_stream(dev)]
            if stream is not None and stream != srcs[0]:
                srcs.append(stream)
        self.complete[b] = (srcs, "ready" if stream is not None else "hook")
        self._drain()

    def _drain(self):
        """Issue, in issue_order, every bucket whose predecessors are out and that is complete."""
        while self._next < len(self.issue_order):
            b = self.issue_order[self._next]
            if self.issued[b]:
                self._next += 1
                continue
            if self.complete[b] is None:
                return
            srcs, by = self.complete[b]
            self.issued_early += 1
            self.early_buckets.append(b)
            self.early_by.append(by)
            ex = self._exchange_stream() if self.early_mode == "stream" else None
            if ex is not None:
                for src in srcs:
                    if src != ex:
                        ex.wait_stream(src)          # the gradients were produced up to here on `src`
                with torch.cuda.stream(ex):
                    self._issue(b)
            elif len(srcs) < 2:
                self._issue(b)
            else:
                # the collective is ordered behind the stream the gradients were accumulated on
                srcs[1].wait_stream(srcs[0])
                with torch.cuda.stream(srcs[1]):
                    self._issue(b)
            self._next += 1

    def _on_grad(self, p):
        if self.armed and self.overlap and (self.world > 1 or self.force):
            self._done(p)

    def ready(self, params, stream=None):
        """``params`` were accumulated in place (no autograd hook fires for them) by work enqueued
        on ``stream`` (None = the current stream) up to this moment."""
        if not self.armed or not self.overlap or (self.world <= 1 and not self.force):
            return
        self._ready_calls += 1
        for p in params:
            self._done(p, stream)

    def finish(self):
        """Flush the buckets that are not out yet (parameters finalised at the very end of the
        backward pass, unused parameters), then wait for every bucket."""
        if self.world > 1 or self.force:
            ex = self._exchange_stream() if self.early_mode == "stream" else None
            if ex is not None:
                # the early collectives (and the gradients accumulated on that stream) first; the rest follows
                # on the caller's stream, where the optimiser step is enqueued next
                torch.cuda.current_stream(self.flat.grad.device).wait_stream(ex)
            for b in self.issue_order:
                if not self.issued[b]:
                    self._issue(b)
            for h in self.handles:
                h.wait()
        self.handles = []
        self._seen = set()
        self.pending = list(self.expected)
        self.issued = [False] * len(self.bounds)
        self.complete = [None] * len(self.bounds)
        self._next = 0
        self.last_issued_early, self.issued_early = self.issued_early, 0
        self.last_early_buckets, self.early_buckets = self.early_buckets, []
        self.last_early_by, self.early_by = self.early_by, []
        self.ready_calls, self._ready_calls = self._ready_calls, 0
        return 1.0 / self.world

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def shard_batch(tensors, rank, world):
    """Contiguous equal shards along dim 0 (what DataParallel.scatter does)."""
    out = []
    for t in tensors:
        n = t.shape[0]
        per = (n + world - 1) // world
        out.append(t[rank * per:min(n, (rank + 1) * per)])
    return out
