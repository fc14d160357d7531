"""Training step of the engine: features -> Transducer -> loss -> backward -> gradient exchange
-> Adam, all on the device.

Mirrors ``Trainer.train_step`` of the reference's log-mel trainer (cli/baseline.py:214-248;
cli/train.py:223-271 is the same loop around the FrontEnd variant): the batch is processed in
``sub_batch_size`` slices, each slice's mean loss is divided by the number of slices and
back-propagated, then gradients are optionally clipped and the optimiser steps.  Differences, all
MI355X-motivated: the log-mel front-end runs on the GPU inside the step (the reference computes
it in DataLoader workers on the CPU), gradients live in one flat buffer, data parallelism is one
process per GPU with bucketed RCCL all-reduce overlapped with the backward pass, and Adam is a
single kernel.
"""
import torch
import torch.distributed as dist

from . import config
from .dp import BucketedAllReduce
from .features import StackedLogFbank
from .flags import model_kwargs
from .models import Transducer
from .optim import FlatParams, FusedAdam, ReduceLROnPlateau, WarmupLR


class TrainEngine:
    def __init__(self, flags, vocab_size=None, device="cuda", compute_dtype=None,
                 process_group=None, state_dict=None):
        self.flags = flags
        self.device = torch.device(device)
        # The library's three internal HIP streams must exist BEFORE any other stream of this process
        # (RCCL creates its own at the first collective - the parameter broadcast below; torch's
        # stream pool at the first torch.cuda.Stream()): HIP hands out its hardware queues in stream
        # creation order and late-comers share one.  Measured: 27 ms/step vs 32-65 ms with 1-3
        # foreign streams created first (tools/stream_order_probe.py).
        from . import side
        with torch.cuda.device(self.device):
            side.stream(self.device)
        cd = config._parse(compute_dtype) if compute_dtype is not None else config.get_compute_dtype()
        self.compute_dtype = cd
        self.features = StackedLogFbank(
            n_frame=flags.downsample, pad_to_divisible=True, out_dtype=torch.float32,
            sample_rate=getattr(flags, "sample_rate", 16000), win_length=flags.win_length,
            hop_length=flags.hop_length, n_fft=flags.n_fft, n_filt=flags.feature_size,
            dither=getattr(flags, "dither", 1e-5)).to(self.device)
        # the reference's train transform ends with TimeMasking + FrequencyMasking
        # (rnnt/transforms.py:196-200); here one kernel on the resident stacked batch
        self.spec_augment = None
        if getattr(flags, "T_mask", 0) and getattr(flags, "T_num_mask", 0) or \
                getattr(flags, "F_mask", 0) and getattr(flags, "F_num_mask", 0):
            from .transforms import SpecAugment
            self.spec_augment = SpecAugment(getattr(flags, "T_mask", 0), getattr(flags, "T_num_mask", 0),
                                            getattr(flags, "F_mask", 0), getattr(flags, "F_num_mask", 0))
        self.model = Transducer(**model_kwargs(flags, vocab_size=vocab_size))
        if state_dict is not None:
            self.model.load_state_dict(state_dict)
        self.model.to(self.device)
        self.model.compute_dtype = cd
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if self.world > 1:  # identical initial weights on every rank
            for p in self.model.parameters():
                dist.broadcast(p.data, src=0, group=process_group)
        self.flat = FlatParams(self.model)
        self.optim = FusedAdam(self.flat, lr=flags.lr, max_grad_norm=getattr(flags, "gradclip", None))
        # bucket boundaries follow the order backward finishes things in: the joint, the prediction
        # network, then one bucket per encoder layer (its four LSTM tensors are contiguous in the flat
        # buffer), the small LayerNorm tensors last
        enc = self.model.encoder
        cuts = [self.model.joint.joint[0].weight, self.model.decoder.embed.weight]
        if hasattr(enc.lstm, "lstms"):
            cuts += [m.layer(0)[0] for m in enc.lstm.lstms] + [enc.lstm.projs[0][0].weight]
        # parameters whose gradients are only final at the very END of the backward pass and that nobody
        # reports (the stack's LayerNorm gradients are summed from partial rows by its last kernels): their
        # buckets are issued last on every rank, so the issue order never depends on which mechanism
        # finalised what (dp.BucketedAllReduce issues strictly in that order)
        late = [enc.norm.weight, enc.norm.bias]
        if hasattr(enc.lstm, "projs"):
            late += [q for m in enc.lstm.projs for q in m.parameters()]
        self.reducer = BucketedAllReduce(self.flat, process_group, boundaries=cuts, late=late)
        from . import dp
        # in-place accumulated gradients report here; set UNCONDITIONALLY so that an engine without an exchange
        # clears the hook of an earlier one (it would pin that engine's flat buffers)
        dp.READY_HOOK = self.reducer.ready if (self.world > 1 or self.reducer.force) else None
        # device-visible give-up words of the encoder stack's bounded in-kernel waits: the Adam kernel skips
        # its update when one is set (advisor r2: bad gradients were applied before anything raised)
        self._guard = None
        import os
        if self.device.type == "cuda" and os.environ.get("EDGEDICT_ADAM_GUARD", "1") != "0":
            from . import _lib
            with torch.cuda.device(self.device):
                self._guard = _lib.load().edgedict_stack_error_words(0)
        self.sub_batch_size = getattr(flags, "sub_batch_size", None)
        # learning-rate control of the reference's loop (cli/train.py:142-146,189-191)
        self.warmup = WarmupLR(self.optim, flags.lr, getattr(flags, "warmup_step", 0) or 0)
        self.sched = None
        if getattr(flags, "sched", False):
            self.sched = ReduceLROnPlateau(self.optim, patience=getattr(flags, "sched_patience", 1),
                                           factor=getattr(flags, "sched_factor", 0.5),
                                           min_lr=getattr(flags, "sched_min_lr", 1e-6))
        self.step_count = 0
        self._pref = None        # (key, xs, xlen, event): the front-end of the NEXT batch, computed during this step

    # ---- front-end (dither -> log-mel -> frame stacking -> SpecAugment) and its prefetch.  The reference computes the
    # features of batch n + 1 in DataLoader workers while the GPU trains on batch n (rnnt/dataset.py:102-103, num_workers
    # in cli/train.py:119-137); here the front-end is a GPU kernel chain of ~0.6 ms at the head of the step, in front of a
    # recurrence that leaves most of the chip idle for its first launches.  train_step(..., next_batch=(wave, wave_len))
    # runs the next batch's front-end on the auxiliary stream under this step's encoder forward; the next call finds it.
    # Every batch's features are computed exactly once, in batch order (same dither seeds, same mask draws as without
    # the prefetch).
    def _front_end(self, wave, wave_len):
        xs, xlen = self.features(wave, wave_len)
        if self.spec_augment is not None:
            xs = self.spec_augment(xs, xlen)
        return xs, xlen

    @staticmethod
    def _batch_key(wave, wave_len):
        return (wave.data_ptr(), tuple(wave.shape), wave._version,
                None if wave_len is None else (wave_len.data_ptr(), wave_len._version))

    def _prefetch(self, wave, wave_len):
        from . import side
        cur = torch.cuda.current_stream(self.device)
        aux = side.stream(self.device)
        aux.wait_stream(cur)                 # the waveform may have been produced on the current stream
        with torch.cuda.stream(aux):
            xs, xlen = self._front_end(wave, wave_len)
            ev = aux.record_event()
        wave.record_stream(aux)
        if wave_len is not None and wave_len.is_cuda:
            wave_len.record_stream(aux)
        self._pref = (self._batch_key(wave, wave_len), xs, xlen, ev)

    def _take_prefetched(self, wave, wave_len):
        pref, self._pref = self._pref, None
        if pref is None or pref[0] != self._batch_key(wave, wave_len):
            return None
        _, xs, xlen, ev = pref
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        xs.record_stream(cur)
        if torch.is_tensor(xlen) and xlen.is_cuda:
            xlen.record_stream(cur)
        return xs, xlen

    # ---- checkpoints in the reference's layout (cli/train.py:321-351): {'optim', 'model', 'sched'}
    def save(self, path):
        ckpt = {"optim": self.optim.state_dict(), "model": self.model.state_dict(),
                "step": self.step_count}
        if self.sched is not None:
            ckpt["sched"] = self.sched.state_dict()
        torch.save(ckpt, path)

    def load(self, path, load_optim=True):
        """Accepts the reference's checkpoints (``{'model': ...}`` with optional 'optim'/'sched',
        or a Lightning checkpoint, rnnt/models.py:368-380) and this engine's own."""
        from .models import convert_lightning2normal
        ckpt = convert_lightning2normal(torch.load(path, map_location="cpu"))
        sd = ckpt["model"] if "model" in ckpt else ckpt
        self.model.load_state_dict(sd)          # copies INTO the flat parameter buffer views
        config.bump_param_epoch()
        if load_optim and "optim" in ckpt:
            self.optim.load_state_dict(ckpt["optim"])
        if self.sched is not None and "sched" in ckpt:
            self.sched.load_state_dict(ckpt["sched"])
        self.step_count = int(ckpt.get("step", 0)) if isinstance(ckpt, dict) else 0

    def close(self):
        """Detach this engine from the process-wide hooks (the in-place gradient report of the encoder
        stack) and drop its autograd hooks; call before building another engine in the same process."""
        from . import dp
        if dp.READY_HOOK == self.reducer.ready:
            dp.READY_HOOK = None
        self.reducer.close()

    def check(self):
        """Raise if a bounded in-kernel wait of an earlier step gave up (its optimiser step was skipped on
        the device).  Reads a pinned host word: free; call after a synchronize for an up-to-date answer."""
        from . import encoder_stack
        encoder_stack.check_wsr_error()

    def validation_end(self, val_loss):
        """Call with the validation loss after each evaluation (cli/train.py:182-184,206-208)."""
        if self.sched is not None:
            self.sched.step(val_loss)

    def train_step(self, wave, wave_len, ys, ylen, next_batch=None):
        """wave f32[B,N] (device), wave_len i32[B] samples (or None), ys i32[B,U], ylen i32[B].
        next_batch: optional (wave, wave_len) of the batch the NEXT call will be given - its front-end then runs on the
        auxiliary stream during this step (one slice per step only; a different batch at the next call just discards it).
        Returns the mean loss of the local batch as a device tensor (no host sync)."""
        from . import ops
        ops.mark("step:enter")
        self.step_count += 1
        self.warmup.step(self.step_count)
        self.model.train()
        self.optim.zero_grad()
        B = wave.shape[0]
        sub = self.sub_batch_size or B
        starts = list(range(0, B, sub))
        total = None
        for s in starts:
            e = min(B, s + sub)
            self.reducer.armed = (s == starts[-1])   # exchange once, after the last accumulation
            got = self._take_prefetched(wave, wave_len) if len(starts) == 1 else None
            if got is None:
                got = self._front_end(wave[s:e], None if wave_len is None else wave_len[s:e])
            xs, xlen = got
            if next_batch is not None and len(starts) == 1:
                # the next batch's front-end: enqueued onto the auxiliary stream right in front of the joint's logits product
                # (models.py _JointLossFn) - beside that matrix-bound kernel it is nearly free, beside the encoder's
                # recurrence it cost the recurrence what it saved at the head of the step (profiles/r6_prefetch.txt)
                ops.set_hook("before_logits_gemm", lambda nb=next_batch: self._prefetch(nb[0], nb[1]))
            try:
                loss = self.model(xs, ys[s:e], xlen, ylen[s:e])
            finally:
                late = ops.pop_hook("before_logits_gemm")
            if late is not None:
                late()                         # a forward pass that did not reach the hook (unfused path): behind it
            loss = loss / len(starts)
            ops.mark("backward:enter")
            loss.backward()
            ops.mark("backward:exit")
            total = loss.detach() if total is None else total + loss.detach()
        scale = self.reducer.finish()
        self.optim.step(grad_scale=scale, guard=self._guard)
        if self.compute_dtype == torch.bfloat16 and config.USE_ENCODER_STACK:
            # next step's weight images, beside its front-end (after Adam in stream order)
            from . import encoder_stack, side
            aux = side.stream(self.device)
            aux.wait_stream(torch.cuda.current_stream(self.device))
            encoder_stack.prepack(self.model.encoder, aux)
        ops.mark("step:exit")
        if self.step_count == 1:
            # Everything long-lived (modules, plans, weight images, torch itself) exists now.  A full
            # cyclic-GC pass over those ~millions of objects takes 20-35 ms of host time and lands in
            # the middle of a step every few dozen steps (tools/step_trend.py: 27 ms steps with 40-60
            # ms outliers); frozen, later collections only look at young objects.
            import gc
            gc.collect()
            gc.freeze()
        return total
