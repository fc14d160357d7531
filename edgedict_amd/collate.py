"""Batch assembly either side of the hot path (host side).

Mirrors the reference's collate functions (rnnt/dataset.py:202-240): ``zero_pad_concat`` (features
padded with zeros to the longest item), ``end_pad_concat`` (token rows padded with PAD=1),
``seq_collate`` -> ``(xs f32[B,maxT,...], ys i32[B,maxU], xlen i32[B], ylen i32[B])`` with the
same dtypes and values.  What is MI355X-specific:

* the padded batch is written ONCE, straight into pinned host memory (``pin=True``), so the upload
  is a single asynchronous DMA and the lengths stay on the host (``Transducer.forward`` takes host
  lengths and then runs joint + loss on the packed lattice without a device sync);
* ``wave_collate`` is the same for RAW waveforms: the engine computes log-mel features on the GPU
  inside the step (``TrainEngine.train_step(wave, wave_len, ys, ylen)``), where the reference runs
  them in DataLoader workers on the CPU (rnnt/dataset.py:98-104);
* ``shard_by_length`` deals utterances to data-parallel ranks so that every rank gets the same
  number of utterances and nearly the same number of frames (the recurrence's cost is set by the
  LONGEST utterance of a rank's batch, the joint's by the sum), where ``DataParallel.scatter``
  (cli/train.py:152-153) cuts contiguous chunks of a length-sorted batch
  (``reverse_sorted_by_length``, rnnt/dataset.py:78-80) and gives rank 0 all the long ones.
"""
import numpy as np
import torch

PAD = 1   # rnnt/tokenizer.py: NUL=0 (blank), PAD=1, BOS=2, UNK=3


def _alloc(shape, dtype, pin):
    out = torch.zeros(shape, dtype=dtype)
    if pin and torch.cuda.is_available():
        out = out.pin_memory()
    return out


def zero_pad_concat(feats, pin=False):
    """list of [T_i, ...] tensors -> f32 [B, max T_i, ...], zero padded (rnnt/dataset.py:202-211)."""
    max_t = max(len(f) for f in feats)
    out = _alloc((len(feats), max_t) + tuple(feats[0].shape[1:]), torch.float32, pin)
    for i, f in enumerate(feats):
        out[i, :len(f)] = f
    return out


def end_pad_concat(texts, pin=False, dtype=torch.long):
    """list of 1-D token tensors -> [B, max U_i] padded with PAD (rnnt/dataset.py:214-222)."""
    max_u = max(len(t) for t in texts)
    out = _alloc((len(texts), max_u), dtype, pin)
    out.fill_(PAD)
    for i, t in enumerate(texts):
        out[i, :len(t)] = torch.as_tensor(t)
    return out


def seq_collate(results, pin=False):
    """[(feat [T_i, F], tokens [U_i]), ...] -> (xs, ys i32, xlen i32, ylen i32), all on the host
    (rnnt/dataset.py:225-240)."""
    xs = [r[0] for r in results]
    ys = [r[1] for r in results]
    xlen = torch.from_numpy(np.array([len(x) for x in xs])).int()
    ylen = torch.from_numpy(np.array([len(y) for y in ys])).int()
    return zero_pad_concat(xs, pin), end_pad_concat(ys, pin, torch.int32), xlen, ylen


def wave_collate(results, pin=False):
    """[(waveform f32[N_i], tokens [U_i]), ...] -> (wave f32[B, max N_i], wave_len i32[B], ys i32,
    ylen i32): the batch the GPU front-end takes."""
    waves = [torch.as_tensor(r[0], dtype=torch.float32).reshape(-1) for r in results]
    ys = [r[1] for r in results]
    wave_len = torch.from_numpy(np.array([len(w) for w in waves])).int()
    ylen = torch.from_numpy(np.array([len(y) for y in ys])).int()
    return zero_pad_concat(waves, pin), wave_len, end_pad_concat(ys, pin, torch.int32), ylen


def to_device(batch, device):
    """Upload the padded tensors of a collated batch asynchronously; the length vectors stay on the
    host (as pinned copies) for the sync-free slicing in ``Transducer.forward``."""
    out = []
    for t in batch:
        if t.dim() == 1 and t.dtype == torch.int32:
            out.append(t)
        else:
            out.append(t.to(device, non_blocking=True))
    return tuple(out)


def reverse_sorted_by_length(lengths):
    """Indices in decreasing length order, ties in input order (the ``sorted(..., reverse=True)``
    of rnnt/dataset.py:78-80 is stable)."""
    lengths = np.asarray(lengths)
    return np.argsort(-lengths, kind="stable")


def shard_by_length(lengths, world_size):
    """Deal ``len(lengths)`` utterances (a multiple of ``world_size``) to ranks: returns
    ``world_size`` index arrays of equal size.  Utterances are taken in decreasing length; each goes
    to the rank with the smallest frame total among those that still have room, so both the
    per-rank maximum and the per-rank sum are balanced (longest-processing-time rule)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if n % world_size:
        raise ValueError("batch of %d does not split over %d ranks" % (n, world_size))
    per = n // world_size
    shards = [[] for _ in range(world_size)]
    load = np.zeros(world_size, dtype=np.int64)
    for i in reverse_sorted_by_length(lengths):
        free = [r for r in range(world_size) if len(shards[r]) < per]
        r = min(free, key=lambda q: (load[q], q))
        shards[r].append(int(i))
        load[r] += lengths[i]
    return [np.array(s, dtype=np.int64) for s in shards]
