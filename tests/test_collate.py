"""Host side: batch assembly (edgedict_amd/collate.py) against the reference's golden vectors
(tests/golden/collate.npz, made by the reference's own functions), the oracle restatement, and the
balance properties of the rank sharding."""
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from edgedict_amd import collate
from oracle import collate_ref

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "collate.npz"))
NCASES = len([k for k in GOLD.files if k.endswith("_n")])


def _case(ci):
    meta = GOLD["c%d_meta" % ci]
    n = int(GOLD["c%d_n" % ci][0])
    seed, F = int(meta[0]), int(meta[1])
    Ts, Us = meta[2:2 + n].tolist(), meta[2 + n:2 + 2 * n].tolist()
    g = torch.Generator().manual_seed(seed)     # same draws as oracle/make_golden_collate.py
    feats = [torch.randn(t, F, generator=g) for t in Ts]
    toks = [torch.randint(4, 2048, (u,), generator=g) for u in Us]
    return feats, toks


@pytest.mark.parametrize("ci", range(NCASES))
def test_seq_collate_matches_reference_golden(ci):
    feats, toks = _case(ci)
    xs, ys, xlen, ylen = collate.seq_collate(list(zip(feats, toks)))
    assert xs.dtype == torch.float32 and ys.dtype == torch.int32
    assert xlen.dtype == torch.int32 and ylen.dtype == torch.int32
    assert np.array_equal(xs.numpy(), GOLD["c%d_xs" % ci])
    assert np.array_equal(ys.numpy(), GOLD["c%d_ys" % ci])
    assert np.array_equal(xlen.numpy(), GOLD["c%d_xlen" % ci])
    assert np.array_equal(ylen.numpy(), GOLD["c%d_ylen" % ci])


@settings(max_examples=30, deadline=None, derandomize=True)
@given(st.lists(st.tuples(st.integers(1, 9), st.integers(1, 7)), min_size=1, max_size=6), st.integers(1, 4))
def test_seq_collate_matches_oracle_on_ragged_batches(shape, F):
    rng = np.random.default_rng(len(shape) * 131 + F)
    res = [(rng.standard_normal((t, F)).astype(np.float32), rng.integers(4, 99, u)) for t, u in shape]
    xs, ys, xlen, ylen = collate.seq_collate([(torch.from_numpy(f), torch.from_numpy(y)) for f, y in res])
    rx, ry, rxl, ryl = collate_ref.seq_collate(res)
    assert np.array_equal(xs.numpy(), rx) and np.array_equal(ys.numpy(), ry)
    assert np.array_equal(xlen.numpy(), rxl) and np.array_equal(ylen.numpy(), ryl)


def test_end_pad_concat_default_is_long_like_the_reference():
    out = collate.end_pad_concat([torch.tensor([5, 6]), torch.tensor([7])])
    assert out.dtype == torch.long and out.tolist() == [[5, 6], [7, collate.PAD]]


def test_wave_collate_pads_raw_audio():
    res = [(torch.arange(5.0), torch.tensor([4, 5])), (torch.arange(3.0) + 10, torch.tensor([6]))]
    wave, wl, ys, yl = collate.wave_collate(res)
    assert wave.tolist() == [[0, 1, 2, 3, 4], [10, 11, 12, 0, 0]]
    assert wl.tolist() == [5, 3] and yl.tolist() == [2, 1] and ys.tolist() == [[4, 5], [6, 1]]
    assert wl.dtype == torch.int32 and ys.dtype == torch.int32


def test_reverse_sorted_by_length_is_stable():
    assert collate.reverse_sorted_by_length([3, 9, 3, 9, 1]).tolist() == [1, 3, 0, 2, 4]


@settings(max_examples=50, deadline=None, derandomize=True)
@given(st.integers(1, 8), st.integers(1, 8), st.integers(0, 10 ** 6))
def test_shard_by_length_partitions_and_balances(world, per, seed):
    rng = np.random.default_rng(seed)
    lengths = rng.integers(50, 402, world * per)
    shards = collate.shard_by_length(lengths, world)
    allidx = np.concatenate(shards)
    assert sorted(allidx.tolist()) == list(range(world * per))       # a partition
    assert all(len(s) == per for s in shards)                        # equal utterance counts
    sums = np.array([lengths[s].sum() for s in shards])
    assert sums.max() - sums.min() <= lengths.max()                  # LPT bound with equal counts
    # never worse than the reference's contiguous cut of the length-sorted batch
    order = collate.reverse_sorted_by_length(lengths)
    contig = np.array([lengths[order[r * per:(r + 1) * per]].sum() for r in range(world)])
    assert sums.max() <= contig.max()


def test_shard_by_length_rejects_uneven_batches():
    with pytest.raises(ValueError):
        collate.shard_by_length([1, 2, 3], 2)
