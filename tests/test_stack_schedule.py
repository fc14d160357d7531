"""CPU: the encoder stack's dynamic wavefront schedules (csrc/encoder_stack.hip) run DRY through
``edgedict_stack_schedule`` - no device, nothing launched - and checked for the invariants the
kernels rely on: every layer-step is carried by exactly one launch, in recurrence order; a layer
never opens a chunk before the side-stream product that feeds it was enqueued, and that product is
enqueued only after the producing layer finished (and, forward, normalised) the chunk's frames; a
launch never carries more layer-steps than the launch structure holds; the E6D2 launch counts.

Two forward schedules: the launch-per-step one (EDGEDICT_STACK_LPW=0: every layer steps at most once per
launch) and the default launch-persistent one (forward_lpw: a launch carries up to EDGEDICT_LPW_STEPS
CONSECUTIVE steps of each of its layers, never across a chunk boundary).  The backward pass is launch-per-step."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from edgedict_amd import encoder_stack as es

MAX_SLOTS = 8


def _geometry(T0, reductions, chunk):
    L = len(reductions)
    Ts, T = [], T0
    for r in reductions:
        Ts.append(T)
        T = (T + r - 1) // r
    f = int(np.prod(reductions))
    fl = []
    for r in reductions:
        fl.append(f)
        f //= r
    cf = [chunk * x for x in fl]
    return L, Ts, cf


def _sched(lpw, *args, **kw):
    """lpw: the launch-persistent forward kernel AND the split-K BPTT kernel (both carry several steps of a
    layer per launch); False: the launch-per-step kernels of both passes."""
    keys = ("EDGEDICT_STACK_LPW", "EDGEDICT_STACK_BWD_SK")
    old = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ[k] = "1" if lpw else "0"
    try:
        return es.schedule(*args, **kw)
    finally:
        for k in keys:
            if old[k] is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = old[k]


def _check_lpw(T0, reductions, chunk, nsub, H=64, B=3):
    """forward_lpw: macro-steps of <= nsub consecutive steps."""
    L, Ts, cf = _geometry(T0, reductions, chunk)
    old = os.environ.get("EDGEDICT_LPW_STEPS")
    os.environ["EDGEDICT_LPW_STEPS"] = str(nsub)
    try:
        steps, enq, n, max_slots = _sched(True, T0, 64, H, reductions, B=B, chunk=chunk, backward=False)
    finally:
        if old is None:
            os.environ.pop("EDGEDICT_LPW_STEPS", None)
        else:
            os.environ["EDGEDICT_LPW_STEPS"] = old
    wgs = (H // 16) * ((B + 63) // 64)
    assert 1 <= max_slots <= min(MAX_SLOTS, 256 // wgs)
    slots = np.zeros(n, dtype=int)
    longest = 0
    for l in range(L):
        s = steps[l].astype(int)
        assert len(s) == Ts[l] and (s >= 0).all() and (s < n).all()
        assert (np.diff(s) >= 0).all(), "recurrence order"
        for w in np.unique(s):
            ts = np.nonzero(s == w)[0]
            assert (np.diff(ts) == 1).all(), "a launch carries CONSECUTIVE steps of a layer"
            assert ts[0] // cf[l] == ts[-1] // cf[l], "never across a chunk boundary"
            longest = max(longest, len(ts))
            slots[w] += 1
    assert slots.max() == max_slots and slots.min() >= 1
    for l in range(1, L):
        for k in range(len(enq[l])):
            opening = int(steps[l][k * cf[l]])
            e = int(enq[l][k])
            last = int(steps[l - 1][min(Ts[l - 1], (k + 1) * cf[l - 1]) - 1])
            assert e == last + 1, "the product is enqueued right behind the launch that finishes the chunk"
            # margin 2: normally a whole launch lies between; when nothing else is runnable the scheduler does
            # not wait with empty launches - the consumer's workgroups then poll the chunk's flag themselves
            assert opening >= e, "the consumer's launch is enqueued after the product"
    return n, longest


def _check_sk(T0, reductions, chunk, nsub, H=64, B=3):
    """the split-K BPTT's macro-steps: <= nsub consecutive steps of a layer per launch, descending t."""
    L, Ts, cf = _geometry(T0, reductions, chunk)
    old = os.environ.get("EDGEDICT_SK_STEPS")
    os.environ["EDGEDICT_SK_STEPS"] = str(nsub)
    try:
        steps, enq, n, max_slots = _sched(True, T0, 64, H, reductions, B=B, chunk=chunk, backward=True)
    finally:
        if old is None:
            os.environ.pop("EDGEDICT_SK_STEPS", None)
        else:
            os.environ["EDGEDICT_SK_STEPS"] = old
    wgs = (H // 64) * 4
    assert 1 <= max_slots <= min(MAX_SLOTS, 256 // wgs)
    slots = np.zeros(n, dtype=int)
    longest = 0
    for l in range(L):
        s = steps[l].astype(int)
        assert len(s) == Ts[l] and (s >= 0).all() and (s < n).all()
        assert (np.diff(s[::-1]) >= 0).all(), "recurrence order (last frame first)"
        for w in np.unique(s):
            ts = np.nonzero(s == w)[0]
            assert (np.diff(ts) == 1).all(), "a launch carries CONSECUTIVE steps of a layer"
            assert ts[0] // cf[l] == ts[-1] // cf[l], "never across a chunk boundary"
            longest = max(longest, len(ts))
            slots[w] += 1
    assert slots.max() == max_slots
    for l in range(L - 1):
        for k in range(len(enq[l])):
            hi = min(Ts[l], (k + 1) * cf[l]) - 1
            opening, e = int(steps[l][hi]), int(enq[l][k])
            last = int(steps[l + 1][k * cf[l + 1]])          # the producer's lowest frame of the chunk runs last
            assert e >= last + 1 and opening >= e, (l, k, e, last, opening)
    return n, longest


def _check(T0, reductions, chunk, backward):
    L, Ts, cf = _geometry(T0, reductions, chunk)
    steps, enq, n, max_slots = _sched(False, T0, 64, 64, reductions, B=3, chunk=chunk, backward=backward)
    assert 1 <= max_slots <= MAX_SLOTS
    per_launch = np.zeros(n, dtype=int)
    for l in range(L):
        s = steps[l].astype(int)
        assert len(s) == Ts[l] and (s >= 0).all() and (s < n).all()
        order = s if not backward else s[::-1]
        assert (np.diff(order) > 0).all(), "a layer steps at most once per launch, in recurrence order"
        np.add.at(per_launch, s, 1)
    assert per_launch.max() == max_slots and (per_launch >= 0).all()
    for l in range(L):
        nch = len(enq[l])
        for k in range(nch):
            lo, hi = k * cf[l], min(Ts[l], (k + 1) * cf[l]) - 1
            opening = steps[l][hi] if backward else steps[l][lo]      # first step of the chunk in run order
            e = int(enq[l][k])
            src = (l + 1) if backward else (l - 1)
            if src < 0 or src >= L:
                assert e <= opening                                  # fed from the input / the loss side
                continue
            assert e >= 0 and opening >= e, "consumer launch comes after the product was enqueued"
            # the producing layer's frames of this chunk (same chunk index, its own frame rate)
            plo, phi = k * cf[src], min(Ts[src], (k + 1) * cf[src]) - 1
            last = int(steps[src][plo] if backward else steps[src][phi])
            # backward: enqueued right after the producer's launch; forward: after the launch that
            # carries the LayerNorm of the producer's last frame, i.e. one launch later
            assert e >= last + (1 if backward else 2), (l, k, e, last)
    return n


@pytest.mark.parametrize("backward", [False, True])
def test_e6d2_schedule(backward):
    n = _check(401, [1, 2, 1, 1, 1, 1], 12, backward)
    # static-lag schedule: 547 / 546 launches; dynamic: layers behind the reduction run every launch
    # when no faster layer runs beside them
    assert n == (501 if backward else 487)


@settings(max_examples=60, deadline=None, derandomize=True)
@given(st.integers(1, 90), st.lists(st.sampled_from([1, 1, 2]), min_size=1, max_size=6),
       st.integers(1, 6), st.booleans())
def test_schedule_invariants_on_random_geometries(T0, reductions, chunk, backward):
    if int(np.prod(reductions)) > 4:          # at most two time reductions (period 4), as tested on the GPU
        reductions = [1 if i > 1 else r for i, r in enumerate(reductions)]
    _check(T0, reductions, chunk, backward)


def test_e6d2_launch_persistent_forward_schedule():
    # 64 x H=1024: 64 workgroups per layer, at most 4 layers per launch (256 CUs)
    n, longest = _check_lpw(401, [1, 2, 1, 1, 1, 1], 12, 12, H=1024, B=64)
    assert longest == 12 and n <= 60
    n6, longest6 = _check_lpw(401, [1, 2, 1, 1, 1, 1], 12, 6, H=1024, B=64)
    assert longest6 == 6 and n6 == 83


@settings(max_examples=60, deadline=None, derandomize=True)
@given(st.integers(1, 90), st.lists(st.sampled_from([1, 1, 2]), min_size=1, max_size=6),
       st.sampled_from([2, 4, 6, 8]), st.sampled_from([2, 4, 6, 12]))
def test_launch_persistent_schedule_invariants_on_random_geometries(T0, reductions, chunk, nsub):
    if int(np.prod(reductions)) > 4:
        reductions = [1 if i > 1 else r for i, r in enumerate(reductions)]
    n, longest = _check_lpw(T0, reductions, chunk, nsub)
    assert longest <= nsub


def test_e6d2_split_k_bptt_schedule():
    # 64 x H=1024: 64 workgroups per layer (16 unit blocks x 4 K quarters), at most 4 layers per launch
    n, longest = _check_sk(401, [1, 2, 1, 1, 1, 1], 12, 12, H=1024, B=64)
    assert longest in (12, 13) and n <= 50
    n6, longest6 = _check_sk(401, [1, 2, 1, 1, 1, 1], 12, 6, H=1024, B=64)
    assert longest6 in (6, 7) and n6 <= 90
    # the benched default: chunks of 16 output frames, a chunk per launch.  Layers 0 and 1 have 401 = 25 x 16 + 1 frames:
    # the left-over frame rides along with the first launch of its chunk (17 steps) instead of costing the layer - and the
    # critical path below it - a launch of its own (37 launches before round 6)
    n16, longest16 = _check_sk(401, [1, 2, 1, 1, 1, 1], 16, 16, H=1024, B=64)
    assert longest16 == 17 and n16 == 36


@settings(max_examples=60, deadline=None, derandomize=True)
@given(st.integers(1, 90), st.lists(st.sampled_from([1, 1, 2]), min_size=1, max_size=6),
       st.sampled_from([2, 4, 6, 8]), st.sampled_from([2, 4, 6, 12]))
def test_split_k_bptt_schedule_invariants_on_random_geometries(T0, reductions, chunk, nsub):
    if int(np.prod(reductions)) > 4:
        reductions = [1 if i > 1 else r for i, r in enumerate(reductions)]
    n, longest = _check_sk(T0, reductions, chunk, nsub)
    assert longest <= nsub + 1          # (+ 1: a single left-over frame of a chunk rides along with the launch before)
