"""`asyncflow_amd.results.differing_scenarios` on CPU tensors: what the whole-batch GPU parity tests rest on."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from asyncflow_amd import _abi
from asyncflow_amd.results import differing_scenarios


def _batch(n=7, cap=11, ticks=9, pitch=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    counts = torch.zeros((n, _abi.CNT_SLOTS), dtype=torch.int32)
    counts[:, _abi.CNT_COMPLETED] = torch.randint(0, cap + 1, (n,), generator=g, dtype=torch.int32)
    counts[:, _abi.CNT_TICKS] = torch.randint(0, ticks + 1, (n,), generator=g, dtype=torch.int32)
    counts[:, _abi.CNT_EVENTS] = -5          # a u32 above 2^31 stored in int32
    clock = torch.rand((n, cap, 2), generator=g, dtype=torch.float64)
    samples = torch.randint(0, 50, (n, ticks, pitch), generator=g, dtype=torch.int32)
    return counts, clock, samples


def test_identical_batches_and_garbage_behind_the_counts():
    a = _batch()
    b = tuple(t.clone() for t in a)
    for i in range(a[0].shape[0]):       # rows a scenario did not reach are uninitialised memory in the real buffers
        b[1][i, int(a[0][i, _abi.CNT_COMPLETED]):] = float("nan")
        b[2][i, int(a[0][i, _abi.CNT_TICKS]):] = -1
    assert differing_scenarios(*a, *b).size == 0
    assert differing_scenarios(*a, *b, chunk=2).size == 0


@pytest.mark.parametrize("chunk", [1, 3, 512])
def test_single_differences_are_found(chunk):
    a = _batch(n=9)
    a[0][:, _abi.CNT_COMPLETED] = 6
    a[0][:, _abi.CNT_TICKS] = 5
    b = tuple(t.clone() for t in a)
    b[1].view(torch.int64)[2, 5, 1] ^= 1             # last completed row, one mantissa bit
    b[2][4, 4, 3] += 1                               # last tick reached
    b[0][7, _abi.CNT_FLAGS] = 16
    b[0][8, _abi.CNT_MAX_LIVE] = 99                  # a diagnostic of one kernel family: not a result
    assert differing_scenarios(*a, *b, chunk=chunk).tolist() == [2, 4, 7]


def test_negative_zero_and_nan_are_bit_patterns():
    a = _batch(n=2)
    a[0][:, _abi.CNT_COMPLETED] = 3
    b = tuple(t.clone() for t in a)
    a[1][0, 0, 0], b[1][0, 0, 0] = 0.0, -0.0
    a[1][1, 1, 1] = b[1][1, 1, 1] = float("nan")
    assert differing_scenarios(*a, *b).tolist() == [0]


def test_shapes_must_agree():
    a, b = _batch(n=3), _batch(n=4)
    with pytest.raises(ValueError, match="batches of"):
        differing_scenarios(*a, *b)
    with pytest.raises(ValueError, match="kept an output"):
        differing_scenarios(a[0], a[1], None, a[0], a[1], a[2])
    assert differing_scenarios(a[0], None, None, a[0], None, None).size == 0
    assert isinstance(differing_scenarios(*a, *a), np.ndarray)
