"""Scalar statistics collection with the reference's interface (torch_utils/training_stats.py:
``init_multiprocessing``, ``report``, ``report0``, ``Collector``).  Observability, not compute; kept
because the losses call ``report`` on the step path and ADA reads 'Loss/signs/real' through a
Collector.  One all_reduce of [n_names, 3] float64 per ``Collector.update()`` (:262), as upstream.

MI355X note: ``report`` is a flatten, a cast, three reductions, a stack, a cast and an add per call upstream (~75 calls per
training step).  For GPU tensors the three moments are accumulated into the float64 counter by ONE native launch
(csrc/stats.hip, ``mg_stats_accumulate``); CPU tensors and python scalars take the aten expression.
``enabled = False`` turns reporting into a no-op.
"""
import re

import numpy as np
import torch

from . import native

enabled = True
_num_moments = 3
_reduce_dtype = torch.float32
_counter_dtype = torch.float64
_rank = 0
_sync_device = None
_sync_called = False
_counters = dict()      # name -> device -> float64[3] running [count, sum, sum of squares]
_cumulative = dict()    # name -> float64[3] on the CPU, updated by _sync()
_native_dtypes = (torch.float32, torch.float16, torch.bfloat16)


def init_multiprocessing(rank, sync_device):
    global _rank, _sync_device
    assert not _sync_called
    _rank, _sync_device = rank, sync_device


def report(name, value):
    if name not in _counters:
        _counters[name] = dict()
    if not enabled:
        return value
    elems = torch.as_tensor(value)
    if elems.numel() == 0:
        return value
    if elems.is_cuda and elems.dtype in _native_dtypes:
        # MI355X path: ONE native launch adds [count, sum, sum of squares] to the float64 counter (csrc/stats.hip)
        x = elems.detach()
        if not x.is_contiguous():
            x = x.contiguous()
        slot = _counters[name].get(x.device)
        if slot is None:
            slot = _counters[name][x.device] = torch.zeros([_num_moments], dtype=_counter_dtype, device=x.device)
        with native.guard(x.device):
            native.check(native.lib().mg_stats_accumulate(x.data_ptr(), native.dtype_code(x.dtype), x.numel(), slot.data_ptr(),
                                                          native.stream_ptr(x.device)))
        return value
    elems = elems.detach().flatten().to(_reduce_dtype)
    moments = torch.stack([elems.new_full([], float(elems.numel())), elems.sum(), elems.square().sum()]).to(_counter_dtype)
    slot = _counters[name].get(moments.device)
    if slot is None:
        _counters[name][moments.device] = moments.clone()
    else:
        slot.add_(moments)
    return value


def report0(name, value):
    report(name, value if _rank == 0 else [])
    return value


class Collector:
    """Long-term averages of everything reported under names matching `regex` (reference :105-229)."""

    def __init__(self, regex='.*', keep_previous=True):
        self._regex = re.compile(regex)
        self._keep_previous = keep_previous
        self._cumulative = dict()
        self._moments = dict()
        self.update()
        self._moments.clear()

    def names(self):
        return [name for name in _counters if self._regex.fullmatch(name)]

    def update(self):
        if not self._keep_previous:
            self._moments.clear()
        for name, cumulative in _sync(self.names()):
            if name not in self._cumulative:
                self._cumulative[name] = torch.zeros([_num_moments], dtype=_counter_dtype)
            delta = cumulative - self._cumulative[name]
            self._cumulative[name].copy_(cumulative)
            if float(delta[0]) != 0:
                self._moments[name] = delta

    def _get_delta(self, name):
        assert self._regex.fullmatch(name)
        if name not in self._moments:
            self._moments[name] = torch.zeros([_num_moments], dtype=_counter_dtype)
        return self._moments[name]

    def num(self, name):
        return int(self._get_delta(name)[0])

    def mean(self, name):
        d = self._get_delta(name)
        return float('nan') if int(d[0]) == 0 else float(d[1] / d[0])

    def std(self, name):
        d = self._get_delta(name)
        if int(d[0]) == 0 or not np.isfinite(float(d[1])):
            return float('nan')
        if int(d[0]) == 1:
            return 0.0
        mean, raw_var = float(d[1] / d[0]), float(d[2] / d[0])
        return np.sqrt(max(raw_var - np.square(mean), 0))

    def as_dict(self):
        return {name: dict(num=self.num(name), mean=self.mean(name), std=self.std(name)) for name in self.names()}

    def __getitem__(self, name):
        return self.mean(name)


def _sync(names):
    """Fold the per-device counters into the CPU cumulative table; across ranks with ONE all_reduce."""
    if len(names) == 0:
        return []
    global _sync_called
    _sync_called = True
    device = _sync_device if _sync_device is not None else torch.device('cpu')
    deltas = []
    for name in names:
        delta = torch.zeros([_num_moments], dtype=_counter_dtype, device=device)
        for counter in _counters[name].values():
            delta.add_(counter.to(device))
            counter.zero_()
        deltas.append(delta)
    deltas = torch.stack(deltas)
    if _sync_device is not None:
        torch.distributed.all_reduce(deltas)
    deltas = deltas.cpu()
    out = []
    for idx, name in enumerate(names):
        if name not in _cumulative:
            _cumulative[name] = torch.zeros([_num_moments], dtype=_counter_dtype)
        _cumulative[name].add_(deltas[idx])
        out.append((name, _cumulative[name]))
    return out
