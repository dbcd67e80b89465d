"""The parts of the reference's torch_utils/misc.py that sit on the training-step path
(``constant``, ``nan_to_num``, ``assert_shape``, ``profiled_function``,
``InfiniteSampler``, ``copy_params_and_buffers``, ``ddp_sync``, ``check_ddp_consistency``).
Same names and argument meaning as montage_gan/torch_utils/misc.py; written for torch >= 2."""
import contextlib
import re
import warnings

import numpy as np
import torch

_constant_cache = dict()
_constant_fast = dict()


def constant(value, shape=None, dtype=None, device=None, memory_format=None):
    """Cached constant tensor: avoids a host->device copy per use (misc.py:22-43)."""
    fast = None
    if type(value) in (int, float):       # python scalars (ADA's matrix entries): skip the array round trip for the key
        fast = (value, type(value), shape if shape is None else tuple(shape), dtype, device, memory_format)
        hit = _constant_fast.get(fast)
        if hit is not None:
            return hit
    value = np.asarray(value)
    shape = None if shape is None else tuple(shape)
    dtype = dtype or torch.get_default_dtype()
    device = torch.device('cpu') if device is None else torch.device(device)
    memory_format = memory_format or torch.contiguous_format
    key = (value.shape, value.dtype, value.tobytes(), shape, dtype, device, memory_format)
    hit = _constant_cache.get(key)
    if hit is None:
        hit = torch.as_tensor(value.copy(), dtype=dtype, device=device)
        if shape is not None:
            hit = hit.broadcast_to(shape)
        hit = hit.contiguous(memory_format=memory_format)
        _constant_cache[key] = hit
    if fast is not None and len(_constant_fast) < 65536:
        _constant_fast[fast] = hit
    return hit


nan_to_num = torch.nan_to_num      # misc.py:49-60 (the pre-1.8 emulation is not needed)


class suppress_tracer_warnings(warnings.catch_warnings):
    def __enter__(self):
        super().__enter__()
        warnings.simplefilter('ignore', category=torch.jit.TracerWarning)
        return self


def assert_shape(tensor, ref_shape):
    """None in ref_shape = any size (misc.py:82-97)."""
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}')
    for idx, (size, ref) in enumerate(zip(tensor.shape, ref_shape)):
        if ref is not None and int(size) != int(ref):
            raise AssertionError(f'Wrong size for dimension {idx}: got {size}, expected {ref}')


profiling = False       # set True to emit a record_function scope per profiled op (visible to rocprofv3 via roctx)


def profiled_function(fn):
    """record_function scope named after fn (misc.py:104-110).  The scope costs ~15 us of host time per call
    (6 k calls per training step), so it is only entered while ``misc.profiling`` is on."""
    def wrapped(*args, **kwargs):
        if not profiling:
            return fn(*args, **kwargs)
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    wrapped.__name__ = fn.__name__
    wrapped.__doc__ = fn.__doc__
    return wrapped


_pinned_ring = []
_pinned_next = 0


def upload_small(array, device):
    """float32 numpy array (<= 64 values) -> device tensor WITHOUT a blocking copy: the values are staged in a ring
    of pinned host buffers and copied with non_blocking=True on the current stream.  A pageable host->device copy
    (torch.tensor / as_tensor with device=...) makes the host wait for everything queued on that stream before it;
    inside a training step that drains the launch pipeline every time."""
    global _pinned_next
    device = torch.device(device)
    a = np.ascontiguousarray(array, dtype=np.float32)
    if device.type != 'cuda':
        return torch.from_numpy(a.copy()).to(device)
    assert a.size <= 64
    if not _pinned_ring:
        for _ in range(256):
            _pinned_ring.append([torch.empty([64], dtype=torch.float32).pin_memory(), None])
    slot = _pinned_ring[_pinned_next % len(_pinned_ring)]
    _pinned_next += 1
    if slot[1] is not None:
        slot[1].synchronize()           # 256 uploads ago: long done
    buf = slot[0][:a.size]
    buf.copy_(torch.from_numpy(a.reshape(-1)))
    out = buf.to(device, non_blocking=True).reshape(a.shape)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    slot[1] = ev
    return out


class InfiniteSampler(torch.utils.data.Sampler):
    """Endless index stream; rank r takes every num_replicas-th draw; optional local
    shuffling inside a sliding window (misc.py:117-148)."""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        order = np.arange(len(self.dataset))
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        idx = 0
        while True:
            i = idx % order.size
            if idx % self.num_replicas == self.rank:
                yield order[i]
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            idx += 1


def params_and_buffers(module):
    return list(module.parameters()) + list(module.buffers())


def named_params_and_buffers(module):
    return list(module.named_parameters()) + list(module.named_buffers())


@torch.no_grad()
def copy_params_and_buffers(src_module, dst_module, require_all=False):
    src = dict(named_params_and_buffers(src_module))
    for name, tensor in named_params_and_buffers(dst_module):
        assert (name in src) or (not require_all), name
        if name in src:
            tensor.copy_(src[name].detach()).requires_grad_(tensor.requires_grad)


ddp_sync_listener = None        # set by phase_engine.GradSync while a phase's bucketed gradient exchange is armed


@contextlib.contextmanager
def ddp_sync(module, sync):
    """Gradient synchronisation switch (misc.py:178-185): the forward pass run under ``ddp_sync(module, True)`` is the
    one whose backward pass exchanges `module`'s gradients.  Understands torch DDP wrappers (``no_sync``) and this repo's
    flat-buffer exchange (``phase_engine.GradSync``, told through ``ddp_sync_listener`` which parameters become final)."""
    assert isinstance(module, torch.nn.Module)
    if sync and ddp_sync_listener is not None:
        ddp_sync_listener(module)
    if sync or not hasattr(module, 'no_sync'):
        yield
    else:
        with module.no_sync():
            yield


def check_ddp_consistency(module, ignore_regex=None):
    """Every parameter/buffer equal to rank 0's (misc.py:191-200)."""
    for name, tensor in named_params_and_buffers(module):
        fullname = type(module).__name__ + '.' + name
        if ignore_regex is not None and re.fullmatch(ignore_regex, fullname):
            continue
        tensor = tensor.detach()
        other = tensor.clone()
        torch.distributed.broadcast(tensor=other, src=0)
        assert (nan_to_num(tensor) == nan_to_num(other)).all(), fullname
