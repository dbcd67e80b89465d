"""Per-sample channel mixing of an image batch: ``y[n] = A[n] @ x[n] + t[n]`` over <= 4 channels.

The colour transforms of the ADA pipe end in ``C[:, :3, :3] @ images + C[:, :3, 3:]`` (reference
training/augment.py:403-406): a batched GEMM with M = 3 that the BLAS library runs at a few percent of the memory
rate, plus a slice, a broadcast add and a concatenation for RGBA.  ``channel_mix`` is one native pass; linear in x,
so every order of gradient is the same kernel with A transposed and no offset.  A / t that require grad, CPU
tensors and float64 take the reference expression.
"""
import torch

from .. import native


def _ref(x, A, t):
    n, c = x.shape[:2]
    y = A.to(x.dtype) @ x.reshape(n, c, -1)
    if t is not None:
        y = y + t.to(x.dtype).reshape(n, c, 1)
    return y.reshape(x.shape)


class _Mix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, A, t):
        n, c = x.shape[:2]
        x = x.contiguous()
        y = torch.empty_like(x)
        if x.numel():
            with native.guard(x.device):
                native.check(native.lib().mg_channel_mix(native.ptr(x), native.ptr(A), native.ptr(t), native.ptr(y),
                                                         native.dtype_code(x.dtype), n, c, x.numel() // (n * c), native.stream_ptr(x.device)))
        ctx.save_for_backward(A)
        return y

    @staticmethod
    def backward(ctx, dy):
        A, = ctx.saved_tensors
        dx = _Mix.apply(dy, A.transpose(1, 2).contiguous(), None) if ctx.needs_input_grad[0] else None
        return dx, None, None


def channel_mix(x, A, t=None, impl='cuda'):
    """x [N,C,...]; A [N,C,C]; t [N,C] or None."""
    assert isinstance(x, torch.Tensor) and x.ndim >= 3 and A.shape == (x.shape[0], x.shape[1], x.shape[1])
    if (impl == 'cuda' and x.device.type == 'cuda' and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and x.shape[1] <= 4
            and not A.requires_grad and (t is None or not t.requires_grad)):
        return _Mix.apply(x, A.detach().to(torch.float32).contiguous(), t.detach().to(torch.float32).contiguous() if t is not None else None)
    return _ref(x, A, t)
