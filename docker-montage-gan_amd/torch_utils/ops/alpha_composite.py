"""Differentiable "over" compositing of an RGBA layer stack.

Boundary added by this build: upstream the global GAN calls the python
loops of ``custom_utils.image_utils.alpha_composite_pytorch`` (image_utils.py:112-163,
call site custom/loss_aio.py:251) -- B*(L-1) rounds of ~8 tiny kernels plus cat/stack.
Here it is one kernel forward, one backward (csrc/composite.hip):

    alpha_composite(blchw[B,L,4,H,W] in [0,1]) -> [B,4,H,W]         (also [L,4,H,W] -> [4,H,W])

``impl='ref'`` (or CPU tensors) evaluates the same recurrence with batched aten ops and is
what autograd differentiates on the CPU.  On a GPU tensor the native library is
required (raises otherwise).
"""
import torch

from .. import native
from .. import misc


def _a_over_b(top, canvas):
    # image_utils.py:129-134 on [..., 4, H, W] tensors
    c1, a1 = top[..., :3, :, :], top[..., 3:, :, :]
    c2, a2 = canvas[..., :3, :, :], canvas[..., 3:, :, :]
    rest = a2 * (1 - a1)
    a_out = a1 + rest
    c_out = torch.nan_to_num((c1 * a1 + c2 * rest) / a_out)
    return torch.cat([c_out, a_out], dim=-3)


@misc.profiled_function
def _alpha_composite_ref(x):
    canvas = x[..., 0, :, :, :]
    for li in range(1, x.shape[-4]):
        canvas = _a_over_b(x[..., li, :, :, :], canvas)
    return canvas


class _AlphaCompositeHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, safe_grad):
        b, l, c, h, w = x.shape
        x = x.contiguous()
        y = torch.empty([b, 4, h, w], dtype=x.dtype, device=x.device)
        with native.guard(x.device):
            native.check(native.lib().mg_alpha_composite_fwd(native.ptr(x), native.ptr(y), native.dtype_code(x.dtype),
                                                             b, l, h, w, native.stream_ptr(x.device)))
        ctx.save_for_backward(x)
        ctx.safe_grad = bool(safe_grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        b, l, c, h, w = x.shape
        if torch.is_grad_enabled():
            # create_graph=True (the R1 penalty of the global discriminator differentiates its input gradient again,
            # reference custom/loss_aio.py:327-335): the gradient as a differentiable function of (x, dy), from the aten
            # recurrence.  The native kernel below returns a constant as far as autograd is concerned.
            with torch.enable_grad():
                y = _alpha_composite_ref(x)
                dx, = torch.autograd.grad([y], [x], [dy], create_graph=True)
            return dx, None
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        with native.guard(x.device):
            native.check(native.lib().mg_alpha_composite_bwd(native.ptr(x), native.ptr(dy), native.ptr(dx),
                                                             native.dtype_code(x.dtype), b, l, h, w,
                                                             1 if ctx.safe_grad else 0, native.stream_ptr(x.device)))
        return dx, None


def alpha_composite(blchw_lchw, safe_grad=False, impl='cuda'):
    """Bottom-up composite: layer 0 is the canvas, every later layer goes OVER it.

    safe_grad=False reproduces the reference's autograd bit for bit in where it yields NaN
    (pixels whose accumulated alpha is exactly 0 give 0/0 in the backward of the division);
    safe_grad=True gives those pixels a zero gradient instead (conscious deviation, opt-in).
    Second-order gradients (create_graph=True) go through the aten recurrence."""
    assert isinstance(blchw_lchw, torch.Tensor) and blchw_lchw.ndim in (4, 5)
    assert impl in ['ref', 'cuda']
    x = blchw_lchw
    assert x.shape[-3] == 4, 'RGBA layers expected'
    if impl == 'cuda' and x.device.type == 'cuda':
        assert x.dtype in (torch.float32, torch.float16, torch.bfloat16)
        batched = x if x.ndim == 5 else x.unsqueeze(0)
        y = _AlphaCompositeHip.apply(batched, safe_grad)
        return y if x.ndim == 5 else y.squeeze(0)
    return _alpha_composite_ref(x)
