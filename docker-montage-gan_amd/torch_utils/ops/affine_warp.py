"""Affine resampling of an image batch: ``grid_sample(x, affine_grid(theta, size))`` in one native pass.

The ADA pipe resamples every augmented batch through
``torch.nn.functional.affine_grid`` + ``grid_sample_gradfix.grid_sample`` (reference training/augment.py:376-377,
bilinear / zeros / align_corners=False).  ``affine_warp`` is the same function of (x, theta) without the grid
tensor in between, and its gradient with respect to x is a gather (csrc/augment.hip) instead of aten's
scatter of float atomics.  The op is linear in x, so its backward is again differentiable (R1 regularisation
differentiates the discriminator input twice through the pipe): backward-of-backward is the forward op.

theta is treated as a constant (ADA's matrices come from random draws); a theta that requires grad, CPU tensors,
float64 and more than 4 channels take the reference composition.
"""
import torch

from .. import native
from . import grid_sample_gradfix


def _ref(x, theta, out_h, out_w):
    grid = torch.nn.functional.affine_grid(theta=theta, size=[x.shape[0], x.shape[1], out_h, out_w], align_corners=False)
    return grid_sample_gradfix.grid_sample(x, grid)


def _launch(fn, src, theta, dst, n, c, ih, iw, oh, ow):
    if src.numel() and dst.numel():
        with native.guard(src.device):
            native.check(fn(native.ptr(src), native.ptr(theta), native.ptr(dst), native.dtype_code(src.dtype), n, c, ih, iw, oh, ow,
                            native.stream_ptr(src.device)))
    elif dst.numel():
        dst.zero_()


class _Warp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, theta, out_h, out_w):
        n, c, ih, iw = x.shape
        x = x.contiguous()
        y = torch.empty([n, c, out_h, out_w], dtype=x.dtype, device=x.device)
        _launch(native.lib().mg_affine_warp_fwd, x, theta, y, n, c, ih, iw, out_h, out_w)
        ctx.save_for_backward(theta)
        ctx.in_size = (ih, iw)
        return y

    @staticmethod
    def backward(ctx, dy):
        theta, = ctx.saved_tensors
        dx = _WarpAdjoint.apply(dy, theta, *ctx.in_size) if ctx.needs_input_grad[0] else None
        return dx, None, None, None


class _WarpAdjoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, theta, in_h, in_w):
        n, c, oh, ow = dy.shape
        dy = dy.contiguous()
        dx = torch.empty([n, c, in_h, in_w], dtype=dy.dtype, device=dy.device)
        _launch(native.lib().mg_affine_warp_bwd, dy, theta, dx, n, c, in_h, in_w, oh, ow)
        ctx.save_for_backward(theta)
        ctx.out_size = (oh, ow)
        return dx

    @staticmethod
    def backward(ctx, ddx):
        theta, = ctx.saved_tensors
        ddy = _Warp.apply(ddx, theta, *ctx.out_size) if ctx.needs_input_grad[0] else None
        return ddy, None, None, None


def affine_warp(x, theta, out_h, out_w, impl='cuda'):
    """x [N,C,H,W]; theta [N,2,3] (normalised coordinates, as for affine_grid); returns [N,C,out_h,out_w]."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4 and theta.shape == (x.shape[0], 2, 3)
    if (impl == 'cuda' and x.device.type == 'cuda' and x.dtype in (torch.float32, torch.float16, torch.bfloat16)
            and x.shape[1] <= 4 and not theta.requires_grad):
        return _Warp.apply(x, theta.detach().to(torch.float32).contiguous(), int(out_h), int(out_w))
    return _ref(x, theta, int(out_h), int(out_w))
