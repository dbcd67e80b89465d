"""Bilinear ``grid_sample`` (zero padding, align_corners=False) whose input gradient can be differentiated again.

Interface of montage_gan/torch_utils/ops/grid_sample_gradfix.py (:23, :28-32): the module switch ``enabled`` and
``grid_sample(input, grid)``.  Upstream's version only engages on torch 1.7-1.9 (:37-44) and silently becomes
``F.grid_sample`` elsewhere; this one targets the present ``aten::grid_sampler_2d_backward`` (output-mask signature).
Sampling is linear in the image, so the adjoint of the image gradient is the sampler itself -- that is all the
second-order support R1 regularisation through AugmentPipe needs; like upstream there is no second-order term for the
grid.  (The hot ADA call site no longer comes through here on the GPU: see ops/affine_warp.py.)
"""
import torch

enabled = False     # module switch, as upstream: False -> plain torch.nn.functional.grid_sample

_MODE = dict(mode='bilinear', padding_mode='zeros', align_corners=False)
_BILINEAR, _ZEROS = 0, 0            # aten enum values of the two settings above


def _sample(image, grid):
    return torch.nn.functional.grid_sample(input=image, grid=grid, **_MODE)


def grid_sample(input, grid):
    return _Sampler.apply(input, grid) if enabled else _sample(input, grid)


class _Sampler(torch.autograd.Function):
    """y = sample(image, grid)"""

    @staticmethod
    def forward(ctx, image, grid):
        if image.ndim != 4 or grid.ndim != 4:
            raise ValueError('grid_sample expects [N,C,H,W] images and [N,Ho,Wo,2] grids')
        ctx.save_for_backward(image, grid)
        return _sample(image, grid)

    @staticmethod
    def backward(ctx, dy):
        image, grid = ctx.saved_tensors
        return _SamplerAdjoint.apply(dy, image, grid)


class _SamplerAdjoint(torch.autograd.Function):
    """(d_image, d_grid) for an upstream gradient dy; differentiable once more w.r.t. dy."""

    @staticmethod
    def forward(ctx, dy, image, grid):
        want = [bool(ctx.needs_input_grad[1]), bool(ctx.needs_input_grad[2])]
        d_image, d_grid = torch.ops.aten.grid_sampler_2d_backward(dy, image, grid, _BILINEAR, _ZEROS, False, want)
        ctx.save_for_backward(grid)
        return d_image, d_grid

    @staticmethod
    def backward(ctx, dd_image, dd_grid):
        if ctx.needs_input_grad[2]:
            raise RuntimeError('no second-order gradient w.r.t. the sampling grid')
        grid, = ctx.saved_tensors
        # d_image = A^T dy with A = the (linear) sampling operator  =>  its adjoint applied to dd_image is A dd_image
        d_dy = _Sampler.apply(dd_image, grid) if ctx.needs_input_grad[0] else None
        return d_dy, None, None
