"""grid_sample (bilinear, zeros padding, align_corners=False) with gradients of any
order between input and output.

Same interface as the reference (montage_gan/torch_utils/ops/grid_sample_gradfix.py:23,28-32):
module switch ``enabled`` and ``grid_sample(input, grid)``.  The reference is pinned to
torch 1.7-1.9 (:37-44) and otherwise falls back to ``F.grid_sample``; this one is written
against today's ``aten::grid_sampler_2d_backward`` (which takes an output mask) and is the
op AugmentPipe calls.  As upstream, there is no gradient of the gradient w.r.t. ``grid``.
"""
import torch

enabled = False  # Enable the custom op by setting this to true.


def grid_sample(input, grid):
    if _should_use_custom_op():
        return _GridSample2dForward.apply(input, grid)
    return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros',
                                           align_corners=False)


def _should_use_custom_op():
    return bool(enabled)


class _GridSample2dForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid):
        assert input.ndim == 4 and grid.ndim == 4
        ctx.save_for_backward(input, grid)
        return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros',
                                               align_corners=False)

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        return _GridSample2dBackward.apply(grad_output, input, grid)


class _GridSample2dBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_output, input, grid):
        ctx.save_for_backward(grid)
        mask = [bool(ctx.needs_input_grad[1]), bool(ctx.needs_input_grad[2])]
        grad_input, grad_grid = torch.ops.aten.grid_sampler_2d_backward(grad_output, input, grid, 0, 0, False, mask)
        return grad_input, grad_grid

    @staticmethod
    def backward(ctx, grad2_grad_input, grad2_grad_grid):
        grid, = ctx.saved_tensors
        grad2_grad_output = None
        if ctx.needs_input_grad[0]:
            # the input gradient is linear in grad_output: its adjoint is the forward sampler
            grad2_grad_output = _GridSample2dForward.apply(grad2_grad_input, grid)
        assert not ctx.needs_input_grad[2], 'no second-order gradient w.r.t. the sampling grid'
        return grad2_grad_output, None, None
