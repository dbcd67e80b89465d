"""The STN's sampling step for pure translations, without a materialised grid.

Boundary added by this build.  Upstream (fukuwarai/networks.py:235-258):

    theta = convert_translate_to_2x3(translation)            # python loops, B*L tiny tensors
    grid  = F.affine_grid(theta.view(-1,2,3), x2.size(), align_corners=False)
    x2    = F.grid_sample(x2 + 1, grid, align_corners=False) - 1        # STNv2c workaround

``translate_sample(x[M,C,H,W], t[M,2], shift)`` computes exactly that for
theta = [[1,0,t_x],[0,1,t_y]] with gradients w.r.t. both x and t (csrc/translate.hip:
gather-only input gradient, block-reduced translation gradient).  ``impl='ref'`` / CPU
tensors run the aten ops above.
"""
import numpy as np
import torch

from .. import native
from .. import misc


def translation_to_theta(t):
    """[..., 2] -> [..., 2, 3] = [[1,0,tx],[0,1,ty]]; what image_utils.convert_translate_to_2x3
    (:316-335) builds with a double python loop."""
    eye = misc.constant(np.eye(2), dtype=t.dtype, device=t.device).expand(*t.shape[:-1], 2, 2)      # cached: no host->device copy per call
    return torch.cat([eye, t.unsqueeze(-1)], dim=-1)


@misc.profiled_function
def _translate_sample_ref(x, t, shift=0.0):
    theta = translation_to_theta(t.to(x.dtype))
    grid = torch.nn.functional.affine_grid(theta, list(x.shape), align_corners=False)
    xs = x + shift if shift != 0 else x
    y = torch.nn.functional.grid_sample(xs, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    return y - shift if shift != 0 else y


class _TranslateSampleHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, shift):
        m, c, h, w = x.shape
        x = x.contiguous()
        t32 = t.detach().to(torch.float32).contiguous()
        y = torch.empty_like(x)
        with native.guard(x.device):
            native.check(native.lib().mg_translate_sample_fwd(native.ptr(x), native.ptr(t32), native.ptr(y),
                                                              native.dtype_code(x.dtype), m, c, h, w, float(shift),
                                                              native.stream_ptr(x.device)))
        ctx.save_for_backward(x, t32)
        ctx.shift = float(shift)
        ctx.t_dtype = t.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, t32 = ctx.saved_tensors
        m, c, h, w = x.shape
        dy = dy.contiguous()
        dx = dt = None
        L, code, s = native.lib(), native.dtype_code(x.dtype), native.stream_ptr(x.device)
        with native.guard(x.device):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                native.check(L.mg_translate_sample_bwd_input(native.ptr(dy), native.ptr(t32), native.ptr(dx), code,
                                                             m, c, h, w, s))
            if ctx.needs_input_grad[1]:
                dt = torch.zeros([m, 2], dtype=torch.float32, device=x.device)
                native.check(L.mg_translate_sample_bwd_shift(native.ptr(x), native.ptr(dy), native.ptr(t32),
                                                             native.ptr(dt), code, m, c, h, w, ctx.shift, s))
                dt = dt.to(ctx.t_dtype)
        return dx, dt, None


def translate_sample(x, t, shift=0.0, impl='cuda'):
    """x [M,C,H,W]; t [M,2] translations (dx, dy) in normalised [-1,1] units (positive = content
    moves towards the top-left, as in torch's affine_grid); shift: STNv2c adds 1 before sampling and
    subtracts it afterwards so that out-of-frame pixels read -1 instead of 0.  First-order gradients."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(t, torch.Tensor) and t.shape == (x.shape[0], 2)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda':
        assert x.dtype in (torch.float32, torch.float16, torch.bfloat16)
        return _TranslateSampleHip.apply(x, t, shift)
    return _translate_sample_ref(x, t, shift)
