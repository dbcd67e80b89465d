"""Minibatch standard deviation (reference training/networks.py:667-691) as one native launch each way.

``minibatch_std(x, group_size, num_channels)`` returns ``cat([x, y], dim=1)`` with y the per-group standard deviation averaged over
the channels of each of the ``num_channels`` blocks and the pixels -- the whole ``MinibatchStdLayer.forward``.  On a dense float32 device
tensor (the discriminator epilogue always runs in float32) that is ``mg_mbstd_fwd`` / ``mg_mbstd_bwd`` (csrc/mbstd.hip) instead of ten
aten launches forward and about fifteen backward on a 1 MB tensor.  Under ``create_graph=True`` (the R1 penalty differentiates the
discriminator twice) the gradient is taken through the aten composition, which autograd can differentiate again; CPU tensors, other
dtypes and group sizes without a compiled instance take the composition throughout.
"""
import torch

from .. import native

_GROUPS = (1, 2, 3, 4, 8, 16)


def _ref(x, g, f):
    n, c, h, w = x.shape
    y = x.reshape(g, -1, f, c // f, h, w)                   # [G, n/G, F, c, H, W]
    y = (y - y.mean(dim=0)).square().mean(dim=0)           # variance over the group
    y = (y + 1e-8).sqrt().mean(dim=[2, 3, 4])              # [n/G, F]
    y = y.reshape(-1, f, 1, 1).repeat(g, 1, h, w)
    return torch.cat([x, y], dim=1)


class _MinibatchStd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, f):
        n, c, h, w = x.shape
        out = torch.empty([n, c + f, h, w], dtype=x.dtype, device=x.device)
        with native.guard(x.device):
            native.check(native.lib().mg_mbstd_fwd(native.ptr(x), native.ptr(out), n, c, h * w, g, f, native.stream_ptr(x.device)))
        ctx.save_for_backward(x)
        ctx.gf = (g, f)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, = ctx.saved_tensors
        g, f = ctx.gf
        if not ctx.needs_input_grad[0]:
            return None, None, None
        if torch.is_grad_enabled():
            # create_graph=True: the gradient as a differentiable function of (x, dout), from the aten composition
            # (the native kernel below returns a constant as far as autograd is concerned)
            with torch.enable_grad():
                dx, = torch.autograd.grad([_ref(x, g, f)], [x], [dout], create_graph=True)
            return dx, None, None
        n, c, h, w = x.shape
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        with native.guard(x.device):
            native.check(native.lib().mg_mbstd_bwd(native.ptr(dout), native.ptr(x), native.ptr(dx), n, c, h * w, g, f, native.stream_ptr(x.device)))
        return dx, None, None


def minibatch_std(x, group_size, num_channels=1):
    n, c, h, w = x.shape
    g = min(int(group_size), n) if group_size is not None else n
    f = int(num_channels)
    if (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and g in _GROUPS and n % g == 0 and c % f == 0 and x.numel() > 0):
        return _MinibatchStd.apply(x, g, f)
    return _ref(x, g, f)
