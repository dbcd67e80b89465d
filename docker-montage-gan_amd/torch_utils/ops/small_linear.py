"""Fully connected layer on a small batch as one native launch: ``y = x @ w.t() (+ b)``, float32, at most 64 rows.

``FullyConnectedLayer`` (reference training/networks.py:104-133) calls ``torch.addmm`` / ``matmul`` on
[32 x 512] x [512 x 512]-sized operands about three thousand times per training step (mapping network, per-layer style
affines, discriminator heads); the BLAS library's general tiles need 9-10 us for each of those 1-us problems.
csrc/linear.hip runs them with one wave per output feature.  The op is bilinear, so its backward under
``create_graph=True`` (path-length regularisation reaches the style affines twice) is written with plain matmuls,
which autograd differentiates again; the plain backward is two native launches (input gradient; weight + bias gradient).
Anything else -- more rows, other dtypes, CPU tensors -- takes the reference expression.
"""
import torch

from .. import native

# OFF by default.  Measured on MI355X in the training step (profiles/ notes in DESIGN.md): 1770 launches at 138 us on average
# against ~10 us for the BLAS library -- the input-gradient kernel walks all output features serially in eight waves and is
# latency-bound; it needs the reduction split across many more waves (and an LDS or atomic combine) before this pays.
# The kernels are exact and tested (test_small_linear_native_vs_float64); switch on with small_linear.enabled = True.
enabled = False
MAX_ROWS = 64
MAX_IN = 2048           # wider layers (the discriminator's 8192 -> 512 head) stay on the BLAS library


class _SmallLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x, w = x.contiguous(), w.contiguous()
        m, ic = x.shape
        oc = w.shape[0]
        y = torch.empty([m, oc], dtype=torch.float32, device=x.device)
        bb = b.contiguous() if b is not None else None
        with native.guard(x.device):
            native.check(native.lib().mg_linear_fwd(x.data_ptr(), w.data_ptr(), bb.data_ptr() if bb is not None else None, y.data_ptr(),
                                                    m, ic, oc, native.stream_ptr(x.device)))
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_b and ctx.needs_input_grad[2]
        if torch.is_grad_enabled():
            return (g @ w if need_x else None), (g.t() @ x if need_w else None), (g.sum(0) if need_b else None)
        g = g.contiguous()
        m, ic = x.shape
        oc = w.shape[0]
        gx = torch.empty_like(x) if need_x else None
        gw = torch.empty_like(w) if need_w else None
        gb = torch.empty([oc], dtype=torch.float32, device=x.device) if need_b else None
        if need_x or need_w or need_b:
            with native.guard(x.device):
                native.check(native.lib().mg_linear_bwd(g.data_ptr(), x.data_ptr(), w.data_ptr(), gx.data_ptr() if need_x else None,
                                                        gw.data_ptr() if need_w else None, gb.data_ptr() if need_b else None,
                                                        m, ic, oc, native.stream_ptr(x.device)))
        return gx, gw, gb


def small_linear(x, w, b=None, impl='cuda'):
    """x [M,I], w [O,I], b [O] or None -> [M,O] = x @ w.t() + b."""
    if (enabled and impl == 'cuda' and x.device.type == 'cuda' and x.ndim == 2 and w.ndim == 2 and x.dtype == torch.float32 and w.dtype == torch.float32
            and (b is None or b.dtype == torch.float32) and 0 < x.shape[0] <= MAX_ROWS and x.shape[1] <= MAX_IN and x.shape[1] == w.shape[1]):
        return _SmallLinear.apply(x, w, b)
    if b is not None:
        return torch.addmm(b.unsqueeze(0), x, w.t())
    return x.matmul(w.t())
