"""a * b + c with broadcasting and exact un-broadcast gradients
(reference montage_gan/torch_utils/ops/fma.py:16-61).  The networks use it for
``x * dcoefs[N,C,1,1] + noise[1,1,H,W]``; on the GPU the fused epilogue
(``modconv_epilogue``) removes this extra HBM round trip altogether."""
import torch


def fma(a, b, c):
    return _FusedMultiplyAdd.apply(a, b, c)


def _unbroadcast(x, shape):
    """Sum x over the axes along which a tensor of `shape` was broadcast to x.shape."""
    lead = x.ndim - len(shape)
    assert lead >= 0
    axes = [i for i in range(x.ndim) if x.shape[i] > 1 and (i < lead or shape[i - lead] == 1)]
    if axes:
        x = x.sum(dim=axes, keepdim=True)
    if lead:
        x = x.reshape(-1, *x.shape[lead + 1:])
    assert x.shape == shape
    return x


class _FusedMultiplyAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        need = ctx.needs_input_grad
        da = _unbroadcast(dout * b, a.shape) if need[0] else None
        db = _unbroadcast(dout * a, b.shape) if need[1] else None
        dc = _unbroadcast(dout, ctx.c_shape) if need[2] else None
        return da, db, dc
