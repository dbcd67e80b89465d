"""Batched preparation of layer weights.

Every layer of the reference networks derives the tensor it convolves / multiplies with from its parameter on each
call: ``weight * weight_gain`` (equalised learning rate, ref training/networks.py:122,161), ``.to(x.dtype)`` for the
reduced-precision blocks (:83,179), ``bias * bias_gain`` (:126).  On a GPU each of those is a 3-4 us kernel in
the forward pass and another in the backward pass -- about 11 000 launches per training step of the 9-GAN loop,
13 % of the kernel time and a fifth of the host's dispatch work.

``scope(entries)`` computes all of them for one network in a handful of multi-tensor launches
(``torch._foreach_mul`` with a scalar list, ``torch._foreach_copy_`` across dtypes) behind ONE autograd node whose
backward is again multi-tensor; layers pick their tensor up with ``get(param, gain, dtype)``.  The values are bit
identical to the per-layer expressions (a float32 multiply, then the cast); a combination that was not prepared --
CPU tensors, a layer called outside its network, an unusual dtype -- is simply computed in place, as before.
"""
import os

import torch

_stack = []
enabled = os.environ.get('MG_WEIGHT_PREP', '1') != '0'          # module switch (MG_WEIGHT_PREP=0 turns it off; tests compare both ways)


class _ScaleCast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gains, dtypes, *params):
        ctx.gains, ctx.dtypes = gains, dtypes
        ctx.set_materialize_grads(False)        # an output nobody differentiates must leave its parameter's .grad at None
        out = list(torch._foreach_mul(list(params), list(gains)))
        low = [i for i, d in enumerate(dtypes) if d != out[i].dtype]
        if low:
            dst = [torch.empty_like(out[i], dtype=dtypes[i]) for i in low]        # empty_like keeps channels_last strides
            torch._foreach_copy_(dst, [out[i] for i in low])
            for i, d in zip(low, dst):
                out[i] = d
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        gains = ctx.gains
        res = [None] * len(grads)
        idx = [i for i, g in enumerate(grads) if g is not None and ctx.needs_input_grad[2 + i]]
        if torch.is_grad_enabled():
            # create_graph=True (path-length / R1 regularisation): stay differentiable, per tensor
            for i in idx:
                res[i] = grads[i].to(torch.float32) * gains[i]
        elif idx:
            src = []
            low = [i for i in idx if grads[i].dtype != torch.float32]
            if low:
                buf = [torch.empty_like(grads[i], dtype=torch.float32) for i in low]
                torch._foreach_copy_(buf, [grads[i] for i in low])
                wide = dict(zip(low, buf))
            else:
                wide = {}
            src = [wide.get(i, grads[i]) for i in idx]
            for i, r in zip(idx, torch._foreach_mul(src, [gains[i] for i in idx])):
                res[i] = r
        return (None, None, *res)


class Plan:
    """The static part of a scope: which parameters, gains and dtypes one network prepares (built once per network)."""

    def __init__(self, entries):
        seen, self.params, gains, dtypes = set(), [], [], []
        for p, g, d in entries:
            if p is None or p.dtype != torch.float32:
                continue
            k = (id(p), float(g), d)
            if k not in seen:           # one entry per distinct (param, gain, dtype)
                seen.add(k)
                self.params.append(p); gains.append(float(g)); dtypes.append(d)
        self.gains, self.dtypes = tuple(gains), tuple(dtypes)


class scope:
    """``with scope(plan_or_entries):`` -- prepare, serve ``get`` inside, forget on exit."""

    def __init__(self, plan):
        self.plan = plan if isinstance(plan, Plan) else Plan(plan)

    def __enter__(self):
        plan, table = self.plan, {}
        if enabled and plan.params and plan.params[0].device.type == 'cuda':
            outs = _ScaleCast.apply(plan.gains, plan.dtypes, *plan.params)
            # keyed by the identity of the CURRENT parameter objects (a deep-copied network carries a copied plan)
            table = {(id(p), g, d): o for p, g, d, o in zip(plan.params, plan.gains, plan.dtypes, outs)}
        _stack.append(table)
        return self

    def __exit__(self, *exc):
        _stack.pop()
        return False


def get(param, gain, dtype):
    """``(param * gain).to(dtype)`` -- from the innermost scope that prepared it, else computed here."""
    if _stack:
        key = (id(param), float(gain), dtype)
        for table in reversed(_stack):
            hit = table.get(key)
            if hit is not None:
                return hit
    w = param * gain if gain != 1 else param
    return w.to(dtype)
