"""Phase machinery shared by ``training_loop`` and ``training_loop_aio`` (and bench.py).

What the reference does inline in its loops (training/training_loop.py:206-232, 288-326;
custom/training_loop_aio.py:431-509, 614-668) is collected here so that both loops and the
benchmark execute the SAME code:

  * ``build_phases``    lazy-regularisation split: lr and betas rescaled by i/(i+1), one optimizer
                        shared by the main and reg phase of a module (:211-222)
  * ``Phase.begin/finish`` zero_grad(set_to_none) + requires_grad toggles, then gradient
                        post-processing and the optimizer step
  * ``GradSync``        replaces torch DDP.  One flat fp32 buffer per phase, ONE all-reduce(SUM)/N
                        over RCCL after the last accumulation round (the ``sync`` round upstream,
                        :302), then ``nan_to_num(nan=0, posinf=1e5, neginf=-1e5)`` on the flat buffer
                        -- one kernel instead of one per parameter (:310-312) -- and scatter back.
                        This is the intended behaviour of the aio loop, whose own DDP wrap is
                        discarded upstream.
  * ``update_ema``      multi-tensor lerp + buffer copy (:319-326)

xGMI note: a ring all-reduce is per-link bound (~153 GB/s/link); the flat buffers here are 69-550 MB,
large enough for RCCL's bandwidth-optimal algorithms, and one call per phase keeps latency terms off
the step path.
"""
import torch

from .. import dnnlib
from ..torch_utils import misc


def _modules_of(module):
    return list(module) if isinstance(module, (list, tuple)) else [module]


def _params_of(module):
    seen, out = set(), []
    for m in _modules_of(module):
        if m is None:
            continue
        for p in m.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


class GradSync:
    """Gradient post-processing for one optimizer step: cross-rank mean + NaN/Inf scrub."""

    def __init__(self, world_size=1, process_group=None):
        self.world_size, self.process_group = world_size, process_group
        self._agreed = set()      # gradient signatures all ranks were seen to share

    @torch.no_grad()
    def __call__(self, params):
        grads = [p.grad for p in params if p.grad is not None]
        if self.world_size > 1:
            # Every rank runs the same phase code, so the SAME parameters carry gradients everywhere (a phase that
            # computes nothing -- global_Greg upstream -- leaves them all None and Adam skips them, as on one GPU).
            # One tiny all-reduce checks that before the big one; a mismatch would otherwise hang or mis-add.  Which
            # parameters get a gradient is a property of the phase's code path, not of the data, so each distinct
            # signature is verified once (the comparison reads the result back, i.e. it drains the launch queue:
            # doing it every phase would serialise host and device 36 times per step).
            key = (len(params), len(grads), sum(g.numel() for g in grads))
            agree = key in self._agreed
            if not agree:
                sig = torch.tensor(key[1:], dtype=torch.int64, device=self._device(params))
                lo, hi = sig.clone(), sig.clone()
                torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN, group=self.process_group)
                torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX, group=self.process_group)
                agree = torch.equal(lo, hi)
                if agree:
                    self._agreed.add(key)
            if not agree:
                # ranks disagree: fall back to the union, missing gradients count as zeros
                grads = []
                for p in params:
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                    grads.append(p.grad)
        if not grads:
            return 0
        flat = torch._utils._flatten_dense_tensors(grads)
        if self.world_size > 1:
            torch.distributed.all_reduce(flat, group=self.process_group)
            flat.div_(self.world_size)
        torch.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)
        torch._foreach_copy_(grads, list(torch._utils._unflatten_dense_tensors(flat, grads)))
        return flat.numel()

    @staticmethod
    def _device(params):
        for p in params:
            return p.device
        return torch.device('cpu')


class Phase(dnnlib.EasyDict):
    """name, module (or list of modules), opt, interval, + begin()/finish()."""

    def params(self):
        if '_params' not in self:
            self['_params'] = _params_of(self.module)
        return self['_params']

    def begin(self):
        self.opt.zero_grad(set_to_none=True)
        for m in _modules_of(self.module):
            if m is not None:
                m.requires_grad_(True)

    def finish(self, grad_sync):
        for m in _modules_of(self.module):
            if m is not None:
                m.requires_grad_(False)
        with torch.autograd.profiler.record_function(self.name + '_opt'):
            grad_sync(self.params())
            self.opt.step()


def make_optimizer(params, opt_kwargs, device):
    """construct_class_by_name on the caller's kwargs; on the GPU torch.optim.Adam is asked for its
    fused (single multi-tensor kernel) implementation unless the caller decided otherwise."""
    kw = dict(opt_kwargs)
    if kw.get('class_name') == 'torch.optim.Adam' and torch.device(device).type == 'cuda' and 'fused' not in kw and 'foreach' not in kw:
        kw['fused'] = True
    return dnnlib.util.construct_class_by_name(params, **kw)


def build_phases(specs, device):
    """specs: iterable of (name_fmt, module_or_list, params, opt_kwargs, reg_interval, main_interval) where
    name_fmt has one '{}' for the phase kind ('G{}' -> 'Gmain', 'local_D{}_hair' -> 'local_Dreg_hair').
    Returns the Phase list in execution order: 'both', or 'main' followed by 'reg', per spec."""
    phases = []
    for name, module, params, opt_kwargs, reg_interval, main_interval in specs:
        params = list(params)
        if reg_interval is None:
            opt = make_optimizer(params, opt_kwargs, device)
            phases.append(Phase(name=name.format('both'), module=module, opt=opt, interval=main_interval))
        else:
            mb_ratio = reg_interval / (reg_interval + 1)
            kw = dnnlib.EasyDict(opt_kwargs)
            kw.lr = kw.lr * mb_ratio
            kw.betas = [beta ** mb_ratio for beta in kw.betas]
            opt = make_optimizer(params, kw, device)
            phases.append(Phase(name=name.format('main'), module=module, opt=opt, interval=main_interval))
            phases.append(Phase(name=name.format('reg'), module=module, opt=opt, interval=reg_interval))
    return phases


@torch.no_grad()
def update_ema(module_ema, module, ema_beta):
    """p_ema <- lerp(p, p_ema, beta); buffers copied (training_loop.py:319-326), multi-tensor."""
    for m_ema, m in zip(_modules_of(module_ema), _modules_of(module)):
        if m is None:
            continue
        p_ema, p = list(m_ema.parameters()), list(m.parameters())
        if p:
            torch._foreach_copy_(p_ema, torch._foreach_lerp(p, p_ema, ema_beta))
        b_ema, b = list(m_ema.buffers()), list(m.buffers())
        if b:
            torch._foreach_copy_(b_ema, b)


def ema_beta_for(batch_size, cur_nimg, ema_kimg, ema_rampup):
    ema_nimg = ema_kimg * 1000
    if ema_rampup is not None:
        ema_nimg = min(ema_nimg, cur_nimg * ema_rampup)
    return 0.5 ** (batch_size / max(ema_nimg, 1e-8))


def broadcast_module_state(modules, src=0):
    """What the DDP constructor does at start-up: every rank takes rank 0's parameters and buffers."""
    for m in _modules_of(modules):
        if m is None:
            continue
        for t in misc.params_and_buffers(m):
            torch.distributed.broadcast(t.detach(), src=src)
