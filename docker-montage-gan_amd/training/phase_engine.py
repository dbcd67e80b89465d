"""Phase machinery shared by ``training_loop`` and ``training_loop_aio`` (and bench.py).

What the reference does inline in its loops (training/training_loop.py:206-232, 288-326;
custom/training_loop_aio.py:431-509, 614-668) is collected here so that both loops and the
benchmark execute the SAME code:

  * ``build_phases``    lazy-regularisation split: lr and betas rescaled by i/(i+1), one optimizer
                        shared by the main and reg phase of a module (:211-222)
  * ``Phase.begin/finish`` zero_grad(set_to_none) + requires_grad toggles, then gradient
                        post-processing and the optimizer step
  * ``GradSync``        replaces torch DDP.  One persistent flat fp32 buffer per optimizer; all-reduce(SUM)/N
                        over RCCL of the last accumulation round's gradients (the ``sync`` round upstream,
                        :302) -- in buckets started from inside the backward pass (eager phases) or as one
                        call between the two recorded halves of a phase (HIP graphs) --, then ``nan_to_num(nan=0, posinf=1e5, neginf=-1e5)`` on the flat buffer
                        -- one kernel instead of one per parameter (:310-312) -- and scatter back.
                        This is the intended behaviour of the aio loop, whose own DDP wrap is
                        discarded upstream.
  * ``update_ema``      multi-tensor lerp + buffer copy (:319-326)

xGMI note: a ring all-reduce is per-link bound (~153 GB/s/link); the flat buffers here are 69-550 MB,
large enough for RCCL's bandwidth-optimal algorithms, and one call per phase keeps latency terms off
the step path.
"""
import contextlib
import gc
import os

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
    """Gradient post-processing for one optimizer step: cross-rank mean + NaN/Inf scrub.

    Each optimizer owns ONE persistent flat float32 buffer (allocated on first use, shared by its main and reg
    phase).  ``pack`` gathers the phase's gradients into it with a single multi-tensor copy, ``exchange`` is the
    one all-reduce(SUM)/N of the phase (skipped on one GPU), ``unpack`` scrubs the buffer in place and re-points
    every ``p.grad`` at its slice of the buffer -- the optimizer reads the reduced values where they lie, nothing is
    copied back.  ``wire_dtype=torch.bfloat16`` halves the bytes on the xGMI links (sum in bf16, off by default).
    The three stages are separate so that a phase recorded into HIP graphs can keep the collective outside them.
    """

    VERIFY_FIRST = 2        # signature checks per optimizer key; a call COUNT, hence identical on every rank

    def __init__(self, world_size=1, process_group=None, wire_dtype=None, overlap=None, bucket_bytes=32 << 20):
        self.world_size, self.process_group, self.wire_dtype = world_size, process_group, wire_dtype
        # bucketed exchange behind the backward pass (eager phases); MG_GRAD_OVERLAP=0 falls back to one all-reduce after it
        self.overlap = (world_size > 1 and os.environ.get('MG_GRAD_OVERLAP', '1') != '0') if overlap is None else bool(overlap)
        self.bucket_bytes = bucket_bytes
        self._flat = dict()         # key -> persistent flat buffer
        self._layout = dict()       # key -> tuple of (param index, offset, numel) of the gradients packed last
        self._verified = dict()     # key -> number of signature checks done
        self._plans = dict()        # key -> _BucketPlan (overlapped exchange)
        self._active = None         # the plan whose backward is running
        self.buckets_launched_early = 0     # all-reduces started from inside a backward pass (statistics / tests)
        self.stale_phases = 0               # phases whose early buckets had to be exchanged again (a loss outside the contract)

    # -- stages ------------------------------------------------------------------------------------
    @torch.no_grad()
    def pack(self, params, key=None):
        """Gradients of `params` -> the persistent flat buffer of `key`.  Returns the packed view or None."""
        key = id(params) if key is None else key
        present = [i for i, p in enumerate(params) if p.grad is not None]
        if self.world_size > 1:
            present = self._agree(params, present, key)
        if not present:
            self._layout[key] = ()
            return None
        total = sum(params[i].numel() for i in present)
        flat = self._flat.get(key)
        if flat is None or flat.numel() < total:
            cap = sum(p.numel() for p in params)
            flat = self._flat[key] = torch.empty([cap], dtype=torch.float32, device=params[present[0]].device)
        layout, off = [], 0
        for i in present:
            layout.append((i, off, params[i].numel()))
            off += params[i].numel()
        self._layout[key] = tuple(layout)
        torch._foreach_copy_([self._slice(flat, params[i], o) for i, o, n in layout], [params[i].grad for i in present])
        return flat[:total]

    @staticmethod
    def _slice(flat, p, offset):
        """The slice of the flat buffer that stands for p's gradient, with p's own (dense) strides -- channels-last
        weights keep their layout, which the fused optimizer requires of (param, grad) pairs."""
        return torch.as_strided(flat, p.size(), p.stride(), offset)

    @torch.no_grad()
    def exchange(self, flat):
        if flat is None or self.world_size <= 1:
            return
        if self.wire_dtype is not None and self.wire_dtype != flat.dtype:
            wire = flat.to(self.wire_dtype)
            torch.distributed.all_reduce(wire, group=self.process_group)
            flat.copy_(wire)
        else:
            torch.distributed.all_reduce(flat, group=self.process_group)
        flat.div_(self.world_size)

    @torch.no_grad()
    def unpack(self, params, flat, key=None):
        """nan_to_num(nan=0, posinf=1e5, neginf=-1e5) on the whole buffer (one kernel instead of one per parameter,
        reference training_loop.py:310-312), then p.grad = its slice."""
        key = id(params) if key is None else key
        if flat is None:
            return 0
        torch.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)
        full = self._flat[key]
        for i, o, n in self._layout[key]:
            params[i].grad = self._slice(full, params[i], o)
        return flat.numel()

    def __call__(self, params, key=None):
        flat = self.pack(params, key)
        self.exchange(flat)
        return self.unpack(params, flat, key)

    # -- overlapped exchange ---------------------------------------------------------------------------
    # DDP's idea without DDP (the reference's own wrap is discarded): the flat buffer is cut into buckets of
    # ~bucket_bytes in parameter order; a post-accumulate hook per parameter counts a bucket's gradients in, and a bucket
    # whose gradients are all final is copied into its range of the flat buffer (one multi-tensor copy) and all-reduced
    # asynchronously -- RCCL's own stream -- while the backward pass continues towards the input.  Buckets start strictly
    # from the last to the first, the same order on every rank.  ``finish`` starts what is left, waits, and hands the flat
    # buffer to ``unpack`` exactly as ``pack`` + ``exchange`` would have.
    # WHEN a gradient is final follows the reference's own contract (training_loop.py:300-305, loss.py:50-59, misc.py:178-185):
    # a loss calls backward() several times per round (Dmain: generated, then real images; Gboth: main, then path length;
    # Renderer: generated, then real), and it says which backward is the synchronising one by running that module's FORWARD
    # under ``misc.ddp_sync(module, sync=True)`` -- with torch DDP that forward is the one whose backward all-reduces.  Here
    # ``ddp_sync(module, True)`` arms the module's parameters (``sync_forward``); the hook counts a parameter only once it is
    # armed, i.e. in the first backward pass after its module's synchronising forward.  A module that never runs under
    # ``ddp_sync(.., True)`` is exchanged from ``finish`` as a whole.  A loss that breaks the contract (another backward into an
    # armed parameter after its bucket left) is detected by the same hook and the phase falls back to one exchange of the
    # final gradients from ``finish`` -- late, never wrong.
    def arm(self, params, key=None):
        """Call after zero_grad and before the phase's backward passes (eager phases, more than one rank)."""
        key = id(params) if key is None else key
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = _BucketPlan(self, params, key)
        flat = self._flat.get(key)
        if flat is None or flat.numel() < plan.total:
            flat = self._flat[key] = torch.empty([plan.total], dtype=torch.float32, device=params[0].device)
        plan.reset(flat)
        self._active = plan
        misc.ddp_sync_listener = self.sync_forward
        return plan

    def begin_round(self, last):
        """The loops announce every accumulation round; gradients are final only in the last one."""
        if self._active is not None:
            self._active.last_round = bool(last)

    def sync_forward(self, module):
        """``misc.ddp_sync(module, sync=True)`` was entered: the next backward pass into `module`'s parameters is the
        synchronising one of this phase (see the block comment above)."""
        plan = self._active
        if plan is not None and plan.last_round:
            for p in module.parameters():
                i = plan.index.get(id(p))
                if i is not None:
                    plan.armed[i] = True

    @torch.no_grad()
    def finish(self, params, key=None):
        """After the last backward pass: -> the reduced flat buffer (or None when no parameter has a gradient)."""
        key = id(params) if key is None else key
        plan, self._active = self._plans[key], None
        misc.ddp_sync_listener = None
        present = [i for i, p in enumerate(params) if p.grad is not None]
        present = self._agree(params, present, key)
        if not present:
            assert not plan.works, 'a bucket was exchanged although no parameter carries a gradient'
            self._layout[key] = ()
            return None
        plan.launch_rest(params)
        plan.wait()
        if plan.stale:
            # a gradient changed after its bucket had left (contract broken by the loss): exchange the final values.  Which
            # hooks fire is a property of the phase's autograd graph -- the same code on every rank -- so every rank takes
            # this branch together; no agreement round (it would cost a host read-back per phase).
            self.stale_phases += 1
            plan.next_bucket = len(plan.buckets) - 1
            plan.launch_rest(params)
            plan.wait()
        plan.flat.div_(self.world_size)
        self._layout[key] = tuple((i, plan.offsets[i], params[i].numel()) for i in present)
        return plan.flat[:plan.total]

    # -- cross-rank agreement ------------------------------------------------------------------------
    def _agree(self, params, present, key):
        """Every rank runs the same phase code, so the SAME parameters carry gradients everywhere (a phase that
        computes nothing -- global_Greg upstream -- leaves them all None and Adam skips them, as on one GPU).  A
        mismatch would mis-add or hang the big all-reduce, so the first VERIFY_FIRST packs of every key compare a
        signature (count, numel, CRC of the presence bitmap) across ranks with one small MAX all-reduce of
        [sig, -sig].  Whether the check runs depends only on the call count -- the same on every rank -- never on a
        rank-local cache of outcomes.  On disagreement the union is used: missing gradients count as zeros."""
        done = self._verified.get(key, 0)
        if done >= self.VERIFY_FIRST or (params and params[0].is_cuda and torch.cuda.is_current_stream_capturing()):
            return present          # (a recording happens at the same call count on every rank, so this is rank-consistent too)
        self._verified[key] = done + 1
        import zlib
        bitmap = bytearray(len(params))
        for i in present:
            bitmap[i] = 1
        sig = [len(present), sum(params[i].numel() for i in present), zlib.crc32(bytes(bitmap))]
        dev = params[0].device if params else torch.device('cpu')
        t = torch.tensor(sig + [-v for v in sig], dtype=torch.int64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.process_group)
        t = t.tolist()
        if all(t[j] == -t[j + 3] for j in range(3)):
            return present
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        return list(range(len(params)))


class _BucketPlan:
    """Fixed offsets of every parameter of one optimizer in its flat buffer, the bucket boundaries and the per-phase progress."""

    def __init__(self, sync, params, key):
        self.sync, self.params, self.key = sync, params, key
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += p.numel()
        self.total = off
        self.buckets, members, lo = [], [], 0           # (lo, hi, member indices), in parameter order
        for i, p in enumerate(params):
            members.append(i)
            hi = self.offsets[i] + p.numel()
            if (hi - lo) * 4 >= sync.bucket_bytes or i == len(params) - 1:
                self.buckets.append((lo, hi, tuple(members)))
                members, lo = [], hi
        self.bucket_of = [b for b, (_, _, m) in enumerate(self.buckets) for _ in m]
        self.index = {id(p): i for i, p in enumerate(params)}
        self.flat, self.works, self.last_round, self.stale = None, [], False, False
        self.counted, self.armed, self.pending, self.next_bucket = [], [], [], -1
        for i, p in enumerate(params):
            p.register_post_accumulate_grad_hook(lambda p, i=i: self._on_grad(i))

    def reset(self, flat):
        self.flat, self.works, self.last_round, self.stale = flat, [], False, False
        self.counted = [False] * len(self.params)
        self.armed = [False] * len(self.params)         # set by GradSync.sync_forward (misc.ddp_sync(module, True))
        self.pending = [len(m) for _, _, m in self.buckets]
        self.next_bucket = len(self.buckets) - 1        # the backward pass reaches the last parameters first

    def _on_grad(self, i):
        if self.sync._active is not self:
            return
        if self.counted[i]:
            if self.bucket_of[i] > self.next_bucket:    # its bucket is already on the wire with the previous value
                self.stale = True
            return
        if not self.last_round or not self.armed[i]:
            return
        self.counted[i] = True
        self.pending[self.bucket_of[i]] -= 1
        while self.next_bucket >= 0 and self.pending[self.next_bucket] == 0:
            self._launch(self.next_bucket, self.params)
            self.sync.buckets_launched_early += 1

    @torch.no_grad()
    def _launch(self, b, params):
        lo, hi, members = self.buckets[b]
        have = [i for i in members if params[i].grad is not None]
        if len(have) < len(members):
            self.flat[lo:hi].zero_()                    # a parameter without a gradient counts as zeros, as in pack()'s union case
        if have:
            torch._foreach_copy_([GradSync._slice(self.flat, params[i], self.offsets[i]) for i in have], [params[i].grad for i in have])
        view, wire = self.flat[lo:hi], None
        if self.sync.wire_dtype is not None and self.sync.wire_dtype != view.dtype:
            wire = view.to(self.sync.wire_dtype)
        work = torch.distributed.all_reduce(wire if wire is not None else view, group=self.sync.process_group, async_op=True)
        self.works.append((work, wire, view))
        self.next_bucket = b - 1

    def launch_rest(self, params):
        while self.next_bucket >= 0:
            self._launch(self.next_bucket, params)

    def wait(self):
        for work, wire, view in self.works:
            work.wait()
            if wire is not None:
                view.copy_(wire)
        self.works = []


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

    def end_backward(self):
        for m in _modules_of(self.module):
            if m is not None:
                m.requires_grad_(False)

    def step_params(self):
        """The parameters the optimizer owns (what the reference's nan_to_num loop walks, training_loop.py:310)."""
        if '_step_params' not in self:
            self['_step_params'] = [p for g in self.opt.param_groups for p in g['params']]
        return self['_step_params']

    def finish(self, grad_sync):
        self.end_backward()
        with torch.autograd.profiler.record_function(self.name + '_opt'):
            grad_sync(self.step_params(), key=id(self.opt))
            self.opt.step()


def phase_state(phase, modules=(), extra=(), flat=None, names=None):
    """Every tensor a phase may modify, as (name, tensor, strict) triples: the parameters its optimizer owns, the
    optimizer's state, the buffers of the modules it runs (``w_avg`` moves in every generator pass, also inside
    discriminator phases), the flat gradient buffer and `extra` (``pl_mean``, statistics counters) -- what PhaseGraphs
    saves and restores around the verification replays.  strict = compared tightly.  Parameters are compared loosely:
    Adam with beta1 = 0 moves a weight by +-lr whatever the size of its gradient, so rounding noise on a near-zero
    gradient (float atomics) flips whole steps; the gradients themselves, the moments and everything else are smooth."""
    names = names or {}
    out = [(names.get(id(p), f'param{i}') + str(list(p.shape)), p, False) for i, p in enumerate(phase.step_params())]
    for i, (p, st) in enumerate(phase.opt.state.items()):
        out += [(f'opt.{k}[{names.get(id(p), i)}]', v, True) for k, v in st.items() if isinstance(v, torch.Tensor)]
    for mi, m in enumerate(modules):
        if m is not None:
            out += [(f'{type(m).__name__}{mi}.{n}', b, True) for n, b in m.named_buffers()]
    if flat is not None:
        out.append(('flat_gradients', flat, True))
    out += [(f'extra{i}', t, True) for i, t in enumerate(extra) if isinstance(t, torch.Tensor)]
    seen, uniq = set(), []
    for name, t, strict in out:
        if t.is_cuda and t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            uniq.append((name, t, strict))
    return uniq


def run_phase(phase, accumulate, grad_sync, graphs=None, world_size=1, state_fn=None, overlap=True):
    """One phase of the loop (reference training_loop.py:288-317): zero_grad + requires_grad on, ``accumulate()`` (every
    accumulation round's forward/backward), requires_grad off, gradient exchange + scrub, optimizer step -- eagerly, or
    through ``graphs`` (PhaseGraphs).  With more than one rank the collective stays between two recorded halves.
    ``overlap=False`` keeps an eager phase on the plain exchange (one all-reduce after the backward pass): what a loop built for
    phase graphs uses for the steps it runs eagerly (bench.py's profiled steps), so that they exchange as the recorded ones do."""
    params, key = phase.step_params(), id(phase.opt)

    def forward_backward():
        phase.begin()
        accumulate()
        phase.end_backward()
        phase['_flat'] = grad_sync.pack(params, key)

    def step():
        grad_sync.unpack(params, phase['_flat'], key)
        phase.opt.step()

    def has_work():
        # a phase that computes nothing (global_Greg upstream) has nothing to record
        return phase['_flat'] is not None

    if graphs is None and world_size > 1 and grad_sync.overlap and overlap:
        with torch.autograd.profiler.record_function(phase.name):
            phase.begin()
            grad_sync.arm(params, key)
            accumulate()
            phase.end_backward()
            phase['_flat'] = grad_sync.finish(params, key)
            step()
    elif graphs is None:
        with torch.autograd.profiler.record_function(phase.name):
            forward_backward()
            grad_sync.exchange(phase['_flat'])
            step()
    elif world_size == 1:
        graphs.run(phase.name, lambda: (forward_backward(), step()), record_if=has_work, state_fn=state_fn)
    else:
        graphs.run((phase.name, 'fwd_bwd'), forward_backward, record_if=has_work, state_fn=state_fn)
        grad_sync.exchange(phase['_flat'])
        graphs.run((phase.name, 'step'), step, record_if=has_work, state_fn=state_fn)


def make_optimizer(params, opt_kwargs, device, capturable=False):
    """construct_class_by_name on the caller's kwargs; on the GPU torch.optim.Adam is asked for its
    fused (single multi-tensor kernel) implementation unless the caller decided otherwise."""
    kw = dict(opt_kwargs)
    if 'betas' in kw and not any(isinstance(b, torch.Tensor) for b in kw['betas']):
        kw['betas'] = tuple(float(b) for b in kw['betas'])          # upstream passes [0, 0.99]: today's Adam rejects an int beta
    if kw.get('class_name') == 'torch.optim.Adam' and torch.device(device).type == 'cuda' and 'fused' not in kw and 'foreach' not in kw:
        kw['fused'] = True
        if capturable:          # step counter on the device: the step can be recorded into a HIP graph (same arithmetic)
            kw.setdefault('capturable', True)
    return dnnlib.util.construct_class_by_name(params, **kw)


def build_phases(specs, device, capturable=False):
    """specs: iterable of (name_fmt, module_or_list, params, opt_kwargs, reg_interval, main_interval) where
    name_fmt has one '{}' for the phase kind ('G{}' -> 'Gmain', 'local_D{}_hair' -> 'local_Dreg_hair').
    Returns the Phase list in execution order: 'both', or 'main' followed by 'reg', per spec."""
    phases = []
    for name, module, params, opt_kwargs, reg_interval, main_interval in specs:
        params = list(params)
        if reg_interval is None:
            opt = make_optimizer(params, opt_kwargs, device, capturable)
            phases.append(Phase(name=name.format('both'), module=module, opt=opt, interval=main_interval))
        else:
            mb_ratio = reg_interval / (reg_interval + 1)
            kw = dnnlib.EasyDict(opt_kwargs)
            kw.lr = kw.lr * mb_ratio
            kw.betas = [beta ** mb_ratio for beta in kw.betas]
            opt = make_optimizer(params, kw, device, capturable)
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


def ada_adjust(signs_real_mean, ada_target, batch_size, ada_interval, ada_kimg):
    """Increment of the augmentation probability: sign(E[sign(D(real))] - target) * batch * interval / (ada_kimg * 1000)
    (reference training_loop.py:333-337)."""
    import numpy as np
    return float(np.sign(signs_real_mean - ada_target) * (batch_size * ada_interval) / (ada_kimg * 1000))


def broadcast_module_state(modules, src=0):
    """What the DDP constructor does at start-up: every rank takes rank 0's parameters and buffers."""
    for m in _modules_of(modules):
        if m is None:
            continue
        for t in misc.params_and_buffers(m):
            torch.distributed.broadcast(t.detach(), src=src)


class PhaseGraphs:
    """Whole phases recorded into HIP graphs and replayed.

    A phase of the loop (reference training_loop.py:288-317: zero_grad, forward, backward of every accumulation
    round, gradient post-processing, optimizer step) is ~1-3 thousand kernel launches whose arguments never change
    between iterations once the inputs live at fixed addresses: the host spends more time issuing them than the
    MI355X spends executing them at small per-GPU batches.  ``run(key, fn)`` executes ``fn`` eagerly the first
    ``warmup`` times (solver selection, lazy state, statistics counters), records it right after the last eager
    execution on a capture stream (``hipStreamBeginCapture`` through torch.cuda.graph) and from then on replays the
    graph: one host call per phase.  Everything a body reads must be a persistent tensor that the caller refreshes in
    place (reals, latents); everything it updates (parameters, Adam moments and step counters, EMA inputs, ``pl_mean``,
    ``w_avg``, statistics counters, the ADA probability ``p``) is device state updated by the recorded kernels.

    ``verify=True`` (default) proves every recording before it is trusted.  The state the phase may modify (``state_fn()``)
    and the generator state are saved before the eager execution.  The phase is then executed eagerly a SECOND time from that
    saved state: what the two eager results differ by is the phase's own run-to-run spread (float atomics in the library's
    split-K kernels and in the fused reductions, in bf16; 0.5-1.5 % of the update for the full-width local phases, 1e-7 for
    float32 toy networks -- profiles/r03_graph_probe_*.log).  After recording, the graph is replayed ``verify_replays`` (3)
    times from the same saved state and every replay must land within max(verify_tol, spread_factor x spread) of the eager
    result, measured on the STRICTLY compared state (the flat gradient buffer, Adam moments and counters, buffers, statistics:
    everything before Adam's beta1 = 0 turns rounding noise on near-zero gradients into +-lr steps); the weights' own deviation is
    kept as diagnostic output (``report``) and bounded loosely.  Also checked: the state must not move during the capture itself
    (a capture executes nothing; whatever moved ran outside the recording -- an autograd accumulator keyed to another stream).
    A phase whose two EAGER executions already differ by more than ``chaos_tol`` (0.25) of the update cannot be proved either way
    by comparing results -- at full width that is ``global_Gmain``: the composite's c/alpha terms make its gradient ill-conditioned
    and two eager executions from identical state and random draws differ by 1.2-6.5x the update (a replay: 1.4-6.5x; the same
    phase with OTHER random draws: 105x) -- and stays eager, listed with that measurement.  Round 2 read this as a replay fault of
    that one phase and kept it eager by name; the measurements of round 3 show eager re-execution to be no closer to itself.
    A library routine that prepares a buffer outside the captured stream fails a later replay; that phase stays eager (``rejected``).

    Multi-GPU: a collective is not recorded.  The phase is split at the gradient exchange -- graph 1 ends with the
    gradients packed into the phase's persistent flat buffer, the all-reduce runs eagerly on the same stream, graph 2
    (scrub, optimizer step) follows.
    Memory: every graph keeps its own private pool by default (``share_pool=False``): a graph's scratch buffers are then
    written by that graph alone, which is what makes two back-to-back verification replays representative of every later
    replay; sized for 288 GB of HBM3E (the 36 phase graphs of the 8-layer step at batch 32 hold well under half of it).
    ``share_pool=True`` lets the graphs reuse each other's scratch memory (they replay strictly one after another).
    """

    def __init__(self, device, warmup=1, verify=True, verify_tol=0.02, verify_tol_loose=0.5, share_pool=False, verify_replays=3,
                 never_record=(), spread_factor=4.0, chaos_tol=0.25):
        self.device = torch.device(device)
        self.never_record = tuple(never_record)         # phase names kept eager on request (none by default)
        self.warmup, self.verify, self.verify_tol, self.verify_tol_loose = warmup, verify, verify_tol, verify_tol_loose
        self.verify_replays, self.spread_factor, self.chaos_tol = verify_replays, spread_factor, chaos_tol
        self._seen = dict()         # key -> eager executions so far
        self._graphs = dict()       # key -> torch.cuda.CUDAGraph
        self.rejected = dict()      # key -> reason (stays eager)
        self.report = dict()        # key -> dict(eager_spread=, replays=[strict deviation per replay], loose=[...]) of the verification
        self._discarded = []
        self.share_pool = share_pool
        self._pool = None
        self.stream = torch.cuda.Stream(device=self.device)
        self.captured = 0
        self.replays = 0

    @contextlib.contextmanager
    def stream_scope(self):
        """Everything the loop does while graphs are in use -- eager warm-up executions, recordings, replays, the phases
        that stay eager -- runs on ONE side stream, the recording stream.  Autograd keys each parameter's gradient
        accumulator to the stream it was created on; an accumulator that an earlier eager pass created on the default
        stream and that is still alive when a phase is recorded would do its work on the default stream, i.e. OUTSIDE the
        recording: the replayed graph then never writes those gradients (torch warns "the AccumulateGrad node's stream
        does not match ... break CUDA graph capture").  With a single stream for all of it there is nothing to mismatch."""
        outer = torch.cuda.current_stream(self.device)
        if outer == self.stream:
            yield
            return
        self.stream.wait_stream(outer)
        with torch.cuda.stream(self.stream):
            yield
        outer.wait_stream(self.stream)

    def _capture(self, fn):
        gc.collect()            # autograd graphs of earlier eager passes that only a reference cycle keeps alive
        g = torch.cuda.CUDAGraph()
        if self.share_pool:
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(g, pool=self._pool, stream=self.stream):
                fn()
        else:
            with torch.cuda.graph(g, stream=self.stream):
                fn()
        return g

    @staticmethod
    def _same(a, b):
        """Bitwise-equal contents, a NaN in the same place on both sides included (a statistics counter that holds the sum of a NaN
        penalty compares unequal to itself under torch.equal)."""
        if a.is_floating_point():
            return bool(((a == b) | ((a != a) & (b != b))).all())
        return torch.equal(a, b)

    @staticmethod
    def _deviation(after, got, before):
        """|after - got| / |after - before| per tensor (float64 accumulation).  Non-finite values count as equal when both
        sides hold the same one at the same place (the R1 penalty of a fully transparent pixel is NaN upstream as well)."""
        big = 3.0e38
        a, g, b = (torch.nan_to_num(t.double(), nan=big, posinf=big / 2, neginf=-big / 2) for t in (after, got, before))
        num, den = float((a - g).square().sum()), float((a - b).square().sum())
        if num != num or num == float('inf'):
            return float('inf'), num, den
        return (num, den)[0] ** 0.5 / max(den, 1e-60) ** 0.5 if num > 0 else 0.0, num, den

    def run(self, key, fn, record_if=None, state_fn=None):
        """Execute fn(): eagerly the first `warmup` times, through its graph afterwards.  The recording is made right
        after the last eager execution (recording executes nothing on the device), so a phase that fires rarely --
        Dreg, every 16th iteration -- is replayed from its second firing on instead of being recorded in the middle
        of the run.  Returns True when the graph was replayed."""
        g = self._graphs.get(key)
        if g is not None:
            g.replay()
            self.replays += 1
            return True
        if key in self.rejected:
            fn()
            return False
        if (key[0] if isinstance(key, tuple) else key) in self.never_record:
            self.rejected[key] = 'not recorded on request (never_record)'
            fn()
            return False
        n = self._seen.get(key, 0) + 1
        self._seen[key] = n
        last = n >= self.warmup
        check = last and self.verify and state_fn is not None
        if check:
            before0 = {t.data_ptr(): t.detach().clone() for _, t, _ in state_fn()}          # keyed by storage address
            rng_before = torch.cuda.get_rng_state(self.device)
        fn()
        if not (last and (record_if is None or record_if())):
            return False
        if check:
            torch.cuda.synchronize(self.device)
            # state created lazily inside fn (Adam moments and step counters, statistics counters) started from zero
            entries = state_fn()
            state = [t.detach() for _, t, _ in entries]
            before = [before0[t.data_ptr()] if t.data_ptr() in before0 else torch.zeros_like(t) for t in state]
            after = [t.clone() for t in state]
            rng_after = torch.cuda.get_rng_state(self.device)

            def restart():
                torch._foreach_copy_(state, before)
                torch.cuda.set_rng_state(rng_before, self.device)

            def measure():
                """(strict deviation, loose deviation, worst strict tensors) of the current state against the eager result."""
                torch.cuda.synchronize(self.device)
                groups, offenders = {True: [0.0, 0.0], False: [0.0, 0.0]}, []
                for (name, _, strict), a_, s_, b_ in zip(entries, after, state, before):
                    dev, num, den = self._deviation(a_, s_, b_)
                    groups[strict][0] += num
                    groups[strict][1] += den
                    if strict and dev > self.verify_tol:
                        offenders.append((dev, name))
                out = []
                for strict in (True, False):
                    num, den = groups[strict]
                    out.append(float('inf') if (num != num or num == float('inf')) else (num / max(den, 1e-60)) ** 0.5 if num > 0 else 0.0)
                return out[0], out[1], ', '.join(f'{n} {d:.3g}' for d, n in sorted(offenders, reverse=True)[:4])

            # the phase's own run-to-run spread: a second eager execution from the same state and random draws
            restart()
            fn()
            spread, spread_loose, _ = measure()
            rep = self.report[key] = dict(eager_spread=spread, eager_spread_weights=spread_loose, replays=[], replays_weights=[])
            torch._foreach_copy_(state, after)
            if spread > self.chaos_tol:
                torch.cuda.set_rng_state(rng_after, self.device)
                self.rejected[key] = (f'not recorded: two EAGER executions from the same state and random draws differ by {spread:.3g} of the '
                                      f'update (ill-conditioned gradients under float atomics) -- a recording cannot be proved against that')
                return False
            restart()
            torch.cuda.synchronize(self.device)
            pre = [t.clone() for t in state]
        g = self._capture(fn)
        if check:
            why = ''
            torch.cuda.synchronize(self.device)
            moved = [name for (name, _, _), s_, p_ in zip(entries, state, pre) if not self._same(s_, p_)]
            del pre
            if moved:
                why = f'{len(moved)} tensors changed during the capture (work outside the recording): ' + ', '.join(moved[:4])
            tol = max(self.verify_tol, self.spread_factor * spread)
            for r in range(self.verify_replays if not why else 0):
                restart()
                g.replay()
                dev, dev_loose, top = measure()
                rep['replays'].append(dev)
                rep['replays_weights'].append(dev_loose)
                if dev > tol or dev_loose > max(self.verify_tol_loose, self.spread_factor * spread_loose):
                    why = (f'replay {r + 1} deviates from the eager execution by {dev:.3g} of the update on the strictly compared state '
                           f'(tolerance {tol:.3g} = max({self.verify_tol}, {self.spread_factor} x the eager spread {spread:.3g}); weights {dev_loose:.3g}); worst: {top}')
                    break
            torch._foreach_copy_(state, after)                  # continue from the eager result
            torch.cuda.set_rng_state(rng_after, self.device)
            if why:
                self.rejected[key] = why
                if self.share_pool:
                    self._discarded.append(g)       # keeps the shared pool's use count above zero; its blocks are reused by later recordings
                return False
        self._graphs[key] = g
        self.captured += 1
        return False
