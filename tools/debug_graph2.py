#!/usr/bin/env python3
"""Diagnostic: per phase of a tiny train_aio trainer, eager execution vs replay of its recorded graph from identical state."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
DEV = torch.device('cuda', 0)


def mod(n):
    importlib.import_module(PKG)
    return importlib.import_module(f'{PKG}.{n}')


def main():
    ada = '--ada' in sys.argv
    share_pool = '--no-share' not in sys.argv
    TL, DS, AU, PE = mod('custom.training_loop_aio'), mod('custom.dataset_aio'), mod('training.augment'), mod('training.phase_engine')
    AU.fixed_margin = True
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=2, num_fp16_res=2, fp16_dtype=torch.bfloat16)
    for k in ('local_G_kwargs', 'local_D_kwargs', 'global_D_kwargs'):
        kw[k].channel_base, kw[k].channel_max = 256, 16
    kw.mapping_kwargs.num_layers = 2
    kw.training_set_kwargs.num_layers = 2
    if ada:
        kw.augment_kwargs = dict(class_name='training.augment.AugmentPipe', **AU.augpipe_specs['bgc'])
        kw.augment_p = 0.4
    reals = [DS.synthetic_layers(2, 2, 256, seed=5 + i, device=DEV) for i in range(2)]
    t = TL.AioTrainer(device=DEV, cfg=dict(TL.config, bypass_renderer=True), resident_batches=reals, random_seed=3, graphs=True, **kw)
    if not share_pool:
        orig = t.graphs._capture

        def cap(fn):
            g = torch.cuda.CUDAGraph()
            if t.graphs._stream is None:
                t.graphs._stream = torch.cuda.Stream(device=DEV)
            with torch.cuda.graph(g, stream=t.graphs._stream):
                fn()
            t.graphs.captured += 1
            return g
        t.graphs._capture = cap
    t.run_iteration()           # eager + record every phase
    torch.cuda.synchronize()
    mods = [t.mapping_network, *t.local_G_list, *t.local_D_list, t.pos_estimator, t.global_D]
    state = [x for m in mods for x in list(m.parameters()) + list(m.buffers())]
    for ph in t.phases:
        for st_ in ph.opt.state.values():
            state += [v for v in st_.values() if isinstance(v, torch.Tensor)]
    state += [l.pl_mean for l in t.loss.local_GAN_loss_list]
    state = list({id(x): x for x in state}.values())
    st = t._stage_inputs()
    graphs = t.graphs
    for phase, gz in zip(t.phases, st.z_rounds):
        if phase.name not in graphs._graphs:
            print(f'{phase.name:28s} not recorded'); continue
        snap = [x.detach().clone() for x in state]
        S = torch.cuda.get_rng_state(DEV)
        t.graphs = None
        t._run_phase(phase, st, gz)         # eager
        torch.cuda.synchronize()
        after_e = [x.detach().clone() for x in state]
        ge = [p.grad.detach().clone() for p in phase.step_params() if p.grad is not None]
        for x, s in zip(state, snap):
            x.detach().copy_(s)
        torch.cuda.set_rng_state(S, DEV)
        t.graphs = graphs
        t._run_phase(phase, st, gz)         # replay
        torch.cuda.synchronize()
        after_g = [x.detach().clone() for x in state]
        gg = [p.grad.detach().clone() for p in phase.step_params() if p.grad is not None]
        for x, s in zip(state, snap):
            x.detach().copy_(s)
        torch.cuda.set_rng_state(S, DEV)
        t.graphs = None
        t._run_phase(phase, st, gz)         # eager again: the spread
        torch.cuda.synchronize()
        after_e2 = [x.detach().clone() for x in state]
        t.graphs = graphs

        def dist(a, b, ref):
            num = sum(float((x.double() - y.double()).square().sum()) for x, y in zip(a, b)) ** 0.5
            den = sum(float((x.double() - r.double()).square().sum()) for x, r in zip(a, ref)) ** 0.5
            return num / max(den, 1e-30)
        gn = sum(float((x.double() - y.double()).square().sum()) for x, y in zip(ge, gg)) ** 0.5 / max(sum(float(x.double().square().sum()) for x in ge) ** 0.5, 1e-30) if len(ge) == len(gg) else float('nan')
        print(f'{phase.name:28s} state-delta: graph-vs-eager {dist(after_e, after_g, snap):.3e}  eager-vs-eager {dist(after_e, after_e2, snap):.3e}   grads rel diff {gn:.3e}  n_grads {len(ge)}/{len(gg)}', flush=True)


if __name__ == '__main__':
    main()
