#!/usr/bin/env python3
"""Diagnostic: single phases of the tiny trainer recorded on their own; structure of the recorded graph (DOT dump)."""
import importlib
import os
import re
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
    TL, DS, AU, PE = mod('custom.training_loop_aio'), mod('custom.dataset_aio'), mod('training.augment'), mod('training.phase_engine')
    AU.fixed_margin = True
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=2, num_fp16_res=2, fp16_dtype=torch.bfloat16)
    for k in ('local_G_kwargs', 'local_D_kwargs', 'global_D_kwargs'):
        kw[k].channel_base, kw[k].channel_max = 256, 16
    kw.mapping_kwargs.num_layers = 2
    kw.training_set_kwargs.num_layers = 2
    reals = [DS.synthetic_layers(2, 2, 256, seed=5 + i, device=DEV) for i in range(2)]
    t = TL.AioTrainer(device=DEV, cfg=dict(TL.config, bypass_renderer=True), resident_batches=reals, random_seed=3, graphs=True,
                      graph_warmup=10 ** 9, **kw)
    t.run_iteration()           # eager, nothing recorded
    torch.cuda.synchronize()
    mods = [t.mapping_network, *t.local_G_list, *t.local_D_list, t.pos_estimator, t.global_D]
    state = [x for m in mods for x in list(m.parameters()) + list(m.buffers())]
    for ph in t.phases:
        for st_ in ph.opt.state.values():
            state += [v for v in st_.values() if isinstance(v, torch.Tensor)]
    state += [l.pl_mean for l in t.loss.local_GAN_loss_list]
    state = list({id(x): x for x in state}.values())
    st = t._stage_inputs()
    names = sys.argv[1:] or ['local_Dmain_layer0', 'local_Gmain_layer0', 'local_Dmain_layer0']
    saved_graphs = t.graphs
    for trial, name in enumerate(names):
        mode = 'global'
        if ':' in name:
            name, mode = name.split(':')
        idx = [p.name for p in t.phases].index(name)
        phase, gz = t.phases[idx], st.z_rounds[idx]
        t.graphs = None

        def body():
            t._run_phase(phase, st, gz)
        snap = [x.detach().clone() for x in state]
        S = torch.cuda.get_rng_state(DEV)
        body(); torch.cuda.synchronize()
        after_e = [x.detach().clone() for x in state]
        ge = [p.grad.detach().clone() for p in phase.step_params() if p.grad is not None]
        for x, s in zip(state, snap):
            x.detach().copy_(s)
        g = torch.cuda.CUDAGraph()
        g.enable_debug_mode()
        with torch.cuda.graph(g, stream=torch.cuda.Stream(device=DEV), capture_error_mode=mode):
            body()
        dot = f'gpurun_out/r2_graph_{trial}_{name}.dot'
        try:
            g.debug_dump(dot)
        except Exception as e:
            print('debug_dump failed', e)
        for rep in range(2):
            for x, s in zip(state, snap):
                x.detach().copy_(s)
            torch.cuda.set_rng_state(S, DEV)
            g.replay(); torch.cuda.synchronize()
            after_g = [x.detach().clone() for x in state]
            gg = [p.grad.detach().clone() for p in phase.step_params() if p.grad is not None]

            def dist(a, b, ref):
                num = sum(float((x.double() - y.double()).square().sum()) for x, y in zip(a, b)) ** 0.5
                den = sum(float((x.double() - r.double()).square().sum()) for x, r in zip(a, ref)) ** 0.5
                return num / max(den, 1e-30)
            gn = sum(float((x.double() - y.double()).square().sum()) for x, y in zip(ge, gg)) ** 0.5 / max(sum(float(x.double().square().sum()) for x in ge) ** 0.5, 1e-30)
            bad = [(i, float((x.double() - y.double()).abs().max())) for i, (x, y) in enumerate(zip(ge, gg)) if not torch.allclose(x, y, rtol=1e-2, atol=1e-5)]
            print(f'{name:24s} mode={mode:12s} replay {rep}: state-delta graph-vs-eager {dist(after_e, after_g, snap):.3e}  grads rel diff {gn:.3e}  bad grads {len(bad)}/{len(ge)} {bad[:6]}', flush=True)
        if os.path.exists(dot):
            txt = open(dot).read()
            nodes = set(re.findall(r'^\\s*"?(\\w+)"?\\s*\\[', txt, flags=re.M))
            edges = re.findall(r'"?(\\w+)"?\\s*->\\s*"?(\\w+)"?', txt)
            has_pred = {b for a, b in edges}
            has_succ = {a for a, b in edges}
            print(f'   dot: {len(txt)} bytes, nodes~{len(nodes)}, edges {len(edges)}, without predecessor {len(nodes - has_pred)}, without successor {len(nodes - has_succ)}', flush=True)
            os.system(f'head -c 1500 {dot}; echo; grep -c KERNEL {dot}; grep -c MEMCPY {dot}; grep -c MEMSET {dot}')
            if len(txt) > 2000000:
                os.remove(dot)
    t.graphs = saved_graphs


if __name__ == '__main__':
    main()
