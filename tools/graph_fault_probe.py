#!/usr/bin/env python3
"""Diagnostic for the phase-graph replay fault: records ONE phase of the aio loop and compares, tensor by tensor,
  * a second EAGER execution from the same saved state (the run-to-run noise floor),
  * the state right after the capture (a capture must execute nothing: anything that moved ran outside the recording),
  * R replays from the same saved state, against the eager result and against each other,
and summarises the topology of the captured graph (chain, or forks/joins and the kernels at them).

    python tools/graph_fault_probe.py [--phase global_Gmain] [--layers 2] [--batch 4] [--replays 5] [--res 256] [--channels full|tiny]
                                      [--fused-epilogue 0|1] [--dot 0|1]
"""
import argparse
import collections
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


def dev_of(after, got, before):
    a, g, b = (torch.nan_to_num(t.double(), nan=3e38, posinf=1e38, neginf=-1e38) for t in (after, got, before))
    num, den = float((a - g).square().sum()) ** 0.5, float((a - b).square().sum()) ** 0.5
    return num / max(den, 1e-30), num, den


def head(t, k=4):
    return [float(f'{v:.6g}') for v in t.detach().double().flatten()[:k].tolist()]


def dot_summary(path):
    """Node / edge statistics of a hipGraphDebugDotPrint file."""
    txt = open(path, errors='replace').read()
    edges = re.findall(r'"?([\w]+)"?\s*->\s*"?([\w]+)"?', txt)
    labels = dict(re.findall(r'"?([\w]+)"?\s*\[[^\]]*label="([^"]*)"', txt))
    indeg, outdeg = collections.Counter(), collections.Counter()
    for a, b in edges:
        outdeg[a] += 1
        indeg[b] += 1
    nodes = set(labels) | set(indeg) | set(outdeg)
    forks = [n for n in nodes if outdeg[n] > 1]
    joins = [n for n in nodes if indeg[n] > 1]
    roots = [n for n in nodes if indeg[n] == 0]
    leaves = [n for n in nodes if outdeg[n] == 0]
    kinds = collections.Counter()
    for n, l in labels.items():
        l = l.replace('\\n', ' ')
        kinds['memset' if 'emset' in l else 'memcpy' if 'emcpy' in l else 'kernel' if ('ernel' in l or '(' in l) else 'other'] += 1
    out = dict(nodes=len(nodes), edges=len(edges), forks=len(forks), joins=len(joins), roots=len(roots), leaves=len(leaves), kinds=dict(kinds))
    out['fork_labels'] = [labels.get(n, n)[:100] for n in forks[:8]]
    out['join_labels'] = [labels.get(n, n)[:100] for n in joins[:8]]
    out['root_labels'] = [labels.get(n, n)[:100] for n in roots[:8]]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--phase', default='global_Gmain')
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--replays', type=int, default=5)
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--channels', default='full')
    ap.add_argument('--fused-epilogue', type=int, default=1)
    ap.add_argument('--ada', type=int, default=1)
    ap.add_argument('--dot', type=int, default=0)
    ap.add_argument('--tol', type=float, default=0.02)
    ap.add_argument('--top', type=int, default=6)
    ap.add_argument('--eager-repeats', type=int, default=1)
    ap.add_argument('--wrong-rng', type=int, default=0)
    ap.add_argument('--fp16-res', type=int, default=4)
    ap.add_argument('--nhwc', type=int, default=1)
    ap.add_argument('--out', default='gpurun_out/graph_probe')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    TL, DS, AU, NW, PE = mod('custom.training_loop_aio'), mod('custom.dataset_aio'), mod('training.augment'), mod('training.networks'), mod('training.phase_engine')
    mod('tuning').use_tuned_miopen_db()
    AU.fixed_margin = True
    NW.fused_epilogue = bool(a.fused_epilogue)
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=a.batch, num_fp16_res=a.fp16_res, fp16_dtype=torch.bfloat16)
    kw.training_set_kwargs.num_layers = a.layers
    if a.channels == 'tiny':
        for k in ('local_G_kwargs', 'local_D_kwargs', 'global_D_kwargs'):
            kw[k].channel_base, kw[k].channel_max = 256, 16
        kw.mapping_kwargs.num_layers = 2
    if a.nhwc:
        kw.local_G_kwargs.fp16_channels_last = True
        kw.local_D_kwargs.block_kwargs.fp16_channels_last = True
        kw.global_D_kwargs.block_kwargs.fp16_channels_last = True
    if a.ada:
        kw.augment_kwargs = dict(class_name='training.augment.AugmentPipe', **AU.augpipe_specs['bgc'])
        kw.augment_p = 0.3
    reals = [DS.synthetic_layers(a.batch, a.layers, a.res, seed=i, device=DEV) for i in range(2)]
    t = TL.AioTrainer(device=DEV, cfg=dict(TL.config, bypass_renderer=True), resident_batches=reals, random_seed=0, graphs=True, **kw)
    torch.backends.cudnn.benchmark = False
    G = t.graphs
    st = t._stage_inputs()
    names = [p.name for p in t.phases]
    assert a.phase == 'all' or a.phase in names, names
    print(f'# probe: phase {a.phase}, layers {a.layers}, batch {a.batch}, channels {a.channels}, fused_epilogue {a.fused_epilogue}, ada {a.ada}', flush=True)
    with G.stream_scope():
        for phase, gz in zip(t.phases, st.z_rounds):
            if a.phase != 'all' and phase.name != a.phase:
                # the phases in front of it run eagerly once (their optimizer steps precede the probed phase in the loop)
                saved, t.graphs = t.graphs, None
                t._run_phase(phase, st, gz)
                t.graphs = saved
                continue
            if phase.name == 'global_Greg':
                continue
            print(f'## {phase.name}', flush=True)
            state_fn = lambda: t._phase_state(phase)
            rounds = len(st.blchw)

            def accumulate():
                for r, (real_layer, gen_z, real_list) in enumerate(zip(st.blchw, gz, st.lists)):
                    t.loss.accumulate_gradients(phase=phase.name, real_blchw=real_layer, real_list_of_bchw=real_list, gen_z=gen_z,
                                                sync=(r == rounds - 1), gain=phase.interval)
            params, key = phase.step_params(), id(phase.opt)

            def fn():
                phase.begin()
                accumulate()
                phase.end_backward()
                phase['_flat'] = t.grad_sync.pack(params, key)
                t.grad_sync.unpack(params, phase['_flat'], key)
                phase.opt.step()

            before0 = {x.data_ptr(): x.detach().clone() for _, x, _ in state_fn()}
            rng_before = torch.cuda.get_rng_state(DEV)
            fn()
            torch.cuda.synchronize()
            entries = state_fn()
            state = [x.detach() for _, x, _ in entries]
            names_ = [n for n, _, _ in entries]
            before = [before0[x.data_ptr()] if x.data_ptr() in before0 else torch.zeros_like(x) for x in state]
            after = [x.clone() for x in state]
            rng_after = torch.cuda.get_rng_state(DEV)

            def restore():
                torch._foreach_copy_(state, before)
                torch.cuda.set_rng_state(rng_before, DEV)

            def report(tag, got, ref=after, top=10):
                rows = []
                for n, a_, g_, b_ in zip(names_, ref, got, before):
                    d, num, den = dev_of(a_, g_, b_)
                    if d > a.tol:
                        rows.append((d, n, a_, g_, b_))
                rows.sort(key=lambda r: -r[0])
                # PhaseGraphs' accept criterion: sqrt(sum |ref - got|^2 / sum |ref - before|^2) over the strictly / loosely compared groups
                grp = {True: [0.0, 0.0], False: [0.0, 0.0]}
                for (n, _, strict), a_, g_, b_ in zip(entries, ref, got, before):
                    _, num, den = dev_of(a_, g_, b_)
                    grp[strict][0] += num * num
                    grp[strict][1] += den * den
                gs_, gl_ = ((grp[k][0] / max(grp[k][1], 1e-60)) ** 0.5 for k in (True, False))
                per = sorted(dev_of(a_, g_, b_)[0] for (n, _, strict), a_, g_, b_ in zip(entries, ref, got, before) if strict and a_.numel() > 1)
                q = [per[int(f * (len(per) - 1))] for f in (0.1, 0.25, 0.5, 0.75, 0.9)] if per else []
                print(f'[{tag}] group deviation strict {gs_:.4g} loose {gl_:.4g}; tensors over {a.tol}: {len(rows)} of {len(names_)}; '
                      f'per-tensor deviation (strict, numel > 1) quantiles 10/25/50/75/90 %: ' + ' '.join(f'{v:.3g}' for v in q), flush=True)
                for d, n, a_, g_, b_ in rows[:top]:
                    print(f'    {d:9.3g}  {n}  ref {head(a_)} got {head(g_)} before {head(b_)}', flush=True)
                return rows

            # (1) the noise floor: the same phase eagerly again, from the same state
            restore()
            fn()
            torch.cuda.synchronize()
            report('eager again vs eager', [x.clone() for x in state], top=a.top)
            for extra in range(a.eager_repeats - 1):
                restore()
                fn()
                torch.cuda.synchronize()
                report(f'eager again ({extra + 2}) vs eager', [x.clone() for x in state], top=0)
            if a.wrong_rng:
                # what a WRONG replay looks like under the same measure: the same phase with different random draws
                torch._foreach_copy_(state, before)
                torch.cuda.set_rng_state(rng_after, DEV)
                fn()
                torch.cuda.synchronize()
                report('eager with OTHER random draws vs eager', [x.clone() for x in state], top=0)
            # (2) capture: nothing may execute
            restore()
            torch.cuda.synchronize()
            pre = [x.clone() for x in state]
            g = torch.cuda.CUDAGraph()
            if a.dot:
                g.enable_debug_mode()
            import gc
            gc.collect()
            with torch.cuda.graph(g, stream=G.stream):
                fn()
            torch.cuda.synchronize()
            moved = [(n, float((x.double() - y.double()).abs().max())) for n, x, y in zip(names_, state, pre) if not torch.equal(x, y)]
            print(f'[capture] tensors that changed although a capture executes nothing: {len(moved)}', moved[:12], flush=True)
            if a.dot:
                path = os.path.join(a.out, f'{a.phase}.dot')
                try:
                    g.debug_dump(path)
                    print('[graph]', dot_summary(path), flush=True)
                    os.remove(path)
                except Exception as e:      # noqa
                    print('[graph] debug_dump failed:', repr(e), flush=True)
            # (3) replays
            outs = []
            for rep in range(a.replays):
                restore()
                g.replay()
                torch.cuda.synchronize()
                got = [x.clone() for x in state]
                report(f'replay {rep + 1} vs eager', got, top=a.top)
                if outs:
                    report(f'replay {rep + 1} vs replay 1', got, ref=outs[0], top=min(a.top, 2))
                outs.append(got)
            # (4) replay WITHOUT restoring (what the loop does): state continues from the last replay
            torch._foreach_copy_(state, after)
            torch.cuda.set_rng_state(rng_after, DEV)
            del g
            if a.phase != 'all':
                break
    print('# done', flush=True)


if __name__ == '__main__':
    main()
