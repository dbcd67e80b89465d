#!/usr/bin/env python3
"""Diagnostic: which random tensors of global_Gmain differ between the eager execution and the replay of its recording."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
DEV = torch.device('cuda', 0)


def mod(n):
    importlib.import_module(PKG)
    return importlib.import_module(f'{PKG}.{n}')


def main():
    batch, layers = 4, 2
    TL, DS, AU, NW, PE = mod('custom.training_loop_aio'), mod('custom.dataset_aio'), mod('training.augment'), mod('training.networks'), mod('training.phase_engine')
    mod('tuning').use_tuned_miopen_db()
    AU.fixed_margin = True
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=batch, num_fp16_res=4, fp16_dtype=torch.bfloat16)
    kw.training_set_kwargs.num_layers = layers
    kw.local_G_kwargs.fp16_channels_last = True
    kw.local_D_kwargs.block_kwargs.fp16_channels_last = True
    kw.global_D_kwargs.block_kwargs.fp16_channels_last = True
    reals = [DS.synthetic_layers(batch, layers, 256, seed=i, device=DEV) for i in range(2)]
    t = TL.AioTrainer(device=DEV, cfg=dict(TL.config, bypass_renderer=True), resident_batches=reals, random_seed=0, graphs=True, **kw)
    torch.backends.cudnn.benchmark = False
    # record every random tensor drawn through torch.randn / torch.rand / random_ in call order into persistent buffers
    bufs, counter = [], [0]
    orig_randn, orig_rand, orig_randn_like = torch.randn, torch.rand, torch.randn_like

    def keep(tag, v):
        i = counter[0]; counter[0] += 1
        if i >= len(bufs):
            bufs.append((tag, torch.zeros_like(v, dtype=torch.float32)))
        bufs[i][1].copy_(v.float())
        return v
    torch.randn = lambda *a, **k: keep('randn' + str(list(a[0]) if a and isinstance(a[0], (list, tuple)) else a), orig_randn(*a, **k))
    torch.rand = lambda *a, **k: keep('rand' + str(list(a[0]) if a and isinstance(a[0], (list, tuple)) else a), orig_rand(*a, **k))
    torch.randn_like = lambda x, **k: keep('randn_like' + str(list(x.shape)), orig_randn_like(x, **k))
    if '--fill' in sys.argv:
        orig_zeros = torch.zeros
        torch.zeros = lambda *a, **k: torch.empty(*a, **k).fill_(0)           # a fill KERNEL instead of a memset node
        torch.zeros_like = lambda x, **k: torch.empty_like(x, **k).fill_(0)
    orig_begin = PE.Phase.begin

    def begin(self):
        counter[0] = 0
        return orig_begin(self)
    PE.Phase.begin = begin
    orig_state = t._phase_state

    def state_with_draws(phase):
        return orig_state(phase) + [(f'draw{i}:{tag}', b, True) for i, (tag, b) in enumerate(bufs)]
    t._phase_state = state_with_draws
    st = t._stage_inputs()
    with t.graphs.stream_scope():
        for phase, gz in zip(t.phases, st.z_rounds):
            if phase.name not in ('global_Gmain', 'local_Gmain_layer1'):
                continue
            counter[0] = 0
            bufs.clear()
            t._run_phase(phase, st, gz)
            torch.cuda.synchronize()
            print(phase.name, 'draws recorded:', len(bufs), '->', 'REJECTED: ' + t.graphs.rejected[phase.name][:1500] if phase.name in t.graphs.rejected else 'verified')


if __name__ == '__main__':
    main()
