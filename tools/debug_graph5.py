#!/usr/bin/env python3
"""Diagnostic: the bench configuration (full channel counts) with phase graphs + verification: which phases verify."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
DEV = torch.device('cuda', 0)


def mod(n):
    importlib.import_module(PKG)
    return importlib.import_module(f'{PKG}.{n}')


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    TL, DS, AU = mod('custom.training_loop_aio'), mod('custom.dataset_aio'), mod('training.augment')
    mod('tuning').use_tuned_miopen_db()
    AU.fixed_margin = True
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=batch, num_fp16_res=4, fp16_dtype=torch.bfloat16)
    kw.augment_kwargs = dict(class_name='training.augment.AugmentPipe', **AU.augpipe_specs['bgc'])
    kw.augment_p = 0.3
    kw.training_set_kwargs.num_layers = layers
    kw.local_G_kwargs.fp16_channels_last = True
    kw.local_D_kwargs.block_kwargs.fp16_channels_last = True
    kw.global_D_kwargs.block_kwargs.fp16_channels_last = True
    reals = [DS.synthetic_layers(batch, layers, 256, seed=i, device=DEV) for i in range(2)]
    t = TL.AioTrainer(device=DEV, cfg=dict(TL.config, bypass_renderer=True), resident_batches=reals, random_seed=0, graphs=True, **kw)
    torch.backends.cudnn.benchmark = False
    for i in range(3):
        t0 = time.time()
        t.run_iteration()
        torch.cuda.synchronize()
        print(f'iteration {i}: {time.time() - t0:.2f} s; recorded {t.graphs.captured}, rejected {len(t.graphs.rejected)}, replays {t.graphs.replays}', flush=True)
    for k, v in t.graphs.rejected.items():
        print('REJECTED', k, v)
    print('recorded:', sorted(map(str, t.graphs._graphs)))


if __name__ == '__main__':
    main()
