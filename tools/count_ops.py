#!/usr/bin/env python3
"""Count aten ops dispatched per source line of the package during one tiny train_aio iteration (CPU or GPU)."""
import collections
import importlib
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_site = collections.Counter()
        self.by_op = collections.Counter()
        self.site_ops = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        site = 'autograd/other'
        for fr in reversed(traceback.extract_stack(limit=14)):
            if PKG in fr.filename and 'count_ops' not in fr.filename:
                site = f'{fr.filename.split(PKG + "/")[-1]}:{fr.lineno} {fr.name}'
                break
        self.by_site[site] += 1
        self.by_op[name] += 1
        self.site_ops[site][name] += 1
        return func(*args, **(kwargs or {}))


def main():
    dev = 'cuda' if torch.cuda.is_available() else 'cpu'
    importlib.import_module(PKG)
    TL = importlib.import_module(f'{PKG}.custom.training_loop_aio')
    AU = importlib.import_module(f'{PKG}.training.augment')
    DS = importlib.import_module(f'{PKG}.custom.dataset_aio')
    AU.fixed_margin = True
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=4, num_fp16_res=0 if dev == 'cpu' else 4, fp16_dtype=torch.bfloat16)
    for k in ('local_G_kwargs', 'local_D_kwargs', 'global_D_kwargs'):
        kw[k].channel_base, kw[k].channel_max = 512, 8
    kw.training_set_kwargs.num_layers = 2
    kw.augment_kwargs = dict(class_name='training.augment.AugmentPipe', **AU.augpipe_specs['bgc'])
    kw.augment_p = 0.3
    reals = [DS.synthetic_layers(4, 2, 256, seed=1, device=dev)]
    t = TL.AioTrainer(device=dev, cfg=dict(TL.config, bypass_renderer=True), resident_batches=reals, **kw)
    t.run_iteration()
    only = sys.argv[1] if len(sys.argv) > 1 else None
    c = Counter()
    with c:
        t.run_iteration()           # batch_idx 1: main phases only
    tot = sum(c.by_site.values())
    print('total aten ops', tot)
    for site, n in c.by_site.most_common(60):
        top = ', '.join(f'{k.replace("aten.", "")}x{v}' for k, v in c.site_ops[site].most_common(4))
        print(f'{n:6d} {100 * n / tot:5.1f}%  {site}   [{top}]')
    print('--- by op')
    for op, n in c.by_op.most_common(40):
        print(f'{n:6d}  {op}')


if __name__ == '__main__':
    main()
