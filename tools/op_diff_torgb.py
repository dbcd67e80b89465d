#!/usr/bin/env python3
"""Which aten ops does one phase dispatch with the fused ToRGB op on and off?  (host-side torch.profiler, op name + stack top)"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
import bench  # noqa: E402

ns = argparse.Namespace(batch_gpu=8, layers=1, res=256, dtype='bf16', tiny=False, no_ada=True, ada_p=0.3, ada_margin='side',
                        layout='nhwc', nhwc_1x1='conv', workload='aio', graphs='off')
bench.mod('tuning').use_tuned_miopen_db()
torch.cuda.set_device(0)
NW = bench.mod('training.networks')
trainer, name = bench.build_trainer(ns, torch.device('cuda', 0), 0, 1)
torch.backends.cudnn.benchmark = False
st = trainer._stage_inputs()
res = {}
for phase_name in sys.argv[1:] or ['local_Greg_layer0', 'local_Gmain_layer0']:
    for flag in (True, False):
        NW.fused_torgb = flag
        ph = [(p, z) for p, z in zip(trainer.phases, st.z_rounds) if p.name == phase_name][0]
        for _ in range(2):
            trainer._run_phase(ph[0], st, ph[1])
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
            trainer._run_phase(ph[0], st, ph[1])
            torch.cuda.synchronize()
        c = collections.Counter()
        for e in prof.key_averages(group_by_stack_n=3):
            top = ' <- '.join(s.split('/')[-1][:50] for s in e.stack[:2]) if e.stack else ''
            c[(e.key, top)] += e.count
        res[flag] = c
    keys = set(res[True]) | set(res[False])
    rows = sorted(((res[True][k] - res[False][k], k) for k in keys if res[True][k] != res[False][k]), reverse=True)
    print(f'## {phase_name}: ops with the fused ToRGB minus without ({sum(res[True].values())} vs {sum(res[False].values())} in total)')
    for d, k in rows[:40]:
        print(f'{d:+5d}  {k[0]:40s} {k[1]}')
    for d, k in rows[-15:]:
        print(f'{d:+5d}  {k[0]:40s} {k[1]}')
