#!/usr/bin/env python3
"""Which aten ops (and with which input shapes) does one training iteration dispatch?  torch.profiler on the host side.
usage: op_census.py [bench args]   (runs 2 warm-up iterations, profiles 1)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
import bench  # noqa: E402
import argparse  # noqa: E402

ns = argparse.Namespace(batch_gpu=32, layers=8, res=256, dtype='bf16', tiny=False, no_ada=False, ada_p=0.3, ada_margin='side',
                        layout='nhwc', nhwc_1x1='conv', workload='aio', graphs='off')
bench.mod('tuning').use_tuned_miopen_db()
torch.cuda.set_device(0)
trainer, name = bench.build_trainer(ns, torch.device('cuda', 0), 0, 1)
torch.backends.cudnn.benchmark = False
bench.mod('torch_utils.training_stats').enabled = True
for _ in range(2):
    trainer.run_iteration()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    trainer.run_iteration()
    torch.cuda.synchronize()
want = ('aten::copy_', 'aten::fill_', 'aten::mul', 'aten::add_', 'aten::add', 'aten::sum', 'aten::pow', 'aten::zeros', 'aten::clone', 'aten::contiguous')
rows = prof.key_averages(group_by_input_shape=True, group_by_stack_n=4)
rows = sorted(rows, key=lambda e: -e.count)
for e in rows[:140]:
    if e.key in want:
        st = ' <- '.join(s.split('/')[-1][:60] for s in e.stack[:3]) if e.stack else ''
        print(f'{e.count:6d}  {e.key:18s} {str(e.input_shapes)[:60]:60s} {st}')

print('---- all ops by count (no shape grouping)')
tot = 0
for e in sorted(prof.key_averages(), key=lambda e: -e.count)[:60]:
    print(f'{e.count:6d}  {e.key:48s} self cpu {e.self_cpu_time_total / 1e3:8.1f} ms   total cpu {e.cpu_time_total / 1e3:8.1f} ms')
print('---- by self cpu time')
for e in sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)[:40]:
    print(f'{e.count:6d}  {e.key:48s} self cpu {e.self_cpu_time_total / 1e3:8.1f} ms')

print('---- GEMM-like ops by input shapes')
for e in sorted([e for e in prof.key_averages(group_by_input_shape=True) if e.key in ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm', 'aten::einsum', 'aten::matmul', 'aten::linear')], key=lambda e: -e.count)[:40]:
    print(f'{e.count:6d}  {e.key:16s} {str(e.input_shapes)[:110]:110s} self cpu {e.self_cpu_time_total / 1e3:7.2f} ms')
