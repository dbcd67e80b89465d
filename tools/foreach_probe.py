#!/usr/bin/env python3
"""How many kernels do the multi-tensor ops this build relies on launch?"""
import torch
from torch.profiler import profile, ProfilerActivity

dev = 'cuda:0'
ws = [torch.randn(512, 512, 3, 3, device=dev).to(memory_format=torch.channels_last) for _ in range(20)] + [torch.randn(512, 512, device=dev) for _ in range(20)]
gains = [0.1 + 0.01 * i for i in range(len(ws))]


def count(fn, name):
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        fn(); torch.cuda.synchronize()
    ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print(f'{name:50s} kernels: {len(ks):4d}   ', sorted({e.name[:60] for e in ks})[:3])


count(lambda: torch._foreach_mul(ws, gains), '_foreach_mul(list, scalarlist)')
count(lambda: [w * g for w, g in zip(ws, gains)], 'per-tensor mul')
sc = torch._foreach_mul(ws, gains)
dst = [torch.empty_like(w, dtype=torch.bfloat16) for w in sc]
count(lambda: torch._foreach_copy_(dst, sc), '_foreach_copy_ fp32 -> bf16')
count(lambda: [w.to(torch.bfloat16) for w in sc], 'per-tensor .to(bf16)')
dst32 = [torch.empty_like(w) for w in sc]
count(lambda: torch._foreach_copy_(dst32, dst), '_foreach_copy_ bf16 -> fp32')
assert all(d.is_contiguous(memory_format=torch.channels_last) for d in dst[:20])
assert all(torch.equal(d, (w * g).to(torch.bfloat16)) for d, w, g in zip(dst, ws, gains))
print('values identical to per-tensor mul + cast')
sq = torch._foreach_mul(ws[:20], ws[:20])
count(lambda: torch._foreach_mul(ws[:20], ws[:20]), '_foreach_mul(list, list) (squares)')
count(lambda: [w.square().sum(dim=[2, 3]) for w in ws[:20]], 'per-tensor square().sum([2,3])')
