#!/usr/bin/env python3
"""GPU time of multi-tensor vs per-tensor weight scaling for a StyleGAN2-sized parameter set."""
import torch
dev = 'cuda:0'
shapes = [(512, 512, 3, 3)] * 7 + [(256, 512, 3, 3), (256, 256, 3, 3), (128, 256, 3, 3), (128, 128, 3, 3), (64, 128, 3, 3), (64, 64, 3, 3)] + [(512, 512)] * 21 + [(512,)] * 14
ws = [torch.randn(*s, device=dev) for s in shapes]
gains = [0.5] * len(ws)
print('tensors', len(ws), 'MB', sum(w.numel() for w in ws) * 4 / 1e6)


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print(f'_foreach_mul            {t(lambda: torch._foreach_mul(ws, gains)):8.1f} us')
print(f'per-tensor mul          {t(lambda: [w * g for w, g in zip(ws, gains)]):8.1f} us')
sc = torch._foreach_mul(ws, gains)
dst = [torch.empty_like(w, dtype=torch.bfloat16) for w in sc]
print(f'_foreach_copy_ to bf16  {t(lambda: torch._foreach_copy_(dst, sc)):8.1f} us')
print(f'per-tensor .to(bf16)    {t(lambda: [w.to(torch.bfloat16) for w in sc]):8.1f} us')
