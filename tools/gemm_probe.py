#!/usr/bin/env python3
"""Time the skinny float32 GEMMs of the discriminator epilogue / mapping layers in a few formulations."""
import torch


def t(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


dev = 'cuda:0'
for (m, k, n) in [(32, 8192, 512), (32, 512, 512), (32, 512, 8192), (64, 8192, 512), (32, 2048, 512)]:
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev)
    b = torch.randn(n, device=dev)
    g = torch.randn(m, n, device=dev)
    wt = w.t().contiguous()
    print(f'M={m} K={k} N={n}')
    print(f'  addmm(b, x, w.t())        {t(lambda: torch.addmm(b.unsqueeze(0), x, w.t())):8.1f} us')
    print(f'  x @ w.t()                 {t(lambda: x.matmul(w.t())):8.1f} us')
    print(f'  (w @ x.t()).t()           {t(lambda: (w @ x.t()).t()):8.1f} us')
    print(f'  x @ wt (pre-transposed)   {t(lambda: x @ wt):8.1f} us')
    print(f'  bf16 x @ w.t()            {t(lambda: x.bfloat16() @ w.bfloat16().t()):8.1f} us')
    print(f'  grad_w = g.t() @ x        {t(lambda: g.t() @ x):8.1f} us')
    print(f'  grad_w = (x.t() @ g).t()  {t(lambda: (x.t() @ g).t()):8.1f} us')
    print(f'  grad_x = g @ w            {t(lambda: g @ w):8.1f} us')
    print(f'  grad_x = (w.t() @ g.t()).t() {t(lambda: (w.t() @ g.t()).t()):8.1f} us')
