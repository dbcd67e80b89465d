#!/usr/bin/env python3
"""Is the data gradient of the stride-1 3x3 convolutions faster as a FORWARD convolution of dy with the flipped, transposed
weights (MIOpen's fwd solvers) than through aten::convolution_backward (its bwd-data solvers)?  HIP-event timings per problem."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
importlib.import_module(PKG)
importlib.import_module(f'{PKG}.tuning').use_tuned_miopen_db()
torch.backends.cudnn.benchmark = len(sys.argv) > 1 and sys.argv[1] == 'find'
dev = torch.device('cuda', 0)
iters = 20


def timed(fn):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


cb = torch.ops.aten.convolution_backward
n = 32
for dt, fmt, cases in ((torch.bfloat16, torch.channels_last, [(64, 64, 256), (128, 128, 128), (256, 256, 64), (512, 512, 32), (64, 128, 128), (128, 256, 64)]),
                       (torch.float32, torch.contiguous_format, [(512, 512, 16), (512, 512, 8)])):
    for ci, co, r in cases:
        x = torch.randn([n, ci, r, r], device=dev).to(dt).contiguous(memory_format=fmt)
        w = (torch.randn([co, ci, 3, 3], device=dev) / 30).to(dt).contiguous(memory_format=fmt)
        dy = torch.randn([n, co, r, r], device=dev).to(dt).contiguous(memory_format=fmt)
        args = (dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
        t_fwd = timed(lambda: torch.convolution(x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1))
        t_both = timed(lambda: cb(*args, [True, True, False]))
        t_dx = timed(lambda: cb(*args, [True, False, False]))
        t_dw = timed(lambda: cb(*args, [False, True, False]))

        def dgrad_as_fwd():
            wt = w.flip([2, 3]).transpose(0, 1).contiguous(memory_format=fmt)
            return torch.convolution(dy, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
        t_alt = timed(dgrad_as_fwd)
        ref = cb(*args, [True, False, False])[0].float()
        alt = dgrad_as_fwd().float()
        err = (ref - alt).abs().max().item() / ref.abs().max().item()
        print(f'{str(dt)[6:]:9s} {ci:4d}->{co:4d} {r:4d}^2: fwd {t_fwd:7.1f}  bwd(dx+dw) {t_both:7.1f}  dx {t_dx:7.1f}  dw {t_dw:7.1f}  dx-as-fwd {t_alt:7.1f} us  (rel diff {err:.2e})', flush=True)


print('--- strided / transposed problems: one call (dx+dw) against two calls', flush=True)
for dt, fmt, tr, wshape, xshape, stride, pad in (
        (torch.bfloat16, torch.channels_last, False, [128, 64, 3, 3], [n, 64, 257, 257], 2, 0),
        (torch.bfloat16, torch.channels_last, False, [256, 128, 3, 3], [n, 128, 129, 129], 2, 0),
        (torch.bfloat16, torch.channels_last, False, [512, 256, 3, 3], [n, 256, 65, 65], 2, 0),
        (torch.bfloat16, torch.channels_last, True, [128, 64, 3, 3], [n, 128, 128, 128], 2, 0),
        (torch.bfloat16, torch.channels_last, True, [256, 128, 3, 3], [n, 256, 64, 64], 2, 0),
        (torch.bfloat16, torch.channels_last, True, [512, 256, 3, 3], [n, 512, 32, 32], 2, 0),
        (torch.bfloat16, torch.channels_last, False, [4, 64, 1, 1], [n, 64, 256, 256], 1, 0),
        (torch.bfloat16, torch.channels_last, False, [128, 64, 1, 1], [n, 64, 128, 128], 1, 0),
        (torch.float32, torch.contiguous_format, False, [512, 512, 3, 3], [n, 512, 17, 17], 2, 0),
        (torch.float32, torch.contiguous_format, True, [512, 512, 3, 3], [n, 512, 8, 8], 2, 0),
        (torch.float32, torch.contiguous_format, False, [512, 513, 3, 3], [n, 513, 8, 8], 1, 1)):
    x = torch.randn(xshape, device=dev).to(dt).contiguous(memory_format=fmt)
    w = (torch.randn(wshape, device=dev) / 30).to(dt).contiguous(memory_format=fmt)
    y = torch.convolution(x, w, None, [stride] * 2, [pad] * 2, [1, 1], tr, [0, 0], 1)
    dy = torch.randn_like(y)
    args = (dy, x, w, None, [stride] * 2, [pad] * 2, [1, 1], tr, [0, 0], 1)
    t_both = timed(lambda: cb(*args, [True, True, False]))
    t_dx = timed(lambda: cb(*args, [True, False, False]))
    t_dw = timed(lambda: cb(*args, [False, True, False]))
    print(f'{str(dt)[6:]:9s} {"convT" if tr else "conv "} w{wshape} x{xshape} s{stride}: bwd(dx+dw) {t_both:7.1f}  dx {t_dx:7.1f}  dw {t_dw:7.1f}  sum {t_dx + t_dw:7.1f} us', flush=True)
