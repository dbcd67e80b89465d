#!/usr/bin/env python3
"""ToRGB + image-path cast: the one-pass native op (ops/torgb.py) against the four launches it replaces, forward and backward,
at the step's shapes (HIP events, median of 20 after 5 warm-up calls; algorithmic bytes = x (+ dx) + the float32 planes)."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
importlib.import_module(PKG)
TG = importlib.import_module(f'{PKG}.torch_utils.ops.torgb')
native = importlib.import_module(f'{PKG}.torch_utils.native')
importlib.import_module(f'{PKG}.tuning').use_tuned_miopen_db()
DEV = 'cuda:0'


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    for (n, c, r) in [(32, 64, 256), (32, 128, 128), (32, 256, 64), (32, 512, 32), (4, 64, 256)]:
        x = torch.randn([n, c, r, r], device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn([4, c, 1, 1], device=DEV) / c ** 0.5).to(torch.bfloat16).requires_grad_(True)
        s = (torch.randn([n, c], device=DEV) + 1).requires_grad_(True)
        b = torch.zeros([4], device=DEV).to(torch.bfloat16).requires_grad_(True)
        gy = torch.randn([n, 4, r, r], device=DEV)
        row = []
        for flag in (True, False):
            TG.enabled = flag
            y = TG.torgb(x, w, s, b, clamp=256)
            fwd = timeit(lambda: TG.torgb(x, w, s, b, clamp=256))
            bwd = timeit(lambda: torch.autograd.grad([y], [x, w, s, b], [gy], retain_graph=True))
            row += [fwd, bwd]
        TG.enabled = True
        # the kernels alone (HIP events inside the launch scope of the library: no Python between the two events)
        native.prof_reset(); native.prof_enable(True)
        for _ in range(10):
            y = TG.torgb(x, w, s, b, clamp=256)
        torch.cuda.synchronize()
        kf = native.prof_collect()['torgb']
        native.prof_reset()
        for _ in range(10):
            torch.autograd.grad([y], [x, w, s, b], [gy], retain_graph=True)
        torch.cuda.synchronize()
        kb = native.prof_collect()['torgb']
        native.prof_enable(False); native.prof_reset()
        print(f'[{n},{c},{r},{r}] bf16: forward kernel {kf["ms"] / kf["launches"] * 1e3:6.1f} us = {kf["bytes"] / kf["ms"] / 1e9:5.2f} TB/s '
              f'(call {row[0]:6.1f} us vs {row[2]:6.1f} us for the four launches); backward kernel {kb["ms"] / kb["launches"] * 1e3:6.1f} us = '
              f'{kb["bytes"] / kb["ms"] / 1e9:5.2f} TB/s (call {row[1]:6.1f} us vs {row[3]:6.1f} us composed)', flush=True)


if __name__ == '__main__':
    main()
