#!/usr/bin/env python3
"""How far apart are two EAGER runs of the tiny aio loop of tests/test_train_step_gpu.py::test_recorded_phases_equal_the_eager_loop?
Builds the eager arm N times from the same seed and prints the pairwise distances of the gradients of the second iteration's
global_Dmain phase (what the test calls "first replayed gradients") and of the weights after six iterations."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
from test_train_step_gpu import _tiny_aio, mod       # noqa: E402


def main():
    n, ada = int(sys.argv[1]) if len(sys.argv) > 1 else 6, True
    AU, TS = mod('training.augment'), mod('torch_utils.training_stats')
    AU.fixed_margin, AU.side_stream_margin = True, False
    arms = []
    for i in range(n):
        torch.manual_seed(11)
        TS._counters.clear()
        t = _tiny_aio(False, ada)
        t.run_iteration(); t.run_iteration()
        torch.cuda.synchronize()
        g = [p.grad.detach().float().clone() for p in t.global_D.parameters()]
        for _ in range(4):
            t.run_iteration()
        torch.cuda.synchronize()
        w = [p.detach().float().clone() for m in [t.mapping_network, *t.local_G_list, *t.local_D_list, t.pos_estimator, t.global_D] for p in m.parameters()]
        arms.append((g, w))
        del t

    def dist(xs, ys):
        num = sum(float((a - b).square().sum()) for a, b in zip(xs, ys)) ** 0.5
        return num / max(sum(float(a.square().sum()) for a in xs) ** 0.5, 1e-30)
    for k, name in ((0, 'gradients of global_D, iteration 1'), (1, 'weights after 6 iterations')):
        print(name)
        for i in range(n):
            print('   ', ' '.join(f'{dist(arms[i][k], arms[j][k]):9.3g}' for j in range(n)))


if __name__ == '__main__':
    main()
