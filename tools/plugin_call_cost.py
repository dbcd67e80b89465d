#!/usr/bin/env python3
"""Host time per plugin call: the pybind11 / ATen shim (csrc/torch_plugin.cpp, what get_plugin returns) against the pure-ctypes
binding of the same C ABI (MONTAGE_PLUGIN=ctypes; round 1's boundary).  10 000 calls of each op on a tensor small enough that the
device is never the bottleneck; the queue is drained every 500 calls so that back-pressure does not enter the figure.
usage: plugin_call_cost.py            (runs both bindings in two child processes and prints one table)"""
import importlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import torch
    sys.path.insert(0, ROOT)
    pkg = 'docker-montage-gan_amd'
    importlib.import_module(pkg)
    co = importlib.import_module(f'{pkg}.torch_utils.custom_ops')
    co.verbosity = 'none'
    up, ba = co.get_plugin('upfirdn2d_plugin'), co.get_plugin('bias_act_plugin')
    dev = torch.device('cuda', 0)
    x = torch.randn([2, 8, 16, 16], device=dev)
    f = torch.ones([4, 4], device=dev) / 16
    b = torch.randn([8], device=dev)
    e = torch.empty([0], device=dev)
    calls = {'upfirdn2d': lambda: up.upfirdn2d(x, f, 1, 1, 1, 1, 2, 1, 2, 1, False, 1.0),
             'bias_act': lambda: ba.bias_act(x, b, e, e, e, 0, 1, 3, 0.2, 1.41421356, -1.0)}
    out = {}
    for name, fn in calls.items():
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        t = 0.0
        for _ in range(20):
            t0 = time.perf_counter()
            for _ in range(500):
                fn()
            t += time.perf_counter() - t0
            torch.cuda.synchronize()
        out[name] = t / 10000 * 1e6
    print('RESULT', os.environ.get('MONTAGE_PLUGIN', 'torch'), ' '.join(f'{k}={v:.2f}' for k, v in out.items()), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
        sys.exit(0)
    rows = []
    for binding in ('ctypes', 'torch'):
        env = dict(os.environ, MONTAGE_PLUGIN=binding)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:])
            sys.exit(1)
        rows.append(line[0])
    print('host microseconds per call (10 000 calls, [2,8,16,16] float32):')
    for ln in rows:
        _, binding, *vals = ln.split()
        print(f'  {"pybind11 / ATen shim" if binding == "torch" else "ctypes binding (round 1)":28s} ' + '  '.join(vals))
