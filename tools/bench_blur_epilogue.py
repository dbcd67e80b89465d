"""HIP-event timings of the fused blur + epilogue pass (mg_upfirdn2d_epilogue) against blur then epilogue, at the step's shapes."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
importlib.import_module(PKG)
E = importlib.import_module(f'{PKG}.torch_utils.ops.modconv_epilogue')
U = importlib.import_module(f'{PKG}.torch_utils.ops.upfirdn2d')
DEV = torch.device('cuda:0')
cl = torch.channels_last


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


f = U.setup_filter([1, 3, 3, 1]).to(DEV)
for (c, r) in [(64, 256), (128, 128), (256, 64), (512, 32)]:
    n = 32
    x = torch.randn([n, c, r + 3, r + 3], device=DEV).to(torch.bfloat16).contiguous(memory_format=cl)
    d = torch.rand([n, c], device=DEV) + 0.5
    nz = torch.randn([r, r], device=DEV)
    st = torch.ones([], device=DEV)
    b = torch.randn([c], device=DEV).to(torch.bfloat16)
    pad = [0, 0, 0, 0]
    with torch.no_grad():
        for on in (True, False):
            E.blur_fusion = on
            us = timeit(lambda: E.blur_epilogue(x, f, pad, fgain=4.0, dcoefs=d, noise=nz, b=b, act='lrelu', clamp=256, noise_strength=st))
            nbytes = (x.numel() + n * c * r * r) * 2
            print(f'[{n},{c},{r + 3}]->[{r}] fused={on}: {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s of in+out', flush=True)
        us = timeit(lambda: U.upfirdn2d(x, f, padding=pad, gain=4.0))
        print(f'[{n},{c},{r + 3}]->[{r}] blur alone: {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s', flush=True)
