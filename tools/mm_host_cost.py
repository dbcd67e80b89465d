"""Host time per call of the small GEMMs of the mapping network (fp32), in the forms autograd issues them and in alternatives."""
import time
import torch
dev = 'cuda:0'
x = torch.randn([64, 512], device=dev)
dy = torch.randn([64, 512], device=dev)
w = torch.randn([512, 512], device=dev)
dyt_c, xt_c = dy.t().contiguous(), x.t().contiguous()
out = torch.empty([512, 512], device=dev)
forms = {
    'fwd   x @ w.t()                      [64,512]x[512,512]': lambda: x.mm(w.t()),
    'dgrad dy @ w                         [64,512]x[512,512]': lambda: dy.mm(w),
    'wgrad dy.t() @ x   (autograd)        [512,64]x[64,512]': lambda: dy.t().mm(x),
    'wgrad (x.t() @ dy).t()': lambda: x.t().mm(dy).t(),
    'wgrad dy.t().contiguous() @ x': lambda: dy.t().contiguous().mm(x),
    'wgrad precontiguous dyT @ x': lambda: dyt_c.mm(x),
    'wgrad mm(out=)': lambda: torch.mm(dy.t(), x, out=out),
    'wgrad bmm batch 1': lambda: torch.bmm(dy.t().unsqueeze(0), x.unsqueeze(0)),
    'wgrad einsum': lambda: torch.einsum('no,ni->oi', dy, x),
    'wgrad addmm beta=0': lambda: torch.addmm(out, dy.t(), x, beta=0),
}
for name, fn in forms.items():
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name:62s} host {1e6 * (t1 - t0) / 500:7.1f} us/call   with drain {1e6 * (t2 - t0) / 500:7.1f} us/call', flush=True)
