import torch, gc
DEV = torch.device('cuda', 0)
s = torch.cuda.Stream(device=DEV)
x = torch.zeros([1000], device=DEV)
w = torch.randn([1000], device=DEV, requires_grad=True)
opt = torch.optim.Adam([w], lr=0.1, fused=True, capturable=True)

def fn():
    w.grad = None
    (w * 2).sum().backward()
    x.add_(w.grad)
    opt.step()

for mode in ('outer_default', 'outer_same'):
    x.zero_()
    outer = torch.cuda.current_stream()
    if mode == 'outer_same':
        s.wait_stream(outer)
        ctx = torch.cuda.stream(s)
        ctx.__enter__()
    fn(); torch.cuda.synchronize()
    print(mode, 'eager x[0]', float(x[0]), 'w[0]', float(w[0]))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    for i in range(3):
        g.replay(); torch.cuda.synchronize()
        print(mode, 'replay', i, 'x[0]', float(x[0]), 'w[0]', float(w[0]), 'current==s', torch.cuda.current_stream() == s)
    if mode == 'outer_same':
        ctx.__exit__(None, None, None)
