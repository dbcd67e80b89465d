import time, torch
DEV = torch.device('cuda', 0)
x = torch.zeros([64], device=DEV)
big = torch.zeros([64 * 1024 * 1024], device=DEV)
for n_nodes in (2000,):
    for name, body in (('tiny kernels', lambda: x.add_(1)), ('50us kernels', lambda: big.add_(1))):
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n_nodes):
                body()
        torch.cuda.synchronize()
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            g.replay()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f'{name}: {n_nodes} nodes: host {t_host / reps / n_nodes * 1e6:.2f} us/node, wall {t_all / reps / n_nodes * 1e6:.2f} us/node')
        # eager for comparison
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_nodes):
            body()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f'{name}: eager: host {t_host / n_nodes * 1e6:.2f} us/launch, wall {t_all / n_nodes * 1e6:.2f} us/launch')
