import torch, time
x = torch.zeros(64, device="cuda")
big = torch.zeros(64*1024*1024, device="cuda")
def chain(n, t):
    for _ in range(n): t.add_(1.0)
for name, t, n in (("tiny", x, 400), ("256MB", big, 40)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(10, t)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain(n, t)
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(name, "nodes", n, "us per node %.2f" % (dt / n * 1e6))
