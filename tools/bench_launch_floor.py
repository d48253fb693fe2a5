import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from padt_amd import ops
x = torch.zeros(16, 2048, device="cuda", dtype=torch.bfloat16); y = torch.zeros_like(x)
def tiny(): ops.pack_rows(x, y, 16, to_packed=True)
def timeit_graph(fn, n):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("tiny dependent kernel in a graph chain: %.2f us per launch (200 launches)" % timeit_graph(tiny, 200))
print("tiny dependent kernel in a graph chain: %.2f us per launch (1000 launches)" % timeit_graph(tiny, 1000))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(10): tiny()
torch.cuda.synchronize(); e0.record()
for _ in range(1000): tiny()
e1.record(); torch.cuda.synchronize()
print("eager stream: %.2f us per launch" % (e0.elapsed_time(e1)))
