"""What a dependent kernel node costs inside a replayed HIP graph on this box: N trivial launches in a chain, per-node time.  python tools/micro/graph_node_floor.py"""
import torch
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
for n in (20, 100):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): x.add_(1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n): x.add_(1.0)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): g.replay()
    e1.record(); torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 50 * 1e3
    print("graph of %d dependent 64-element adds: %.1f us per replay = %.2f us per node" % (n, per, per / n), flush=True)
