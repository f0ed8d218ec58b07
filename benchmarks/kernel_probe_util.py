import torch


def timeit(fn, n=200):
    """µs per call of fn replayed n times back to back inside one hipGraph."""
    for _ in range(5):
        rc = fn()
        assert not isinstance(rc, int) or rc == 0, f"launch refused with status {rc}"   # never time a refused launch
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
