import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from robotic_manipulator_rloa_amd import parallel
dev = torch.device("cuda", 0)
n = 81152
for W in (2, 4, 6, 8):
    streams = [torch.cuda.Stream() for _ in range(W)]
    comms = parallel.XgmiAllReduce.local_group(W, n, dev, timeout_s=1.0)
    ins = [torch.full((n,), float(r + 1), device=dev) for r in range(W)]
    outs = [torch.empty(n, device=dev) for _ in range(W)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(3):
        for r in range(W):
            with torch.cuda.stream(streams[r]):
                comms[r].all_reduce(ins[r], outs[r])
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    want = float(W * (W + 1) // 2)
    print(W, "ok" if all(bool((o == want).all()) for o in outs) else "BAD", [c.status() for c in comms], f"{dt:.2f}s", flush=True)
    for c in comms:
        c.close(collective=False)
