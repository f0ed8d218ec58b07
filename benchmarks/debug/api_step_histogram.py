"""Where the per-timestep path's run-to-run spread comes from: host time stamps of every timestep of one long run of
benchmarks/host_api_steps.py's loop — the mean per block of 1000 timesteps, percentiles, and what the pipeline counted.
usage: api_step_histogram.py [batch] [timesteps]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.chdir(tempfile.mkdtemp())
import logging
import numpy as np
import torch
from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
logging.getLogger('robotic_manipulator_rloa.utils.logger').setLevel(40)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
env = SyntheticEnvironment(6)
agent = NAFAgent(env, 21, 6, 256, BATCH, 100000, 1e-3, 1e-3, 0.99, 1, 1, 500, torch.device("cuda:0"), 0)
m = agent.memory
rng = np.random.default_rng(1)
r = np.zeros((100000, m.row_floats), np.float32)
r[:, :m.S] = rng.standard_normal((100000, m.S))
r[:, m.S:m.S + m.A] = rng.uniform(-1, 1, (100000, m.A))
r[:, m.S + m.A] = -rng.random(100000)
r[:, m.off_s2:m.off_s2 + m.S] = r[:, :m.S]
m.add_rows_device(torch.from_numpy(r).cuda(), 100000)
state = env.reset(False)
# ORDER=auto (the product: the prefetch goes first on a tick whose verdict the host had to wait for) | graph | prefetch (always that one first)
ORDER = os.environ.get("ORDER", "auto")
ENV_US = float(os.environ.get("ENV_US", "0"))
def force_order():
    pipe = getattr(agent._chunk, "pipe", None)
    if pipe is None or ORDER == "auto":
        return
    inner = pipe.collect
    def collect():
        inner()
        pipe.waited = ORDER == "prefetch"
    pipe.collect = collect
ts = np.zeros(N + 1)
ph = np.zeros((N, 3))          # host time inside act() (waits for the action), env.step(), step() (publishes the row, launches)
def steps(n, rec):
    global state
    pc = time.perf_counter
    for i in range(n):
        t0 = pc()
        a = agent.act(state)
        t1 = pc()
        nxt, rw, d = env.step(a)
        while ENV_US and (pc() - t1) * 1e6 < ENV_US:      # (a slower environment: busy waiting, as bench.py's env_100us leg)
            pass
        t2 = pc()
        agent.step(state, a, rw, nxt, d)
        state = env.reset(False) if d else nxt
        t3 = pc()
        if rec:
            ts[i + 1] = t3
            ph[i] = (t1 - t0, t2 - t1, t3 - t2)
steps(4 * BATCH + 60, False)
force_order()
torch.cuda.synchronize()
ts[0] = time.perf_counter()
steps(N, True)
torch.cuda.synchronize()
d = np.diff(ts) * 1e6
print(f"batch {BATCH}: {N / (ts[-1] - ts[0]):.0f} timesteps/s; per timestep us: mean {d.mean():.1f} p10 {np.percentile(d, 10):.1f} p50 {np.percentile(d, 50):.1f} "
      f"p90 {np.percentile(d, 90):.1f} p99 {np.percentile(d, 99):.1f} max {d.max():.0f}")
print("mean us per block of 1000:", " ".join(f"{d[i:i + 1000].mean():.1f}" for i in range(0, N, 1000)))
print("act() / env.step() / step() us per block of 1000:", " ".join(f"{ph[i:i + 1000, 0].mean() * 1e6:.1f}/{ph[i:i + 1000, 1].mean() * 1e6:.1f}/{ph[i:i + 1000, 2].mean() * 1e6:.1f}"
                                                                      for i in range(0, N, 1000)))
try:
    print("cpu of this thread at the end:", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?", "affinity:", len(os.sched_getaffinity(0)), "cpus")
except Exception as e:
    print("cpu query failed:", e)
print("timesteps over 60 us:", int((d > 60).sum()), "over 200 us:", int((d > 200).sum()))
ch = agent._chunk
print("pipeline:", getattr(ch, "prefetch_stats", None) and ch.prefetch_stats(), "order", ORDER, "ticks with the prefetch first:",
      getattr(getattr(ch, "pipe", None), "side_first_runs", None))
