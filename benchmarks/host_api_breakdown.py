"""Where a timestep of the reference-API path goes (one host env, NAFAgent.act / env.step / NAFAgent.step, B=64, N=1e5):
wall-clock per phase over 2000 timesteps, host side, with the phases separated by timers only (no extra synchronisation)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(tempfile.mkdtemp())
import logging
import torch
from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
logging.getLogger('robotic_manipulator_rloa.utils.logger').setLevel(40)
env = SyntheticEnvironment(6)
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 64
agent = NAFAgent(env, 21, 6, 256, BATCH, 100000, 1e-3, 1e-3, 0.99, 1, 1, 500, torch.device("cuda:0"), 0)
state = env.reset(False)
T = {"act": 0.0, "env.step": 0.0, "agent.step": 0.0}
def steps(n, timed):
    global state
    pc = time.perf_counter
    for _ in range(n):
        t0 = pc(); a = agent.act(state)
        t1 = pc(); nxt, r, d = env.step(a)
        t2 = pc(); agent.step(state, a, r, nxt, d)      # the product's own path: the row is appended by the update's graph
        t3 = pc()
        if timed:
            T["act"] += t1 - t0; T["env.step"] += t2 - t1; T["agent.step"] += t3 - t2
        state = env.reset(False) if d else nxt
steps(300, False)
torch.cuda.synchronize(); t0 = time.perf_counter(); steps(2000, True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{2000/dt:.0f} timesteps/s, {dt/2000*1e6:.1f} us per timestep")
for k, v in T.items():
    print(f"  {k:12s} {v/2000*1e6:7.1f} us")
# the update alone, back to back (what the GPU needs per timestep once nothing waits for the host)
ch = agent._chunk
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000):
    ch.run()          # head_rows = 0: the graph's append node finds a zero row count and appends nothing
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"  chunk (flush-less) back to back: {dt/2000*1e6:.1f} us per update incl. the riding act()")
