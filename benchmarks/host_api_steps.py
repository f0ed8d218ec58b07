"""Reference-API path (BASELINE configs[0] shape): one host environment, NAFAgent.act + env.step + NAFAgent.step
(add, sample, learn) per timestep, batch 64 / buffer 1e5 — timesteps per second through the drop-in interface."""
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
def steps(n):
    global state
    for _ in range(n):
        a = agent.act(state)
        nxt, r, d = env.step(a)
        agent.step(state, a, r, nxt, d)
        state = env.reset(False) if d else nxt
steps(300)
torch.cuda.synchronize(); t0 = time.time(); steps(3000); torch.cuda.synchronize(); dt = time.time() - t0
print(f"host-API path: {3000/dt:.0f} timesteps/s ({dt/3000*1e6:.0f} us per act+env.step+add+sample+learn), optimizer steps {int(agent.learner.step_dev.item())}")
