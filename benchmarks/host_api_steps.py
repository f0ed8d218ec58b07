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


def prefill(agent, rows):
    """the steady state SURVEY.md section 8(d) asks for: the ring filled (here: to `rows` transitions of the stand-in env's value
    ranges) before anything is timed — the sampler's redraw rounds and the gather's locality are then those of a long run"""
    import numpy as np
    if rows <= 0:
        return
    m = agent.memory
    rng = np.random.default_rng(1)
    r = np.zeros((rows, m.row_floats), np.float32)
    r[:, :m.S] = rng.standard_normal((rows, m.S))
    r[:, m.S:m.S + m.A] = rng.uniform(-1, 1, (rows, m.A))
    r[:, m.S + m.A] = -rng.random(rows)
    r[:, m.off_s2:m.off_s2 + m.S] = r[:, :m.S] + 0.05 * rng.standard_normal((rows, m.S))
    m.add_rows_device(torch.from_numpy(r).cuda(), rows)
    torch.cuda.synchronize()


FILL = int(os.environ.get("NAF_BENCH_FILL", "100000"))
prefill(agent, FILL)
def steps(n):
    global state
    for _ in range(n):
        a = agent.act(state)
        nxt, r, d = env.step(a)
        agent.step(state, a, r, nxt, d)
        state = env.reset(False) if d else nxt
steps(max(300, 4 * BATCH + 60))      # (past the dense regime of the sampler: population >= 4 B)
torch.cuda.synchronize(); t0 = time.time(); steps(3000); torch.cuda.synchronize(); dt = time.time() - t0
print(f"host-API path: {3000/dt:.0f} timesteps/s ({dt/3000*1e6:.0f} us per act+env.step+add+sample+learn), optimizer steps {int(agent.learner.step_dev.item())}")
