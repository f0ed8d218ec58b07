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
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 64
JOINTS = int(sys.argv[2]) if len(sys.argv) > 2 else 6          # (the reference's state: 9 + 2 A floats, environment.py:261)
LAYER = int(sys.argv[3]) if len(sys.argv) > 3 else 256


class ScriptedEnvironment:
    """more joints than the kinematic stand-in models (8): a table of transitions replayed in order — the protocol of the reference's
    Environment, about a microsecond of host time per step, no dynamics (labelled in the output)"""
    def __init__(self, A, n=4096):
        import numpy as np
        rng = np.random.default_rng(3)
        self.S, self.A, self.t = 9 + 2 * A, A, 0
        self.states = rng.standard_normal((n, self.S))
        self.rewards = -rng.random(n)
        self.observation_space, self.action_space = np.zeros(self.S), np.zeros(A)

    def reset(self, verbose=False):
        return self.states[self.t % len(self.states)]

    def step(self, action):
        self.t += 1
        return self.states[self.t % len(self.states)], float(self.rewards[self.t % len(self.rewards)]), 0


env = SyntheticEnvironment(JOINTS) if JOINTS <= 8 else ScriptedEnvironment(JOINTS)
agent = NAFAgent(env, 9 + 2 * JOINTS, JOINTS, LAYER, BATCH, 100000, 1e-3, 1e-3, 0.99, 1, 1, 500, torch.device("cuda:0"), 0)
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
ch = agent._chunk
print(f"joints {JOINTS} ({type(env).__name__}) layer_size {LAYER} batch {BATCH} chain {agent.learner.chain} form "
      f"{'pipelined' if getattr(ch, 'pipelined', False) else ('fused' if getattr(ch, 'fused_prep', False) or getattr(ch, 'fused_tail', False) else 'separate')}: ", end="")
print(f"host-API path: {3000/dt:.0f} timesteps/s ({dt/3000*1e6:.0f} us per act+env.step+add+sample+learn), optimizer steps {int(agent.learner.step_dev.item())}")
