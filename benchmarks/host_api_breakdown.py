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
DELAY = float(os.environ.get("NAF_BENCH_ENV_DELAY_US", "0")) * 1e-6
prefill(agent, FILL)
T = {"act": 0.0, "env.step": 0.0, "agent.step": 0.0}
def steps(n, timed):
    global state
    pc = time.perf_counter
    for _ in range(n):
        t0 = pc(); a = agent.act(state)
        t1 = pc(); nxt, r, d = env.step(a)
        if DELAY:                                        # (a slower environment: NAF_BENCH_ENV_DELAY_US of busy waiting per step)
            while pc() - t1 < DELAY:
                pass
        t2 = pc(); agent.step(state, a, r, nxt, d)      # the product's own path: the row is appended by the update's graph
        t3 = pc()
        if timed:
            T["act"] += t1 - t0; T["env.step"] += t2 - t1; T["agent.step"] += t3 - t2
        state = env.reset(False) if d else nxt
steps(max(300, 4 * BATCH + 60), False)      # (past the dense regime of the sampler: population >= 4 B)
# inside agent.step(): the launch call alone (TrainChunk.run_row: the row into device memory + hipGraphLaunch)
_ch = agent._chunk
if _ch is not None and _ch._seq_np is not None:
    _rr = _ch.run_row
    T["  step: launch"] = 0.0
    def _timed_run_row():
        t0 = time.perf_counter(); _rr(); T["  step: launch"] += time.perf_counter() - t0
    _ch.run_row = _timed_run_row
    _wt = _ch.wait_tail
    T["  act: wait"] = 0.0
    def _timed_wait():
        t0 = time.perf_counter(); _wt(); T["  act: wait"] += time.perf_counter() - t0
    _ch.wait_tail = _timed_wait
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
if getattr(ch, "pipelined", False):
    print(f"  the start-over graph (draw + chain + act + prefetch + chain: 12 launches, TWO chains) back to back: {dt/2000*1e6:.1f} us; "
          f"timesteps above: {ch.fast_runs} on the prefetched minibatch (6 launches), {ch.slow_runs - 2000} started over")
else:
    print(f"  chunk (flush-less) back to back: {dt/2000*1e6:.1f} us per update incl. the riding act()")
