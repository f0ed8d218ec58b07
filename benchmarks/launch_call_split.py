"""What the per-timestep path's launch call (TrainChunk.run_row -> naf_host_publish_launch) is made of, on the host: the store of the
transition row into device memory + fence, and hipGraphLaunch of the seven-launch graph — each timed alone, with the GPU idle for a
given time before the call (does the runtime's completion handling of the previous graph get in the launch's way?).

    python benchmarks/launch_call_split.py [batch]
"""
import os, sys, time, tempfile
os.environ["NAF_STEP_PIPELINE"] = "0"      # (the seven-launch graph: this script launches the chunk's graph itself, timestep by timestep)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(tempfile.mkdtemp())
import logging
import numpy as np
import torch
from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
logging.getLogger('robotic_manipulator_rloa.utils.logger').setLevel(40)
env = SyntheticEnvironment(6)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
agent = NAFAgent(env, 21, 6, 256, B, 100000, 1e-3, 1e-3, 0.99, 1, 1, 500, torch.device("cuda:0"), 0)
state = env.reset(False)
for _ in range(4 * B + 60):
    a = agent.act(state)
    nxt, r, d = env.step(a)
    agent.step(state, a, r, nxt, d)
    state = env.reset(False) if d else nxt
ch = agent._chunk
assert ch._exec is not None and ch.head_dev is not None
lib = agent.learner.lib
st = torch.cuda.current_stream().cuda_stream
pc = time.perf_counter
print(f"B = {B}, prefetch {'on' if ch.spec_rec is not None else 'off'}; host microseconds, median of 400 calls")
for idle_us in (0, 5, 20, 100):
    t_pub, t_launch, t_seen = [], [], []
    for _ in range(400):
        ch._head_count_np[0] = 1
        ch._seq_prev = int(ch._seq_np[0])
        t0 = pc()
        lib.naf_host_publish(ch._head_dst, ch._head_src, ch._head_bytes)
        t1 = pc()
        lib.naf_host_publish_launch(None, None, 0, ch._exec, st)
        t2 = pc()
        ch._inflight = True
        ch.wait_tail()
        t3 = pc()
        t_pub.append(t1 - t0); t_launch.append(t2 - t1); t_seen.append(t3 - t2)
        while pc() - t3 < idle_us * 1e-6:
            pass
    med = lambda v: float(np.median(v)) * 1e6
    print(f"  {idle_us:4d} us after the previous action was seen: store + fence {med(t_pub):5.2f}   hipGraphLaunch {med(t_launch):5.2f}   "
          f"launch returned -> action seen {med(t_seen):5.2f}")
