"""Where the two fused launches of the per-timestep path (csrc/step_path.hip) spend their time: wall-clock marks the kernels leave
themselves (csrc/common.h NAF_TL), averaged over many timesteps of the reference-API loop (one host env, B given).

    NAF_BUILD_DEFINES=-DNAF_TIMELINE python benchmarks/step_timeline.py [batch] [reps]
"""
import ctypes as C
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(tempfile.mkdtemp())
import logging
import numpy as np
import torch
from robotic_manipulator_rloa_amd import _lib
from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent

if "NAF_TIMELINE" not in os.environ.get("NAF_BUILD_DEFINES", ""):
    raise SystemExit("build with NAF_BUILD_DEFINES=-DNAF_TIMELINE")
logging.getLogger('robotic_manipulator_rloa.utils.logger').setLevel(40)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
env = SyntheticEnvironment(6)
agent = NAFAgent(env, 21, 6, 256, B, 100000, 1e-3, 1e-3, 0.99, 1, 1, 500, torch.device("cuda:0"), 0)
lib = _lib.load()
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


def step():
    global state
    a = agent.act(state)
    nxt, r, d = env.step(a)
    agent.step(state, a, r, nxt, d)
    state = env.reset(False) if d else nxt


for _ in range(4 * B + 60):
    step()
ch = agent._chunk
PIPELINED = bool(getattr(ch, "pipelined", False))
KIDS = {"bb_layer1": 0, "bb_linear_stats": 1, "bb_layer2_head": 2, "gemm_bundle": 4, "finish": 5, "step_prep": 7, "adam_act": 8,
        "l1_riders": 2048}      # (layer 1 riding on adam_act: slots 0 - 6 as bb_layer1's, 13 = entry, 14 = the step's flags seen)
acc = {k: np.zeros((2, 16)) for k in KIDS}
out = (C.c_longlong * 32)()
n_used = 0
for _ in range(REPS):
    fast_before = ch.fast_runs if PIPELINED else 0
    step()
    torch.cuda.synchronize()
    if PIPELINED and ch.fast_runs == fast_before:
        continue                                   # (a timestep that started over: the other graph, not the one shown)
    n_used += 1
    raw = {}
    for k, kid in KIDS.items():
        assert lib.naf_timeline_read(kid, out) == 0
        raw[k] = np.array(out[:], dtype=np.int64).reshape(2, 16)
    # pipelined: the graph starts with adam_act (its first workgroup's entry is the origin) and the chain follows it
    t0 = raw["adam_act"][0, 0] if PIPELINED else raw["step_prep"][0, 0]
    for k in KIDS:
        v = (raw[k] - t0) / 100.0
        v[raw[k] == 0] = np.nan
        acc[k] += np.nan_to_num(v, nan=0.0)
REPS = max(1, n_used)
if PIPELINED:
    print(f"B = {B}, pipelined graph (adam_act: the waiting gradient's step + act() + commit, then the chain) and, beside it on a stream "
          f"of its own, the prefetch launch (step_prep rows: append + depth-2 prefetch): microseconds since adam_act's entry, mean of "
          f"{REPS} timesteps (first workgroup | last workgroup; 0 = no mark)")
    print("step_prep rows (one workgroup: both rows are it): slots 0 - 6 main phases, 7 - 12 moments, 13 - 15 draw")
    order = ("adam_act", "l1_riders", "step_prep", "bb_layer1", "bb_linear_stats", "bb_layer2_head", "gemm_bundle", "finish")
else:
    print(f"B = {B}: microseconds since step_prep's entry, mean of {REPS} timesteps (first workgroup | last workgroup; 0 = no mark)")
    order = ("step_prep", "bb_layer1", "bb_linear_stats", "bb_layer2_head", "gemm_bundle", "finish", "adam_act")
for k in order:
    m = acc[k] / REPS
    print(f"{k:16s} first: " + " ".join(f"{x:6.2f}" for x in m[0]))
    print(f"{'':16s} last:  " + " ".join(f"{x:6.2f}" for x in m[1]))
