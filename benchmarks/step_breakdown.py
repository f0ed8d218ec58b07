"""What one bench.py step (64-env vector step + 64 updates) is made of: the env-loop graph, sample + gather, and the
graph of 64 updates, each timed alone (back to back, device time by events) and together."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synth_rows
from robotic_manipulator_rloa_amd.engine import DeviceEnvLoop, TrainChunk
from robotic_manipulator_rloa_amd.learner import Learner
from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer

dev = torch.device("cuda")
S, A, H, B, E, N = 21, 6, 256, 256, 64, 1_000_000
L = Learner(S, A, H, B, 1e-3, 1e-3, 0.99, dev)
sd = reference_init_state_dict(S, A, H, seed=0)
L.load_params(0, sd)
L.load_params(1, sd)
replay = ReplayBuffer(N, B, dev, seed=1000, state_size=S, action_size=A)
replay.add_rows_device(synth_rows(N, S, A, replay.row_floats, replay.off_s2, 77, dev), N)
loop = DeviceEnvLoop(L, replay, E, seed=31, max_frames=400)
chunk_out = TrainChunk(L, replay, E, gather_outside_graph=True)       # as bench.py runs it
loop.capture()
chunk_out.capture()


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def both():
    loop.step()
    chunk_out.run()


print(f"env-loop graph (act + env step + append)        {timed(loop.step):8.1f} us")
print(f"sample + gather, eager                           {timed(chunk_out._sample_gather):8.1f} us")
print(f"graph of 64 updates                              {timed(chunk_out.graph.replay):8.1f} us")
print(f"chunk.run() = sample + gather + graph            {timed(chunk_out.run):8.1f} us")
print(f"one bench step = env loop + chunk.run()          {timed(both):8.1f} us")
chunk_in = TrainChunk(L, replay, E, gather_outside_graph=False)
chunk_in.capture()
print(f"chunk with sample + gather INSIDE the graph      {timed(chunk_in.run):8.1f} us")
