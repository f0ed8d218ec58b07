"""Updates/s of learn() by the arm's joint count (the reference builds its head for ANY action size, naf_neural_network.py:53-54; its
state is 9 + 2 A floats, environment.py:261): which chain each (A, batch_size) pair runs and what an update costs there, graph-replayed
chunks of 64 updates as bench.py runs them; 9 .. 11 joints also on the unfused chain they ran until round 6 (14 launches per update).
Usage (GPU box): python benchmarks/joint_counts.py [joints] [batches] > gpurun_out/joint_counts.txt"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synth_rows
from robotic_manipulator_rloa_amd.engine import TrainChunk
from robotic_manipulator_rloa_amd.learner import Learner
from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer

dev = torch.device("cuda")
U, N, H = 64, 100_000, 256
joints = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [6, 8, 9, 10, 11, 12]
batches = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 256, 1024, 2048]
print(f"H = {H}, S = 9 + 2 A, chunks of {U} graph-replayed updates, ring of {N} rows")
for A in joints:
    S = 9 + 2 * A
    for B in batches:
        for fuse in ((None, "unfused") if 9 <= A <= 11 else (None,)):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                L = Learner(S, A, H, B, 1e-3, 1e-3, 0.99, dev, fuse=fuse)
            sd = reference_init_state_dict(S, A, H, seed=0)
            L.load_params(0, sd)
            L.load_params(1, sd)
            replay = ReplayBuffer(N, B, dev, seed=1000, state_size=S, action_size=A)
            replay.add_rows_device(synth_rows(N, S, A, replay.row_floats, replay.off_s2, 77, dev), N)
            chunk = TrainChunk(L, replay, U, gather_outside_graph=True)
            chunk.capture()
            for _ in range(5):
                chunk.run()
            torch.cuda.synchronize()
            reps = 40
            t0 = time.perf_counter()
            for _ in range(reps):
                chunk.run()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / (reps * U) * 1e6
            print(f"joints {A:3d}  state {S:3d}  batch {B:5d}  chain {L.chain:8s} {us:7.2f} us/update  {1e6 / us:9.0f} updates/s", flush=True)
            del chunk, replay, L
