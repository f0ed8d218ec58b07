"""Updates/s of learn() at layer sizes other than 256 (the reference takes any `layer_size`; its own agent test builds 128):
which chain each (layer_size, batch_size) pair runs and what an update costs there, graph-replayed chunks of 64 updates as
bench.py runs them. Usage (GPU box): python benchmarks/layer_sizes.py > gpurun_out/layer_sizes.txt"""
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
S, A, U, N = 21, 6, 64, 100_000
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 128, 192, 256, 384, 512]
batches = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 256, 1024]
print(f"S = {S}, A = {A}, chunks of {U} graph-replayed updates, ring of {N} rows")
for H in sizes:
    for B in batches:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            L = Learner(S, A, H, B, 1e-3, 1e-3, 0.99, dev)
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
        print(f"layer_size {H:4d}  batch {B:5d}  chain {L.chain:8s} fuse {','.join(sorted(L.fuse)):16s} {us:7.2f} us/update  "
              f"{1e6 / us:9.0f} updates/s", flush=True)
        del chunk, replay, L
