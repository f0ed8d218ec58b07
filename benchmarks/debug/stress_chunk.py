"""Is the chunked path (graphs of 64 updates, the headline's) the same bits every time it runs from the same state — also free-running,
graph behind graph without a host synchronisation in between, at the batch sizes whose kernels hand statistics between workgroups of
ONE launch (B > 512: folded records with a fall-back after 20 us)?   python benchmarks/debug/stress_chunk.py [repeats]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(tempfile.mkdtemp())
import numpy as np, torch
from bench import synth_rows
from robotic_manipulator_rloa_amd.engine import DeviceEnvLoop, TrainChunk
from robotic_manipulator_rloa_amd.learner import Learner
from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
dev = torch.device("cuda")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for (S, A, B, N, robot, K) in ((21, 6, 256, 200000, "kuka", 30), (21, 6, 1024, 200000, "kuka", 15), (23, 7, 2048, 400000, "panda", 10), (21, 6, 100, 100000, "kuka", 20)):
    L = Learner(S, A, 256, B, 1e-3, 1e-3, 0.99, dev)
    sd = reference_init_state_dict(S, A, 256, seed=0)
    replay = ReplayBuffer(N, B, dev, seed=1000, state_size=S, action_size=A)
    replay.add_rows_device(synth_rows(N, S, A, replay.row_floats, replay.off_s2, 77, dev), N)
    loop = DeviceEnvLoop(L, replay, 64, seed=31, max_frames=400, robot=robot)
    chunk = TrainChunk(L, replay, 64)
    L.load_params(0, sd); L.load_params(1, sd)
    loop.capture(); chunk.capture()
    torch.cuda.synchronize()
    saved = [t.clone() for t in (L.theta2, L.grad, L.adam_m, L.adam_v, L.bn_stats, L.step_dev, L.partials, replay.rows, replay.meta,
                                 replay._sample_ctr, loop.env_state, loop.step_ctr, loop.actor.obs, loop.actor.counter)]
    live = (L.theta2, L.grad, L.adam_m, L.adam_v, L.bn_stats, L.step_dev, L.partials, replay.rows, replay.meta, replay._sample_ctr,
            loop.env_state, loop.step_ctr, loop.actor.obs, loop.actor.counter)
    ref, bad = None, 0
    for rep in range(REPS):
        for t, s_ in zip(live, saved):
            t.copy_(s_)
        torch.cuda.synchronize()
        for _ in range(K):                      # free-running: K vector steps (64 env steps + 64 updates each), no sync in between
            loop.step()
            chunk.run()
        torch.cuda.synchronize()
        out = (L.theta2.clone(), L.bn_stats.clone(), replay.meta.clone(), L.adam_v.clone())
        if ref is None:
            ref = out
        elif not all(torch.equal(a, b) for a, b in zip(ref, out)):
            bad += 1
            print(f"  B = {B}: repeat {rep} differs from repeat 0", flush=True)
    print(f"B = {B} ({robot}): {REPS} repeats of {K} vector steps ({K * 64} updates) from one state, {bad} differ; fold fall-backs {L.fold_fallbacks}", flush=True)
    del loop, chunk, replay, L
