"""Data-parallel NAF training through the reference's own API — what a user of BASELINE configs[2]..[4] starts:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        examples/train_dp.py --robot kuka --batch 256 --envs 64 --episodes 2000 --frames 400

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI). Each rank owns `--envs` environments, its
own HBM replay shard and its own minibatch; the one exchange is a sum all-reduce of the flat gradient inside every
learn() (the one-shot peer-memory form inside a node, RCCL otherwise). The calls are the
reference's (rl_framework.py:431, :478): ManipulatorFramework.initialize_naf_agent -> run_training, with `n_envs`
switching run_training from NAFAgent.run (one env; `--envs 1`) to the many-env loop. Rank 0 writes
checkpoints/{episode}/weights.p, scores.txt and model.p exactly as a single-GPU run does; every rank leaves the loop
after the same vector step.

Without a launcher around it (`python examples/train_dp.py ...`) the same script trains on one GPU.
`NAF_DP_SHARE_GPU=1` rehearses a W-rank launch on ONE GPU (every rank on cuda:0, gloo control plane): functional only.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--robot", default="kuka", choices=["kuka", "xarm6", "xarm6_robot", "panda"])
    ap.add_argument("--environment", default="synthetic", choices=["synthetic", "pybullet"],
                    help="synthetic: the kinematic stand-in, on the device for --envs > 1; pybullet: DIRECT-mode workers")
    ap.add_argument("--envs", type=int, default=64, help="environments per GPU (1 = the reference's one-env loop)")
    ap.add_argument("--episodes", type=int, default=200)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--buffer", type=int, default=1000000)
    ap.add_argument("--obstacle-jitter", type=float, default=0.0, help="configs[3]: per-(rank, env) obstacle offset range")
    ap.add_argument("--checkpoint-frequency", type=int, default=500)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--test-episodes", type=int, default=0, help="test_trained_model on rank 0's weights afterwards")
    args = ap.parse_args()

    from robotic_manipulator_rloa_amd import parallel
    from robotic_manipulator_rloa_amd.presets import ROBOT_PRESETS, pybullet_arguments, synthetic_initial_joints
    from robotic_manipulator_rloa_amd.rl_framework import ManipulatorFramework

    rank, local_rank, world = parallel.init_distributed()
    torch.cuda.set_device(parallel.local_device())
    fw = ManipulatorFramework()
    fw.set_log_level(20 if rank == 0 else 40)
    fw.set_hyperparameter("batch_size", args.batch)
    fw.set_hyperparameter("buffer_size", args.buffer)
    preset = pybullet_arguments(args.robot)
    variation = list(ROBOT_PRESETS[args.robot]["training_variation"])
    if args.environment == "synthetic":
        n = len(preset["involved_joints"])
        fw.initialize_synthetic_environment(n, preset["target_position"], preset["obstacle_position"],
                                            synthetic_initial_joints(args.robot), variation[:n],
                                            obstacle_jitter=args.obstacle_jitter)
    else:
        import pybullet_data
        preset["manipulator_file"] = os.path.join(pybullet_data.getDataPath(), preset["manipulator_file"])
        fw.initialize_environment(initial_positions_variation_range=variation, visualize=False, **preset)
    # identical weights on every rank (same seed + a broadcast); env / noise / sampler streams are offset by the rank
    fw.initialize_naf_agent(checkpoint_frequency=args.checkpoint_frequency, seed=args.seed,
                            n_envs=args.envs if args.envs > 1 else None)
    agent = fw.naf_agent
    scores = fw.run_training(args.episodes, args.frames, verbose=False)
    torch.cuda.synchronize()
    L = agent.learner
    digest = float(L.theta2.double().sum().item())
    line = {"rank": rank, "world": world, "optimizer_steps": int(L.step_dev.item()), "theta_sum": digest,
            "replay_rows": len(agent.memory), "replay_rows_device": agent.memory.device_len(),
            "episodes_recorded": sum(1 for v in scores.values() if v != (0, 0)),
            "grad_exchange": "none" if world == 1 else ("one-shot peer memory" if L.xgmi is not None else "rccl"),
            "stats": {k: v for k, v in (agent.last_run_stats or {}).items() if k in ("env_steps", "updates", "seconds",
                                                                                     "env_steps_per_s")}}
    print("TRAIN_DP " + json.dumps(line), flush=True)
    if args.test_episodes and rank == 0:
        out = fw.test_trained_model(args.test_episodes, args.frames, n_envs=args.envs if args.envs > 1 else None)
        print("TEST " + json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        if L.xgmi is not None:
            L.xgmi.close()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
