"""Does the drop-in LEARN? Train E synthetic arms on the device with NAFAgent.run_vectorized and print the mean episode
score per block of finished episodes (completion order). Not a parity check (tests/ do that against the reference's
goldens): a behavioural sanity check of the whole loop — env kernel, HBM ring, sampler, learn() chain, episode ledger."""
import argparse, json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=3200)
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--evaluate", type=int, default=128, help="episodes of test_trained_model (no noise) behind the training")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    a.out = os.path.abspath(a.out) if a.out else None
    os.chdir(tempfile.mkdtemp())
    from robotic_manipulator_rloa_amd import ManipulatorFramework
    f = ManipulatorFramework()
    f.set_hyperparameter("batch_size", a.batch)
    f.set_hyperparameter("buffer_size", 1_000_000)
    f.initialize_synthetic_environment(n_joints=6)
    f.initialize_naf_agent(checkpoint_frequency=10 ** 9, seed=a.seed, n_envs=a.envs)
    scores = f.run_training(a.episodes, a.frames, verbose=False)
    sc = np.array([scores[k][0] for k in sorted(scores)], dtype=np.float64)
    fr = np.array([scores[k][1] for k in sorted(scores)], dtype=np.float64)
    n = len(sc) // a.blocks
    rows = [{"episodes": f"{i * n + 1}-{(i + 1) * n}", "mean_score": round(float(sc[i * n:(i + 1) * n].mean()), 2),
             "mean_frames": round(float(fr[i * n:(i + 1) * n].mean()), 1),
             "reached": int((fr[i * n:(i + 1) * n] < a.frames).sum())} for i in range(a.blocks)]
    evaluation = f.test_trained_model(a.evaluate, a.frames) if a.evaluate else None     # (the agent's n_envs: batched act, no noise)
    out = {"episodes": len(sc), "evaluation_without_noise": evaluation, "envs": a.envs, "frames": a.frames, "batch": a.batch, "blocks": rows,
           "stats": {k: v for k, v in (f.naf_agent.last_run_stats or {}).items() if k not in ("scores", "checkpoints")}}
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
