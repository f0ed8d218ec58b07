"""Worker of tests/test_agent_gpu.py::test_dp_run_loop_two_ranks_* — W processes (torch.distributed.run) that all use
cuda:0 (NAF_DP_SHARE_GPU=1: gloo control plane, the peer-memory gradient exchange) and drive NAFAgent.run() — the
reference's training loop (naf_algorithm.py:228-292) — under data parallel with environments whose episodes END EARLY
and at different frames on every rank (tests/scripted_env.py: 3..6 steps of a budget of 8), so that run() pads with idle
ticks. Checked on every rank: the replay ring holds exactly the transitions its own env produced, each once and in order
(an idle tick must not re-append the last row); the scores are the scripted ones; every rank took the same
number of optimizer steps and holds bit-identical parameters; no wait on a peer timed out; only rank 0 wrote files.

Prints DP_LOOP_OK_<rank>; on success.
"""
import os
import sys

ROOT = os.environ.get("NAF_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from robotic_manipulator_rloa_amd import parallel  # noqa: E402
from scripted_env import ScriptedEnvironment  # noqa: E402


def main():
    os.environ["NAF_DP_SHARE_GPU"] = "1"
    rank, _, world = parallel.init_distributed()
    assert world > 1 and dist.get_backend() == "gloo"
    dev = parallel.local_device()
    os.makedirs(f"rank{rank}", exist_ok=True)
    os.chdir(f"rank{rank}")                                    # who writes what is visible per rank
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    env = ScriptedEnvironment(offset=rank, scale=int(os.environ.get("NAF_TEST_DP_LEN_SCALE", "1")))
    frames, episodes, B = int(os.environ.get("NAF_TEST_DP_FRAMES", "8")), int(os.environ.get("NAF_TEST_DP_EPISODES", "14")), \
        int(os.environ.get("NAF_TEST_DP_BATCH", "8"))
    # NAF_XGMI=0 puts the gradient on torch.distributed's all-reduce — gloo in this rehearsal, which (unlike RCCL) cannot be
    # captured into a graph: the same launches run eagerly there
    agent = NAFAgent(env, env.S, env.A, 256, B, 1000, 1e-3, 1e-3, 0.99, 1, 1, 5, dev, 0,
                     use_graph=os.environ.get("NAF_XGMI", "1") != "0")
    assert agent.world_size == world and agent.rank == rank
    scores = agent.run(frames, episodes, verbose=False)
    torch.cuda.synchronize()
    # round 6: the PIPELINED per-timestep graph under data parallel (the ranks vote per tick on which graph runs: an idle tick or a
    # prefetch that did not hold on ANY rank starts the tick over on ALL of them); with the collective through the host (NAF_XGMI=0:
    # gloo cannot be captured) the launches run eagerly and nothing is pipelined
    ch = agent._chunk
    want_pipe = (os.environ.get("NAF_XGMI", "1") != "0" and os.environ.get("NAF_STEP_FORM", "pipelined") == "pipelined" and
                 B >= 16)                                     # (below 16 rows the column-tile chain runs: no fused launches to pipeline)
    assert ch.pipelined == want_pipe, (ch.form, want_pipe)
    if ch.pipelined:
        runs = [None] * world
        dist.all_gather_object(runs, (ch.fast_runs, ch.slow_runs))
        assert len(set(runs)) == 1, f"the ranks ran different graphs: {runs}"
        assert ch.slow_runs >= episodes and ch.error_words() == {"act_poll_timeouts": 0, "pipe_errors": 0, "verdict_waits_synchronised": 0}
        min_fast = int(os.environ.get("NAF_TEST_MIN_FAST", "0"))
        assert ch.fast_runs >= min_fast, (ch.fast_runs, ch.slow_runs)
        print(f"DP_PIPE rank {rank}: {ch.fast_runs} ticks on the six-launch graph, {ch.slow_runs} started over", flush=True)

    # ---- the scores dict is run()'s: {episode: (score, last frame)} of THIS rank's env -----------------------------------
    want_rows = []
    for k in range(episodes):
        sc, L = env.expected(k)
        assert scores[k + 1] == (sc, L), (k, scores[k + 1], (sc, L))
        want_rows += [(rank, k, t) for t in range(1, L + 1)]

    # ---- the ring: every transition once, in order; host and device agree on the fill --------------------------------------
    m = agent.memory
    assert m.device_len() == len(m) == len(want_rows), (m.device_len(), len(m), len(want_rows))
    got = m.rows[:len(want_rows), m.off_s2:m.off_s2 + 3].cpu().numpy()             # next_state = (offset, k, t)
    np.testing.assert_array_equal(got, np.array(want_rows, dtype=np.float32))
    assert (m.rows[len(want_rows):] == 0).all()

    # ---- lock-step: same number of optimizer steps, bit-identical parameters and Adam state on every rank ------------------
    L_ = agent.learner
    steps = int(L_.step_dev.item())
    ticks = episodes * frames
    assert steps == ticks - B, (steps, ticks, B)              # the gate is the tick count under data parallel: ticks > B
    state = torch.cat([L_.theta2.reshape(-1), L_.adam_m, L_.adam_v]).cpu()
    gathered = [torch.empty_like(state) for _ in range(world)]
    dist.all_gather(gathered, state)
    for r in range(world):
        assert torch.equal(gathered[r], gathered[0]), f"rank {r} differs from rank 0"
    all_steps = [None] * world
    dist.all_gather_object(all_steps, steps)
    assert len(set(all_steps)) == 1, all_steps
    assert torch.isfinite(state).all()
    if L_.xgmi is not None:
        assert L_.xgmi.status()[1] == 0                        # no timed-out wait
    assert L_.fold_fallbacks >= 0

    # ---- files: rank 0 only (checkpoint_frequency 5 -> episodes 5, 10) -----------------------------------------------------
    have = sorted(os.listdir("checkpoints")) if os.path.isdir("checkpoints") else []
    if rank == 0:
        assert have == ["10", "5"] and os.path.isfile("model.p") and os.path.isfile("checkpoints/5/scores.txt"), have
    else:
        assert have == [] and not os.path.exists("model.p"), have
    dist.barrier()
    if L_.xgmi is not None:
        L_.xgmi.close()
    print(f"DP_LOOP_OK_{rank};", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
