"""The host vector env around the GPU learner (bench.py's `host_vector_env` entry) on its own: E = 64 stand-in environments in
worker processes, B = 256, sync and async policy; prints env-steps/s and what the box gives (usable CPUs, workers).

    python benchmarks/host_vector_bench.py [workers]
"""
import logging, os, sys, tempfile, time
from functools import partial
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
from robotic_manipulator_rloa_amd.environment.vector_env import HostVectorEnv, usable_cpus
from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent


def main():
    os.chdir(tempfile.mkdtemp())
    logging.getLogger('robotic_manipulator_rloa.utils.logger').setLevel(40)
    S, A, E = 21, 6, 64
    cpus = usable_cpus()
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, min(E, cpus // 2))
    per = (E + workers - 1) // workers
    print(f"usable cpus {cpus} (os.cpu_count() {os.cpu_count()}), {(E + per - 1) // per} workers x {per} envs")
    dev = torch.device("cuda:0")
    for mode in (False, True):
        agent = NAFAgent(None, S, A, 256, 256, 100000, 1e-3, 1e-3, 0.99, 1, 1, 500, dev, 0)
        vec = HostVectorEnv(partial(SyntheticEnvironment, A), E, S, A, envs_per_worker=per, max_frames=400, seed=1)
        try:
            agent.run_host_vectorized(vec, 20, async_policy=mode)
            t0 = time.perf_counter()
            for _ in range(200):
                vec.step(vec.arr["actions"].copy())
            dt_env = (time.perf_counter() - t0) / 200
            r = agent.run_host_vectorized(vec, 300, async_policy=mode)
        finally:
            vec.close()
        print(f"{'async' if mode else 'sync '} policy: {r['env_steps_per_s']:.0f} env-steps/s ({1e6 * 64 / r['env_steps_per_s']:.0f} us per vector step; "
              f"the vector env's step alone {1e6 * dt_env:.0f} us)")
        del agent


if __name__ == "__main__":          # (the workers are spawned: they import this file)
    main()
