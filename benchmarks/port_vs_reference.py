"""Is oracle/torch_cpu_port.py (bench.py's `cpu_baseline`, kind "port") a fair stand-in for the reference on CPU?

BUILD CONTAINER ONLY: imports the unmodified reference from /root/reference (stub pybullet, exactly as
tests/golden/make_golden.py does) and times, ALTERNATING port and reference, the two quantities the bench reports —
learn() alone and the whole per-timestep path act + add + sample + learn (naf_algorithm.py:129-215) — at (B=256, N=1e5)
and (B=256, N=1e6, the benched fill), a fixed thread count, >= 5 repetitions each. Medians go to
profiles/port_vs_reference.json.  Usage: python benchmarks/port_vs_reference.py [--reps 5] [--seconds 4]
"""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time
import types
from unittest.mock import MagicMock

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def install_reference():
    pb = types.ModuleType("pybullet")
    pb.error = type("error", (Exception,), {})
    pb.GUI, pb.DIRECT, pb.POSITION_CONTROL, pb.VELOCITY_CONTROL = 1, 2, 2, 0
    pb.__getattr__ = lambda name: MagicMock()
    pbd = types.ModuleType("pybullet_data")
    pbd.getDataPath = lambda: "/nonexistent/pybullet_data"
    sys.modules["pybullet"], sys.modules["pybullet_data"] = pb, pbd
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--seconds", type=float, default=4.0, help="per timed leg")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--shapes", default="256:100000,256:1000000")
    args = ap.parse_args()
    install_reference()
    os.chdir(tempfile.mkdtemp(prefix="naf_pvr_"))
    import logging
    import torch
    from robotic_manipulator_rloa.naf_components.naf_algorithm import NAFAgent
    from oracle.torch_cpu_port import TorchCpuAgent
    logging.disable(logging.CRITICAL)
    torch.set_num_threads(args.threads)
    S, A, H = 21, 6, 256
    out = {"threads": args.threads, "host_cpus": os.cpu_count(), "torch": torch.__version__, "reps": args.reps,
           "seconds_per_leg": args.seconds, "shapes": []}
    for spec in args.shapes.split(","):
        B, N = (int(x) for x in spec.split(":"))
        rng = np.random.default_rng(0)
        st = rng.standard_normal((N + 4097, S))
        ac = rng.uniform(-1, 1, (N + 4097, A)).astype(np.float32)
        ref = NAFAgent(object(), S, A, H, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, torch.device("cpu"), 0)
        port = TorchCpuAgent(S, A, H, B, N, seed=0)
        for i in range(N):                                 # both deques full, every record owning its own array views
            ref.memory.add(st[i], ac[i], -0.5, st[i + 1], 0)
            port.add(st[i], ac[i], -0.5, st[i + 1], 0)

        def leg_learn(agent, sample):
            ex = sample()
            agent.learn(ex)
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < args.seconds:
                agent.learn(ex)
                n += 1
            return n / (time.perf_counter() - t0)

        def leg_step(agent):
            s = st[N]
            t0, n = time.perf_counter(), 0
            while time.perf_counter() - t0 < args.seconds:
                a = agent.act(s)
                s2 = st[N + ((n + 1) & 4095)]
                agent.step(s, a, -0.5, s2, 0)
                s = s2
                n += 1
            return n / (time.perf_counter() - t0)

        res = {"ref_learn": [], "port_learn": [], "ref_step": [], "port_step": []}
        for _ in range(args.reps):                         # alternate, so drift of the noisy vCPUs hits both alike
            res["ref_learn"].append(leg_learn(ref, ref.memory.sample))
            res["port_learn"].append(leg_learn(port, port.sample))
            res["ref_step"].append(leg_step(ref))
            res["port_step"].append(leg_step(port))
        med = {k: statistics.median(v) for k, v in res.items()}
        rec = {"B": B, "N": N, "median_per_s": {k: round(v, 2) for k, v in med.items()},
               "all_per_s": {k: [round(x, 2) for x in v] for k, v in res.items()},
               "port_over_ref_learn": round(med["port_learn"] / med["ref_learn"], 3),
               "port_over_ref_step": round(med["port_step"] / med["ref_step"], 3)}
        out["shapes"].append(rec)
        print(json.dumps(rec), flush=True)
        del ref, port
    path = os.path.join(ROOT, "profiles", "port_vs_reference.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
