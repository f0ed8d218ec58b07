"""A long free-running run of the per-timestep loop in its shipped form (pipelined) and through the twelve separate launches, on the
same scripted transitions: the SHA-256 of every action taken, the final parameters, optimizer state, BatchNorm buffers and ring
must be equal.   python benchmarks/debug/soak.py [timesteps] [batch] [ring] [joints] [layer_size] [state_size]   (state = 9 + 2 joints unless given)"""
import hashlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.chdir(tempfile.mkdtemp())
import logging
import numpy as np, torch
from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
from synth_data import make_transitions
logging.getLogger('robotic_manipulator_rloa.utils.logger').setLevel(40)
DEV = torch.device("cuda:0")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
A = int(sys.argv[4]) if len(sys.argv) > 4 else 6
H = int(sys.argv[5]) if len(sys.argv) > 5 else 256
S = int(sys.argv[6]) if len(sys.argv) > 6 else 9 + 2 * A          # (a state size other than the reference's 9 + 2 joints)
CH = min(50000, STEPS)


def run(form):
    os.environ["NAF_STEP_FORM"] = form
    agent = NAFAgent(object(), S, A, H, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
    h = hashlib.sha256()
    state, done, t0 = None, 0, time.perf_counter()
    while done < STEPS:
        st_, ac, rw, ns, dn = make_transitions(CH + 1, S, A, seed=7000 + done)
        if state is None:
            state = st_[0].astype(np.float64)
        acts = np.empty((CH, A), np.float32)
        for t in range(CH):
            a = agent.act(state)
            acts[t] = a
            nxt = ns[t].astype(np.float64)
            agent.step(state, a, float(rw[t]), nxt, 0)
            state = nxt
        h.update(acts.tobytes())
        done += CH
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L, ch = agent.learner, agent._chunk
    out = dict(digest=h.hexdigest(), theta=L.theta2.clone(), m=L.adam_m.clone(), v=L.adam_v.clone(), bn=L.bn_stats.clone(),
               ring=agent.memory.rows.clone(), meta=agent.memory.meta.clone(), step=int(L.step_dev.item()), err=[int(e) for e in L.err_host[:3]],
               runs=(getattr(ch, "fast_runs", 0), getattr(ch, "slow_runs", 0)), us=dt / done * 1e6, finite=bool(torch.isfinite(L.theta2).all()))
    print(f"NAF_STEP_FORM={form}: {done} timesteps, {out['us']:.1f} us each (scripted transitions, no environment), {out['step']} optimizer steps, "
          f"pipelined graph / start-over graph {out['runs']}, error words {out['err']}, parameters finite {out['finite']}, actions {out['digest'][:16]}", flush=True)
    return out


a, b = run("pipelined"), run("separate")
same = a["digest"] == b["digest"] and all(torch.equal(a[k], b[k]) for k in ("theta", "m", "v", "bn", "ring", "meta")) and a["step"] == b["step"]
print(f"{A} joints, state {S}, layer size {H}, B = {B}, ring {N}: every action, theta, theta', m, v, BatchNorm buffers, ring and counters equal: {same}")
sys.exit(0 if same else 1)
