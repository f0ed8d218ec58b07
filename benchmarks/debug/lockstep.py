"""Two agents in lockstep on the same transitions — one through the fused per-timestep launches (form by the environment:
NAF_STEP_FORM), one through the twelve separate launches — compared after EVERY timestep: which quantity
differs first when they part (an intermittent mismatch, ~1e-5 per timestep, seen by tests/test_step_path_gpu.py)?"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.chdir(tempfile.mkdtemp())
import logging
import numpy as np, torch
from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
from synth_data import make_transitions
logging.getLogger('robotic_manipulator_rloa.utils.logger').setLevel(40)
DEV = torch.device("cuda:0")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
SYNC = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # compare every SYNC timesteps
S, A, H, B, N = 21, 6, 256, 64, 300
CH = 5000
FORM = os.environ.get("NAF_STEP_FORM", "pipelined")
fa = NAFAgent(object(), S, A, H, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
os.environ["NAF_STEP_FORM"] = "separate"
ua = NAFAgent(object(), S, A, H, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
print(f"form {FORM} against the separate launches; compare every {SYNC}", flush=True)


def state_of(ag):
    L, ch = ag.learner, ag._chunk
    d = dict(theta=L.theta2, m=L.adam_m, v=L.adam_v, bn=L.bn_stats, step=L.step_dev, ring=ag.memory.rows, meta=ag.memory.meta,
             ctr=ag.memory._sample_ctr)
    if ch is not None:
        d.update(idx=ch.idx)
        if not getattr(ch, "pipelined", False) and ch.spec_rec is None:
            d.update(batch=ch.batch, mom=ch.moments)
    return d


t_global, parted = 0, False
state = None
while t_global < STEPS and not parted:
    st_, ac, rw, ns, dn = make_transitions(CH + 1, S, A, seed=1000 + t_global)
    if state is None:
        state = st_[0].astype(np.float64)
    for t in range(CH):
        a1 = fa.act(state)
        a2 = ua.act(state)
        a1 = np.array(a1, copy=True); a2 = np.array(a2, copy=True)
        nxt = ns[t].astype(np.float64)
        if not np.array_equal(a1, a2):
            print(f"timestep {t_global}: ACTIONS differ {a1} vs {a2}", flush=True)
            parted = True
        fa.step(state, a1, float(rw[t]), nxt, 0)
        ua.step(state, a2, float(rw[t]), nxt, 0)
        state = nxt
        t_global += 1
        if parted or t_global % SYNC == 0:
            torch.cuda.synchronize()
            sf, su = state_of(fa), state_of(ua)
            diff = [k for k in sf if k in su and not torch.equal(sf[k], su[k])]
            if diff or parted:
                print(f"timestep {t_global}: differ in {diff}", flush=True)
                for k in diff:
                    x, y = sf[k].flatten().double().cpu(), su[k].flatten().double().cpu()
                    w = torch.nonzero(x != y).flatten()
                    print(f"   {k}: {len(w)} of {x.numel()} elements, first at {int(w[0])}: {float(x[w[0]])} vs {float(y[w[0]])}", flush=True)
                ch = fa._chunk
                print("   fused chunk fast/slow", getattr(ch, "fast_runs", 0), getattr(ch, "slow_runs", 0), "err", [int(e) for e in fa.learner.err_host[:3]], flush=True)
                parted = True
                break
    print(f"{t_global} timesteps, parted: {parted}", flush=True)
