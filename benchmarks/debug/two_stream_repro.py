"""Needs benchmarks/probe/two_streams_product.patch applied (NAF_STEP_TWO_STREAMS=1: the graphs of consecutive timesteps on two alternating
streams; NOTEBOOK section 11.20): three agents in a row, both launch orders, the streams' handles and the order words printed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, tempfile
os.chdir(tempfile.mkdtemp())
from robotic_manipulator_rloa_amd import engine
from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
from synth_data import make_transitions
DEV = torch.device("cuda:0")
S, A, N, T, B = 21, 6, 20000, 400, int(sys.argv[1]) if len(sys.argv) > 1 else 256
inner = engine._Pipeline.collect
for rep, first in enumerate((True, False, True)):
    def collect(self, _first=first):
        inner(self)
        self.waited = _first
    engine._Pipeline.collect = collect
    os.environ["NAF_STEP_FORM"] = "pipelined"
    agent = NAFAgent(object(), S, A, 256, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
    m = agent.memory
    rng = np.random.default_rng(5)
    r = np.zeros((6000, m.row_floats), np.float32)
    r[:, :S] = rng.standard_normal((6000, S)); r[:, S:S + A] = rng.uniform(-1, 1, (6000, A)); r[:, S + A] = -rng.random(6000)
    r[:, m.off_s2:m.off_s2 + S] = r[:, :S]
    m.add_rows_device(torch.from_numpy(r).to(DEV), 6000)
    st_, ac, rw, ns, dn = make_transitions(T + 1, S, A, seed=33)
    state = st_[0].astype(np.float64)
    try:
        for t in range(T):
            a = agent.act(state)
            nxt = ns[t].astype(np.float64)
            agent.step(state, a, float(rw[t]), nxt, 0)
            state = nxt
        torch.cuda.synchronize()
        ok = "ok"
    except Exception as e:
        ok = f"FAILED at t={t}: {str(e)[:80]}"
    p = agent._chunk.pipe
    torch.cuda.synchronize()
    print(f"run {rep} prefetch_first={first}: {ok}; two={p.two} streams cur={torch.cuda.current_stream().cuda_stream:#x} side={p.side.cuda_stream:#x} "
          f"alt={(p.alt.cuda_stream if p.alt is not None else 0):#x} n_graphs={p.n_graphs} order[0]={int(p.order[0].item())} order[32]={int(p.order[32].item())} "
          f"fast/slow={p.fast_runs}/{p.slow_runs} need_word={int(p._need_np[0]) if p._need_np is not None else None}", flush=True)
