"""Rate of the intermittent mismatch of the fused per-timestep loop against the unfused one, per variant (argv[1]):
  plain      fused launches, no prefetch            nostore    + NAF_HOST_STORE=0 (the kernel reads the pinned row itself)
  syncwait   + a stream synchronise in wait_tail    presync    + a stream synchronise before every launch
  prefetch / pipelined: those forms, as shipped
  oldwait    plain, but the host waits as the round's first form did: polls the ordinal WORD the launch stores behind the action's
             plain words (host_seq) and reads those words (the launch still writes both) — the protocol that let stale actions in
Usage: python benchmarks/debug/stress_variants.py <variant> [runs]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.chdir(tempfile.mkdtemp())
import logging
import numpy as np, torch
from robotic_manipulator_rloa_amd import engine
from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
from synth_data import make_transitions
logging.getLogger('robotic_manipulator_rloa.utils.logger').setLevel(40)
DEV = torch.device("cuda:0")
VAR = sys.argv[1] if len(sys.argv) > 1 else "plain"
RUNS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
S, A, H, B, N, T = 21, 6, 256, 64, 300, 600
st_, ac, rw, ns, dn = make_transitions(B + T + 1, S, A, seed=33)


def drive(agent):
    acts, state = [], st_[0].astype(np.float64)
    for t in range(B + T):
        a = agent.act(state)
        acts.append(np.array(a, copy=True))
        nxt = ns[t].astype(np.float64)
        agent.step(state, a, float(rw[t]), nxt, 0)
        state = nxt
    torch.cuda.synchronize()
    return np.array(acts)


def run(env):
    for k in ("NAF_STEP_FUSED", "NAF_STEP_PREFETCH", "NAF_STEP_PIPELINE", "NAF_HOST_STORE"):
        os.environ[k] = env.get(k, "1")
    agent = NAFAgent(object(), S, A, H, B, N, 1e-3, 1e-3, 0.99, 1, 1, 500, DEV, 0)
    acts = drive(agent)
    return acts, agent.learner.theta2.clone()


ref_acts, ref_theta = run({"NAF_STEP_FUSED": "0"})
ref2 = run({"NAF_STEP_FUSED": "0"})
assert np.array_equal(ref_acts, ref2[0]) and torch.equal(ref_theta, ref2[1])
env = {"plain": {"NAF_STEP_PREFETCH": "0", "NAF_STEP_PIPELINE": "0"}, "nostore": {"NAF_STEP_PREFETCH": "0", "NAF_STEP_PIPELINE": "0", "NAF_HOST_STORE": "0"},
       "syncwait": {"NAF_STEP_PREFETCH": "0", "NAF_STEP_PIPELINE": "0"}, "presync": {"NAF_STEP_PREFETCH": "0", "NAF_STEP_PIPELINE": "0"},
       "prefetch": {"NAF_STEP_PIPELINE": "0"}, "pipelined": {},
       "oldwait": {"NAF_STEP_PREFETCH": "0", "NAF_STEP_PIPELINE": "0"}}[VAR]
if VAR == "syncwait":
    _wt = engine.TrainChunk.wait_tail
    def wait_tail(self):
        _wt(self)
        torch.cuda.current_stream().synchronize()
    engine.TrainChunk.wait_tail = wait_tail
if VAR == "presync":
    _rr = engine.TrainChunk.run_row
    def run_row(self):
        torch.cuda.current_stream().synchronize()
        _rr(self)
    engine.TrainChunk.run_row = run_row
if VAR == "oldwait":
    def wait_tail(self):
        if self._seq_np is None:
            torch.cuda.current_stream().synchronize()
            return
        if not self._inflight:
            return
        a = self._tail_actor
        if not hasattr(self, "_old_prev"):
            self._old_prev = None
        sq = a.seq_np                                   # the ordinal word (host_seq), stored behind the action's words + vmcnt(0)
        want = int(self._seq_prev) + 1                  # (chunk 0's ordinal before the launch, + 1: the launch's own)
        while sq[0] != want:
            pass
        self._inflight = False                          # a.actions_np: the plain words, as the old host read them
    engine.TrainChunk.wait_tail = wait_tail
    _rr = engine.TrainChunk.run_row
    def run_row(self):
        a = self._tail_actor
        # (the old bookkeeping: the previous ordinal is the ordinal word's value — wait for the chunk too so that _seq_prev is right)
        while self._seq_np[0] != a.seq_np[0]:
            pass
        _rr(self)
    engine.TrainChunk.run_row = run_row
bad = 0
for r in range(RUNS):
    acts, theta = run(env)
    d = np.where((acts != ref_acts).any(axis=1))[0]
    if len(d) or not torch.equal(theta, ref_theta):
        bad += 1
        print(f"{VAR} run {r}: first differing timestep {d[0] if len(d) else None} ({len(d)} differ)", flush=True)
    if r % 100 == 99:
        print(f"{VAR}: {r + 1} runs, {bad} mismatching", flush=True)
print(f"{VAR}: {RUNS} runs of {T} timesteps, {bad} mismatching")
