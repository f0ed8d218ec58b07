"""NAFAgent with the reference's interface (naf_components/naf_algorithm.py:23-292), running its per-timestep
hot path — act / step (add, gate, sample, learn) / soft_update — on the MI355X through the flat-buffer Learner,
the HBM ReplayBuffer and captured HIP graphs.

Kept from the reference, on purpose: learning starts when len(memory) > batch_size (strict, :150);
the update gate is (t+1) % update_freq == 0 (:147-148); `dones` is stored and sampled but never used in the target
(:199); both networks run BatchNorm in training mode inside learn() and only parameters() are soft-updated; act()
returns a NOISY action even at test time; sampled actions are truncated toward zero (`.long()`), unless
action_mode='float'. Files written by run(): checkpoints/{episode}/weights.p, scores.txt, model.p — same names,
same JSON, same state_dict keys, so they interchange with the reference.
"""
from __future__ import annotations

import json
import os
import random
import time
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib
from ..engine import DeviceEnvLoop, EpisodeLedger, TimestepGraph, UpdateChunk
from ..learner import ActPath, Learner
from ..utils.exceptions import MissingWeightsFile
from ..utils.logger import get_global_logger
from ..utils.replay_buffer import ReplayBuffer
from .naf_neural_network import NAF, _P_MODES

logger = get_global_logger()

_ACTION_MODES = {"trunc_int": _lib.ACTION_TRUNC_INT, "float": _lib.ACTION_FLOAT, 0: 0, 1: 1}


class FlatAdam:
    """What `agent.optimizer` is here: Adam's state lives in two flat buffers next to the parameters and is stepped
    by naf_adam_polyak_fused; this object only exposes it (torch.optim.Adam defaults, naf_algorithm.py:83)."""

    def __init__(self, learner: Learner):
        self._L = learner
        self.defaults = {"lr": learner.lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0}
        self.param_groups = [dict(self.defaults, params=list(range(14)))]

    def zero_grad(self, set_to_none: bool = True) -> None:   # every gradient segment is overwritten by each backward
        return None

    def state_dict(self) -> dict:
        L = self._L
        return {"step": int(L.step_dev.item()), "exp_avg": L.adam_m.clone(), "exp_avg_sq": L.adam_v.clone(),
                "param_groups": self.param_groups}


class NAFAgent:

    MODEL_PATH = 'model.p'

    def __init__(self, environment, state_size: int, action_size: int, layer_size: int, batch_size: int,
                 buffer_size: int, learning_rate: float, tau: float, gamma: float, update_freq: int, num_updates: int,
                 checkpoint_frequency: int, device, seed: int, *, p_mode="hadamard", action_mode="trunc_int",
                 data_parallel: Optional[bool] = None, use_graph: bool = True) -> None:
        """Positional arguments as the reference (naf_algorithm.py:27-41). Keyword-only extras default to the
        reference's behaviour. data_parallel=None: all-reduce gradients iff torch.distributed is initialised with
        more than one rank."""
        _lib.require_gpu()                               # fail before touching the file system
        if torch.device(device).type != "cuda":
            raise _lib.NafHipError(f"NAFAgent needs the MI355X (got device={device}); there is no CPU fallback")
        os.makedirs('checkpoints/', exist_ok=True)       # as the reference does at construction (:61)
        self.environment = environment
        self.state_size, self.action_size, self.layer_size = state_size, action_size, layer_size
        self.buffer_size, self.learning_rate = buffer_size, learning_rate
        random.seed(seed)
        self.device = torch.device(device)
        self.tau, self.gamma = tau, gamma
        self.update_freq, self.num_updates = update_freq, num_updates
        self.batch_size, self.checkpoint_frequency = batch_size, checkpoint_frequency
        self.seed = seed

        import torch.distributed as dist
        world, pg = 1, None
        if data_parallel or (data_parallel is None and dist.is_available() and dist.is_initialized()):
            world = dist.get_world_size()
        self.world_size = world
        self.rank = dist.get_rank() if world > 1 else 0

        self.learner = Learner(state_size, action_size, layer_size, batch_size, learning_rate, tau, gamma, self.device,
                               p_mode=_P_MODES[p_mode], world_size=world, process_group=pg)
        L = self.learner
        # both constructors draw from torch's global RNG exactly as the reference's do (:79-80): identical nets
        self.qnetwork_main = NAF(state_size, action_size, layer_size, seed, self.device, p_mode=p_mode,
                                 _flat=L.theta2[0], _bn=L.bn_stats[0])
        self.qnetwork_target = NAF(state_size, action_size, layer_size, seed, self.device, p_mode=p_mode,
                                   _flat=L.theta2[1], _bn=L.bn_stats[1])
        self.qnetwork_main._tracked_hook = lambda: int(L.step_dev.item())
        self.qnetwork_target._tracked_hook = lambda: int(L.step_dev.item())
        if world > 1:
            dist.broadcast(L.theta2, src=0)
        self.optimizer = FlatAdam(L)
        # per-rank sampler stream; rank 0 uses the user's seed
        self.memory = ReplayBuffer(buffer_size, batch_size, self.device, seed + 7919 * self.rank, state_size=state_size,
                                   action_size=action_size, action_mode=_ACTION_MODES[action_mode])
        if self.rank:
            random.seed(seed)  # ReplayBuffer re-seeded Python's RNG with the rank offset; keep the reference's value
        self.update_t_step = 0
        self._dp_ticks = 0                 # update-schedule ticks (step() calls + idle ticks): the gate under data parallel
        self._last_loss_from = None        # "chunk" | "learn": which path ran the most recent update
        self._ahead = None                 # the state the last step()'s graph already ran the policy on (see _update_tick)
        self._fast = None                  # (chunk, pinned row) once every step() is "write the row, replay the graph" (step())
        self.use_graph = use_graph
        self._chunk: Optional[TimestepGraph] = None
        self._actor1: Optional[ActPath] = None
        self._act_graph = None
        self._obs_pinned = torch.zeros(1, state_size, dtype=torch.float32).pin_memory()
        self._act_pinned = torch.zeros(1, action_size, dtype=torch.float32).pin_memory()
        self._learn_rows = torch.zeros(batch_size + 1, L.lay.batch_row_floats, dtype=torch.float32, device=self.device)[:batch_size]
        self._learn_loss = torch.zeros(L.n_loss_wg, dtype=torch.float32, device=self.device)
        # step()'s transition, read by the graph: [row | count (int32, kept by TimestepGraph) | pad]
        self._row_pin = torch.zeros(1, L.lay.row_floats + 4, dtype=torch.float32).pin_memory()
        self._row_np = self._row_pin.numpy()
        self.last_run_stats: Optional[dict] = None    # counters of the most recent run_vectorized / run_host_vectorized

    # ---- pretrained weights (naf_algorithm.py:91-127) ----------------------------------------------------------
    def _load_weights(self, path: str) -> None:
        sd = torch.load(path, map_location="cpu")
        self.qnetwork_main.load_state_dict(sd)
        self.qnetwork_target.load_state_dict(sd)

    def initialize_pretrained_agent_from_episode(self, episode: int) -> None:
        path = f'checkpoints/{episode}/weights.p'
        if not os.path.isfile(path):
            raise MissingWeightsFile
        self._load_weights(path)
        logger.info(f'Loaded weights from trained naf_components on episode {episode}')

    def initialize_pretrained_agent_from_weights_file(self, weights_path: str) -> None:
        if not os.path.isfile(weights_path):
            raise MissingWeightsFile
        self._load_weights(weights_path)
        logger.info('Loaded pre-trained weights for the NN')

    # ---- per-timestep path ---------------------------------------------------------------------------------------
    def step(self, state, action, reward: float, next_state, done: int) -> None:
        """Store the experience and, every update_freq steps once len(memory) > batch_size, run num_updates
        (sample + learn) (naf_algorithm.py:129-156). The updates are one captured graph:
        [append this transition] -> sample num_updates minibatches -> one gather -> num_updates x learn -> [act(next_state)];
        with one update per timestep (the reference's own loop) the append, the draw, the gather and the moments are one launch
        (naf_step_prep), the optimizer step and the policy's forward on the next state another (naf_adam_polyak_act) around the five
        of the row-split chain — and on one GPU the graph is PIPELINED (engine.TimestepGraph / _Pipeline): the launch that ends a timestep also
        draws the next timestep's minibatch, the chain runs on it while the host steps the environment, and this call's graph
        starts with the launch that appends the row, applies the gradient that is waiting and acts: six launches, of which act()
        waits for the first."""
        f = self._fast
        if f is not None and self.memory._pending == 0:
            # every tick updates (update_freq = 1, gate open): the append is the first node of the update's graph, reading
            # the transition from a pinned row of its own — same order (add, then sample) as the reference's step()
            ch, row = f
            ch.wait_pinned_free()     # the previous graph's append has read the row (run()'s act() waited already)
            S, A, o2 = self._S, self._A, self._o2
            row[:S] = state[0] if isinstance(state, tuple) else state
            row[S:S + A] = action
            row[S + A] = reward
            row[o2:o2 + S] = next_state
            row[o2 + S] = done
            self.memory._total_added += 1
            self._dp_ticks += 1                                    # (update_t_step stays 0: update_freq = 1)
            if done or ch.tail is None:
                self._ahead = None
            else:
                # the graph ends with the policy's forward on the state the loop will ask about next (_update_tick) — taken
                # from the row's own next_state columns where both ends of the graph are the fused launches
                if not self._obs_in_row:
                    self._obs1[:] = next_state
                self._ahead = self._obs1
            if ch._seq_np is not None and self.world_size == 1:
                ch.run_row()
            else:
                ch.run(head_rows=1)
            self._last_loss_from = "chunk"
            return
        if self._row_in_graph():
            m, row = self.memory, self._row_np[0]
            S, A = m.S, m.A
            self._chunk.wait_pinned_free()
            row[:S] = state[0] if isinstance(state, tuple) else state
            row[S:S + A] = action
            row[S + A] = reward
            row[m.off_s2:m.off_s2 + S] = next_state
            row[m.off_s2 + S] = done
            m._total_added += 1
            # from here on every step() takes the short way above (the gate never closes again; a row waiting in the staging
            # area — memory._pending — is checked there)
            if self.use_graph and self._chunk.graph is not None:
                self._S, self._A, self._o2 = S, A, m.off_s2
                self._obs_in_row = self._chunk.row_dev is not None
                self._obs1 = (row[m.off_s2:m.off_s2 + S] if self._obs_in_row else
                              (self._actor1.obs_np[0] if (self._actor1 is not None and self._actor1.host_io) else None))
                self._fast = (self._chunk, row)
            self._update_tick(None if done else next_state, row_in_graph=True)
            return
        self.memory.add(state, action, reward, next_state, done)
        self._update_tick(None if done else next_state)

    def _row_in_graph(self) -> bool:
        m = self.memory
        return (self.update_freq == 1 and self.use_graph and m._handle is not None and m._pending == 0 and
                (len(m) > self.batch_size if self.world_size == 1 else (self._dp_ticks > self.batch_size and len(m) > 0)) and
                self._chunk is not None and self._chunk.head_row is not None)

    def _update_tick(self, next_state=None, row_in_graph: bool = False) -> None:
        """The update schedule of step() (naf_algorithm.py:144-156) without the add. Data parallel: every learn() holds a
        gradient all-reduce, so every rank must run the SAME number of ticks and open the gate at the same tick — the
        gate is therefore the tick count (identical on all ranks; equal to len(memory) whenever every tick added a row,
        i.e. always on one GPU), not the local fill level; run() pads episodes that ended early with idle ticks. A rank
        that falls out of step is caught by the exchange's time-out (TimestepGraph.run raises).

        next_state: where run()'s loop will ask act() next. The chunk's graph ends with the policy's forward on it — behind
        the updates, so it has seen them, as the reference's ordering demands (naf_algorithm.py:249-261) — and the next
        act(next_state) only waits for the graph and reads the action from pinned memory: one launch, one host round trip
        and the idle gap between them less per timestep."""
        self.update_t_step = (self.update_t_step + 1) % self.update_freq
        self._dp_ticks += 1
        ready = len(self.memory) > self.batch_size if self.world_size == 1 else \
            (self._dp_ticks > self.batch_size and len(self.memory) > 0)
        if self.update_t_step == 0 and ready:
            if not row_in_graph:
                self.memory.flush()
            if self._chunk is None:
                a = self._actor()
                tail = a.act if a.host_io else None        # (only the one-launch act() reads / writes pinned memory itself)
                # update_freq = 1: from the first update on every tick runs this graph, and the tick's transition is appended
                # by the graph itself (step()); other schedules keep ReplayBuffer.add's staging in front of it
                head = self._row_pin if (self.update_freq == 1 and self.use_graph) else None
                if head is not None and not row_in_graph:
                    # the tick that builds the graph came through memory.add (the gate was closed before it): build the graph
                    # with the append node, but run THIS tick's updates eagerly — its row is in the ring already
                    self._chunk = TimestepGraph(self.learner, self.memory, self.num_updates, use_graph=True, tail=tail,
                                             tail_state=(a.counter, a._ticket) if tail else (), head_row=head)
                    once = TimestepGraph(self.learner, self.memory, self.num_updates, use_graph=False, tail=tail)
                    if tail is not None and next_state is not None:
                        self._actor1.obs_np[0] = next_state
                        self._ahead = self._actor1.obs_np[0]
                    else:
                        self._ahead = None
                    once.run()
                    # (the eager chunk's tail writes the same pinned action / ordinal words the graph's will: what act() waits
                    #  for next is THIS run)
                    self._chunk._seq_prev, self._chunk._inflight = once._seq_prev, once._inflight
                    self._chunk.loss_parts = once.loss_parts       # (last_loss() of this tick)
                    self._chunk.idx.copy_(once.idx)                # (... and the minibatch it drew, where a reader looks for it)
                    self._last_loss_from = "chunk"
                    return
                self._chunk = TimestepGraph(self.learner, self.memory, self.num_updates, use_graph=self.use_graph,
                                         tail=tail, tail_state=(a.counter, a._ticket) if tail else (), head_row=head)
            if self._chunk.tail is not None and next_state is not None:
                self._chunk.wait_pinned_free()     # (a no-op behind step()'s own wait; idle ticks and staged rows come here)
                if self._chunk.row_dev is not None:        # (the fused launches: the observation rides in the pinned row)
                    o2 = self.memory.off_s2
                    self._ahead = self._row_np[0][o2:o2 + self.memory.S]
                    self._ahead[:] = next_state
                else:
                    self._actor1.obs_np[0] = next_state
                    self._ahead = self._actor1.obs_np[0]   # (the pinned observation itself: what the graph's act() runs on)
            else:
                self._ahead = None
            # the graph's first node appends the pinned row only when this tick brought one (step()'s fast path): an idle
            # tick of a data-parallel run() or a row that went through memory.add's staging must not re-append the last one
            self._chunk.run(head_rows=1 if row_in_graph else 0)
            self._last_loss_from = "chunk"

    def _actor(self) -> ActPath:
        if self._actor1 is None:
            self._actor1 = ActPath(self.learner, 1, seed=(self.seed * 2654435761 + 12345 + self.rank) & 0xFFFFFFFFFFFFFFFF,
                                   host_io=True)
            if self.use_graph and not self._actor1.fused:      # one launch needs no graph around it
                a = self._actor1
                saved = (a.counter.clone(),)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    a.act()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    a.act()
                a.counter.copy_(saved[0])
                self._act_graph = g
        return self._actor1

    def act(self, state) -> np.ndarray:
        """Noisy clamped action for one state, main net in eval mode (naf_algorithm.py:158-178)."""
        a = self._actor1 or self._actor()
        if a.host_io:
            ahead, self._ahead = self._ahead, None
            if ahead is not None:
                # the last step()'s graph already ran the policy on the state it was told comes next: wait for its action
                # (a spin on the pinned word the graph's last launch writes; the stream for graphs without that launch)
                self._chunk.wait_tail()
                s32 = np.asarray(state, dtype=np.float32)
                if s32.shape == ahead.shape and (s32 == ahead).all():
                    out = a.actions_np[0].copy()
                    return out.squeeze() if out.size == 1 else out
            elif self._chunk is not None:
                self._chunk.wait_pinned_free()             # (the graph's tail may still be reading the pinned observation)
            # the kernel reads the state from, and writes the action to, pinned host memory: no copies to enqueue
            a.obs_np[0] = state
            if self._act_graph is not None:
                self._act_graph.replay()
            else:
                a.act()
            torch.cuda.current_stream().synchronize()
            return a.actions_np[0].copy().squeeze()
        self._obs_pinned[0].copy_(torch.from_numpy(np.asarray(state, dtype=np.float32)))
        a.obs.copy_(self._obs_pinned, non_blocking=True)
        if self._act_graph is not None:
            self._act_graph.replay()
        else:
            a.act()
        self._act_pinned.copy_(a.actions, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self._act_pinned.numpy().squeeze().copy()

    def learn(self, experiences: Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]) -> None:
        """One update from an explicit (states, actions, rewards, next_states, dones) tuple
        (naf_algorithm.py:180-215). actions may be int64 (the reference's sample() contract) or float."""
        states, actions, rewards, next_states, dones = experiences
        L, lay, B = self.learner, self.learner.lay, self.batch_size
        if states.shape[0] != B:
            raise ValueError(f"learn() expects minibatches of batch_size={B} rows, got {states.shape[0]}")
        r = self._learn_rows
        S, A = lay.S, lay.A
        r[:, :S] = states.to(self.device, torch.float32)
        r[:, S:S + A] = actions.to(self.device, torch.float32)
        r[:, S + A] = rewards.to(self.device, torch.float32).view(B)
        r[:, lay.off_s2:lay.off_s2 + S] = next_states.to(self.device, torch.float32)
        r[:, lay.off_d] = dones.to(self.device, torch.float32).view(B)
        L.learn_rows(r, self._learn_loss)
        self._last_loss_from = "learn"

    def last_loss(self) -> float:
        """MSE loss of the most recent update, whichever path ran it (host sync). The reference computes it and drops
        it (:215)."""
        if self._last_loss_from == "chunk":
            return float(self._chunk.losses()[-1].item())
        if self._last_loss_from == "learn":
            return float(self._learn_loss.sum().item())
        raise _lib.NafHipError("last_loss(): no update has run yet")

    def soft_update(self, main_nn, target_nn) -> None:
        """theta_target = tau*theta_main + (1 - tau)*theta_target over parameters() (naf_algorithm.py:217-226)."""
        if main_nn is self.qnetwork_main and target_nn is self.qnetwork_target:
            self.learner.soft_update()
            return
        lib, st = self.learner.lib, torch.cuda.current_stream().cuda_stream
        for tp, mp in zip(target_nn.parameters(), main_nn.parameters()):
            if tp.is_contiguous() and mp.is_contiguous() and tp.data_ptr() % 16 == 0 and mp.data_ptr() % 16 == 0:
                _lib.check(lib.naf_polyak_update(tp.data_ptr(), mp.data_ptr(), self.tau, float(1.0 - self.tau), tp.numel(), st),
                           "naf_polyak_update")
            else:   # strided views of a foreign flat buffer
                tp.data.copy_(self.tau * mp.data + (1. - self.tau) * tp.data)

    # ---- training loop with a host-side environment (naf_algorithm.py:228-292) ---------------------------------
    def run(self, frames: int = 1000, episodes: int = 1000, verbose: bool = True) -> Dict[int, Tuple[float, int]]:
        logger.info('Training started')
        scores = {episode: (0, 0) for episode in range(1, episodes + 1)}
        for episode in range(episodes):
            logger.info(f'Running Episode {episode + 1}')
            start = time.time()
            state = self.environment.reset(verbose)
            score, mean = 0, list()
            frame = 0
            for frame in range(1, frames + 1):
                if verbose:
                    logger.info(f'Running frame {frame} in episode {episode + 1}')
                action = self.act(state)
                next_state, reward, done = self.environment.step(action)
                self.step(state, action, reward, next_state, done)
                state = next_state
                score += reward
                mean.append(reward)
                if verbose:
                    logger.info(f'Reward: {reward}  -  Cumulative reward: {score}\n')
                if done:
                    break
            if self.world_size > 1:
                # data parallel: every rank's episode costs exactly `frames` ticks of the update schedule, whatever its
                # own environment did — the ranks' learn() calls (one gradient all-reduce each) stay paired
                for _ in range(frame, frames):
                    self._update_tick()
            scores[episode + 1] = (score, frame)
            logger.info(f'Reward:                             {score}')
            logger.info(f'Number of frames:                   {frame}')
            logger.info(f'Mean of rewards on this episode:    {sum(mean) / frames}')
            logger.info(f'Time taken for this episode:        {round(time.time() - start, 3)} secs\n')
            if (episode + 1) % self.checkpoint_frequency == 0 and self.rank == 0:
                os.makedirs(f'checkpoints/{episode + 1}/', exist_ok=True)
                torch.save(self._cpu_state_dict(), f'checkpoints/{episode + 1}/weights.p')
                with open(f'checkpoints/{episode + 1}/scores.txt', 'w') as f:
                    f.write(json.dumps(scores))
        if self.rank == 0:
            torch.save(self._cpu_state_dict(), self.MODEL_PATH)
            logger.info(f'Model has been successfully saved in {self.MODEL_PATH}')
        return scores

    def _cpu_state_dict(self):
        return type(self.qnetwork_main.state_dict())((k, v.cpu()) for k, v in self.qnetwork_main.state_dict().items())

    # ---- training loop with E on-device synthetic envs (the many-env path of BASELINE configs[1..4]) -----------
    def _ledger(self, episodes: Optional[int]) -> EpisodeLedger:
        return EpisodeLedger(episodes, self.checkpoint_frequency, self._cpu_state_dict, write=self.rank == 0,
                             model_path=self.MODEL_PATH)

    def _stop_agreed(self, stop: bool) -> bool:
        """Data parallel: every chunk of updates holds a gradient all-reduce, so all ranks must leave the loop after the
        same vector step — rank 0's verdict (its envs fill the scores dict) is broadcast where the ranks compare notes."""
        if self.world_size == 1:
            return stop
        import torch.distributed as dist
        pg = self.learner.pg
        flag = torch.tensor([1 if stop else 0], dtype=torch.int32, device=self.device)
        dist.broadcast(flag, src=0 if pg is None else dist.get_global_rank(pg, 0), group=pg)
        return bool(flag.item())

    def run_vectorized(self, vector_steps: Optional[int] = None, n_envs: int = 64, max_frames: int = 400,
                       noise_scale: float = 1.0, robot: str = "kuka", obstacle_jitter: float = 0.0, *,
                       episodes: Optional[int] = None, preset=None, variation=None, drain_every: int = 64,
                       verbose: bool = False) -> dict:
        """NAFAgent.run (naf_algorithm.py:228-292) re-hosted for E synthetic arms on the GPU feeding the HBM replay ring;
        each vector step is followed by E * num_updates / update_freq learn() calls, i.e. the reference's update-to-data
        ratio. Everything stays on the device, no host sync per step.

        What run() produces is produced here too: per-env episode bookkeeping rides in the env-step kernel (running score
        in double, frame count, auto-reset on done / after `max_frames` = run()'s `frames`), finished episodes reach the host
        every `drain_every` vector steps in (step, env) order and are numbered in completion order into
        `scores = {episode: (score, last_frame)}`; `checkpoints/{episode}/weights.p` + `scores.txt` every
        `checkpoint_frequency` finished episodes and `model.p` at the end (rank 0 only under data parallel) — the files
        `plot_training_rewards` and `initialize_pretrained_agent_from_episode` read. A checkpoint's weights are those at
        the drain that saw its episode, at most 2 x drain_every vector steps after the episode ended (drain_every=1: at
        the step itself, one host sync per vector step).

        Stops after `vector_steps` vector steps and/or once `episodes` episodes have finished (checked at the drains:
        training may run up to 2 x drain_every steps past the last episode; episodes finishing there are counted in
        `episodes_finished` but not recorded, as run() records exactly `episodes`).
        Returns the counters of earlier rounds plus 'scores', 'episodes_finished', 'checkpoints'."""
        E = int(n_envs)
        if (E * self.num_updates) % self.update_freq != 0:
            raise ValueError("n_envs * num_updates must be a multiple of update_freq")
        if vector_steps is None and episodes is None:
            raise ValueError("run_vectorized: give vector_steps and/or episodes")
        U = E * self.num_updates // self.update_freq
        loop = DeviceEnvLoop(self.learner, self.memory, E, seed=self.seed + 104729 * self.rank, max_frames=max_frames,
                             noise_scale=noise_scale, use_graph=self.use_graph, robot=robot, obstacle_jitter=obstacle_jitter,
                             preset=preset, variation=variation, records=True, drain_every=drain_every)
        chunk = UpdateChunk(self.learner, self.memory, U, use_graph=self.use_graph)
        ledger = self._ledger(episodes)
        self.memory.flush()
        logger.info(f'Training started ({E} environments on the device)')
        t0 = time.time()
        updates = steps = 0

        def book(final=False):
            for score, frames, *_ in loop.drain(final):
                ledger.add(score, frames)
                if verbose:
                    logger.info(f'Episode {ledger.count + ledger.extra}: Reward {score}  Number of frames {frames}')

        while vector_steps is None or steps < vector_steps:
            loop.step()
            steps += 1
            if len(self.memory) > self.batch_size:
                chunk.run()
                updates += U
            if steps % loop.drain_every == 0:
                book()
                if episodes is not None and self._stop_agreed(ledger.complete):
                    break
        torch.cuda.synchronize()
        dt = time.time() - t0
        book(final=True)
        scores = ledger.finish()
        if self.rank == 0:
            logger.info(f'Model has been successfully saved in {self.MODEL_PATH}')
        self.last_run_stats = {
            "env_steps": steps * E, "updates": updates, "seconds": dt, "env_steps_per_s": steps * E / dt,
            "last_loss": float(chunk.losses()[-1].item()) if updates else None, "scores": scores,
            "episodes_finished": ledger.count + ledger.extra, "checkpoints": list(ledger.checkpoints)}
        return self.last_run_stats

    # ---- training loop with E host environments in worker processes (PyBullet or any env with the reference protocol) --
    def run_host_vectorized(self, vec_env, vector_steps: Optional[int] = None, async_policy: bool = False,
                            noise_scale: float = 1.0, *, episodes: Optional[int] = None, verbose: bool = False) -> dict:
        """vec_env: environment.vector_env.HostVectorEnv with E envs. Per vector step: batched act() on the GPU for the E
        current states -> workers step their envs -> E transitions packed into pinned memory -> one H2D copy -> HBM
        replay ring -> E * num_updates / update_freq learn() calls (the reference's update-to-data ratio).
        async_policy=False keeps the reference's ordering (the policy of step t has seen every update of step t-1);
        async_policy=True enqueues the learn() chunk of step t-1 behind act(t), so the GPU learns while the workers
        simulate (the policy then lags by one vector step: "asynchronous many-env training").
        Scores / checkpoints / model.p as run() writes them (see run_vectorized; here the rewards pass through the host
        every step, so an episode is booked — and a checkpoint written — in the very step it ends). The frame budget per
        episode is vec_env.max_frames. Stops after `vector_steps` steps and/or `episodes` finished episodes."""
        E = vec_env.E
        if (E * self.num_updates) % self.update_freq != 0:
            raise ValueError("n_envs * num_updates must be a multiple of update_freq")
        if vector_steps is None and episodes is None:
            raise ValueError("run_host_vectorized: give vector_steps and/or episodes")
        U = E * self.num_updates // self.update_freq
        L, lay = self.learner, self.learner.lay
        actor = ActPath(L, E, seed=(self.seed * 40503 + 7 + self.rank) & 0xFFFFFFFFFFFFFFFF)
        chunk = UpdateChunk(L, self.memory, U, use_graph=self.use_graph)
        obs_pin = torch.zeros(E, lay.S, dtype=torch.float32).pin_memory()
        act_pin = torch.zeros(E, lay.A, dtype=torch.float32).pin_memory()
        rows_pin = torch.zeros(E, lay.row_floats, dtype=torch.float32).pin_memory()
        rows_dev = torch.zeros(E, lay.row_floats, dtype=torch.float32, device=self.device)
        rows_np, obs_np = rows_pin.numpy(), obs_pin.numpy()
        ledger = self._ledger(episodes)
        ep_score, ep_frames = np.zeros(E, np.float64), np.zeros(E, np.int64)
        self.memory.flush()
        obs = vec_env.reset()
        logger.info(f'Training started ({E} environments in worker processes)')
        t0 = time.time()
        updates, reward_sum, pending_learn, steps = 0, 0.0, False, 0
        dropped_transitions = dropped_episodes = 0
        respawns0 = getattr(vec_env, "respawns", 0)
        stream = torch.cuda.current_stream()
        while vector_steps is None or steps < vector_steps:
            obs_np[...] = obs
            actor.obs.copy_(obs_pin, non_blocking=True)
            actor.act(noise_scale)
            act_pin.copy_(actor.actions, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            if async_policy and pending_learn:
                chunk.run()                                   # learns on step t-1's data while the workers simulate step t
                updates += U
                pending_learn = False
            ev.synchronize()
            _, _, rewards, _, _, obs = vec_env.step(act_pin.numpy())
            steps += 1
            reward_sum += float(rewards.sum())
            ep_score += rewards                               # score += reward, per env (naf_algorithm.py:264)
            ep_frames += 1
            ended = np.nonzero(vec_env.arr["episode_end"])[0]
            n_rows = vec_env.pack_rows(rows_np, lay.off_s2)
            if n_rows is None:
                n_rows = E                                    # (a vector env that predates `valid`: every transition exists)
            if n_rows < E:
                # a worker died or hung during this step and was replaced (HostVectorEnv): its envs' transitions do not exist and
                # their episodes in flight are gone — nothing of them is booked or learned from; the step's updates run all the same
                lost = list(vec_env.respawned_envs)
                dropped_transitions += len(lost)
                dropped_episodes += int((ep_frames[lost] > 0).sum())
                reward_sum -= float(rewards[lost].sum())
                ep_score[lost], ep_frames[lost] = 0.0, 0
            if n_rows:
                rows_dev[:n_rows].copy_(rows_pin[:n_rows], non_blocking=True)
                self.memory.add_rows_device(rows_dev, n_rows)
            if len(self.memory) > self.batch_size:
                if async_policy:
                    pending_learn = True
                else:
                    chunk.run()
                    updates += U
            stream.synchronize()                              # rows_pin / obs_pin are rewritten next iteration
            for e in ended:                                   # booked after the step's updates, as run() does (:266-287)
                ledger.add(float(ep_score[e]), int(ep_frames[e]))
                if verbose:
                    logger.info(f'Episode {ledger.count + ledger.extra}: Reward {ep_score[e]}  Number of frames {ep_frames[e]}')
                ep_score[e], ep_frames[e] = 0.0, 0
            # (data parallel: the verdict is a blocking broadcast — compared every 8th vector step, not on every one; the
            # loop may run up to 7 steps past the last episode, whose episodes are counted but not recorded)
            if episodes is not None and (self.world_size == 1 or steps % 8 == 0) and self._stop_agreed(ledger.complete):
                break
        if pending_learn:
            chunk.run()
            updates += U
        torch.cuda.synchronize()
        dt = time.time() - t0
        scores = ledger.finish()
        if self.rank == 0:
            logger.info(f'Model has been successfully saved in {self.MODEL_PATH}')
        self.last_run_stats = {
            "env_steps": steps * E, "updates": updates, "seconds": dt, "env_steps_per_s": steps * E / dt,
            "mean_reward": reward_sum / max(1, steps * E), "episodes_finished": vec_env.episodes_finished,
            "last_loss": float(chunk.losses()[-1].item()) if updates else None, "scores": scores,
            "checkpoints": list(ledger.checkpoints),
            # env workers found dead or hung and replaced during the run, and what went with them (never booked, never learned from)
            "worker_respawns": getattr(vec_env, "respawns", 0) - respawns0, "dropped_transitions": dropped_transitions,
            "dropped_episodes": dropped_episodes}
        return self.last_run_stats

    # ---- evaluation with E environments (rl_framework.py:319-367 re-hosted) -------------------------------------------
    @staticmethod
    def _episode_quota(n_episodes: int, E: int) -> np.ndarray:
        """Episodes each env contributes: its FIRST q_e episodes, sum = n_episodes. (Taking the first n_episodes to finish
        anywhere would favour short episodes — collisions and quick successes finish before the time-outs do.)"""
        q = np.full(E, n_episodes // E, np.int64)
        q[:n_episodes % E] += 1
        return q

    def evaluate_vectorized(self, n_episodes: int, frames: int, n_envs: int = 64, noise_scale: float = 1.0,
                            robot: str = "kuka", obstacle_jitter: float = 0.0, preset=None, variation=None,
                            drain_every: int = 32) -> List[Tuple[bool, int, bool]]:
        """test_trained_model's episode loop for E device envs: one batched act() (eval-mode BatchNorm, noisy as every
        act() of the reference is) and one env step per vector step, nothing appended to the replay ring, no learning.
        Returns [(completed, last frame index, done)] — completed iff the episode ended `done` with reward == 250; an
        episode that used all `frames` steps without `done` is (False, frames - 1, False) (rl_framework.py:341-355); done
        and not completed = a collision — ordered by (env's episode ordinal, env)."""
        E = int(n_envs)
        quota = self._episode_quota(int(n_episodes), E)
        loop = DeviceEnvLoop(self.learner, None, E, seed=self.seed + 15485863 + 104729 * self.rank, max_frames=frames,
                             noise_scale=noise_scale, use_graph=self.use_graph, robot=robot, obstacle_jitter=obstacle_jitter,
                             preset=preset, variation=variation, records=True, drain_every=drain_every)
        seen = np.zeros(E, np.int64)
        results = []
        while (seen < quota).any():
            for _ in range(loop.drain_every):
                loop.step()
            for score, nfr, done, last_reward, env, ordinal, step in loop.drain(final=True):
                if seen[env] < quota[env]:
                    seen[env] += 1
                    results.append((ordinal, env, bool(done) and last_reward == 250, nfr - 1, bool(done)))
        results.sort()
        return [tuple(r[2:]) for r in results]

    def evaluate_host_vectorized(self, vec_env, n_episodes: int, noise_scale: float = 1.0) -> List[Tuple[bool, int, bool]]:
        """The same for E host environments in worker processes (frame budget = vec_env.max_frames)."""
        E = vec_env.E
        quota = self._episode_quota(int(n_episodes), E)
        actor = ActPath(self.learner, E, seed=(self.seed * 69069 + 11 + self.rank) & 0xFFFFFFFFFFFFFFFF, host_io=True)
        seen, ordinal, fr = np.zeros(E, np.int64), np.zeros(E, np.int64), np.zeros(E, np.int64)
        results = []
        obs = vec_env.reset()
        stream = torch.cuda.current_stream()
        while (seen < quota).any():
            if actor.host_io:
                actor.obs_np[...] = obs
                actor.act(noise_scale)
                stream.synchronize()
                actions = actor.actions_np
            else:
                actor.obs.copy_(torch.from_numpy(np.asarray(obs, np.float32)))
                actions = actor.act(noise_scale).cpu().numpy()
            _, _, rewards, _, dones, obs = vec_env.step(actions)
            fr += 1
            lost = list(getattr(vec_env, "respawned_envs", ()))
            if lost:
                fr[lost] = 0                                  # (a replaced worker: its envs' episodes in flight start over)
            for e in np.nonzero(vec_env.arr["episode_end"])[0]:
                ordinal[e] += 1
                if seen[e] < quota[e]:
                    seen[e] += 1
                    results.append((int(ordinal[e]), int(e), bool(dones[e]) and rewards[e] == 250, int(fr[e]) - 1,
                                    bool(dones[e])))
                fr[e] = 0
        results.sort()
        return [tuple(r[2:]) for r in results]
