#!/usr/bin/env python3
"""bench.py — env-steps/s + learn() updates/s of the NAF hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): KUKA IIWA 6-DoF shapes (S=21, A=6, H=256), 64 envs per GPU, batch 256,
HBM replay of 1e6 transitions (pre-filled, so the ring is full and evicting), HIP NAF head, Hadamard P and
truncated actions = the reference's semantics. One "step" = one vector-env step of the per-timestep hot path
(reference naf_algorithm.py:249-261 for 64 envs): act(64 states) -> env step -> 64 transitions appended ->
64 x (sample 256 + learn()), i.e. the reference's update-to-data ratio (update_freq = num_updates = 1).
The simulator is the on-device synthetic stand-in (PyBullet is not installable here) — labelled in `data`.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, gradients all-reduced over RCCL each update)

Prints ONE JSON line (rank 0). `value` = env-steps/s summed over all ranks (= learn() updates/s).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synth_rows(n, S, A, row_floats, off_s2, seed, device):
    """Transition rows of the BASELINE value ranges, generated on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rows = torch.zeros(n, row_floats, device=device)
    chunk = 1 << 18
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        s = torch.randn(m, S, generator=g, device=device).clamp_(-3.1416, 3.1416)
        s2 = (s + 0.05 * torch.randn(m, S, generator=g, device=device)).clamp_(-3.1416, 3.1416)
        a = torch.rand(m, A, generator=g, device=device) * 2 - 1
        sat = torch.rand(m, A, generator=g, device=device) < 0.10
        a = torch.where(sat, torch.sign(a), a)
        r = -1.5 * torch.rand(m, generator=g, device=device)
        ev = torch.rand(m, generator=g, device=device)
        r = torch.where(ev < 0.0025, torch.full_like(r, 250.0), r)
        r = torch.where((ev >= 0.0025) & (ev < 0.005), torch.full_like(r, -1000.0), r)
        blk = rows[lo:lo + m]
        blk[:, :S], blk[:, S:S + A], blk[:, S + A] = s, a, r
        blk[:, off_s2:off_s2 + S], blk[:, off_s2 + S] = s2, (ev < 0.005).float()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--buffer", type=int, default=1_000_000)
    ap.add_argument("--p-mode", choices=["hadamard", "matmul"], default="hadamard")
    ap.add_argument("--robot", choices=["kuka", "xarm6", "panda"], default="kuka",
                    help="shapes only: kuka/xarm6 S=21 A=6, panda S=23 A=7 (BASELINE configs[3], [4])")
    ap.add_argument("--obstacle-jitter", type=float, default=0.0, help="per-env obstacle randomisation (configs[3])")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # NAF_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo as the control plane — the N > 1 code path of this file on
    # a 1-GPU box (RCCL refuses two ranks on one device). Never a measurement; the JSON line says so.
    rehearsal = os.environ.get("NAF_BENCH_REHEARSAL") == "1" and world > 1
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from robotic_manipulator_rloa_amd import _lib
    from robotic_manipulator_rloa_amd.engine import DeviceEnvLoop, TrainChunk
    from robotic_manipulator_rloa_amd.learner import Learner
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer

    S, A = (23, 7) if args.robot == "panda" else (21, 6)
    H, B, E, N = 256, args.batch, args.envs, args.buffer
    p_mode = _lib.P_HADAMARD if args.p_mode == "hadamard" else _lib.P_MATMUL
    L = Learner(S, A, H, B, 1e-3, 1e-3, 0.99, dev, p_mode=p_mode, world_size=world)
    sd = reference_init_state_dict(S, A, H, seed=0)          # same seed on every rank: replicas start identical
    L.load_params(0, sd)
    L.load_params(1, sd)
    if world > 1:
        if rehearsal:
            host = L.theta2.cpu()
            dist.broadcast(host, src=0)
            L.theta2.copy_(host)
        else:
            dist.broadcast(L.theta2, src=0)
    replay = ReplayBuffer(N, B, dev, seed=1000 + rank, state_size=S, action_size=A)
    rows = synth_rows(N, S, A, replay.row_floats, replay.off_s2, seed=77 + rank, device=dev)
    replay.add_rows_device(rows, N)
    del rows
    loop = DeviceEnvLoop(L, replay, E, seed=31 + rank, max_frames=400, use_graph=not args.no_graph, robot=args.robot,
                         obstacle_jitter=args.obstacle_jitter)
    U = E   # update_freq = num_updates = 1: one learn() per env transition (naf_algorithm.py:147-156)
    chunk = TrainChunk(L, replay, U, use_graph=not args.no_graph, gather_outside_graph=True)
    graph_note = "hipGraph"
    try:
        if not args.no_graph:
            loop.capture()
            chunk.capture()
    except Exception as e:  # e.g. a collective that cannot be captured: run the same launches eagerly
        if rank == 0:
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
        loop.use_graph = chunk.use_graph = False
        graph_note = "eager"

    def one_step():
        loop.step()
        chunk.run()

    for _ in range(args.warmup):
        one_step()
    # ---- timed region: exactly K steps between barrier + synchronize on both sides -------------------------
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev0 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        chunk.gather_events, chunk.empty_events = ev[k], ev0[k]
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    chunk.gather_events = chunk.empty_events = None
    t = torch.tensor([elapsed], device="cpu" if rehearsal else dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    gather_bracket_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    # what an EMPTY event bracket reads on this stream: the part of every bracket that is not the kernel. Recorded in the
    # timed loop itself, right behind each gather bracket (same stream, same surroundings), averaged the same way
    event_overhead_ms = sum(a.elapsed_time(b) for a, b in ev0) / len(ev0)
    gather_ms = max(gather_bracket_ms - event_overhead_ms, 1e-6)

    finite = bool(torch.isfinite(L.theta2).all().item())
    bad = replay.bad_index_count()
    env_steps = args.steps * E * world
    updates = args.steps * U * world
    value = env_steps / elapsed

    out = {
        "metric": "env-steps/s (= learn() updates/s at update_freq=1,num_updates=1), KUKA 6-DoF NAF batch=256",
        "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic (on-device kinematic stand-in env; replay pre-filled "
        "with synthetic transitions; random-init weights, seed 0)",
        "config": {"workload": f"{'configs[1]: ' if (args.robot, B, N, E) == ('kuka', 256, 1000000, 64) else ''}{args.robot} shapes "
                               f"S={S} A={A} H=256, {E} envs/GPU, batch {B}, HBM replay {N}, "
                               f"HIP NAF head ({args.p_mode} P), {U} learn() per vector step",
                   "launch": graph_note + (" [REHEARSAL: all ranks share cuda:0, not a measurement]" if rehearsal else ""),
                   "parallelism": f"dp{world}" if world > 1 else "single",
                   "grad_exchange": ("none" if world == 1 else
                                     "one-shot peer-memory all-reduce over xGMI (csrc/xgmi_reduce.hip)" if L.xgmi is not None
                                     else "RCCL all-reduce")},
        "updates_per_s": round(updates / elapsed, 1),
        "sanity": {"params_finite": finite, "bad_replay_indices": bad, "optimizer_steps": int(L.step_dev.item())},
    }
    if L.xgmi is not None:
        out["sanity"]["xgmi_allreduces"], out["sanity"]["xgmi_timed_out_waits"] = L.xgmi.status()
    if world > 1:
        # data-parallel replicas must have stayed in lock-step: same parameters, bit for bit, on every rank
        chk = torch.stack([L.theta2.double().sum(), L.theta2.double().abs().sum(), L.adam_v.double().sum()])
        chk = chk.cpu() if rehearsal else chk
        every = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(every, chk)
        out["sanity"]["replicas_identical"] = all(bool(torch.equal(e, every[0])) for e in every)
    # ---- roofline of the replay gather (the kernel north_star names), measured live with events -----------
    rows_per_launch = U * B
    alg_bytes = rows_per_launch * (4 * (2 * S + A + 2) * 2 + 4)       # 400 B/row read+written + 4 B index
    out["roofline"] = {"kernel": "replay_gather_rows_kernel", "bound": "hbm",
                       "achieved": round(alg_bytes / (gather_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                       "frac": round(alg_bytes / (gather_ms * 1e-3) / 8e12, 4), "traffic": None,
                       "rows_per_launch": rows_per_launch, "alg_bytes_per_launch": alg_bytes,
                       "avg_launch_ms": round(gather_ms, 5), "avg_event_bracket_ms": round(gather_bracket_ms, 5),
                       "empty_event_bracket_ms": round(event_overhead_ms, 5),
                       "note": "one launch per vector step inside the timed loop, bracketed by HIP events on its stream; "
                               "avg_launch_ms = bracket - empty bracket (an empty bracket is recorded right behind every gather bracket, inside the timed loop); rocprofv3 --kernel-trace average of the same "
                               "command: profiles/r01_bench_kernel_stats.csv (replay_gather_rows_kernel<1, 0>)"}
    out["roofline"]["traffic"] = pmc_traffic(rows_per_launch)
    if rank == 0 and world == 1:
        out["roofline_bulk"] = bulk_gather(replay, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.torch_cpu_port import time_baseline
        # the reference path is dispatch-bound (~1400 aten calls per update): more threads do not help and torch's
        # default on a 256-cpu host (128 threads) is pathologically slow. Time it at 8 threads and at 1, keep the best.
        runs = [time_baseline(S, A, H, B, N, budget_s=args.cpu_budget * 0.6, threads=8),
                time_baseline(S, A, H, B, N, budget_s=args.cpu_budget * 0.4, threads=1)]
        cb = max(runs, key=lambda d: d["steps_per_s"])
        cb["other"] = {f"threads={d['threads']}": round(d["steps_per_s"], 2) for d in runs}
        out["cpu_baseline"] = {"value": round(cb["steps_per_s"], 2), "unit": "env-steps/s", "cores": cb["threads"],
                               "kind": "port",
                               "sample": f"oracle/torch_cpu_port.py (reference op order incl. deque+random.sample "
                                         f"sampler), act+add+sample+learn for {cb['n_steps']} timesteps, B={B}, "
                                         f"deque filled to N={N}; learn()-only {cb['learn_updates_per_s']:.1f} "
                                         f"updates/s over {cb['n_learn']} calls; host has {cb['host_cpus']} cpus; "
                                         f"steps/s by thread count {cb['other']}",
                               "learn_only_updates_per_s": round(cb["learn_updates_per_s"], 2)}
        out["speedup_vs_cpu_port"] = round(value / cb["steps_per_s"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(rows_per_launch):
    """HBM bytes per launch from the rocprofv3 PMC passes recorded in profiles/gather_traffic.json (FETCH_SIZE doubled
    per the gfx950 correction + WRITE_SIZE); None when no pass was recorded for this launch size."""
    try:
        with open(os.path.join(ROOT, "profiles", "gather_traffic.json")) as f:
            for rec in json.load(f)["launches"]:
                if rec["rows_per_launch"] == rows_per_launch:
                    return int((2 * rec["fetch_size_kb"] + rec["write_size_kb"]) * 1024)
    except (OSError, KeyError, ValueError):
        pass
    return None


def bulk_gather(replay, dev, n_rows=1 << 22, reps=20):
    """The same gather kernel on a launch big enough to be bandwidth- instead of latency-bound: 4 Mi uniformly
    random rows (1.07 GB of row traffic per launch) out of the 1e6-row (256 MB) ring."""
    idx = torch.randint(0, len(replay), (n_rows,), device=dev, dtype=torch.int32)
    out = torch.empty(n_rows, replay.row_floats, device=dev)
    for _ in range(3):
        replay.gather_rows(idx, out, n_rows)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        replay.gather_rows(idx, out, n_rows)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    alg = n_rows * (4 * (2 * replay.S + replay.A + 2) * 2 + 4)
    phys = n_rows * (replay.row_floats * 4 * 2 + 4)
    return {"kernel": "replay_gather_rows_kernel", "rows_per_launch": n_rows, "avg_launch_ms": round(ms, 4),
            "traffic": pmc_traffic(n_rows), "alg_bytes_per_launch": alg,
            "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / 8e12, 4),
            "physical_GBps": round(phys / (ms * 1e-3) / 1e9, 1)}


if __name__ == "__main__":
    main()
