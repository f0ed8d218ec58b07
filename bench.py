#!/usr/bin/env python3
"""bench.py — env-steps/s + learn() updates/s of the NAF hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): KUKA IIWA 6-DoF shapes (S=21, A=6, H=256), 64 envs per GPU, batch 256,
HBM replay of 1e6 transitions (pre-filled, so the ring is full and evicting), HIP NAF head, Hadamard P and
truncated actions = the reference's semantics. One "step" = one vector-env step of the per-timestep hot path
(reference naf_algorithm.py:249-261 for 64 envs): act(64 states) -> env step -> 64 transitions appended ->
64 x (sample 256 + learn()), i.e. the reference's update-to-data ratio (update_freq = num_updates = 1).
The simulator is the on-device synthetic stand-in (PyBullet is not installable here) — labelled in `data`.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU, gradients all-reduced once per update (one-shot peer-memory exchange over xGMI, RCCL as the
fallback). Launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK /
WORLD_SIZE in the environment) — or plainly as `python bench.py --gpus N`: without WORLD_SIZE this process touches no
GPU, starts that launcher as a child process, relays its output and exits with its code.

Prints ONE JSON line (rank 0). `value` = env-steps/s summed over all ranks (= learn() updates/s).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (about 6.3 TB/s achievable)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000, help="timed vector steps (4000 x 64 updates: about 10 s)")
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--buffer", type=int, default=1_000_000)
    ap.add_argument("--p-mode", choices=["hadamard", "matmul"], default="hadamard")
    ap.add_argument("--robot", choices=["kuka", "xarm6", "xarm6_robot", "panda"], default="kuka",
                    help="preset of the on-device env and the shapes: kuka / xarm6 / xarm6_robot S=21 A=6, panda S=23 A=7 "
                         "(BASELINE configs[3]: --robot xarm6_robot --batch 1024 --obstacle-jitter 0.1; configs[4]: --robot "
                         "panda --batch 2048 --buffer 4000000)")
    ap.add_argument("--obstacle-jitter", type=float, default=0.0, help="per-env obstacle randomisation (configs[3])")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the host-vector-env and reference-API-path measurements (extra keys, outside `value`)")
    ap.add_argument("--roofline-ring", type=int, default=4_000_000,
                    help="rows of the ring the bulk gather roofline runs on (4e6 x 256 B = 1.02 GB: beyond the 256 MiB "
                         "Infinity Cache, BASELINE configs[4]'s ring); 0 = skip")
    ap.add_argument("--roofline-hbm-ring", type=int, default=16_000_000,
                    help="rows of a second, larger ring for the same bulk launch (16e6 x 256 B = 4.1 GB = 16 x the Infinity "
                         "Cache: `roofline.hbm_only_frac`); <= --roofline-ring = skip")
    ap.add_argument("--roofline-rows", type=int, default=1 << 24,
                    help="rows gathered per bulk launch (16 Mi: 1.1 ms per launch; the ~9 us of ramp and tail of a launch are 3 %% of a 4 Mi one)")
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start torch.distributed.run as a CHILD process
    (this parent has made no HIP call: counting devices does not initialise the GPU on this image) and hand back its
    exit code. The ranks print; rank 0's JSON line is the last line of stdout."""
    import torch
    have = torch.cuda.device_count()
    rehearsal = os.environ.get("NAF_BENCH_REHEARSAL") == "1"
    if have < args.gpus and not rehearsal:
        print(f"[bench] --gpus {args.gpus} but this host shows {have} GPU(s). Refusing to report a {args.gpus}-GPU number "
              f"from fewer devices (NAF_BENCH_REHEARSAL=1 runs every rank on cuda:0 to exercise the N > 1 code path; "
              f"its line is labelled, never a measurement).", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def synth_rows(n, S, A, row_floats, off_s2, seed, device):
    """Transition rows of the BASELINE value ranges, generated on the device."""
    import torch
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
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}, or "
                         f"plain `python bench.py --gpus {args.gpus}`, which starts that launcher itself)")
    # NAF_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo as the control plane — the N > 1 code path of this file on
    # a 1-GPU box (RCCL refuses two ranks on one device). Never a measurement; the JSON line says so.
    rehearsal = os.environ.get("NAF_BENCH_REHEARSAL") == "1" and world > 1
    if world > 1 and not rehearsal and torch.cuda.device_count() < world:
        raise SystemExit(f"{world} ranks but {torch.cuda.device_count()} GPU(s) visible")
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
    # records=True: the per-env episode bookkeeping (score, frames, record ring drained every 64 vector steps — what
    # NAFAgent.run_vectorized builds its {episode: (score, last_frame)} dict and checkpoints from) rides in the timed loop
    loop = DeviceEnvLoop(L, replay, E, seed=31 + rank, max_frames=400, use_graph=not args.no_graph, robot=args.robot,
                         obstacle_jitter=args.obstacle_jitter, records=True, drain_every=64)
    U = E   # update_freq = num_updates = 1: one learn() per env transition (naf_algorithm.py:147-156)
    # sample -> gather -> U updates as ONE graph per vector step (the launch of the gather alone is bracketed with events in a
    # short loop of its own behind the timed region: `roofline_live`)
    chunk = TrainChunk(L, replay, U, use_graph=not args.no_graph)
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
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device="cpu" if rehearsal else dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    finite = bool(torch.isfinite(L.theta2).all().item())
    opt_steps = int(L.step_dev.item())
    episodes = loop.drain(final=True)            # every episode the E envs finished during warm-up + timed steps
    # the gather launch of a vector step (U*B rows) on its own, bracketed by HIP events on the launching stream: the same
    # sample + gather the graph holds, launched eagerly 200 times behind the timed region (the learner is not touched)
    n_ev = 200
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    for k in range(n_ev):
        chunk.gather_events = ev[k]
        chunk._sample_gather()
    torch.cuda.synchronize()
    chunk.gather_events = None
    gather_bracket_ms = sum(a.elapsed_time(b) for a, b in ev) / n_ev

    bad = replay.bad_index_count()
    env_steps = args.steps * E * world
    updates = args.steps * U * world
    value = env_steps / elapsed

    is_cfg1 = (args.robot, B, N, E) == ("kuka", 256, 1000000, 64)
    out = {
        "metric": "env-steps/s (= learn() updates/s at update_freq=1,num_updates=1), KUKA 6-DoF NAF batch=256",
        "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic (on-device kinematic stand-in env; replay pre-filled "
        "with synthetic transitions; random-init weights, seed 0)",
        "config": {"workload": f"{'configs[1]: ' if is_cfg1 else ''}{args.robot} shapes "
                               f"S={S} A={A} H=256, {E} envs/GPU, batch {B}, HBM replay {N}, "
                               f"HIP NAF head ({args.p_mode} P), {U} learn() per vector step",
                   "launch": graph_note + (" [REHEARSAL: all ranks share cuda:0, not a measurement]" if rehearsal else ""),
                   "parallelism": f"dp{world}" if world > 1 else "single",
                   "chain": L.chain, "fused_kernels": ",".join(sorted(L.fuse)),
                   "grad_exchange": {"none": "none",
                                     "merged": "one-shot peer-memory exchange inside the finish launch (csrc/big_batch.hip)",
                                     "oneshot": "one-shot peer-memory all-reduce over xGMI, a launch of its own (csrc/xgmi_reduce.hip)",
                                     "rccl": "RCCL all-reduce (torch.distributed) + norm launch"}[L.exchange]},
        "updates_per_s": round(updates / elapsed, 1),
        "us_per_update": round(1e6 * elapsed / (args.steps * U), 3),
        "timed_seconds": round(elapsed, 3),
        "sanity": {"params_finite": finite, "bad_replay_indices": bad, "optimizer_steps": opt_steps, "fold_fallbacks": L.fold_fallbacks,
                   "episodes_booked": len(episodes),
                   "episode_frames_booked": int(sum(e[1] for e in episodes))},
    }
    if L.xgmi is not None:
        out["sanity"]["xgmi_allreduces"], out["sanity"]["xgmi_timed_out_waits"] = L.xgmi.status()
    if world > 1:
        # what a SCALE record needs to be read without a second run: which collective back-end and world size the job saw, whether
        # every rank could map every peer's receive slab (hipIpc) and pass the exact self-test, what each form of the exchange took
        # per update ON THIS NODE at start-up (Learner.autotune_exchange; MAX over ranks) and which one the ranks agreed on
        at = L.exchange_autotune or {}
        out["preflight"] = {
            "backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "devices_visible": torch.cuda.device_count(),
            "rehearsal_one_gpu": bool(rehearsal),
            "hipipc_peer_slabs_mapped_and_self_test_passed": L.xgmi is not None,
            "xgmi_slab_memory": getattr(L.xgmi, "mem_kind", None),
            "exchange_forms_available": L.exchange_forms(),
            "exchange_us_per_update_at_startup": {k: v for k, v in at.items() if k in ("oneshot", "merged", "rccl")} or None,
            "exchange_forms_that_failed_at_startup": at.get("errors"),
            "exchange_chosen": L.exchange,
            "exchange_pinned_by_env": os.environ.get("NAF_DP_EXCHANGE", "auto") != "auto",
            "xgmi_timed_out_waits": out["sanity"].get("xgmi_timed_out_waits", 0),
        }
    if world > 1:
        # data-parallel replicas must have stayed in lock-step: same parameters, bit for bit, on every rank
        chk = torch.stack([L.theta2.double().sum(), L.theta2.double().abs().sum(), L.adam_v.double().sum()])
        chk = chk.cpu() if rehearsal else chk
        every = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(every, chk)
        out["sanity"]["replicas_identical"] = all(bool(torch.equal(e, every[0])) for e in every)
    # ---- roofline of the replay gather (the kernel north_star names) ---------------------------------------
    # headline: a launch big enough to be bandwidth-bound, on a ring bigger than the Infinity Cache. HIP events on the
    # stream the kernel is launched on, around `reps` back-to-back launches; profiles/ holds the rocprofv3 kernel-trace
    # stats of this very command, whose average for the same kernel instance must agree.
    row_alg = 4 * (2 * S + A + 2) * 2 + 4                     # 400 B/row read+written + 4 B index
    if rank == 0 and args.roofline_ring > 0:
        del loop, chunk
        out["roofline"] = bulk_gather_roofline(S, A, args.roofline_ring, args.roofline_rows, dev, row_alg)
        # the same launch on a ring 16 times the Infinity Cache (16e6 rows = 4.1 GB: at most 1 / 16 of the row fetches can be
        # served by the 256-MiB LLC): the figure that is HBM and nothing else
        if args.roofline_hbm_ring > args.roofline_ring:
            far = bulk_gather_roofline(S, A, args.roofline_hbm_ring, args.roofline_rows, dev, row_alg)
            out["roofline"]["hbm_only_frac"] = far["frac"]
            out["roofline"]["hbm_only"] = {k: far[k] for k in ("achieved", "avg_launch_ms", "ring_rows", "ring_bytes",
                                                                "infinity_cache_share", "physical_GBps", "bad_indices", "traffic")}
            # both measurements launch the same kernel instance the same number of times (3 warm-up + 20 timed each): a
            # rocprofv3 --kernel-trace --stats summary of this command averages the two populations under ONE name
            near, n_each = out["roofline"], 3 + out["roofline"]["launches_timed"]
            out["roofline"]["kernel_stats_check"] = {
                "calls": 2 * n_each, "expected_avg_ms": round((near["avg_launch_ms"] + far["avg_launch_ms"]) / 2, 5),
                "what": "the stats file's average for this kernel instance = the mean of `avg_launch_ms` (ring of "
                        f"{near['ring_rows']} rows) and `hbm_only.avg_launch_ms` (ring of {far['ring_rows']} rows), {n_each} launches "
                        "each; its MinDuration belongs to the smaller ring"}
    # the launch that sits in the timed loop (one per vector step, U*B rows): latency, not bandwidth — reported as the
    # RAW event bracket (an empty bracket reads ~5 us on this stream, so this overstates the kernel; the profiler's
    # kernel-trace average is the number to quote for it)
    rows_live = U * B
    out["roofline_live"] = {"kernel": "replay_gather_rows_kernel<1, 0, W4>", "rows_per_launch": rows_live,
                            "alg_bytes_per_launch": rows_live * row_alg,
                            "avg_event_bracket_ms": round(gather_bracket_ms, 5),
                            "achieved": round(rows_live * row_alg / (gather_bracket_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                            "frac": round(rows_live * row_alg / (gather_bracket_ms * 1e-3) / (PEAK_HBM_GBPS * 1e9), 4),
                            "note": "latency-bound launch; raw HIP-event bracket of the same sample + gather launched eagerly 200 times behind "
                                    "the timed region (inside it they are nodes of the step's graph), no overhead subtracted"}
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
                                         f"updates/s over {cb['n_learn']} calls; sample() alone {cb['sample_ms']:.2f} ms (deque "
                                         f"indexing: memory latency, the host-dependent part); host has {cb['host_cpus']} cpus; "
                                         f"steps/s by thread count {cb['other']}",
                               "learn_only_updates_per_s": round(cb["learn_updates_per_s"], 2),
                               "sample_only_ms": round(cb["sample_ms"], 3)}
        out["speedup_vs_cpu_port"] = round(value / cb["steps_per_s"], 1)
    if rank == 0 and world == 1 and not args.no_extras:
        try:
            out.update(extras(dev, args))
            out["sanity"]["per_timestep_path"] = out.pop("_api_path_sanity", None)
        except Exception as e:                                   # extra keys never cost the headline line
            out["extras_error"] = f"{type(e).__name__}: {e}"
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(rows_per_launch, ring_rows):
    """HBM bytes per launch as RECORDED by separate rocprofv3 --pmc passes over this command (FETCH_SIZE doubled per the
    gfx950 correction + WRITE_SIZE; profiles/gather_traffic.json names the CSVs). Not measurable from inside the run:
    None when no pass was recorded for this launch shape."""
    try:
        with open(os.path.join(ROOT, "profiles", "gather_traffic.json")) as f:
            for rec in json.load(f)["launches"]:
                if rec["rows_per_launch"] == rows_per_launch and rec.get("ring_rows") == ring_rows:
                    return int((2 * rec["fetch_size_kb"] + rec["write_size_kb"]) * 1024), rec.get("source")
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def bulk_gather_roofline(S, A, ring_rows, n_rows, dev, row_alg, reps=20):
    """replay_gather_rows_kernel<4, 2, W4> (the bulk instance: 4 float4 in flight per lane, nontemporal stores): n_rows
    uniformly random positions out of a full ring of `ring_rows` 256-B rows, gathered into packed minibatch rows."""
    import torch
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    ring = ReplayBuffer(ring_rows, 256, dev, seed=5, state_size=S, action_size=A)
    rows = synth_rows(ring_rows, S, A, ring.row_floats, ring.off_s2, seed=99, device=dev)
    ring.add_rows_device(rows, ring_rows)
    del rows
    brf = ring.batch_row_floats
    idx = torch.randint(0, ring_rows, (n_rows,), device=dev, dtype=torch.int32)
    out = torch.empty(n_rows, brf, device=dev)
    for _ in range(3):
        ring.gather_rows(idx, out, n_rows)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        ring.gather_rows(idx, out, n_rows)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    bad = ring.bad_index_count()
    alg = n_rows * row_alg
    phys = n_rows * (ring.row_floats * 4 + brf * 4 + 4)
    traffic, source = pmc_traffic(n_rows, ring_rows)
    w4 = brf // 4
    return {"kernel": f"replay_gather_rows_kernel<4, 2, {w4}>", "bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 1),
            "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / (PEAK_HBM_GBPS * 1e9), 4),
            "traffic": traffic, "traffic_source": source, "rows_per_launch": n_rows, "alg_bytes_per_launch": alg,
            "alg_bytes_per_row": row_alg, "physical_bytes_per_launch": phys, "avg_launch_ms": round(ms, 5),
            "launches_timed": reps, "ring_rows": ring_rows, "ring_bytes": ring_rows * ring.row_floats * 4,
            # the share of the ring the 256-MiB Infinity Cache can hold: an upper bound of the share of row fetches it can
            # serve (uniformly random rows, each fetched n_rows / ring_rows times per launch); FETCH_SIZE counts those hits too
            "infinity_cache_share": round(min(1.0, 256 * 2 ** 20 / (ring_rows * ring.row_floats * 4)), 4), "bad_indices": bad,
            "physical_GBps": round(phys / (ms * 1e-3) / 1e9, 1),
            "note": "HIP events around back-to-back launches on the launching stream; achieved = algorithmic bytes "
                    "(4*(2S+A+2)*2 + 4 per row: 404 B at S=21/A=6) / average launch time; profiles/r06_ring4e6_kernel_stats.csv / r06_ring16e6_kernel_stats.csv are "
                    "the rocprofv3 kernel-trace averages of this kernel instance, each ring in a process of its own. A ring row is padded "
                    "200 -> 256 B (two whole 128-B lines per random row) and a gathered row 200 -> 208 B, so the launch moves "
                    "1.16 x its algorithmic bytes; `traffic` is the PMC record of that (FETCH_SIZE x 2 + WRITE_SIZE). NOT an "
                    "HBM-only figure on this ring: up to `infinity_cache_share` of the row fetches can hit the 256-MiB LLC "
                    "(configs[4]'s real ring, so the number is the workload's); `hbm_only_frac` is the same launch on a ring 16 x "
                    "the LLC"}


def measure_shape(dev, robot, B, N, E, steps, warmup, jitter=0.0, p_mode="hadamard", fuse=None, layer=256):
    """The timed loop of main() at another BASELINE shape on this one GPU (same engines, same update-to-data ratio):
    updates/s of E envs, batch B, ring N."""
    import torch
    from robotic_manipulator_rloa_amd import _lib
    from robotic_manipulator_rloa_amd.engine import DeviceEnvLoop, TrainChunk
    from robotic_manipulator_rloa_amd.learner import Learner
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    S, A = (23, 7) if robot == "panda" else (21, 6)
    L = Learner(S, A, layer, B, 1e-3, 1e-3, 0.99, dev, p_mode=_lib.P_HADAMARD if p_mode == "hadamard" else _lib.P_MATMUL, fuse=fuse)
    sd = reference_init_state_dict(S, A, layer, seed=0)
    L.load_params(0, sd)
    L.load_params(1, sd)
    replay = ReplayBuffer(N, B, dev, seed=1000, state_size=S, action_size=A)
    rows = synth_rows(N, S, A, replay.row_floats, replay.off_s2, seed=77, device=dev)
    replay.add_rows_device(rows, N)
    del rows
    loop = DeviceEnvLoop(L, replay, E, seed=31, max_frames=400, robot=robot, obstacle_jitter=jitter, records=True)
    chunk = TrainChunk(L, replay, E)
    loop.capture()
    chunk.capture()
    for _ in range(warmup):
        loop.step()
        chunk.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loop.step()
        chunk.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(torch.isfinite(L.theta2).all().item()) and replay.bad_index_count() == 0 and int(L.step_dev.item()) == (steps + warmup) * E
    return {"updates_per_s": round(steps * E / dt, 1), "us_per_update": round(1e6 * dt / (steps * E), 2), "steps": steps,
            "timed_seconds": round(dt, 3), "sane": ok, "chain": L.chain, "fused_kernels": ",".join(sorted(L.fuse))}


def measure_joints(dev, A, B, N, U, reps, fuse=None):
    """learn() updates/s of an arm of A joints (state 9 + 2 A floats, environment.py:261) at batch B: chunks of U graph-replayed
    updates on a ring of N rows — without the device env loop, which models arms of up to 8 joints."""
    import warnings
    import torch
    from robotic_manipulator_rloa_amd.engine import TrainChunk
    from robotic_manipulator_rloa_amd.learner import Learner
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    S = 9 + 2 * A
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        L = Learner(S, A, 256, B, 1e-3, 1e-3, 0.99, dev, fuse=fuse)
    sd = reference_init_state_dict(S, A, 256, seed=0)
    L.load_params(0, sd)
    L.load_params(1, sd)
    replay = ReplayBuffer(N, B, dev, seed=1000, state_size=S, action_size=A)
    replay.add_rows_device(synth_rows(N, S, A, replay.row_floats, replay.off_s2, seed=77, device=dev), N)
    chunk = TrainChunk(L, replay, U)
    chunk.capture()
    for _ in range(5):
        chunk.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        chunk.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(torch.isfinite(L.theta2).all().item()) and replay.bad_index_count() == 0 and int(L.step_dev.item()) == (reps + 5) * U
    return {"updates_per_s": round(reps * U / dt, 1), "us_per_update": round(1e6 * dt / (reps * U), 2), "updates": reps * U,
            "timed_seconds": round(dt, 3), "sane": ok, "chain": L.chain, "state_size": S, "what": "learn() only (no device env loop)"}


def extras(dev, args):
    """Measurements outside `value`: (i) the host vector env — E=64 environments in worker
    processes around the GPU learner, synchronous and asynchronous policy; (ii) the reference-API path — one host env,
    NAFAgent.act / env.step / NAFAgent.step per timestep at configs[0]'s B=64 / N=1e5 — beside the CPU port at the same
    B / N. The env is the synthetic kinematic stand-in (no PyBullet in the image)."""
    import logging
    import tempfile
    import numpy as np
    import torch
    from functools import partial
    from robotic_manipulator_rloa_amd.environment.synthetic import SyntheticEnvironment
    from robotic_manipulator_rloa_amd.environment.vector_env import HostVectorEnv
    from robotic_manipulator_rloa_amd.naf_components.naf_algorithm import NAFAgent
    logging.getLogger('robotic_manipulator_rloa.utils.logger').setLevel(40)
    res = {}
    old = os.getcwd()
    os.chdir(tempfile.mkdtemp(prefix="naf_bench_"))
    try:
        S, A, E = 21, 6, 64
        # (ii) reference-API path: one host env, NAFAgent.act -> env.step -> NAFAgent.step per timestep (naf_algorithm.py:249-261):
        # configs[0]'s literal shape (B = 64, ring 1e5) and the same loop at configs[1]'s batch (B = 256). Steady state as
        # SURVEY.md section 8(d) defines it: the ring filled to capacity before anything is timed (the sampler's redraw rounds and
        # the gather's locality are then those of a long run, not of its first thousand steps).
        def api_path(batch, n_api, env_us=0.0):
            env = SyntheticEnvironment(A)
            agent = NAFAgent(env, S, A, 256, batch, 100000, 1e-3, 1e-3, 0.99, 1, 1, 500, dev, 0)
            m = agent.memory
            rng = np.random.default_rng(1)
            r = np.zeros((100000, m.row_floats), np.float32)
            r[:, :S] = rng.standard_normal((100000, S))
            r[:, S:S + A] = rng.uniform(-1, 1, (100000, A))
            r[:, S + A] = -rng.random(100000)
            r[:, m.off_s2:m.off_s2 + S] = r[:, :S] + 0.05 * rng.standard_normal((100000, S))
            m.add_rows_device(torch.from_numpy(r).to(dev), 100000)
            state = env.reset(False)

            def steps(n, state):
                for _ in range(n):
                    a = agent.act(state)
                    t_env = time.perf_counter()
                    nxt, r_, d = env.step(a)
                    while env_us and (time.perf_counter() - t_env) * 1e6 < env_us:      # (a slower environment: busy waiting)
                        pass
                    agent.step(state, a, r_, nxt, d)
                    state = env.reset(False) if d else nxt
                return state
            state = steps(300, state)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            state = steps(n_api, state)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ch = agent._chunk
            launches = 7 if (ch is not None and ch.fused_prep and ch.fused_tail) else 12
            out = {"value": round(n_api / dt, 1), "unit": "timesteps/s", "timesteps": n_api, "batch": batch, "ring": "1e5 rows, full",
                   "launches_per_timestep": launches, "optimizer_steps": int(agent.learner.step_dev.item())}
            if ch is not None and getattr(ch, "pipelined", False):
                # the pipelined form: naf_adam_polyak_act_layer1 (the waiting gradient's step, act(), commit, and layer 1 of the chain in
                # extra workgroups) + the other four launches of the chain on a minibatch prefetched two timesteps ago = 5 launches
                # in the graph (6 with NAF_STEP_L1_RIDE=0), and the append + depth-2 prefetch as ONE launch beside the graph on a
                # stream of its own; a timestep whose prefetches did not hold starts over with the 13-launch graph
                runs = max(1, ch.fast_runs + ch.slow_runs)
                in_graph = 5 if getattr(ch.pipe, "l1_ride", False) else 6
                out["launches_per_timestep"] = round((in_graph * ch.fast_runs + 13 * ch.slow_runs) / runs, 2)
                out["launches_beside_the_graph_per_timestep"] = round(ch.fast_runs / runs, 2)
                out["pipelined"] = {"timesteps_on_the_prefetched_minibatch": ch.fast_runs, "timesteps_that_started_over": ch.slow_runs,
                                    # (ticks whose prefetch was launched BEFORE the graph because the host had waited for the last
                                    #  verdict: DESIGN 4d, "two stable states")
                                    "timesteps_with_the_prefetch_launched_first": getattr(ch.pipe, "side_first_runs", 0)}
            if ch is not None and hasattr(ch, "error_words"):
                # the path's hand-overs fail loudly: polls inside naf_adam_polyak_act that ran into their bound, timesteps whose
                # record did not hold on the device although the host had read that it does, host-side waits that had to synchronise
                out["errors"] = ch.error_words()
                out["host_store_hand_over"] = ("device memory the host stores into (self-test passed)" if ch.head_dev is not None
                                               else "pinned host memory (no large BAR, NAF_HOST_STORE=0, or the self-test failed)")
            del agent
            return out
        n_api = 3000
        res["reference_api_path"] = api_path(64, n_api)
        res["reference_api_path"]["what"] = ("NAFAgent.act + env.step + NAFAgent.step (add, sample, learn) per timestep, one host env "
                                             "(numpy stand-in), B=64, N=1e5 (configs[0] shape), naf_algorithm.py:249-261")
        res["reference_api_path_b256"] = api_path(256, n_api)
        res["reference_api_path_b256"]["what"] = "the same loop at configs[1]'s batch (B=256, N=1e5)"
        # ... and with an environment that takes 100 us per step (a real simulator takes at least that): what the framework adds to a
        # timestep once the learn() chain runs while the host steps the environment
        slow = api_path(256, 2000, env_us=100.0)
        slow["what"] = "the same loop (B=256) with env.step padded to 100 us by busy waiting"
        slow["framework_us_per_timestep"] = round(1e6 / slow["value"] - 100.0, 1)
        res["reference_api_path_b256_env_100us"] = slow
        res["_api_path_sanity"] = {k: {**res[k].get("errors", {}), **res[k].get("pipelined", {})}
                                   for k in ("reference_api_path", "reference_api_path_b256", "reference_api_path_b256_env_100us")}
        if not args.no_cpu_baseline:
            from oracle.torch_cpu_port import time_baseline
            cb = time_baseline(S, A, 256, 64, 100000, budget_s=6.0, threads=8)
            res["reference_api_path"]["cpu_port_same_shape"] = {"value": round(cb["steps_per_s"], 1), "unit": "timesteps/s",
                                                                "cores": cb["threads"], "env": "none (agent only)"}
        # (i) host vector env: 64 envs in worker processes (PyBullet would need one per process; the light stand-in
        # shares workers), B=256, ring 1e5 pre-filled past the `len > batch` gate by the first vector steps
        from robotic_manipulator_rloa_amd.environment.vector_env import usable_cpus
        workers = max(1, min(E, usable_cpus() // 2))               # (the container's share, not the machine's 256)
        per = (E + workers - 1) // workers
        hv = {}
        for mode in (False, True):
            agent = NAFAgent(None, S, A, 256, 256, 100000, 1e-3, 1e-3, 0.99, 1, 1, 500, dev, 0)
            vec = HostVectorEnv(partial(SyntheticEnvironment, A), E, S, A, envs_per_worker=per, max_frames=400, seed=1)
            try:
                agent.run_host_vectorized(vec, 20, async_policy=mode)               # fills 1280 rows, captures graphs
                r = agent.run_host_vectorized(vec, 150, async_policy=mode)
            finally:
                vec.close()
            hv["async_policy" if mode else "sync_policy"] = round(r["env_steps_per_s"], 1)
            del agent
        res["host_vector_env"] = {"unit": "env-steps/s", **hv, "envs": E, "worker_processes": (E + per - 1) // per,
                                  "what": "NAFAgent.run_host_vectorized: batched act() on the GPU -> E host envs step in "
                                          "worker processes (shared memory) -> E rows over PCIe -> HBM ring -> E learn() "
                                          "updates; env = numpy kinematic stand-in (PyBullet absent: labelled, N2)"}
        # (iii) the other single-GPU shapes of BASELINE.json's configs (their 8-GPU aspect is the driver's to run): the same
        # loop at configs[3]'s batch (xarm6_robot, per-env obstacle jitter, B = 1024) and configs[4]'s (panda 7x7 tiles,
        # B = 2048, ring 4e6) — only when the line itself is configs[1]
        if (args.robot, args.batch, args.buffer, args.envs) == ("kuka", 256, 1000000, 64):
            res["other_configs"] = {
                "configs[3] shape: xarm6_robot S=21 A=6, batch 1024, ring 1e6, obstacle jitter 0.1, 1 GPU":
                    measure_shape(dev, "xarm6_robot", 1024, 1000000, E, 500, 30, jitter=0.1),
                "configs[4] shape: panda S=23 A=7, batch 2048, ring 4e6, 1 GPU":
                    measure_shape(dev, "panda", 2048, 4000000, E, 400, 30),
                "batch 512 (kuka, ring 1e6)": measure_shape(dev, "kuka", 512, 1000000, E, 500, 30),
                # small batches: configs[0]'s batch and ring with 64 device envs, and the reference's default batch
                # (rl_framework.py:33-44) — the row-split chain since the end of round 3; batch sizes that are not whole 64-row blocks
                # or whole 16-row groups (round 4: partial last block / workgroup / MFMA tile on the row-split chain) and one below
                # 64 (a single partial block), beside the column-tile chain at that size
                "configs[0] batch and ring: kuka, batch 64, ring 1e5, 64 device envs": measure_shape(dev, "kuka", 64, 100000, E, 500, 30),
                "reference default batch 128 (kuka, ring 1e5)": measure_shape(dev, "kuka", 128, 100000, E, 500, 30),
                "batch 100 (kuka, ring 1e5): the row-split chain with a partial last 16-row group": measure_shape(dev, "kuka", 100, 100000, E, 500, 30),
                "batch 1000 (kuka, ring 1e6): the same at a large batch": measure_shape(dev, "kuka", 1000, 1000000, E, 500, 30),
                "batch 48 (kuka, ring 1e5): one partial 64-row block on the row-split chain": measure_shape(dev, "kuka", 48, 100000, E, 500, 30),
                "batch 48 on the column-tile chain (fuse = columns: the default for other layer / state sizes)":
                    measure_shape(dev, "kuka", 48, 100000, E, 500, 30, fuse="columns"),
                # north_star's literal head: textbook P = L L^T on 8 x 9 padded LDS tiles (--p-mode matmul)
                "configs[1] with P = L L^T (p_mode matmul)": measure_shape(dev, "kuka", 256, 1000000, E, 500, 30, p_mode="matmul"),
                # ... and at configs[4]'s literal shape: 7 x 7 L / P tiles, batch 2048, ring 4e6
                "configs[4] with P = L L^T (p_mode matmul): panda, batch 2048, ring 4e6":
                    measure_shape(dev, "panda", 2048, 4000000, E, 300, 30, p_mode="matmul"),
                # layer sizes other than the presets' 256 (the reference takes any; its own agent test builds 128): narrower ones are
                # stored zero-padded to 256 and run the row-split chain, wider ones the column-tile / unfused chains
                "layer_size 128 (kuka, batch 256, ring 1e6): stored zero-padded to 256":
                    measure_shape(dev, "kuka", 256, 1000000, E, 500, 30, layer=128),
                "layer_size 128, batch 1024 (kuka, ring 1e6)": measure_shape(dev, "kuka", 1024, 1000000, E, 300, 30, layer=128),
                "layer_size 512 (kuka, batch 256, ring 1e6): the row-split chain on two 256-column halves (round 6; column tiles before)":
                    measure_shape(dev, "kuka", 256, 1000000, E, 300, 30, layer=512),
                "layer_size 512, batch 1024 (kuka, ring 1e6)": measure_shape(dev, "kuka", 1024, 1000000, E, 200, 20, layer=512),
                # arms of more than 8 joints (the reference builds its head for any action size, naf_neural_network.py:53-54): 9 .. 11 run
                # the row-split chain since round 6 (one sample per 16-lane group in the fused layer-2 launch), 12 .. 16 the unfused chain
                "9 joints (state 27), batch 256, ring 1e5: the row-split chain (round 6)": measure_joints(dev, 9, 256, 100000, 64, 300),
                "9 joints, batch 256 on the unfused chain it ran before (fuse = unfused)": measure_joints(dev, 9, 256, 100000, 64, 150, fuse="unfused"),
                "11 joints (state 31), batch 1024, ring 1e5": measure_joints(dev, 11, 1024, 100000, 64, 200),
                "12 joints (state 33: beyond layer 1's K = 32), batch 256: unfused": measure_joints(dev, 12, 256, 100000, 64, 150),
            }
    finally:
        os.chdir(old)
    return res


if __name__ == "__main__":
    main()
