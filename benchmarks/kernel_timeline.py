"""Where one update of the large-batch chain spends its time: phases INSIDE each kernel and the gaps BETWEEN kernels, from
wall-clock marks the kernels leave themselves (csrc/common.h NAF_TL, include/naf_hip.h naf_timeline_read). rocprofv3 gives
per-kernel durations; this gives the inside of them, with no profiler attached, under the graph replay the bench times.

    NAF_BUILD_DEFINES=-DNAF_TIMELINE python benchmarks/kernel_timeline.py --batch 1024 [--robot kuka] [--out file.json]

(the define selects an object directory of its own under csrc/build/, but the linked libnaf_hip.so is shared: rebuild
without the define before measuring throughput — the marks cost a few stores per kernel.)
Times are microseconds since the first mark of the update's first kernel; resolution 0.01 us (100 MHz clock).
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERNELS = ["bb_layer1", "bb_linear_stats", "bb_layer2_head", "(unused)", "gemm_bundle", "bb_layer1_bwd_finish", "adam_polyak"]
MARKS = {
    # (bb_layer1 with the previous update's optimizer step riding on it — every update of a chunk but the first — also leaves raw
    # slots 7 .. 10: operands in LDS | clip scale derived | barrier | parameters evaluated, and 11 / 12: entry / exit of the first
    # and last riding workgroup; NAF_TL_RAW=1 prints them)
    "bb_layer1": ["entry", "operands staged", "statistics from moments", "z tile", "normalise + store"],
    "bb_linear_stats": ["entry", "chunk 0 staged", "chunk 0 MFMA", "chunk 1 staged", "chunk 1 MFMA", "Z2 + statistics partials"],
    "bb_layer2_head": ["entry", "operands + statistics fold", "normalise, V'", "heads MFMA", "halves merged", "NAF head body", "dA2 MFMA + sums", "partials out"],
    "(unused)": ["entry"],
    "gemm_bundle": ["entry", "chunk 0 staged", "K loop", "C stored", "layer-1 backward epilogue", "norm partial"],
    "bb_layer1_bwd_finish": ["entry", "loads, folds, dW1 / slab sums, norm partial"],
    "adam_polyak": ["entry", "norm folded + update"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--robot", default="kuka")
    ap.add_argument("--updates", type=int, default=64)
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=1, help="average the marks of this many replays")
    args = ap.parse_args()
    if "NAF_TIMELINE" not in os.environ.get("NAF_BUILD_DEFINES", ""):
        raise SystemExit("build with NAF_BUILD_DEFINES=-DNAF_TIMELINE")
    import torch
    sys.argv = [sys.argv[0]]
    import bench
    from robotic_manipulator_rloa_amd import _lib
    from robotic_manipulator_rloa_amd.engine import TrainChunk
    from robotic_manipulator_rloa_amd.learner import Learner
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer

    dev = torch.device("cuda", 0)
    S, A = (23, 7) if args.robot == "panda" else (21, 6)
    B, N = args.batch, 200_000
    L = Learner(S, A, 256, B, 1e-3, 1e-3, 0.99, dev, p_mode=_lib.P_HADAMARD)
    sd = reference_init_state_dict(S, A, 256, seed=0)
    L.load_params(0, sd)
    L.load_params(1, sd)
    replay = ReplayBuffer(N, B, dev, seed=1000, state_size=S, action_size=A)
    replay.add_rows_device(bench.synth_rows(N, S, A, replay.row_floats, replay.off_s2, seed=77, device=dev), N)
    chunk = TrainChunk(L, replay, args.updates, use_graph=True, gather_outside_graph=True)
    chunk.capture()
    for _ in range(5):
        chunk.run()
    torch.cuda.synchronize()
    lib = L.lib

    def read_marks():
        raw = {}
        for kid, name in enumerate(KERNELS):
            buf = (C.c_longlong * 32)()
            rc = lib.naf_timeline_read(kid, buf)
            if rc != 0:
                raise SystemExit(f"naf_timeline_read({name}) = {rc}")
            raw[name] = [list(buf[:16]), list(buf[16:])]
        return raw
    raw = read_marks()
    if args.reps > 1:
        # --reps N: the last update of N replays, every mark averaged relative to that replay's first entry (one pass is +- 0.4 us
        # on a kernel: enough to see a phase, not to compare two builds)
        acc = {name: [[0.0] * 16, [0.0] * 16] for name in KERNELS}
        for _ in range(args.reps):
            chunk.run()
            torch.cuda.synchronize()
            r = read_marks()
            t0r = min(r[KERNELS[0]][w][0] for w in (0, 1))
            for name in KERNELS:
                for w in (0, 1):
                    for i in range(16):
                        acc[name][w][i] += (r[name][w][i] - t0r) if r[name][w][i] > 0 else -1e15
        raw = {name: [[acc[name][w][i] / args.reps if acc[name][w][i] > -1e14 else -1e9 for i in range(16)] for w in (0, 1)]
               for name in KERNELS}
    # the marks are those of the LAST update of the last replay: one consistent pass through the chain
    t0 = min(raw[KERNELS[0]][w][0] for w in (0, 1))
    out = {"batch": B, "robot": args.robot, "fuse": sorted(L.fuse), "unit": "us since the first kernel's entry", "kernels": {}}
    print(f"B = {B}, fuse = {sorted(L.fuse)}")
    prev_end = None
    for name in KERNELS:
        if max(raw[name][0][0], raw[name][1][0]) < t0:      # not part of this chain (e.g. stage 2 folded into the bundle: "s2")
            print(f"{name:22s} (not launched in this chain)")
            continue
        n = len(MARKS[name])
        rows = {}
        for w, tag in ((0, "first workgroup"), (1, "last workgroup")):
            rows[tag] = [round((raw[name][w][i] - t0) / 100.0, 2) for i in range(n)]
            for i in range(1, n):           # a mark this workgroup did not pass (the finish launch's slab-reduce blocks)
                ref = rows[tag][0] if (name == "gemm_bundle" and MARKS[name][-1] == "chunk 0 landed") else rows[tag][i - 1]
                if not (ref - 1.0 <= rows[tag][i] <= ref + 1000.0):
                    rows[tag][i] = rows[tag][i - 1]
        start = min(r[0] for r in rows.values())
        end = max(max(r) for r in rows.values())
        out["kernels"][name] = {"marks": MARKS[name], **rows, "start": start, "end": end,
                                "gap_before": None if prev_end is None else round(start - prev_end, 2)}
        gap = "" if prev_end is None else f"   (gap {start - prev_end:+.2f})"
        print(f"{name:22s} {start:7.2f} -> {end:7.2f} = {end - start:5.2f} us{gap}")
        for tag, r in rows.items():
            if name == "gemm_bundle" and n == 6 and MARKS[name][5] == "chunk 0 landed":      # (the ring form's marks are not in time order: absolute times since entry)
                steps = " | ".join(f"{MARKS[name][i]} @{r[i] - r[0]:.2f}" for i in range(1, n))
            else:
                steps = " | ".join(f"{MARKS[name][i]} +{r[i] - r[i - 1]:.2f}" for i in range(1, n))
            print(f"    {tag:16s} entry {r[0]:7.2f} | {steps}")
        prev_end = end
        if os.environ.get("NAF_TL_RAW"):      # every slot as written (extra marks placed while investigating a phase)
            for w, tag in ((0, "first"), (1, "last")):
                print(f"    raw slots ({tag} wg):", [round((t - t0) / 100.0, 2) if t > 0 else None for t in raw[name][w]])
    # gemm_bundle, every workgroup: when it entered and left (a launch larger than the chip's resident set runs in rounds)
    ent, ext = [], []
    for i in range(256):
        buf = (C.c_longlong * 32)()
        if lib.naf_timeline_read(1024 + i, buf) != 0:
            break
        ent += list(buf[:16])
        ext += list(buf[16:])
    # workgroups write their own slot; the rest of the array stays zero — and so do the slots of workgroups that leave no
    # mark (the launch's first workgroups, which fold the BatchNorm-backward sums and exit): None, not a clock value
    n_wg = max([i + 1 for i, t in enumerate(ent) if t > 0], default=0)
    if n_wg:
        ent = [round((t - t0) / 100.0, 2) if t > 0 else None for t in ent[:n_wg]]
        ext = [round((t - t0) / 100.0, 2) if t > 0 else None for t in ext[:n_wg]]
        out["gemm_bundle_workgroups"] = {"entry": ent, "exit": ext}
        print(f"gemm_bundle: {n_wg} workgroups; entry / exit (us) of every 32nd that left marks:")
        fmt = lambda v: "   (none)" if v is None else f"{v:7.2f}"          # noqa: E731
        for i in range(0, n_wg, 32):
            j = next((k for k in range(i, min(i + 32, n_wg)) if ent[k] is not None), None)
            if j is not None:
                print(f"    wg {j:4d}: {fmt(ent[j])} -> {fmt(ext[j])}")
        print(f"    last   : {fmt(ent[-1])} -> {fmt(ext[-1])}; latest exit {max(v for v in ext if v is not None):7.2f}")
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
