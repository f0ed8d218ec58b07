"""Measure, don't argue: what would ONE persistent launch confined to a single XCD buy the B = 256 update?
 The update is 8 dependent launches (~38 us); a persistent form trades its 8 kernel boundaries for
8 barriers over the 32 workgroups of one XCD and keeps every activation in that XCD's L2 — but has a quarter of an eighth
of the chip to compute on. Each ingredient is measured on the real chain, nothing is estimated from a data sheet:

  A. the SAME captured update chain (DeviceEnvLoop + TrainChunk as bench.py runs them) replayed on streams whose CU mask
     confines every kernel to 1, 2, 4 or all 8 XCDs (hipExtStreamCreateWithCUMask; a census kernel checks, eagerly and
     under hipGraph replay, where workgroups really ran): the price of computing on 32 CUs, with the L2 locality it brings.
  B. a persistent kernel that confines itself to one XCD and runs {write payload, 32-workgroup barrier, read a
     neighbour's payload}: microseconds per barrier + hand-off, portable agent-scope fences vs an XCD-local form.
  C. the launch boundary of this chain: a captured graph of 8 trivial dependent kernels.
Persistent estimate = A(1 XCD) - 8 x C + 8 x B.  Results: stdout + profiles/r02_persistent_probe.json.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def build_probe():
    src = os.path.join(ROOT, "benchmarks", "probe", "persistent_probe.hip")
    out = os.path.join(ROOT, "benchmarks", "probe", "libpp.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", src, "-o", out], check=True)
    lib = C.CDLL(out)
    lib.pp_census.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.pp_make_masked_stream.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_int]
    lib.pp_barrier_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.pp_blocker.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    return lib


def census(lib, stream_ptr, n_blocks=1024, lds=0, sync_all=True):
    out = torch.full((n_blocks,), -1, dtype=torch.int32, device="cuda")
    if sync_all:
        torch.cuda.synchronize()
    rc = lib.pp_census(out.data_ptr(), n_blocks, 64, lds, stream_ptr)
    assert rc == 0, rc
    torch.cuda.current_stream().synchronize()
    return torch.bincount(out.clamp(min=0), minlength=8).tolist()


def masked_stream(lib, bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = lib.pp_make_masked_stream(C.byref(s), words, 8)
    assert rc == 0, f"hipExtStreamCreateWithCUMask rc={rc}"
    return s.value


def main():
    lib = build_probe()
    res = {}
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # ---- CU-mask convention: which bits are XCC 0? ---------------------------------------------------------------
    full = census(lib, torch.cuda.current_stream().cuda_stream)
    res["census_default_stream"] = full
    conventions = {"interleaved (bit i -> XCC i % 8)": lambda x: sum(1 << i for i in range(256) if i % 8 in x),
                   "blocked (bit i -> XCC i // 32)": lambda x: sum(1 << i for i in range(256) if i // 32 in x)}
    chosen = None
    for name, mk in conventions.items():
        s = masked_stream(lib, mk({0}))
        got = census(lib, s)
        res[f"census_mask_{name}"] = got
        if sum(1 for g in got if g) == 1:
            chosen = (name, mk)
    print("census:", {k: v for k, v in res.items() if k.startswith("census")})
    if chosen is None:
        print("no CU-mask convention confined the census to one XCC; part A skipped")
    # ---- A. the real chain on masked streams -----------------------------------------------------------------------
    from bench import synth_rows
    from robotic_manipulator_rloa_amd.engine import DeviceEnvLoop, TrainChunk
    from robotic_manipulator_rloa_amd.learner import Learner
    from robotic_manipulator_rloa_amd.naf_components.naf_neural_network import reference_init_state_dict
    from robotic_manipulator_rloa_amd.utils.replay_buffer import ReplayBuffer
    S, A, H, B, E, N = 21, 6, 256, 256, 64, 1_000_000
    L = Learner(S, A, H, B, 1e-3, 1e-3, 0.99, dev)
    sd = reference_init_state_dict(S, A, H, seed=0)
    L.load_params(0, sd)
    L.load_params(1, sd)
    replay = ReplayBuffer(N, B, dev, seed=1000, state_size=S, action_size=A)
    replay.add_rows_device(synth_rows(N, S, A, replay.row_floats, replay.off_s2, 77, dev), N)
    loop = DeviceEnvLoop(L, replay, E, seed=31)
    chunk = TrainChunk(L, replay, E)
    loop.capture()
    chunk.capture()

    def run(stream, steps=300, warm=30):
        with torch.cuda.stream(stream):
            for _ in range(warm):
                loop.step()
                chunk.run()
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loop.step()
                chunk.run()
            stream.synchronize()
            return (time.perf_counter() - t0) / (steps * E) * 1e6

    res["chain_us_per_update"] = {"default stream (8 XCDs)": round(run(torch.cuda.current_stream()), 2)}
    res["chain_us_per_update"]["a non-default stream (8 XCDs)"] = round(run(torch.cuda.Stream()), 2)
    if chosen is not None:
        name, mk = chosen
        res["cu_mask_convention"] = name
        # is the mask honoured under hipGraph replay? capture the census kernel and replay it on the masked stream
        s1 = masked_stream(lib, mk({0}))
        ext1 = torch.cuda.ExternalStream(s1)
        out = torch.full((1024,), -1, dtype=torch.int32, device="cuda")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            lib.pp_census(out.data_ptr(), 1024, 64, 0, torch.cuda.current_stream().cuda_stream)
        out.fill_(-1)
        with torch.cuda.stream(ext1):
            g.replay()
            ext1.synchronize()
        res["census_graph_replay_on_masked_stream"] = torch.bincount(out.clamp(min=0), minlength=8).tolist()
        for xs in ({0}, {0, 1}, {0, 1, 2, 3}, set(range(8))):
            ext = torch.cuda.ExternalStream(masked_stream(lib, mk(xs)))
            ext.wait_stream(torch.cuda.current_stream())
            res["chain_us_per_update"][f"masked stream, {len(xs)} XCD(s)"] = round(run(ext), 2)
    # the CU mask is ignored on this box -> confine with blockers: 1024-thread workgroups, two per CU, that occupy every wave
    # slot of every XCC but the free ones, so that the chain's kernels can only be placed on the free XCC(s)
    flag = torch.zeros(1, dtype=torch.int32).pin_memory()
    side_b, work = torch.cuda.Stream(), torch.cuda.Stream()      # (the null stream would wait for the blockers)
    for n_free in (1, 2, 4):
        flag.zero_()
        resident = torch.zeros(1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        rc = lib.pp_blocker(flag.data_ptr(), 8.0, (1 << n_free) - 1, resident.data_ptr(), 512, side_b.cuda_stream)
        assert rc == 0, rc
        time.sleep(0.05)
        with torch.cuda.stream(work):
            where = census(lib, work.cuda_stream, sync_all=False)   # where may another kernel run now?
        us = run(work)
        flag.fill_(1)
        torch.cuda.synchronize()
        res["chain_us_per_update"][f"{n_free} XCD(s) free, others blocked"] = {"us": round(us, 2), "census_of_lds_kernel": where,
                                                                              "blockers_resident": int(resident.item())}
    print("A. chain:", res["chain_us_per_update"], "| graph replay census:", res.get("census_graph_replay_on_masked_stream"))
    # ---- C. launch boundary: 8 trivial dependent kernels per "update" in a graph -------------------------------------
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    libn = L.lib
    gk = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        libn.naf_counter_add(ctr.data_ptr(), 1, side.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(gk):
        for _ in range(8 * 64):
            libn.naf_counter_add(ctr.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
    for _ in range(5):
        gk.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        gk.replay()
    torch.cuda.synchronize()
    res["boundary_us"] = round((time.perf_counter() - t0) / (100 * 8 * 64) * 1e6, 3)
    print("C. boundary between trivial dependent kernels:", res["boundary_us"], "us")
    # ---- B. 32-workgroup same-XCD barrier + hand-off -------------------------------------------------------------------
    res["barrier_us"] = {}
    for local in (0, 1):
        for payload in (0, 1024, 16384):          # floats per workgroup: 0, 4 KB, 64 KB
            ctrl = torch.zeros(16, dtype=torch.int64, device=dev)
            buf = torch.zeros(2 * 32 * max(payload, 4), device=dev)
            iters = 2000
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = lib.pp_barrier_probe(ctrl.data_ptr(), buf.data_ptr(), iters, 0, payload, local, 256, 100 * 1024,
                                      torch.cuda.current_stream().cuda_stream)
            b.record()
            torch.cuda.synchronize()
            assert rc == 0, rc
            c = ctrl.cpu().tolist()
            key = f"{'xcd-local (vmcnt drain + sc1 loads)' if local else 'agent-scope release/acquire'}, {payload * 4} B"
            res["barrier_us"][key] = {"us_per_round": round(a.elapsed_time(b) * 1e3 / iters, 3), "workgroups": c[0],
                                      "workers_started": c[4], "errors": c[2], "timeouts": c[3]}
    for k, v in res["barrier_us"].items():
        print("B.", k, v)
    conf = res["chain_us_per_update"].get("1 XCD(s) free, others blocked", {})
    confined = isinstance(conf, dict) and sum(1 for g in conf.get("census_of_lds_kernel", []) if g) == 1
    res["chain_confined_to_one_xcd"] = confined
    ok = {k: v["us_per_round"] for k, v in res["barrier_us"].items() if v["errors"] == 0 and v["timeouts"] == 0}
    b0 = min(v for k, v in ok.items() if k.endswith(", 0 B"))
    b4k = min(v for k, v in ok.items() if k.endswith(", 4096 B"))
    res["sync_saving_upper_bound_us"] = round(8 * (res["boundary_us"] - b0), 2)
    print(f"8 boundaries = {8 * res['boundary_us']:.1f} us; 8 same-XCD barriers = {8 * b0:.1f} us bare, {8 * b4k:.1f} us with a 4 KB "
          f"hand-off per workgroup -> at most {res['sync_saving_upper_bound_us']} us per update to win on synchronisation")
    if confined:
        res["persistent_estimate_us"] = round(conf["us"] - 8 * res["boundary_us"] + 8 * b4k, 2)
        print(f"persistent estimate = {conf['us']} - 8 x {res['boundary_us']} + 8 x {b4k} = {res['persistent_estimate_us']} us per update "
              f"(today: {res['chain_us_per_update']['default stream (8 XCDs)']})")
    else:
        print("the chain could NOT be confined to one XCD from an unprivileged process on this box (CU masks ignored, kernels of a "
              "second stream wait for the blockers): its compute-on-32-CUs term is argued from the per-kernel block counts in DESIGN.md")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r02_persistent_probe.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
