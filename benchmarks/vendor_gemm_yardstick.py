"""Yardstick for the backward GEMM launch of the row-split chain: what do the vendor libraries need
for the SAME three f32 products at the batch sizes where `gemm_bundle_kernel` sits at 17 - 24 % of the f32-MFMA peak?

  dA1 = dZ2 [B x 256] . W2 [256 x 256]          (M = B,   N = 256, K = 256)
  dW2 = dZ2^T [256 x B] . A1 [B x 256]          (M = 256, N = 256, K = B)
  dWh = dH^T [32 x B] . A2 [B x 272]            (M = 32,  N = 272, K = B)

Each back-end runs the three products as three back-to-back launches inside one captured graph (REPS trios per replay), timed
with HIP events around the replay; the same process also times this library's own launch on the same operands:
  * `bundle_plain`  naf_gemm_bundle on the three products with the learner's split-K slabs, NO BatchNorm-backward prologue and NO
                    layer-1 epilogue (what the vendor launches compute, plus nothing)
  * the in-situ figure of the full launch (prologue + epilogue) is profiles/r0x_b*_digest.csv's gemm_bundle line.
Back-ends: torch's default (hipBLASLt heuristic), rocBLAS, and TunableOp tuned IN THIS PROCESS over both libraries' solutions
(the best pick per shape a user of the vendor libraries can get). Run under rocprofv3 --kernel-trace --stats the per-kernel
averages of the vendor kernels come out as well (benchmarks/yardstick.sh).

  python benchmarks/vendor_gemm_yardstick.py [--out gpurun_out/vendor_gemm_yardstick.md] [--batches 256,1024,2048]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NAF_BLAS_TUNING_FILE", "none")
import torch

H, HP, NHP = 256, 272, 32
REPS = 40


def operands(B):
    f = dict(device="cuda", dtype=torch.float32)
    g = torch.Generator(device="cuda").manual_seed(B)
    r = lambda *s: torch.randn(*s, generator=g, **f)          # noqa: E731
    return dict(dZ2=r(B, H), W2=r(H, H), A1=r(B, H), dH=r(B, NHP), A2=r(B, HP),
                dA1=torch.empty(B, H, **f), gW2=torch.empty(H, H, **f), gWh=torch.empty(NHP, HP, **f))


def trio(o):
    torch.mm(o["dZ2"], o["W2"], out=o["dA1"])
    torch.mm(o["dZ2"].t(), o["A1"], out=o["gW2"])
    torch.mm(o["dH"].t(), o["A2"], out=o["gWh"])


def time_graph(body, reps=REPS, replays=20):
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            body()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(replays):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / (reps * replays))
    return best


def bundle_plain(o, B):
    """this library's launch on the same three products (split-K slabs as the learner cuts them), no prologue / epilogue"""
    from robotic_manipulator_rloa_amd import _lib
    lib = _lib.load()
    D, ptr = _lib.GemmDesc, _lib.ptr
    ks = B // 256 if B % 256 == 0 and B <= 2048 else 1
    ks_w2 = ks // 2 if B >= 2048 else ks
    f = dict(device="cuda", dtype=torch.float32)
    slab_w2 = torch.zeros(max(ks_w2, 1), H * H, **f)
    slab_wh = torch.zeros(max(ks, 1), NHP * HP, **f)
    descs = (D * 3)(
        D(ptr(o["dZ2"]), ptr(o["W2"]), ptr(o["dA1"]), None, B, H, H, H, H, H, 0, 1, 1, 0, None, None),
        D(ptr(o["dZ2"]), ptr(o["A1"]), ptr(slab_w2), None, H, H, B, H, H, H, 1, 1, ks_w2, H * H, None, None),
        D(ptr(o["dH"]), ptr(o["A2"]), ptr(slab_wh), None, NHP, HP, B, NHP, HP, HP, 1, 1, ks, NHP * HP, None, None))

    def body():
        _lib.check(lib.naf_gemm_bundle(descs, 3, torch.cuda.current_stream().cuda_stream), "gemm_bundle")
    body._keep = (slab_w2, slab_wh, descs)
    # parity of the operands' use (slabs summed = the vendor result)
    body()
    torch.cuda.synchronize()
    ref = o["dZ2"].t() @ o["A1"]
    err = (slab_w2.sum(0).view(H, H) - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-4, err
    return body


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/vendor_gemm_yardstick.md")
    ap.add_argument("--batches", default="256,1024,2048")
    a = ap.parse_args()
    batches = [int(x) for x in a.batches.split(",")]
    res = {}
    flops = lambda B: 2.0 * (B * H * H + H * H * B + NHP * HP * B)      # noqa: E731
    # 1. torch default (hipBLASLt heuristic)
    for B in batches:
        o = operands(B)
        res.setdefault(B, {})["hipblaslt_default"] = time_graph(lambda: trio(o))
    # 2. rocBLAS
    torch.backends.cuda.preferred_blas_library("cublas")
    for B in batches:
        o = operands(B)
        res[B]["rocblas"] = time_graph(lambda: trio(o))
    # 3. TunableOp over both libraries, tuned here
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(150)
    torch.cuda.tunable.set_max_tuning_iterations(40)
    for B in batches:
        o = operands(B)
        trio(o)                      # tunes the three shapes (eagerly, outside any capture)
        torch.cuda.synchronize()
    torch.cuda.tunable.tuning_enable(False)
    for B in batches:
        o = operands(B)
        res[B]["tunableop_best"] = time_graph(lambda: trio(o))
        # each product alone, tuned
        for name, fn in (("dA1", lambda: torch.mm(o["dZ2"], o["W2"], out=o["dA1"])),
                         ("dW2", lambda: torch.mm(o["dZ2"].t(), o["A1"], out=o["gW2"])),
                         ("dWh", lambda: torch.mm(o["dH"].t(), o["A2"], out=o["gWh"]))):
            res[B]["tuned_" + name] = time_graph(fn)
    torch.cuda.tunable.enable(False)
    # 4. this library, same products
    for B in batches:
        o = operands(B)
        res[B]["bundle_plain"] = time_graph(bundle_plain(o, B))
    lines = ["# Vendor-GEMM yardstick for the backward GEMM launch (f32, MI355X)", "",
             "Three products (dA1, dW2, dWh) per trio; microseconds per trio, back-to-back launches replayed from a graph "
             f"({REPS} trios per replay, best of 3 x 20 replays, HIP events). `bundle_plain` = naf_gemm_bundle on the same three "
             "products in ONE launch, without the BatchNorm-backward prologue and the layer-1 epilogue the shipping launch carries.", "",
             "| B | GFLOP | hipBLASLt default | rocBLAS | TunableOp best (3 launches) | tuned dA1 | tuned dW2 | tuned dWh | bundle_plain (1 launch) | bundle_plain TFLOP/s |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for B in batches:
        r = res[B]
        lines.append(f"| {B} | {flops(B) / 1e9:.3f} | {r['hipblaslt_default']:.2f} | {r['rocblas']:.2f} | {r['tunableop_best']:.2f} | "
                     f"{r['tuned_dA1']:.2f} | {r['tuned_dW2']:.2f} | {r['tuned_dWh']:.2f} | {r['bundle_plain']:.2f} | "
                     f"{flops(B) / r['bundle_plain'] / 1e6:.1f} |")
    lines += ["", "TunableOp picks (shape key -> solution):", ""]
    try:
        for row in torch.cuda.tunable.get_results():
            lines.append("    " + " | ".join(str(x) for x in row))
    except Exception as e:                                     # noqa: BLE001
        lines.append(f"    (unavailable: {e})")
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(a.out.replace(".md", ".json"), "w") as f:
        json.dump({str(k): v for k, v in res.items()}, f, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
