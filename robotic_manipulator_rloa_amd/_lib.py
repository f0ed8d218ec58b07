"""ctypes binding of libnaf_hip.so (include/naf_hip.h) — the only way the package reaches the HIP kernels.

There is deliberately NO fallback: if the shared library is missing or a call fails, the caller gets a
NafHipError. The library is built in-tree by `build_library()` (hipcc --offload-arch=gfx950), which
__graft_entry__.build() calls; hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import shutil
import subprocess
from typing import Optional

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libnaf_hip.so")
SOURCES = ["lib.hip", "replay.hip", "naf_head.hip", "bn_relu.hip", "fused_layers.hip", "big_batch.hip", "gemm_bundle.hip", "optim.hip",
           "synth_env.hip", "xgmi_reduce.hip", "policy_act.hip", "step_path.hip", "naf_head_wide.hip"]
HEADERS = ["common.h", "head_body.h", "bn_tile.h", "xgmi_dev.h", "adam_body.h", "bn2bwd_fold.h", "act_body.h", "moments_body.h", "sample_body.h", "replay_dev.h", os.path.join("..", "..", "include", "naf_hip.h")]

P_HADAMARD, P_MATMUL = 0, 1
ACTION_TRUNC_INT, ACTION_FLOAT = 0, 1
NORM_CHUNK = 4096


class NafHipError(RuntimeError):
    pass


USAGE_PATH = LIB_PATH + ".usage.json"     # per-kernel registers / scratch / LDS as the compiler reported them (build_library)


def _build_flags() -> list:
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-variable",
             "-Rpass-analysis=kernel-resource-usage"]
    if os.environ.get("NAF_BUILD_DEFINES"):      # tile-shape experiments / -DNAF_TIMELINE (benchmarks/): e.g. "-DFT_TX=4"
        flags += os.environ["NAF_BUILD_DEFINES"].split()
    return flags


def _flags_stamp() -> str:
    return LIB_PATH + ".flags"


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    # a library linked from objects of other flags (a -DNAF_TIMELINE build left behind) is stale too. No stamp = a
    # library that travelled without one (prebuilt on another box): trusted when no defines are asked for.
    want = " ".join(_build_flags())
    if os.path.exists(_flags_stamp()):
        with open(_flags_stamp()) as f:
            if f.read().strip() != want:
                return True
    elif os.environ.get("NAF_BUILD_DEFINES"):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def header_abi_version() -> int:
    """NAF_HIP_ABI_VERSION as include/naf_hip.h states it (the header is the contract; the library must answer the same)."""
    with open(os.path.join(CSRC, HEADERS[-1])) as f:
        m = re.search(r"^#define\s+NAF_HIP_ABI_VERSION\s+(\d+)", f.read(), re.M)
    if not m:
        raise NafHipError("include/naf_hip.h does not define NAF_HIP_ABI_VERSION")
    return int(m.group(1))


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip into csrc/libnaf_hip.so for gfx950. Returns the library path.
    Safe with several processes importing at once (every rank of a torch.distributed.run launch): one of them builds
    under an exclusive file lock into a temporary name of its own, the others wait for the lock and find the library
    fresh."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise NafHipError("hipcc not found: cannot build libnaf_hip.so")
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():           # another process built it while this one waited
                return LIB_PATH
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            flags = _build_flags()
            # one object per source (csrc/build/, keyed by the flags), compiled in parallel and only when the source
            # or a header is newer: an edit of one kernel file rebuilds in seconds instead of a minute
            import hashlib
            from concurrent.futures import ThreadPoolExecutor
            objdir = os.path.join(CSRC, "build", hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10])
            os.makedirs(objdir, exist_ok=True)
            hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)

            def compile_one(src):
                obj = os.path.join(objdir, src.replace(".hip", ".o"))
                sp = os.path.join(CSRC, src)
                if not force and os.path.exists(obj) and os.path.exists(obj + ".usage") and \
                        os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(sp)):
                    return obj, None
                cmd = [hipcc] + flags + ["-c", sp, "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode == 0:
                    with open(obj + ".usage", "w") as f:     # the compiler's resource remarks of this file's kernels
                        f.write(r.stderr)
                return obj, (None if r.returncode == 0 else r.stdout + r.stderr)

            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
                results = list(ex.map(compile_one, SOURCES))
            errs = [e for _, e in results if e]
            if errs:
                raise NafHipError("hipcc failed:\n" + "\n".join(errs))
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [o for o, _ in results],
                               capture_output=True, text=True)
            if r.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise NafHipError("hipcc (link) failed:\n" + r.stdout + r.stderr)
            os.replace(tmp, LIB_PATH)
            _write_usage([o + ".usage" for o, _ in results])
            with open(_flags_stamp(), "w") as f:
                f.write(" ".join(flags) + "\n")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def _write_usage(remark_files) -> None:
    """libnaf_hip.so.usage.json: {kernel (mangled): {vgprs, agprs, sgprs, scratch_bytes_per_lane, lds_bytes, occupancy,
    source}} from hipcc's -Rpass-analysis=kernel-resource-usage remarks — what tests/test_host_cpu.py holds the library to
    (no kernel may spill to scratch)."""
    import json
    out, cur = {}, None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
            "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds_bytes", "VGPRs Spill": "vgpr_spills", "SGPRs Spill": "sgpr_spills"}
    for path in remark_files:
        with open(path) as f:
            for line in f:
                m = re.search(r"remark: Function Name: (\S+)", line)
                if m:
                    cur = out.setdefault(m.group(1), {"source": os.path.basename(path).replace(".o.usage", ".hip")})
                    continue
                m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
                if m and cur is not None and m.group(1).strip() in keys:
                    cur[keys[m.group(1).strip()]] = int(m.group(2))
    with open(USAGE_PATH, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


_lib: Optional[C.CDLL] = None

_vp, _i, _f, _u64, _sz, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_size_t, C.c_int64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/naf_hip.h one to one
_PROTOS = {
    "naf_hip_abi_version": [],
    "naf_hip_arch": [],
    "naf_timeline_read": [_i, _vp],
    "naf_host_store_supported": [_i],
    "naf_host_publish": [_vp, _vp, _sz],
    "naf_host_publish_launch": [_vp, _vp, _sz, _vp, _vp],
    "naf_replay_row_floats": [_i, _i],
    "naf_replay_row_off_next_state": [_i, _i],
    "naf_replay_create": [_u64, _i, _i, _vp, _vp, C.POINTER(_vp)],
    "naf_replay_destroy": [_vp],
    "naf_replay_size": [_vp, C.POINTER(_u64), _vp],
    "naf_replay_add_batch": [_vp, _vp, _i, _vp],
    "naf_replay_add_counted": [_vp, _vp, _vp, _i, _vp],
    "naf_replay_sample_indices": [_vp, _u64, _vp, _u64, _vp, _i, _i, _i, _vp],
    "naf_replay_sample_scratch_ints": [_i],
    "naf_replay_sample_indices_big": [_vp, _u64, _vp, _u64, _vp, _i, _i, _i, _vp, _vp],
    "naf_replay_batch_row_floats": [_i, _i],
    "naf_replay_gather_rows": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "naf_replay_gather_soa": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "naf_counter_add": [_vp, _u64, _vp],
    "naf_head_fwd": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "naf_head_bwd": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "naf_head_fwd_bwd_mse": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    "naf_head_fwd_bwd_mse_splitk": [_vp, _i64, _vp, _i, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _i, _i, _i, _vp],
    "naf_act_noise": [_vp, _i, _vp, _u64, _vp, _u64, _f, _i, _i, _i, _vp],
    "naf_bn_relu_fwd_train": [_vp, _i64, _i, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _vp, _vp, _i, _i, _i,
                              _f, _f, _vp],
    "naf_bn_relu_fwd_eval": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "naf_bn_relu_bwd": [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "naf_linear_bn_relu_fwd_train": [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _vp, _vp,
                                     _i, _i, _i, _f, _f, _vp],
    "naf_bn_relu_bwd_wgrad": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "naf_bn_relu_bwd_wgrad_push": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                                   _vp, _vp, _sz, _sz, _vp],
    "naf_fused_tile_cols": [],
    "naf_heads_bwd_bn_relu_bwd": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i,
                                  _vp],
    "naf_bn_relu_fwd_heads_partial": [_vp, _i64, _i, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _vp, _vp, _vp, _i64,
                                      _i, _i, _i, _vp, _i64, _vp, _i, _i, _f, _f, _vp],
    "naf_bb_moments_floats": [_i],
    "naf_bb_moments": [_vp, _i64, _i64, _i, _i, _vp, _i, _i, _i, _vp],
    "naf_bb_layer1_adam": [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i, _vp, _vp, _vp, _vp, _i, _i, _i,
                           _f, _f, _vp, _vp],
    "naf_bb_linear_stats_adam": [_vp, _i64, _i, _vp, _vp, _i64, _vp, _i64, _i, _vp, _i, _i, _i, _i, _vp, _vp],
    "naf_bb_layer2_head_rows": [_i],
    "naf_bb_layer2_head_exchange_floats": [_i, _i],
    "naf_bb_layer2_head": [_vp, _i64, _i, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i, _vp, _vp, _vp, _i64, _i, _i, _vp, _i, _vp,
                           _i, _f, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp],
    "naf_bb_layer1_bwd_kp": [_i],
    "naf_bb_layer1_bwd_finish_blocks": [_i],
    "naf_bb_layer1_bwd_finish": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                                 _vp, _i, _vp, _vp, _vp, _sz, _vp],
    "naf_gemm_bundle": [_vp, _i, _vp],
    "naf_grad_norm_partials": [_vp, _sz, _vp, _vp, _vp],
    "naf_adam_polyak_fused": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _f, _f, _vp, _f, _sz, _vp],
    "naf_polyak_update": [_vp, _vp, _f, _f, _sz, _vp],
    "naf_synth_env_step": [_vp, _vp, _vp, _vp, _i, _i, _u64, _vp, _i, _vp, _i, _vp],
    "naf_synth_env_reset": [_vp, _vp, _i, _i, _u64, _u64, _vp, _i, _vp],
    "naf_synth_env_state_floats": [_i],
    "naf_policy_act": [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _f, _i, _vp, _i,
                       _vp, _u64, _vp, _vp, _f, _i, _i, _i, _vp],
    "naf_step_prep": [_vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "naf_adam_polyak_act_sync_ints": [],
    "naf_host_store_alloc": [_sz, C.POINTER(_vp)],
    "naf_host_store_free": [_vp],
    "naf_host_store_selftest": [_vp, _i, _vp],
    "naf_step_prefetch": [_vp, _vp],
    "naf_step_launch": [_vp, _vp, _sz, _vp, _vp, _vp, _vp, C.c_int],
    "naf_adam_polyak_act": [_vp, _vp, _vp, _vp, _vp, _u64, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "naf_adam_polyak_act_layer1": [_vp, _vp, _vp, _vp, _vp, _u64, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "naf_xgmi_chunk_floats": [],
    "naf_xgmi_create": [_i, _i, _sz, C.c_double, C.POINTER(_vp)],
    "naf_xgmi_set_timeout": [_vp, C.c_double],
    "naf_xgmi_mem_kind": [_vp],
    "naf_xgmi_export": [_vp, _vp],
    "naf_xgmi_connect": [_vp, _vp, _vp],
    "naf_xgmi_connect_local": [_vp, _vp],
    "naf_xgmi_allreduce_sum": [_vp, _vp, _vp, _vp, _vp, _vp],
    "naf_xgmi_push_desc": [_vp, _vp],
    "naf_xgmi_push_early": [_vp, _vp, _sz, _sz, _vp],
    "naf_xgmi_allreduce_sum_from": [_vp, _vp, _vp, _vp, _vp, _sz, _vp],
    "naf_xgmi_allreduce_sum_from2": [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _vp],
    "naf_xgmi_status": [_vp, C.POINTER(_u64), C.POINTER(_u64)],
    "naf_xgmi_timeouts_nowait": [_vp, C.POINTER(_u64)],
    "naf_xgmi_disconnect": [_vp],
    "naf_xgmi_destroy": [_vp],
}
_RESTYPES = {"naf_hip_arch": C.c_char_p}
EXPORTED_SYMBOLS = tuple(_PROTOS)


class XgmiPushDesc(C.Structure):
    """naf_xgmi_push_t (include/naf_hip.h)"""
    _fields_ = [("peer_base", C.c_void_p * 8), ("ctrl", C.c_void_p), ("data_off", C.c_uint64), ("n_pad", C.c_uint64),
                ("rank", C.c_int), ("world", C.c_int), ("timeout_ticks", C.c_longlong), ("host_timeouts", C.c_void_p)]


class SlabSeg(C.Structure):
    """naf_bb_slab_seg_t (include/naf_hip.h)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("stride", C.c_int64), ("n", C.c_int), ("n_slabs", C.c_int)]


class GemmDesc(C.Structure):
    """naf_gemm_desc_t (include/naf_hip.h)"""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("sumsq", C.c_void_p), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int), ("a_kmajor", C.c_int), ("b_kmajor", C.c_int),
                ("k_split", C.c_int), ("c_split_stride", C.c_int64), ("epi", C.c_void_p), ("pro", C.c_void_p)]


class AdamArgs(C.Structure):
    """naf_adam_args_t (include/naf_hip.h): the deferred optimizer step carried by the next update's first two launches"""
    _fields_ = [("theta", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("theta_target", C.c_void_p),
                ("partials", C.c_void_p), ("n_partials", C.c_int), ("max_norm", C.c_float), ("lr", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("tau", C.c_float), ("one_minus_tau", C.c_float),
                ("step_dev", C.c_void_p), ("inv_world", C.c_float), ("n", C.c_int64), ("l1_floats", C.c_int64),
                ("bc", C.c_void_p)]


class ActNet(C.Structure):
    """naf_act_net_t (include/naf_hip.h): where the main network's parameters sit in the flat buffers, for naf_adam_polyak_act"""
    _fields_ = [("S", C.c_int), ("A", C.c_int), ("H", C.c_int), ("NHP", C.c_int), ("HP", C.c_int),
                ("off_W1", C.c_int64), ("off_b1", C.c_int64), ("off_g1", C.c_int64), ("off_be1", C.c_int64), ("off_W2", C.c_int64),
                ("off_b2", C.c_int64), ("off_g2", C.c_int64), ("off_be2", C.c_int64), ("off_Wh", C.c_int64),
                ("running_mean1", C.c_void_p), ("running_var1", C.c_void_p), ("running_mean2", C.c_void_p),
                ("running_var2", C.c_void_p), ("eps", C.c_float)]


class BbLayer1(C.Structure):
    """naf_bb_layer1_t (include/naf_hip.h): layer 1 of the next update's chain riding on naf_adam_polyak_act_layer1"""
    _fields_ = [("x", C.c_void_p), ("x_net_stride", C.c_int64), ("ldx", C.c_int), ("K", C.c_int), ("W", C.c_void_p),
                ("bias", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("param_net_stride", C.c_int64),
                ("mom", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("stat_net_stride", C.c_int64),
                ("out", C.c_void_p), ("out_net_stride", C.c_int64), ("ldo", C.c_int), ("save_mean", C.c_void_p),
                ("save_invstd", C.c_void_p), ("wc_out", C.c_void_p), ("xhat_out", C.c_void_p), ("B", C.c_int), ("H", C.c_int),
                ("nets", C.c_int), ("momentum", C.c_float), ("eps", C.c_float)]


class StepCopies(C.Structure):
    """naf_step_copies_t (include/naf_hip.h): up to three word ranges a step launch copies before anything else"""
    _fields_ = [("src", C.c_void_p * 3), ("dst", C.c_void_p * 3), ("n_words", C.c_int * 3)]

    @classmethod
    def of(cls, *triples):
        c = cls()
        for i, (src, dst, n) in enumerate(triples):
            c.src[i], c.dst[i], c.n_words[i] = src, dst, int(n)
        return c


class StepPrefetch(C.Structure):
    """naf_step_prefetch_t (include/naf_hip.h): the next timestep's minibatch, drawn by one more workgroup of naf_adam_polyak_act"""
    _fields_ = [("replay", C.c_void_p), ("seed", C.c_uint64), ("counter_dev", C.c_void_p), ("idx_spec", C.c_void_p),
                ("out_rows", C.c_void_p), ("out_ld", C.c_int), ("action_mode", C.c_int), ("mom", C.c_void_p), ("B", C.c_int),
                ("without_replacement", C.c_int), ("spec_rec", C.c_void_p), ("mode", C.c_int), ("src_row", C.c_void_p),
                ("n_word", C.c_void_p), ("row_out", C.c_void_p), ("idx_out", C.c_void_p), ("host_spec", C.c_void_p),
                ("pipe_errors", C.c_void_p), ("copies", StepCopies), ("depth", C.c_int), ("spec_rec_in", C.c_void_p),
                ("idx_spec_in", C.c_void_p), ("pf_seq", C.c_void_p)]


class GemmBn2Bwd(C.Structure):
    """naf_gemm_bn2bwd_t (include/naf_hip.h)"""
    _fields_ = [("z", C.c_void_p), ("partials", C.c_void_p), ("gamma", C.c_void_p), ("save_mean", C.c_void_p),
                ("save_invstd", C.c_void_p), ("d_gamma", C.c_void_p), ("d_beta", C.c_void_p), ("npb", C.c_int), ("B", C.c_int),
                ("H", C.c_int), ("cst", C.c_void_p), ("epoch", C.c_void_p), ("errors", C.c_void_p)]


class BbStatsOnce(C.Structure):
    """naf_bb_stats_once_t (include/naf_hip.h)"""
    _fields_ = [("records", C.c_void_p), ("epoch", C.c_void_p), ("errors", C.c_void_p), ("exchange", C.c_void_p)]


class GemmL1Bwd(C.Structure):
    """naf_gemm_l1bwd_t (include/naf_hip.h)"""
    _fields_ = [("x", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("a1", C.c_void_p), ("save_mean", C.c_void_p),
                ("save_invstd", C.c_void_p), ("partials", C.c_void_p), ("p_slabs", C.c_void_p), ("ldx", C.c_int), ("K", C.c_int),
                ("kp", C.c_int), ("lda1", C.c_int), ("xhat", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("rows", C.c_int)]


def load(allow_build: bool = True) -> C.CDLL:
    """Load libnaf_hip.so. `import torch` must already have happened in the process when tensors are going to be
    passed, so the library binds to the same HIP runtime torch loaded (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if _stale() and allow_build:
        build_library()          # a failed build raises: a stale library is never loaded silently behind it
    if not os.path.exists(LIB_PATH):
        raise NafHipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                          "This package has no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first)
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export what the header declares
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    want = header_abi_version()
    if lib.naf_hip_abi_version() != want:
        raise NafHipError(f"libnaf_hip.so answers ABI version {lib.naf_hip_abi_version()}, include/naf_hip.h says {want}: "
                          "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code == 0:
        return
    if code < 0:
        raise NafHipError(f"{what}: argument/state error {code}")
    raise NafHipError(f"{what}: hipError_t {code}")


def require_gpu() -> None:
    import torch
    if not torch.cuda.is_available():
        raise NafHipError("no MI355X visible (torch.cuda.is_available() is False): the NAF hot path runs only on "
                          "the GPU through libnaf_hip.so — there is no CPU fallback")


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return t.data_ptr() if t is not None else None
