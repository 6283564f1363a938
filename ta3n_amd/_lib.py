"""ctypes binding of libta3n_hip.so (include/ta3n_hip.h).

The library is loaded on first use.  If it has not been built the call raises -
there is no CPU or eager-PyTorch fallback for the product path.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional, Tuple

# torch bundles its own libamdhip64; it must be in the process BEFORE libta3n_hip.so is
# dlopen'ed so both resolve to ONE HIP runtime (otherwise the second runtime sees no device).
import torch  # noqa: F401

_LIB: Optional[C.CDLL] = None
LIB_PATH = os.path.join(os.environ.get("TA3N_LIBDIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib"), "libta3n_hip.so")

FLAG_ADV_RELATION = 1 << 0
FLAG_ADV_VIDEO = 1 << 1
FLAG_ADV_FRAME = 1 << 2
FLAG_ATTN_ENTROPY = 1 << 3
FLAG_TRANS_ATTN = 1 << 4
FLAG_MCD = 1 << 5              # ens_DA 'MCD': second video classifier (regions Y2 / gY2); unfused entry points
FLAG_FEATURE_GRADS = 1 << 6    # backward also takes a gradient at the pooled video feature (region gV_ext): dis_DA DAN / JAN
FLAG_BN_SHARED = 1 << 7        # use_bn AdaBN / AutoDIAL: BatchNorm1d per domain behind the shared frame FC (regions Z0, bn_batch, bn_run)
FLAG_BF16_MFMA = 1 << 8
FLAG_BF16_STORE = 1 << 9
FLAG_F32_SPLIT = 1 << 10       # fp32-grade contractions as three bf16 MFMAs on operands split hi + lo in registers ("bf16x3")
AGG_TRN_M, AGG_AVGPOOL = 0, 1

# every symbol include/ta3n_hip.h declares (tests check the export list)
SYMBOLS = [
    "ta3n_num_relation_tuples", "ta3n_relation_table", "ta3n_segment_indices", "ta3n_gather_segments", "ta3n_plan_create",
    "ta3n_plan_destroy", "ta3n_num_params", "ta3n_param_info", "ta3n_param_floats", "ta3n_live_param_floats",
    "ta3n_workspace_floats", "ta3n_ws_offset", "ta3n_ws_size", "ta3n_plan_describe", "ta3n_set_hyper",
    "ta3n_init_workspace", "ta3n_forward", "ta3n_loss", "ta3n_backward", "ta3n_has_fused_step", "ta3n_train_step", "ta3n_eval_metrics",
    "ta3n_sgd_step", "ta3n_sgd_step_fused", "ta3n_sgd_range", "ta3n_train_step_join", "ta3n_train_step_range", "ta3n_refresh_bf16", "ta3n_sgd_step_next", "ta3n_gather_segments_into", "ta3n_has_pipelined_step", "ta3n_train_step_after_update", "ta3n_num_phases",
    "ta3n_debug_arrays", "ta3n_debug_struct_sizes", "ta3n_time_phases", "ta3n_time_update_launches", "ta3n_last_error", "ta3n_version",
    "ta3n_comm_unique_id", "ta3n_comm_create", "ta3n_comm_destroy", "ta3n_comm_world", "ta3n_all_reduce_sum", "ta3n_train_step_ddp",
    "ta3n_gather_segments_bf16_into", "ta3n_train_steps", "ta3n_train_steps_multi", "ta3n_chain_status", "ta3n_debug_waits",
    "ta3n_peer_create", "ta3n_peer_handle", "ta3n_peer_connect", "ta3n_peer_all_reduce_sum", "ta3n_peer_status", "ta3n_peer_destroy",
    "ta3n_comm_attach_peer", "ta3n_has_fused_update", "ta3n_train_steps_fused_update",
    "ta3n_gaussian_kernel_scratch_floats", "ta3n_gaussian_kernel", "ta3n_mmd_rowdiff", "ta3n_discrepancy_scratch_floats", "ta3n_discrepancy",
    "ta3n_mcd_source_loss", "ta3n_mcd_second_loss",
    "ta3n_shard_ranges", "ta3n_shard_sumsq", "ta3n_sgd_shard", "ta3n_shard_reduce_scatter", "ta3n_sharded_update", "ta3n_train_steps_sharded",
]


class Config(C.Structure):
    _fields_ = [("batch_source", C.c_int32), ("batch_target", C.c_int32), ("num_segments", C.c_int32),
                ("feature_dim", C.c_int32), ("fc_dim", C.c_int32), ("num_bottleneck", C.c_int32),
                ("num_class", C.c_int32), ("flags", C.c_uint32), ("tile_config", C.c_int32),
                ("phase_tiles", C.c_int32 * 16), ("xcd_aware", C.c_int32), ("aggregation", C.c_int32),
                ("wgrads_late", C.c_int32), ("chain", C.c_int32), ("cost_model", C.c_int32), ("split_k", C.c_int32), ("reserved", C.c_int32 * 1)]


class Hyper(C.Structure):
    _fields_ = [("beta", C.c_float * 3), ("gamma", C.c_float), ("lr", C.c_float), ("momentum", C.c_float),
                ("weight_decay", C.c_float), ("clip", C.c_float), ("p_drop_i", C.c_float), ("p_drop_v", C.c_float),
                ("seed_i", C.c_uint32), ("seed_v", C.c_uint32), ("inv_n_cls", C.c_float), ("inv_n_rel", C.c_float),
                ("inv_n_vid", C.c_float), ("inv_n_frm", C.c_float), ("inv_n_ent", C.c_float),
                ("valid_source", C.c_int32), ("valid_target", C.c_int32), ("train", C.c_int32),
                ("reverse", C.c_int32), ("mu", C.c_float), ("reserved", C.c_int32 * 2)]


class Feed(C.Structure):
    """ta3n_feed (include/ta3n_hip.h): device-side batch assembly of a multi-step call."""
    _fields_ = [("store", C.c_void_p), ("bf16", C.c_int32), ("ids_per_step", C.c_int32), ("first_row", C.c_void_p),
                ("num_frames", C.c_void_p), ("labels", C.c_void_p), ("video_ids", C.c_void_p)]


class StepsJob(C.Structure):
    """ta3n_steps_job (include/ta3n_hip.h): the argument list of one ta3n_train_steps call, as an element of ta3n_train_steps_multi."""
    _fields_ = [("plan", C.c_void_p), ("x", C.c_void_p), ("params", C.c_void_p), ("grads", C.c_void_p), ("momentum", C.c_void_p),
                ("ws", C.c_void_p), ("fused_norm", C.c_int32), ("lr_pending", C.c_float), ("momentum_coef", C.c_float),
                ("weight_decay", C.c_float), ("clip", C.c_float), ("hypers", C.POINTER(Hyper)), ("source", C.POINTER(Feed)),
                ("target", C.POINTER(Feed)), ("comm", C.c_void_p), ("scratch_bf16", C.c_void_p), ("stream", C.c_void_p)]


class Ta3nError(RuntimeError):
    pass


def has_experiments() -> bool:
    """True when the loaded library was built with -DTA3N_EXPERIMENTS=1 (the measured-and-rejected step variants: chained launches,
    split-K tiles, the optimiser inside the gradient tiles, the tall four-wave tiles; `pytest -m gpu_ab`)."""
    return lib().ta3n_version().decode().endswith("+experiments")


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises if it is missing: build it with
    `python -m ta3n_amd.build` (hipcc --offload-arch=gfx950)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise Ta3nError(f"{LIB_PATH} not found: the HIP extension is required (python -m ta3n_amd.build); "
                        "there is no CPU fallback")
    # The binary must be the one these sources build: the hash of csrc/ + include/ stored beside it at link time (ta3n_amd/build.py)
    # against the hash of the tree that is loading it.  A stale or hand-placed library fails here, loudly (TA3N_ALLOW_STALE_LIB=1: A/B
    # runs of an older build directory on purpose).
    try:
        from .build import source_hash
        with open(os.path.join(os.path.dirname(LIB_PATH), ".source_hash")) as fh:
            linked_from = fh.read().strip()
        if linked_from != source_hash() and os.environ.get("TA3N_ALLOW_STALE_LIB") != "1":
            raise Ta3nError(f"{LIB_PATH} was linked from sources {linked_from}, this tree is {source_hash()}: rebuild "
                            "(python -m ta3n_amd.build), or TA3N_ALLOW_STALE_LIB=1 to load it anyway")
    except OSError:
        pass                              # (a library without a recorded hash: built by hand; nothing to compare)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.ta3n_num_relation_tuples.argtypes = [C.c_int]
    L.ta3n_relation_table.argtypes = [C.c_int, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.ta3n_segment_indices.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(i64)]
    L.ta3n_gather_segments.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    L.ta3n_plan_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.ta3n_plan_destroy.argtypes = [vp]
    L.ta3n_plan_destroy.restype = None
    L.ta3n_num_params.argtypes = [vp]
    L.ta3n_param_info.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32),
                                  C.POINTER(i32)]
    for fn in ("ta3n_param_floats", "ta3n_live_param_floats", "ta3n_workspace_floats"):
        getattr(L, fn).argtypes = [vp]
        getattr(L, fn).restype = i64
    for fn in ("ta3n_ws_offset", "ta3n_ws_size"):
        getattr(L, fn).argtypes = [vp, C.c_char_p]
        getattr(L, fn).restype = i64
    L.ta3n_plan_describe.argtypes = [vp, C.c_char_p, i64]
    L.ta3n_plan_describe.restype = i64
    L.ta3n_set_hyper.argtypes = [vp, vp, C.POINTER(Hyper), vp]
    L.ta3n_init_workspace.argtypes = [vp, vp, vp]
    L.ta3n_forward.argtypes = [vp, vp, vp, vp, vp]
    L.ta3n_loss.argtypes = [vp, vp, vp]
    L.ta3n_backward.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ta3n_eval_metrics.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.ta3n_has_fused_step.argtypes = [vp]
    L.ta3n_train_step.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ta3n_sgd_step.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ta3n_sgd_step_fused.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ta3n_sgd_range.argtypes = [vp, vp, vp, vp, vp, i64, i64, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    L.ta3n_train_step_join.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.ta3n_train_step_range.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    L.ta3n_refresh_bf16.argtypes = [vp, vp, vp, vp, vp]
    L.ta3n_has_pipelined_step.argtypes = [vp]
    L.ta3n_train_step_after_update.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp]
    L.ta3n_train_steps.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Hyper), C.c_int,
                                   C.POINTER(Feed), C.POINTER(Feed), vp, vp, vp]
    L.ta3n_train_steps_multi.argtypes = [C.POINTER(StepsJob), C.c_int, C.c_int]
    L.ta3n_has_fused_update.argtypes = [vp]
    L.ta3n_shard_ranges.argtypes = [vp, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    L.ta3n_shard_sumsq.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    L.ta3n_sgd_shard.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Hyper), vp]
    L.ta3n_shard_reduce_scatter.argtypes = [vp, vp, vp, vp, vp]
    L.ta3n_sharded_update.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Hyper), vp]
    L.ta3n_train_steps_sharded.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Hyper), C.c_int,
                                           C.POINTER(Feed), C.POINTER(Feed), vp, vp, vp]
    L.ta3n_train_steps_fused_update.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.POINTER(Hyper), C.c_int,
                                                C.POINTER(Feed), C.POINTER(Feed), vp]
    L.ta3n_gather_segments_into.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.ta3n_gather_segments_bf16_into.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.ta3n_sgd_step_next.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp]
    L.ta3n_num_phases.argtypes = [vp, C.c_int]
    L.ta3n_time_update_launches.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.c_int,
                                            C.POINTER(C.c_float)]
    L.ta3n_time_phases.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(C.c_float), C.POINTER(i32), C.POINTER(i32),
                                   C.c_int]
    L.ta3n_debug_arrays.argtypes = [vp] + [C.POINTER(vp), C.POINTER(i64)] * 3 + [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.ta3n_debug_struct_sizes.argtypes = [C.POINTER(i32)] * 5
    L.ta3n_debug_waits.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.ta3n_chain_status.argtypes = [vp, vp, vp]
    L.ta3n_comm_unique_id.argtypes = [C.c_char_p]
    L.ta3n_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    L.ta3n_comm_destroy.argtypes = [vp]
    L.ta3n_comm_destroy.restype = None
    L.ta3n_comm_world.argtypes = [vp]
    L.ta3n_all_reduce_sum.argtypes = [vp, vp, i64, vp, vp]
    L.ta3n_train_step_ddp.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ta3n_peer_create.argtypes = [C.c_int, C.c_int, i64, C.c_int, C.POINTER(vp)]
    L.ta3n_peer_handle.argtypes = [vp, C.c_char_p]
    L.ta3n_peer_connect.argtypes = [vp, C.c_char_p]
    L.ta3n_peer_all_reduce_sum.argtypes = [vp, vp, i64, vp]
    L.ta3n_peer_status.argtypes = [vp, vp]
    L.ta3n_peer_destroy.argtypes = [vp]
    L.ta3n_peer_destroy.restype = None
    L.ta3n_comm_attach_peer.argtypes = [vp, vp]
    L.ta3n_gaussian_kernel_scratch_floats.argtypes = [C.c_int]
    L.ta3n_gaussian_kernel_scratch_floats.restype = i64
    L.ta3n_gaussian_kernel.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, vp, vp, vp, vp]
    L.ta3n_mmd_rowdiff.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp]
    L.ta3n_mcd_source_loss.argtypes = [vp, vp, vp, vp, vp]
    L.ta3n_mcd_second_loss.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp]
    L.ta3n_discrepancy_scratch_floats.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.ta3n_discrepancy_scratch_floats.restype = i64
    L.ta3n_discrepancy.argtypes = [vp, i64, C.c_int, i64, C.c_int, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_float, vp, i64, vp, vp]
    L.ta3n_last_error.restype = C.c_char_p
    L.ta3n_version.restype = C.c_char_p
    _LIB = L
    return L


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = lib().ta3n_last_error().decode()
        if rc == -1:
            raise ValueError(f"{what}: {msg}")        # the reference raises ValueError for bad configs (models.py:137, 562)
        raise Ta3nError(f"{what}: {msg} (status {rc})")
    return rc


def relation_table(num_frames: int) -> List[List[Tuple[int, ...]]]:
    """Selected frame tuples per scale, scale T first (reference TRNmodule.py:30-41, 60, 68-71)."""
    L = lib()
    n = check(L.ta3n_num_relation_tuples(num_frames), "ta3n_num_relation_tuples")
    tup = (C.c_int32 * (n * num_frames))()
    sl = (C.c_int32 * n)()
    sid = (C.c_int32 * n)()
    check(L.ta3n_relation_table(num_frames, tup, sl, sid), "ta3n_relation_table")
    out: List[List[Tuple[int, ...]]] = [[] for _ in range(num_frames - 1)]
    for r in range(n):
        out[sid[r]].append(tuple(tup[r * num_frames + j] for j in range(sl[r])))
    return out


def segment_indices(num_frames: int, num_segments: int, new_length: int = 1) -> List[int]:
    """TSNDataSet._get_test_indices (reference dataset.py:103-116); 1-based."""
    L = lib()
    out = (C.c_int64 * num_segments)()
    check(L.ta3n_segment_indices(num_frames, num_segments, new_length, out), "ta3n_segment_indices")
    return list(out)


class Plan:
    """Owns a ta3n_plan handle and caches its layout tables."""

    def __init__(self, batch_source: int, batch_target: int, num_segments: int, feature_dim: int, fc_dim: int,
                 num_class: int, flags: int, num_bottleneck: int = 256, tile_config: int = 0,
                 phase_tiles: Optional[List[int]] = None, xcd_aware: int = 0, aggregation: int = 0, wgrads_late: int = 0,
                 chain: int = 0, cost_model: int = 0, split_k: int = 0):
        L = lib()
        self.cfg = Config(batch_source, batch_target, num_segments, feature_dim, fc_dim, num_bottleneck, num_class,
                          flags, tile_config)
        for i, t in enumerate((phase_tiles or [])[:16]):      # (launches past the 16th mirror an earlier one or use tile_config)
            self.cfg.phase_tiles[i] = int(t)
        self.cfg.xcd_aware = int(xcd_aware)
        self.cfg.aggregation = int(aggregation)       # AGG_TRN_M / AGG_AVGPOOL
        self.cfg.wgrads_late = int(wgrads_late)
        self.cfg.chain = int(chain)
        self.cfg.cost_model = int(cost_model)
        self.cfg.split_k = int(split_k)
        h = C.c_void_p()
        check(L.ta3n_plan_create(C.byref(self.cfg), C.byref(h)), "ta3n_plan_create")
        self.handle = h
        self._L = L
        n = L.ta3n_num_params(h)
        self.params: List[Tuple[str, int, Tuple[int, ...], bool]] = []
        for i in range(n):
            name = C.c_char_p(); off = C.c_int64(); r = C.c_int32(); c = C.c_int32(); live = C.c_int32()
            check(L.ta3n_param_info(h, i, C.byref(name), C.byref(off), C.byref(r), C.byref(c), C.byref(live)))
            shape = (r.value, c.value) if c.value else (r.value,)
            self.params.append((name.value.decode(), off.value, shape, bool(live.value)))
        self.param_floats = L.ta3n_param_floats(h)
        self.live_floats = L.ta3n_live_param_floats(h)
        self.ws_floats = L.ta3n_workspace_floats(h)
        n = L.ta3n_plan_describe(h, None, 0)
        buf = C.create_string_buffer(n + 1)
        L.ta3n_plan_describe(h, buf, n + 1)
        self.description = json.loads(buf.value.decode())
        self.regions: Dict[str, Tuple[int, int]] = {k: tuple(v) for k, v in self.description["regions"].items()}
        self.has_fused_step = L.ta3n_has_fused_step(h) == 1
        self.has_fused_update = L.ta3n_has_fused_update(h) == 1

    def region(self, name: str) -> Tuple[int, int]:
        return self.regions[name]

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._L.ta3n_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
