"""Fused TA3N train step on one MI355X: forward, loss assembly, backward, gradient
all-reduce (RCCL, only when world_size > 1), clip + Nesterov SGD - every
arithmetic op is a HIP kernel of libta3n_hip.so; this module only owns device
buffers (torch tensors), the stream and the optional hipGraph capture.

It is the host side of what the reference does in main.train (main.py:348-621)
with VideoModel.forward (models.py:545-722); names follow the reference.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence

import os
import torch

from . import _lib
from . import parallel

ALL_FLAGS = (_lib.FLAG_ADV_RELATION | _lib.FLAG_ADV_VIDEO | _lib.FLAG_ADV_FRAME | _lib.FLAG_ATTN_ENTROPY |
             _lib.FLAG_TRANS_ATTN)


def flags_from_options(place_adv: Sequence[str] = ("Y", "Y", "Y"), add_loss_DA: str = "attentive_entropy",
                       use_attn: str = "TransAttn", adv_DA: str = "RevGrad", use_target: str = "uSv") -> int:
    """opts.py flags -> TA3N_FLAG_* (main.py:508-562 conditions)."""
    f = 0
    adv_on = adv_DA != "none" and use_target != "none"
    if adv_on and place_adv[0] == "Y":
        f |= _lib.FLAG_ADV_RELATION
    if adv_on and place_adv[1] == "Y":
        f |= _lib.FLAG_ADV_VIDEO
    if adv_on and place_adv[2] == "Y":
        f |= _lib.FLAG_ADV_FRAME
    if add_loss_DA == "attentive_entropy" and use_attn != "none" and use_target != "none":
        f |= _lib.FLAG_ATTN_ENTROPY
    if use_attn == "TransAttn":
        f |= _lib.FLAG_TRANS_ATTN
    return f


def beta_dann(p: float) -> float:
    """main.py:351."""
    return 2.0 / (1.0 + math.exp(-10 * p)) - 1


def lr_dann(lr0: float, p: float) -> float:
    """adjust_learning_rate_dann, main.py:800-802."""
    return lr0 / (1.0 + 10 * p) ** 0.75


def dropout_seeds(step: int, rank: int = 0):
    """(seed_i, seed_v) of one step on one rank.  The kernels key a mask element by its LOCAL row index, so the rank has
    to be part of the seed: the reference's DataParallel replicas draw independent masks for their shards (nn.Dropout,
    models.py:574-575, 679-680), two ranks must not share one."""
    s = int(step)
    r = (0xC2B2AE3D * int(rank)) & 0xFFFFFFFF
    return ((0x9E3779B1 * (2 * s + 1)) ^ r) & 0xFFFFFFFF, ((0x85EBCA77 * (2 * s + 2)) ^ ((r * 0x27D4EB2F) & 0xFFFFFFFF)) & 0xFFFFFFFF


_HYPER_DTYPE = None
_LEGACY_HOST_PREP = os.environ.get("TA3N_HOST_PREP", "") == "legacy"      # A/B aid (tools/r5_session16.sh): schedule arrays entry by entry, no split


def _hyper_dtype():
    """(numpy, the structured dtype of ta3n_hyper) - the conversion from the ctypes structure costs ~25 us, so once per process."""
    global _HYPER_DTYPE
    import numpy as np
    if _HYPER_DTYPE is None:
        _HYPER_DTYPE = np.dtype(_lib.Hyper)
    return np, _HYPER_DTYPE


class TrainEngine:
    """Device-resident state of one rank: flat parameters / gradients / momentum,
    workspace, static input buffers.  Source rows come first in every batch
    tensor (rows [0, Bs) source, [Bs, Bs+Bt) target)."""

    def __init__(self, batch_source: int, batch_target: int, num_segments: int = 5, feature_dim: int = 2048,
                 fc_dim: int = 512, num_class: int = 12, flags: Optional[int] = None, dropout_i: float = 0.5,
                 dropout_v: float = 0.5, momentum: float = 0.9, weight_decay: float = 1e-4, clip: float = 20.0,
                 device: Optional[torch.device] = None, tile_config: int = 0, process_group=None,
                 phase_tiles: Optional[Sequence[int]] = None, xcd_aware: int = 0, fused: bool = True,
                 bf16: bool = False, bf16_store: bool = False, aggregation: str = "trn-m", wgrads_late: bool = False,
                 f32_split: bool = False, chain: Optional[bool] = None, grad_transport: Optional[str] = None,
                 dis_DA: str = "none", place_dis: Sequence[str] = ("N", "Y", "N"), alpha: float = 0.0, use_bn: str = "none",
                 ens_DA: str = "none", mu: float = 0.0, split_k: Optional[int] = None, sharded_update: Optional[bool] = None,
                 peer_exchange: Optional[bool] = None, ddp_buckets: Optional[int] = None, share_comm: bool = False):
        if not torch.cuda.is_available():
            raise _lib.Ta3nError("TrainEngine needs a HIP device (no CPU fallback)")
        if flags is None:        # default: the full TA3N configuration for trn-m, the source-only one (BASELINE configs[0]) for avgpool
            flags = ALL_FLAGS if aggregation == "trn-m" else 0
        # dis_DA DAN / JAN (main.py:452-505, loss.py:46-120): a discrepancy loss on the class logits (feat[0]) and / or the pooled
        # video feature (feat[1]), weighted by alpha.  It enters the step as one more gradient at those two tensors: the unfused
        # launch lists (ta3n_forward / ta3n_loss / ta3n_backward) with the gradient entry at the video feature
        # (TA3N_FLAG_FEATURE_GRADS), the loss itself by the HIP kernels of csrc/ta3n_mmd.hip.  Single rank: the loss couples every
        # pair of videos of the batch (the reference computes it on the gathered batch of its DataParallel replicas).
        if dis_DA not in ("none", "DAN", "JAN"):
            raise NotImplementedError(f"dis_DA {dis_DA!r} (built: DAN, JAN)")
        self.dis_DA, self.place_dis, self.alpha = dis_DA, tuple(place_dis), float(alpha)
        self.loss_d = None                       # device scalar: the discrepancy loss of the last step (main.py's loss_d)
        self._disc_scratch = None                # (native path: kernel matrices, stacked rows, the loss scalar in its last float)
        if dis_DA != "none":
            if dis_DA == "DAN" and len(self.place_dis) > 2 and self.place_dis[2] == "Y":
                raise ValueError("place_dis[2]: the reference itself fails on the 3-D frame features (loss.py:49)")
            flags |= _lib.FLAG_FEATURE_GRADS
            fused = False
        # use_bn AdaBN / AutoDIAL (models.py:194-198, 490-543, 569-570): BatchNorm1d per domain between the shared frame FC and its
        # ReLU - batch statistics over the rows of each domain: two pointwise launches (TA3N_FLAG_BN_SHARED) behind the shared-FC product
        # and in front of its weight gradient, in the fused step as in the unfused lists; the running statistics are buffers of this
        # engine (state_dict names as in the reference), moved by a stream-ordered update after every train-mode forward.
        # Single rank: the statistics are taken over the rank's own rows (what each nn.DataParallel replica of the reference
        # does too, but not the same numbers as one GPU on the whole batch).  AutoDIAL's mixing parameter stays at its initial 1.
        if use_bn not in ("none", "AdaBN", "AutoDIAL"):
            raise NotImplementedError(f"use_bn {use_bn!r} (built: AdaBN, AutoDIAL)")
        self.use_bn = use_bn
        if use_bn != "none":       # (round 6: the fused step carries the two BatchNorm launches - 10 launches instead of 17; fused=False keeps the unfused lists)
            flags |= _lib.FLAG_BN_SHARED
        # ens_DA MCD (Maximum Classifier Discrepancy; models.py:276-279, 682-684, 716-720; main.py:447-448, 548-556): a second video
        # classifier, its cross-entropy on the source rows, and a SECOND forward with GradReverse(mu) behind dropout_v whose loss is
        # -mean |softmax(out_target) - softmax(out_target_2)|.  Here: the unfused launch lists with the second classifier
        # (TA3N_FLAG_MCD), a second workspace for the reversed pass, the two small logit-level losses in torch, gradients of the
        # two passes added.  Single rank.
        if ens_DA not in ("none", "MCD"):
            raise NotImplementedError(f"ens_DA {ens_DA!r} (built: MCD)")
        self.ens_DA, self.mu = ens_DA, float(mu)
        self.loss_s = None                       # device scalar: the MCD discrepancy loss of the last step (main.py's loss_s)
        self.loss_c2 = None                      # ... and the second classifier's cross-entropy on the source rows
        self._global_source, self._global_target = int(batch_source), int(batch_target)      # job-wide valid counts of the current step (set_hyper)
        self._job_last_hyper = None      # ta3n_hyper of the last step a _steps_job enqueued (becomes self._hyper in _steps_done)
        self.loss_e_shift = None                 # MCD + attentive entropy: (d total, d loss_e) that moving the target rows' entropy term to the
                                                 # second pass's logits adds to what the loss kernel logged (main.py:549 vs :559-562)
        if ens_DA == "MCD":
            if use_bn != "none":
                raise NotImplementedError("ens_DA MCD with use_bn (the second forward moves the BatchNorm buffers again: the module path)")
            flags |= _lib.FLAG_MCD
            fused = False
        if f32_split:            # fp32-grade contractions as three bf16 MFMAs on operands split hi + lo (ta3n_hip.h) ...
            if bf16:
                raise ValueError("f32_split and bf16 are different arithmetics: set one")
            flags |= _lib.FLAG_F32_SPLIT
            if bf16_store:       # ... read from "pair twins": producers store the hi and the lo plane, nothing is split in the K loops
                flags |= _lib.FLAG_BF16_STORE
        else:
            if bf16 or bf16_store:   # BASELINE configs[1]: contraction operands rounded to bf16, fp32 accumulation and fp32 state
                flags |= _lib.FLAG_BF16_MFMA
            if bf16_store:           # ... and the forward launches of the fused step read bf16 twins instead of rounding on the fly
                flags |= _lib.FLAG_BF16_STORE
        self.bf16 = bool(flags & _lib.FLAG_BF16_MFMA)
        self.bf16_store = bool(flags & _lib.FLAG_BF16_STORE)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if aggregation not in ("trn-m", "avgpool"):
            raise NotImplementedError(f"frame_aggregation {aggregation!r} (built: 'trn-m', and 'avgpool' in the source-only configuration)")
        self.aggregation = aggregation
        if aggregation == "avgpool":     # TemPooling: no relation features, no attention (use_attn none in the reference's script); with
            # use_target none (BASELINE configs[0]) the caller passes no adversarial flag either (flags_from_options)
            flags &= ~(_lib.FLAG_ATTN_ENTROPY | _lib.FLAG_TRANS_ATTN)
        if phase_tiles is None and tile_config == 0 and aggregation == "trn-m":      # measured choices for the benchmarked shapes
            from .tuning import tuned_phase_tiles
            phase_tiles = tuned_phase_tiles(batch_source + batch_target, num_segments, feature_dim, min(fc_dim, feature_dim),
                                            self.bf16, self.bf16_store, split=bool(flags & _lib.FLAG_F32_SPLIT))
        if os.environ.get("TA3N_PHASE_TILES") and tile_config == 0:      # measurement aid: "i:code,i:code" replaces entries of the list (A/B of one launch's tile under bench.py's protocol)
            phase_tiles = list(phase_tiles or []) + [0] * (16 - len(phase_tiles or []))
            for item in os.environ["TA3N_PHASE_TILES"].split(","):
                i, code = item.split(":")
                phase_tiles[int(i)] = int(code)
        if chain is None:        # chained launches (ta3n_config.chain): the fused trn-m step in 5 launches instead of 8
            chain = os.environ.get("TA3N_CHAIN", "0") == "1" and aggregation == "trn-m" and fused
        self._flags = int(flags)
        self.plan = _lib.Plan(batch_source, batch_target, num_segments, feature_dim, fc_dim, num_class, flags,
                              tile_config=tile_config, phase_tiles=list(phase_tiles or []), xcd_aware=xcd_aware,
                              aggregation=_lib.AGG_AVGPOOL if aggregation == "avgpool" else _lib.AGG_TRN_M,
                              wgrads_late=int(wgrads_late), chain=int(chain),
                              cost_model=int(os.environ.get("TA3N_COST_MODEL", "0")),
                              split_k=int(os.environ.get("TA3N_SPLIT_K", "0")) if split_k is None else int(split_k))
        self.chain = bool(chain)
        self.Bs, self.Bt, self.T, self.D, self.C = batch_source, batch_target, num_segments, feature_dim, num_class
        self.B = batch_source + batch_target
        self.F = min(fc_dim, feature_dim)
        self.dropout_i, self.dropout_v = dropout_i, dropout_v
        self.momentum, self.weight_decay, self.clip = momentum, weight_decay, clip
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        # The DA options under more than one rank follow what the reference's nn.DataParallel does with them (main.py:79): the discrepancy
        # loss is taken on the gathered global batch (discrepancy()); MCD's discrepancy is a mean over the global target batch
        # (mcd_second_forward()); BatchNorm statistics are PER REPLICA - each replica normalises its own slice of the batch with its own
        # batch statistics (torch's DataParallel replicates the module; nothing synchronises the statistics) - and the running
        # buffers that persist are replica 0's (sync_buffers()).
        p = self.plan
        with torch.cuda.device(self.device):
            self.P = torch.zeros(p.param_floats, dtype=torch.float32, device=self.device)
            self.G = torch.zeros(p.param_floats, dtype=torch.float32, device=self.device)
            self.M = torch.zeros(p.live_floats, dtype=torch.float32, device=self.device)
            self.ws = torch.zeros(p.ws_floats, dtype=torch.float32, device=self.device)
            self.X = torch.zeros(self.B * self.T, self.D, dtype=torch.float32, device=self.device)
            self._L = _lib.lib()
            _lib.check(self._L.ta3n_init_workspace(p.handle, self.ws.data_ptr(), self._stream()), "ta3n_init_workspace")
        off, n = p.region("labels")
        self._labels = self.ws[off:off + n].view(torch.int32)
        self._mcd_buf: Optional[torch.Tensor] = None
        self.mcd_raw_seeds = None                # ens_DA MCD: the second pass's two dropout stream seeds as given (None: derived from the first pass's)
        self.ws2: Optional[torch.Tensor] = None      # ens_DA MCD: workspace and gradient buffer of the reversed second pass
        self.G2: Optional[torch.Tensor] = None
        if self.ens_DA == "MCD":
            with torch.cuda.device(self.device):
                self.ws2 = torch.zeros(p.ws_floats, dtype=torch.float32, device=self.device)
                self.G2 = torch.zeros(p.param_floats, dtype=torch.float32, device=self.device)
                _lib.check(self._L.ta3n_init_workspace(p.handle, self.ws2.data_ptr(), self._stream()), "ta3n_init_workspace")
                # the loss assembly of the two passes from the library (ta3n_mcd_source_loss / ta3n_mcd_second_loss; TA3N_NATIVE_MCD=0: the torch
                # form of rounds 4-5, kept for A/B): per-row terms [3 B] + the four scalars {loss_c2, loss_s, d total, d loss_e} at the end
                self._mcd_buf = torch.zeros(3 * self.B + 4, dtype=torch.float32, device=self.device)
        self._mcd_native = self.ens_DA == "MCD" and os.environ.get("TA3N_NATIVE_MCD", "1") != "0"
        # fused: forward + loss + backward as ONE C-ABI call (ta3n_train_step, 7 launches) when the plan has it
        self.fused = bool(fused) and p.has_fused_step
        # deferred update: the optimiser step of call s is enqueued at the start of call s + 1, split so that everything but
        # the shared frame FC updates on a side stream beside the next step's first launch (include/ta3n_hip.h: ta3n_sgd_range)
        self._pending = None
        self._side: Optional[torch.cuda.Stream] = None
        # TA3N_DDP_SELFTEST=1: take the N > 1 code path (split launches + RCCL buckets) in a 1-rank process group
        self._ddp_selftest = os.environ.get("TA3N_DDP_SELFTEST") == "1"
        # TA3N_SIDE_UPDATE=0: keep the pipelined update a launch of its own (ta3n_sgd_step_next)
        self._side_update = (os.environ.get("TA3N_SIDE_UPDATE", "1") == "1" and
                             self._L.ta3n_has_pipelined_step(self.plan.handle) == 1)
        # 1 (default): one all-reduce after the last launch; 2: everything but the shared frame FC's gradient is reduced while
        # the last launch runs (worth it only when that launch is longer than an extra collective's fixed cost)
        self._ddp_buckets = int(os.environ.get("TA3N_DDP_BUCKETS", "1")) if ddp_buckets is None else int(ddp_buckets)
        self._n_first = next(off for name, off, _, _ in p.params if not name.startswith("fc_feature_shared_source"))
        # N > 1 (or the 1-rank self-test): RCCL straight from the C ABI on the step's streams (TA3N_DDP_NATIVE=0: through
        # torch.distributed instead).
        self.comm = None
        self.comm_fallback: Optional[str] = None
        self._g16 = None
        self._comm_stream: Optional[torch.cuda.Stream] = None
        rccl_group = self.world == 1 or torch.distributed.get_backend(self.pg) == "nccl"     # gloo (CPU / shared-GPU tests): torch path
        if (self.world > 1 or self._ddp_selftest) and rccl_group and os.environ.get("TA3N_DDP_NATIVE", "1") == "1":
            # Either EVERY rank uses the library's communicator or none does (parallel.NativeComm agrees on that over the torch group
            # before and after ncclCommInitRank): ranks that disagreed would enqueue different collectives and hang.
            try:
                self.comm = (parallel.shared_native_comm if share_comm else parallel.NativeComm)(self.pg if self.world > 1 else None, self.device)
            except Exception as ex:      # noqa: BLE001 - raised on every rank together: still RCCL, through torch.distributed
                self.comm = None
                self.comm_fallback = f"{type(ex).__name__}: {ex}"
                if self.rank == 0:
                    print(f"[ta3n] RCCL communicator of the C ABI unavailable ({self.comm_fallback}); gradient all-reduce goes "
                          f"through torch.distributed (backend nccl = RCCL)", flush=True)
            # Gradient transport: fp32 unless asked otherwise (grad_transport="bf16" / TA3N_DDP_BF16=1: every rank's gradients are
            # rounded to bf16 and SUMMED in bf16 - half the xGMI bytes, but the N-rank result then differs from the 1-rank /
            # fp32-transport one by up to ~N * 2^-9 relative per element, and the clip norm is taken on the rounded sum)
            want16 = (grad_transport == "bf16") if grad_transport is not None else os.environ.get("TA3N_DDP_BF16", "0") == "1"
            if self.comm is not None and want16:
                self._g16 = torch.zeros(p.live_floats, dtype=torch.bfloat16, device=self.device)
        # TA3N_DDP_PEER=1: the exchange as a two-shot all-reduce over peer-mapped buffers (csrc/ta3n_peer.hip) instead of ncclAllReduce;
        # opt-in (only its protocol could be exercised on the one-GPU boxes this was built on); any backend of the process group
        self.peer = None
        want_peer = (os.environ.get("TA3N_DDP_PEER", "0") == "1") if peer_exchange is None else bool(peer_exchange)
        if (self.world > 1 or self._ddp_selftest) and want_peer:
            want16 = (grad_transport == "bf16") if grad_transport is not None else os.environ.get("TA3N_DDP_BF16", "0") == "1"
            try:
                self.peer = parallel.PeerComm(self.pg if self.world > 1 else None, self.device, p.live_floats, bf16=want16)
                if self.comm is not None:      # ta3n_all_reduce_sum / ta3n_train_steps of the communicator go through the peer path
                    _lib.check(self._L.ta3n_comm_attach_peer(self.comm.handle, self.peer.handle), "ta3n_comm_attach_peer")
            except Exception as ex:      # noqa: BLE001 - raised on every rank together (PeerComm): keep the default exchange
                self.peer = None
                if self.rank == 0:
                    print(f"[ta3n] peer all-reduce unavailable ({type(ex).__name__}: {ex}); using the default exchange", flush=True)
        if self.comm is not None and self.peer is None and getattr(self.comm, "shared", False):
            self._L.ta3n_comm_attach_peer(self.comm.handle, None)      # (a shared communicator may still carry the previous engine's transport)
        # use_bn: running [source, target][mean, var][F] (nn.BatchNorm1d: zeros / ones, momentum 0.1, unbiased variance) + batch counter
        self.bn_running: Optional[torch.Tensor] = None
        self.bn_batches = 0
        if self.use_bn != "none":
            # (round 6) the buffers ARE the workspace region the BatchNorm launch reads in eval mode and - since it tracks them itself, on the
            # device, momentum 0.1 / unbiased variance like nn.BatchNorm1d - updates in train mode: a K-step call needs no host in between
            self.bn_running = self.region("bn_run").view(2, 2, self.F)
            self.bn_running[:, 0] = 0.0
            self.bn_running[:, 1] = 1.0
        # Sharded update (TA3N_DDP_SHARDED=1 / sharded_update=True; N > 1 or the 1-rank self-test, fused step): the gradient exchange as
        # reduce-scatter + all-gather around an optimiser pass that touches only this rank's 1 / world of the parameters
        # (include/ta3n_hip.h: ta3n_sharded_update).  Every rank must make the same choice.
        want_sh = (os.environ.get("TA3N_DDP_SHARDED", "0") == "1") if sharded_update is None else bool(sharded_update)
        self._sharded = bool(want_sh and (self.world > 1 or self._ddp_selftest) and self.fused and self.peer is None and
                             self._L.ta3n_has_pipelined_step(self.plan.handle) == 1)
        self._shard_own = self._shard_layout = None
        if self._sharded:
            own, lay = (C.c_int64 * 4)(), (C.c_int64 * 4)()
            _lib.check(self._L.ta3n_shard_ranges(p.handle, self.rank, self.world, own, lay), "ta3n_shard_ranges")
            self._shard_own, self._shard_layout = list(own), list(lay)
            if self._g16 is not None:      # bf16 gradient transport: the scratch covers the padded end of region B
                self._g16 = torch.zeros(max(p.live_floats, self._shard_layout[3]), dtype=torch.bfloat16, device=self.device)
        self._P2: Optional[torch.Tensor] = None      # second parameter buffer of the fused-update steps (train_steps)
        self.step_count = 0
        self.skip_collective = False
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._hyper = _lib.Hyper()

    # ---- plumbing ----
    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def region(self, name: str, shape=None) -> torch.Tensor:
        off, n = self.plan.region(name)
        t = self.ws[off:off + n]
        return t.view(shape) if shape is not None else t

    def param_views(self, src: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        src = self.P if src is None else src
        out = {}
        for name, off, shape, live in self.plan.params:
            n = 1
            for s in shape:
                n *= s
            out[name] = src[off:off + n].view(shape)
        return out

    def load_state(self, state: Dict[str, torch.Tensor]) -> None:
        """Copy reference-named tensors (state_dict keys, models.py:141-294) into the flat buffer."""
        views = self.param_views()
        for k, v in views.items():
            if k in state:
                v.copy_(state[k].to(device=self.device, dtype=torch.float32))
        if self.bn_running is not None:      # BatchNorm buffers (models.py:195-198: bn_shared_S / bn_shared_T)
            for d, dom in enumerate("ST"):
                for j, what in enumerate(("running_mean", "running_var")):
                    key = f"bn_shared_{dom}.{what}"
                    if key in state:
                        self.bn_running[d, j].copy_(state[key].to(device=self.device, dtype=torch.float32))
            if "bn_shared_S.num_batches_tracked" in state:
                self.bn_batches = int(state["bn_shared_S.num_batches_tracked"])
        self.refresh_bf16(params=True)

    def refresh_bf16(self, x: bool = False, params: bool = False) -> None:
        """TA3N_FLAG_BF16_STORE: rebuild the bf16 twins of what the HOST side wrote (features / parameters).  Everything the
        library writes itself keeps its twin up to date.  Call after writing self.X or self.P directly."""
        if not self.bf16_store:
            return
        _lib.check(self._L.ta3n_refresh_bf16(self.plan.handle, self.X.data_ptr() if x else None,
                                             self.P.data_ptr() if params else None, self.ws.data_ptr(), self._stream()),
                   "ta3n_refresh_bf16")

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = {k: v.detach().clone() for k, v in self.param_views().items()}
        if self.bn_running is not None:
            for d, dom in enumerate("ST"):
                out[f"bn_shared_{dom}.running_mean"] = self.bn_running[d, 0].clone()
                out[f"bn_shared_{dom}.running_var"] = self.bn_running[d, 1].clone()
                out[f"bn_shared_{dom}.num_batches_tracked"] = torch.tensor(self.bn_batches, dtype=torch.int64)
        return out

    def live_names(self):
        return [n for n, _, _, live in self.plan.params if live]

    def momentum_views(self) -> Dict[str, torch.Tensor]:
        """Momentum buffers of the live parameters (views into the flat buffer; zero = "no update yet", which is what
        torch.optim.SGD's lazily created buffer amounts to: buf = g on first use = mu * 0 + g)."""
        out = {}
        for name, off, shape, live in self.plan.params:
            if live:
                n = 1
                for s in shape:
                    n *= s
                out[name] = self.M[off:off + n].view(shape)
        return out

    def set_batch(self, source: torch.Tensor, target: torch.Tensor, source_label: torch.Tensor) -> None:
        """[Bs,T,D], [Bt,T,D] float features and int labels into the static device buffers."""
        self.X[: self.Bs * self.T].copy_(source.reshape(-1, self.D), non_blocking=True)
        self.X[self.Bs * self.T:].copy_(target.reshape(-1, self.D), non_blocking=True)
        self._labels[: self.Bs].copy_(source_label.to(torch.int32), non_blocking=True)
        self.refresh_bf16(x=True)

    def set_hyper(self, beta: Sequence[float], gamma: float, lr: float, train: bool = True,
                  valid_source: Optional[int] = None, valid_target: Optional[int] = None,
                  global_source: Optional[int] = None, global_target: Optional[int] = None,
                  seed: Optional[int] = None, upload: bool = True, raw_seeds: Optional[Sequence[int]] = None) -> None:
        """Per-step scalars.  global_* are the job-wide valid video counts (all ranks):
        losses are means over the GLOBAL batch like the reference's DataParallel gather
        (main.py:446, 533; loss.py:24), so ranks divide by global counts and gradients are SUMMED."""
        h = self._hyper
        ns = self.Bs if valid_source is None else valid_source
        nt = self.Bt if valid_target is None else valid_target
        gs = ns * self.world if global_source is None else global_source
        gt = nt * self.world if global_target is None else global_target
        h.beta[0], h.beta[1], h.beta[2] = float(beta[0]), float(beta[1]), float(beta[2])
        h.gamma, h.lr = float(gamma), float(lr)
        h.momentum, h.weight_decay = float(self.momentum), float(self.weight_decay)
        h.clip = float(self.clip) if self.clip is not None else 0.0
        h.p_drop_i, h.p_drop_v = float(self.dropout_i), float(self.dropout_v)
        h.seed_i, h.seed_v = dropout_seeds(self.step_count if seed is None else seed, self.rank)
        if raw_seeds is not None:      # the two dropout stream seeds as given (a caller that draws them itself, like VideoModel.forward does)
            h.seed_i, h.seed_v = int(raw_seeds[0]) & 0xFFFFFFFF, int(raw_seeds[1]) & 0xFFFFFFFF
        for k, v in parallel.loss_normalisers(gs, gt, self.T).items():
            setattr(h, k, v)
        h.valid_source, h.valid_target, h.train = int(ns), int(nt), int(bool(train))
        self._global_source, self._global_target = int(gs), int(gt)
        if not upload:      # the caller delivers self._hyper another way (ta3n_sgd_step_next)
            return
        _lib.check(self._L.ta3n_set_hyper(self.plan.handle, self.ws.data_ptr(), C.byref(h), self._stream()), "ta3n_set_hyper")

    # ---- launches ----
    def forward(self) -> None:
        _lib.check(self._L.ta3n_forward(self.plan.handle, self.X.data_ptr(), self.P.data_ptr(), self.ws.data_ptr(),
                                        self._stream()), "ta3n_forward")
        if self.bn_running is not None and self._hyper.train:
            self._bn_track()

    def _bn_track(self, steps: int = 1) -> None:
        """use_bn: `steps` train-mode forwards were enqueued - the BatchNorm launch moved the running statistics itself (ws["bn_run"], which
        self.bn_running views); what is left to the host is nn.BatchNorm1d's num_batches_tracked."""
        if self.bn_running is not None:
            self.bn_batches += int(steps)

    def loss(self) -> None:
        _lib.check(self._L.ta3n_loss(self.plan.handle, self.ws.data_ptr(), self._stream()), "ta3n_loss")

    def discrepancy(self) -> None:
        """main.py:452-505 between ta3n_loss and ta3n_backward: alpha * (DAN: mmd_rbf per selected feature, in chunks of <= 256
        videos; JAN: the joint kernel of logits and video feature) on the first min(Bs, Bt) valid rows of each domain; its gradient
        is added to the logit gradient (region gY) and written to the feature-gradient entry (gV_ext).  HIP kernels through
        ta3n_amd.loss (ta3n_gaussian_kernel / ta3n_mmd_rowdiff); torch only differentiates the O(n^2) glue."""
        if self.dis_DA == "none":
            return
        ns, nt = int(self._hyper.valid_source), int(self._hyper.valid_target)
        if self.world == 1 and os.environ.get("TA3N_NATIVE_DISCREPANCY", "1") != "0":
            # one rank: the whole term from the library (ta3n_discrepancy) - the torch glue around the same kernels (cat / slices / autograd /
            # a dozen small allocations) cost 0.4 - 1.1 ms of host time per step against 0.2 ms for everything else the step launches
            (oy, _), (ov, nv), (ogy, _), (ogv, _) = (self.plan.region(k) for k in ("Y", "V", "gY", "gV_ext"))
            fv = nv // self.B
            if self._disc_scratch is None:
                n = int(self._L.ta3n_discrepancy_scratch_floats(self.Bs, self.Bt, self.C, fv))
                self._disc_scratch = torch.empty(n + 1, dtype=torch.float32, device=self.device)
            loss = self._disc_scratch[-1:]
            _lib.check(self._L.ta3n_discrepancy(self.ws.data_ptr(), oy, self.C, ov, fv, ogy, ogv, self.Bs, self.Bt, ns, nt,
                                                1 if self.dis_DA == "DAN" else 2, int(self.place_dis[0] == "Y"), int(self.place_dis[1] == "Y"),
                                                float(self.alpha), self._disc_scratch.data_ptr(), self._disc_scratch.numel() - 1, loss.data_ptr(),
                                                self._stream()), "ta3n_discrepancy")
            self.loss_d = loss[0]
            return
        y, v = self.region("Y", (self.B, self.C)), self.region("V", (self.B, -1))
        # more than one rank: the reference takes this loss after DataParallel's gather, on the global batch - the ranks' valid rows are
        # gathered in rank order (one sum all-reduce over per-rank slots) and every rank keeps its own gradient rows (parallel.discrepancy_over_ranks)
        self.loss_d, gy, gv = parallel.discrepancy_over_ranks(self.dis_DA, self.place_dis, self.alpha, y, v, self.Bs, ns, nt,
                                                               self.pg if self.world > 1 else None, world=self.world, rank=self.rank)
        self.region("gY", (self.B, self.C)).add_(gy)
        self.region("gV_ext", (self.B, -1)).copy_(gv)

    def sync_buffers(self, src: int = 0) -> None:
        """use_bn under more than one rank: the BatchNorm running statistics that persist are replica `src`'s - nn.DataParallel re-creates
        its replicas from the module on device 0 before every forward and only that module's buffers survive it (main.py:79) - so before
        a validation pass or a checkpoint every rank takes rank src's.  A collective: every rank calls it (no-op on one rank / without BN)."""
        if self.bn_running is None or self.world == 1:
            return
        packed = torch.cat((self.bn_running.reshape(-1), self.bn_running.new_tensor([float(self.bn_batches)])))
        parallel.broadcast_(packed, src=src, group=self.pg)
        self.bn_running.copy_(packed[:-1].view_as(self.bn_running))
        self.bn_batches = int(packed[-1].item())

    def backward(self) -> None:
        _lib.check(self._L.ta3n_backward(self.plan.handle, self.X.data_ptr(), self.P.data_ptr(), self.G.data_ptr(),
                                         self.ws.data_ptr(), self._stream()), "ta3n_backward")

    def _region2(self, name: str, shape=None) -> torch.Tensor:
        off, n = self.plan.region(name)
        t = self.ws2[off:off + n]
        return t.view(shape) if shape is not None else t

    def mcd_source_loss(self) -> None:
        """main.py:447-448 between ta3n_loss and ta3n_backward: + CrossEntropy(out_source_2, label) over the valid source rows - its
        logit gradient goes to region gY2 (the loss kernel knows nothing of the second classifier)."""
        if self.ens_DA != "MCD":
            return
        if self._mcd_native:      # the library's kernels (ta3n_mcd_source_loss): the torch assembly below cost ~0.2 ms of host time per step
            _lib.check(self._L.ta3n_mcd_source_loss(self.plan.handle, self.ws.data_ptr(), self._mcd_buf.data_ptr(), self._mcd_buf[-4:].data_ptr(),
                                                    self._stream()), "ta3n_mcd_source_loss")
            self.loss_c2 = self._mcd_buf[-4]
            return
        ns = int(self._hyper.valid_source)
        if self._flags & _lib.FLAG_ATTN_ENTROPY:      # the attentive entropy of the TARGET rows is taken on the second pass's logits
            # (mcd_second_forward); right after ta3n_loss those rows of gY carry nothing but that term
            self.region("gY", (self.B, self.C))[self.Bs:] = 0
        y2 = self.region("Y2", (self.B, self.C))
        g2 = self.region("gY2", (self.B, self.C))
        g2.zero_()
        if ns > 0:
            lab = self._labels[:ns].long()
            lp = torch.log_softmax(y2[:ns], 1)
            inv = float(self._hyper.inv_n_cls)
            g = lp.exp()
            g[torch.arange(ns, device=self.device), lab] -= 1.0
            g2[:ns] = g * inv
            self.loss_c2 = -(lp[torch.arange(ns, device=self.device), lab]).sum() * inv
        else:
            self.loss_c2 = y2.new_zeros(())

    def mcd_second_forward(self) -> None:
        """main.py:548-556: the whole model again with reverse=True (GradReverse(mu) between dropout_v and the video heads,
        models.py:682-684; fresh dropout masks) and loss_s = -dis_MCD(out_target, out_target_2) (loss.py:29-30) over the valid target
        rows.  The reference REBINDS out_target to this pass's logits before it assembles the attentive entropy (main.py:549 vs
        :559-562), so the target half of that loss belongs to this pass too: its logit gradient moves from the first pass's gY to
        this one's (where GradReverse(mu) scales it on the way to the features), and the first pass's video-domain logits, which
        weight it, see this pass's entropies."""
        self.loss_e_shift = None                  # (a rank whose target shard is empty this step must not report the previous step's shift)
        h = _lib.Hyper.from_buffer_copy(self._hyper)
        h.reverse, h.mu = 1, float(self.mu)
        h.seed_i, h.seed_v = dropout_seeds(int(self._hyper.seed_i) ^ 0x5bd1e995, self.rank)
        if self.mcd_raw_seeds is not None:      # a caller that draws the second pass's stream seeds itself (main.py: as VideoModel.forward would)
            h.seed_i, h.seed_v = int(self.mcd_raw_seeds[0]) & 0xFFFFFFFF, int(self.mcd_raw_seeds[1]) & 0xFFFFFFFF
        L, plan = self._L, self.plan
        _lib.check(L.ta3n_set_hyper(plan.handle, self.ws2.data_ptr(), C.byref(h), self._stream()), "ta3n_set_hyper")
        if self.bf16_store:      # the second workspace's launches read the parameter / input twins too (round 6: the unfused lists on twins) - the
            # optimiser and set_batch keep them in the FIRST workspace: copied over, 11 MB at the headline shape
            for name in ("p16", "x16", "p16_lo", "x16_lo"):
                if name in plan.regions:
                    off, n = plan.region(name)
                    self.ws2[off:off + n].copy_(self.ws[off:off + n], non_blocking=True)
        _lib.check(L.ta3n_forward(plan.handle, self.X.data_ptr(), self.P.data_ptr(), self.ws2.data_ptr(), self._stream()), "ta3n_forward")
        nt = int(self._hyper.valid_target)
        if self._mcd_native:      # ta3n_mcd_second_loss: clears the second workspace's gradient entries, loss_s, the moved entropy term, all gradients
            _lib.check(L.ta3n_mcd_second_loss(plan.handle, self.ws.data_ptr(), self.ws2.data_ptr(), int(self._global_target), self._mcd_buf.data_ptr(),
                                              self._mcd_buf[-4:].data_ptr(), self._stream()), "ta3n_mcd_second_loss")
            self.loss_s = self._mcd_buf[-3]
            if nt > 0 and (self._flags & _lib.FLAG_ATTN_ENTROPY):
                self.loss_e_shift = (self._mcd_buf[-2], self._mcd_buf[-1])
            return
        for name in ("gY", "gY2", "gPr", "gPv", "gPf", "g_attn", "gV_ext"):
            if name in plan.regions:
                self._region2(name).zero_()
        self.loss_s = self.ws2.new_zeros(())
        if nt == 0:
            return
        rows = slice(self.Bs, self.Bs + nt)
        y = self._region2("Y", (self.B, self.C))[rows].detach().clone().requires_grad_(True)
        y2 = self._region2("Y2", (self.B, self.C))[rows].detach().clone().requires_grad_(True)
        # loss.py:29-30 torch.mean over (target videos x classes) of the GLOBAL batch: ranks divide by the job-wide count and their
        # gradients are summed like every other term (set_hyper)
        loss = -torch.sum(torch.abs(torch.softmax(y, 1) - torch.softmax(y2, 1))) / float(max(self._global_target, 1) * self.C)
        self.loss_s = loss.detach()
        if self._flags & _lib.FLAG_ATTN_ENTROPY:
            def ent(z):
                return torch.sum(-torch.softmax(z, 1) * torch.log_softmax(z, 1), 1)
            scale = float(self._hyper.gamma) * float(self._hyper.inv_n_ent)
            y_first = self.region("Y", (self.B, self.C))[rows].detach()
            pv = self.region("Pv", (self.B, 2))[rows].detach().clone().requires_grad_(True)
            w = 1.0 + ent(pv)
            e_new, e_old = scale * torch.sum(w * ent(y)), scale * torch.sum(w * ent(y_first))
            gpv, = torch.autograd.grad(e_new - e_old, pv, retain_graph=True)
            de = (e_new - e_old).detach()
            self.loss_e_shift = (de, de / float(self._hyper.gamma) if float(self._hyper.gamma) != 0.0 else de * 0)
            self.region("gPv", (self.B, 2))[rows] += gpv   # (zero when the two passes drew the same dropout masks)
            loss = loss + e_new
        g1, g2 = torch.autograd.grad(loss, (y, y2))
        self._region2("gY", (self.B, self.C))[rows] = g1
        self._region2("gY2", (self.B, self.C))[rows] = g2

    def mcd_second_backward(self) -> None:
        _lib.check(self._L.ta3n_backward(self.plan.handle, self.X.data_ptr(), self.P.data_ptr(), self.G2.data_ptr(), self.ws2.data_ptr(),
                                         self._stream()), "ta3n_backward")
        n = self.plan.live_floats
        self.G[:n].add_(self.G2[:n])

    def all_reduce_grads(self) -> None:
        if self.skip_collective:      # measurement only (bench.py: the step without its exchange -> exposed collective time)
            return
        if self.peer is not None and self.comm is None:      # (with a communicator the library routes through the attached peer itself)
            self.peer.all_reduce_sum_(self.G[: self.plan.live_floats])
            return
        if self.comm is not None:      # RCCL on the step's stream, enqueued by the library
            _lib.check(self._L.ta3n_all_reduce_sum(self.comm.handle, self.G.data_ptr(), self.plan.live_floats,
                                                   self._g16.data_ptr() if self._g16 is not None else None, self._stream()),
                       "ta3n_all_reduce_sum")
        elif self.world > 1:
            parallel.all_reduce_sum_(self.G[: self.plan.live_floats], self.pg)
        elif self._ddp_selftest:
            w = parallel.all_reduce_sum_async(self.G[: self.plan.live_floats], self.pg, True)
            if w is not None:
                w.wait()

    def check_exchange(self) -> None:
        """Raise if the peer all-reduce (TA3N_DDP_PEER=1) ever gave up waiting for a rank: its error word is sticky and every later
        exchange delivers NaN (csrc/ta3n_peer.hip).  Synchronises the stream - call it where the host synchronises anyway (a log line,
        before validation / a checkpoint, the end of an epoch).  RCCL needs no such check: its collectives block."""
        if self.peer is not None:
            self.peer.status(self.device)

    def sgd_step_fused(self) -> None:
        """Update with the global norm taken from the fused step's per-tile partials (single rank only)."""
        _lib.check(self._L.ta3n_sgd_step_fused(self.plan.handle, self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(),
                                               self.ws.data_ptr(), self._stream()), "ta3n_sgd_step_fused")

    def sgd_step(self) -> None:
        _lib.check(self._L.ta3n_sgd_step(self.plan.handle, self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(),
                                         self.ws.data_ptr(), self._stream()), "ta3n_sgd_step")

    def fused_step(self) -> None:
        """forward + loss + backward through the fused launch sequence (include/ta3n_hip.h: ta3n_train_step)."""
        _lib.check(self._L.ta3n_train_step(self.plan.handle, self.X.data_ptr(), self.P.data_ptr(), self.G.data_ptr(),
                                           self.ws.data_ptr(), self._stream()), "ta3n_train_step")
        self._bn_track()

    def _fused_step_overlapped_allreduce(self) -> None:
        """N > 1: the all-reduce of every gradient but the shared frame FC's (the last launch's output, 4.2 of the
        13.9 MB) starts before that launch and runs beside it over xGMI; the rest follows; both are joined before
        the update.  Same collectives in the same order on every rank."""
        if self.comm is not None:      # the same schedule inside the library: two HIP streams, three event edges, no framework
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(self.device)
            _lib.check(self._L.ta3n_train_step_ddp(self.plan.handle, self.comm.handle, self.X.data_ptr(), self.P.data_ptr(),
                                                   self.G.data_ptr(), self.ws.data_ptr(),
                                                   self._g16.data_ptr() if self._g16 is not None else None, self._stream(),
                                                   C.c_void_p(self._comm_stream.cuda_stream)), "ta3n_train_step_ddp")
            self._bn_track()
            return
        n = self._L.ta3n_num_phases(self.plan.handle, 4)
        args = (self.plan.handle, self.X.data_ptr(), self.P.data_ptr(), self.G.data_ptr(), self.ws.data_ptr())
        _lib.check(self._L.ta3n_train_step_range(*args, 0, n - 1, self._stream()), "ta3n_train_step_range")
        w1 = parallel.all_reduce_sum_async(self.G[self._n_first: self.plan.live_floats], self.pg, self._ddp_selftest)
        _lib.check(self._L.ta3n_train_step_range(*args, n - 1, 1, self._stream()), "ta3n_train_step_range")
        w2 = parallel.all_reduce_sum_async(self.G[: self._n_first], self.pg, self._ddp_selftest)
        for w in (w1, w2):
            if w is not None:
                w.wait()
        self._bn_track()

    def _enqueue_step(self) -> None:
        if self._sharded:
            self.fused_step()
            self._shard_reduce_scatter()
            h = self._hyper
            self._sharded_update(float(h.lr), float(h.momentum), float(h.weight_decay), float(h.clip), None)
            return
        if self.fused and (self.world > 1 or self._ddp_selftest) and self._ddp_buckets == 2:
            self._fused_step_overlapped_allreduce()
            self.sgd_step()
            return
        if self.fused:
            self.fused_step()
        else:
            self.forward()
            self.loss()
            self.mcd_source_loss()
            self.discrepancy()
            if self.ens_DA == "MCD":
                self.mcd_second_forward()
            self.backward()
            if self.ens_DA == "MCD":
                self.mcd_second_backward()
        self.all_reduce_grads()
        if self.fused and self.world == 1 and not self._ddp_selftest:
            self.sgd_step_fused()       # local gradients are final: their norm partials are already in ws
        else:
            self.sgd_step()

    # ---- deferred, overlapped update ----
    def _apply_pending(self, overlap: bool):
        """Enqueue the pending update.  overlap: returns the event the next step must join after its first launch."""
        if self._pending is None:
            return None
        lr, mu, wd, clip = self._pending
        self._pending = None
        fused_norm = int(self.fused and self.world == 1)
        main = torch.cuda.current_stream(self.device)
        L, h = self._L, self.plan.handle

        def rng(lo, hi, stream):
            _lib.check(L.ta3n_sgd_range(h, self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.ws.data_ptr(), lo, hi,
                                        fused_norm, lr, mu, wd, clip, C.c_void_p(stream.cuda_stream)), "ta3n_sgd_range")
        if not overlap:
            rng(0, self.plan.live_floats, main)
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        rng(0, self._n_first, main)                 # (+ the gradient-norm pass when the norm is not fused)
        fork = torch.cuda.Event()
        fork.record(main)
        self._side.wait_event(fork)
        rng(self._n_first, self.plan.live_floats, self._side)
        join = torch.cuda.Event()
        join.record(self._side)
        return join

    def flush(self) -> None:
        """Apply a deferred update (train_step(defer_update=True)) so that the parameters are current."""
        if self._sharded and self._pending is not None:
            lr, mu, wd, clip = self._pending
            self._pending = None
            self._sharded_update(lr, mu, wd, clip, None)
            return
        self._apply_pending(overlap=False)

    # ---- sharded update (reduce-scatter / own-shard optimiser / all-gather) ----
    def _shard_reduce_scatter(self) -> None:
        """The own shards of self.G receive the job-wide sum (RCCL reduce-scatter from the C ABI; over a torch.distributed group
        without RCCL - gloo tests - an all-reduce, of which only the own shards are then read)."""
        if self.skip_collective:
            return
        if self.comm is not None:
            _lib.check(self._L.ta3n_shard_reduce_scatter(self.plan.handle, self.comm.handle, self.G.data_ptr(),
                                                         self._g16.data_ptr() if self._g16 is not None else None, self._stream()),
                       "ta3n_shard_reduce_scatter")
        elif self.world > 1:
            parallel.all_reduce_sum_(self.G[: self._shard_layout[3]], self.pg)

    def _sharded_update(self, lr: float, mu: float, wd: float, clip: float, nxt) -> None:
        """clip + Nesterov SGD on this rank's shards with the job-wide gradient norm, then everybody's updated shards gathered."""
        h, L = self.plan.handle, self._L
        nx = C.byref(nxt) if nxt is not None else None
        if self.comm is not None and not self.skip_collective:
            _lib.check(L.ta3n_sharded_update(h, self.comm.handle, self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.ws.data_ptr(),
                                             lr, mu, wd, clip, nx, self._stream()), "ta3n_sharded_update")
            return
        _lib.check(L.ta3n_shard_sumsq(h, self.G.data_ptr(), self.ws.data_ptr(), self.rank, self.world, self._stream()), "ta3n_shard_sumsq")
        if self.world > 1 and not self.skip_collective:      # one float per rank; every other slot is zero: a sum is the gather
            off, _ = self.plan.region("norm_part")
            parallel.all_reduce_sum_(self.ws[off:off + self.world], self.pg)
        _lib.check(L.ta3n_sgd_shard(h, self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.ws.data_ptr(), self.rank, self.world,
                                    lr, mu, wd, clip, nx, self._stream()), "ta3n_sgd_shard")
        if self.world > 1 and not self.skip_collective:
            a_chunk, a_end, b_chunk, b_end = self._shard_layout
            for base, chunk in ((0, a_chunk), (a_end, b_chunk)):
                parts = [self.P[base + r * chunk: base + (r + 1) * chunk] for r in range(self.world)]
                torch.distributed.all_gather(parts, parts[self.rank].clone(), group=self.pg)
        self.refresh_bf16(params=True)

    def train_step_deferred(self, beta: Sequence[float], gamma: float, lr: float, **hyper_kw) -> None:
        """train_step whose optimiser update is postponed to the start of the next call, where all of it except the shared
        frame FC overlaps the next step's first launch.  Same arithmetic; call flush() before reading parameters."""
        if not self.fused:
            raise _lib.Ta3nError("deferred updates need the fused step")
        join = self._apply_pending(overlap=True)
        self.set_hyper(beta, gamma, lr, train=True, **hyper_kw)
        ev = C.c_void_p(join.cuda_event) if join is not None else None
        _lib.check(self._L.ta3n_train_step_join(self.plan.handle, self.X.data_ptr(), self.P.data_ptr(), self.G.data_ptr(),
                                                self.ws.data_ptr(), self._stream(), ev), "ta3n_train_step_join")
        self._bn_track()
        self.all_reduce_grads()
        self._pending = (float(lr), float(self.momentum), float(self.weight_decay), float(self.clip) if self.clip is not None else 0.0)
        self.step_count += 1

    def train_step_pipelined(self, beta: Sequence[float], gamma: float, lr: float, **hyper_kw) -> None:
        """train_step whose optimiser update is postponed to the start of the next call, where ONE launch applies it and
        leaves the next step's scalars in the workspace (ta3n_sgd_step_next): no per-step host-to-device copy.  Same
        arithmetic as train_step; call flush() before reading parameters."""
        if not self.fused:
            raise _lib.Ta3nError("pipelined updates need the fused step")
        first = self._pending is None
        self.set_hyper(beta, gamma, lr, train=True, upload=first, **hyper_kw)
        if self._sharded:
            if not first:
                lr_p, mu, wd, clip = self._pending
                self._pending = None
                self._sharded_update(lr_p, mu, wd, clip, self._hyper)
            self.fused_step()
            self._shard_reduce_scatter()
            self._pending = (float(lr), float(self.momentum), float(self.weight_decay), float(self.clip) if self.clip is not None else 0.0)
            self.step_count += 1
            return
        two_buckets = (self.world > 1 or self._ddp_selftest) and self._ddp_buckets == 2
        stepped = False
        if not first:
            lr_p, mu, wd, clip = self._pending
            self._pending = None
            fused_norm = int(self.world == 1 and not self._ddp_selftest)
            if self._side_update and not two_buckets:
                # the update of everything but the shared frame FC rides in the new step's first launch
                _lib.check(self._L.ta3n_train_step_after_update(self.plan.handle, self.X.data_ptr(), self.P.data_ptr(),
                                                                self.G.data_ptr(), self.M.data_ptr(), self.ws.data_ptr(), fused_norm,
                                                                lr_p, mu, wd, clip, C.byref(self._hyper), self._stream()),
                           "ta3n_train_step_after_update")
                self._bn_track()
                stepped = True
            else:
                _lib.check(self._L.ta3n_sgd_step_next(self.plan.handle, self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(),
                                                      self.ws.data_ptr(), fused_norm, lr_p, mu, wd, clip, C.byref(self._hyper),
                                                      self._stream()), "ta3n_sgd_step_next")
        if two_buckets:
            self._fused_step_overlapped_allreduce()
        else:
            if not stepped:
                self.fused_step()
            self.all_reduce_grads()
        self._pending = (float(lr), float(self.momentum), float(self.weight_decay), float(self.clip) if self.clip is not None else 0.0)
        self.step_count += 1

    def hyper_for(self, beta: Sequence[float], gamma: float, lr: float, step: Optional[int] = None, **hyper_kw) -> "_lib.Hyper":
        """The per-step scalars of one step as a fresh ta3n_hyper (what set_hyper would upload)."""
        keep = self._hyper
        self._hyper = _lib.Hyper()
        try:
            self.set_hyper(beta, gamma, lr, train=True, upload=False, seed=step, **hyper_kw)
            return self._hyper
        finally:
            self._hyper = keep

    def hyper_array(self, entries: Sequence[Sequence], step0: int):
        """(ta3n_hyper * len(entries)) for the steps step0, step0 + 1, ... with entries[k] = (beta, gamma, lr): byte for byte what
        hyper_for returns entry by entry (tests/test_schedules_and_rng.py), filled column-wise.  Entry 0 comes from hyper_for - every
        field the way set_hyper writes it - and the fields that change from step to step (beta, gamma, lr, the two dropout seeds) are
        written for all steps at once: the per-entry loop cost 4 - 7 us of host time per step INSIDE a train_steps call, before its first
        launch (round 5: at the 0.1 ms headline step that was 4 - 6 % of the measured step time)."""
        m = len(entries)
        hy = (_lib.Hyper * m)()
        if m == 0:
            return hy
        h0 = self.hyper_for(*entries[0], step=step0)
        size = C.sizeof(_lib.Hyper)
        if m == 1:
            C.memmove(hy, C.byref(h0), size)
            return hy
        if _LEGACY_HOST_PREP:              # A/B aid: the per-entry loop of rounds 3-4
            for k in range(m):
                h = h0 if k == 0 else self.hyper_for(*entries[k], step=step0 + k)
                C.memmove(C.byref(hy, k * size), C.byref(h), size)
            return hy
        np, dt = _hyper_dtype()
        np.frombuffer(hy, dtype=np.uint8).reshape(m, size)[:] = np.frombuffer(h0, dtype=np.uint8)      # every entry = entry 0 ...
        a = np.frombuffer(hy, dtype=dt)
        sched = np.array([(b_[0], b_[1], b_[2], g_, lr_) for b_, g_, lr_ in entries], dtype=np.float32)  # ... but (float -> fp32, RNE like c_float):
        a["beta"] = sched[:, :3]
        a["gamma"] = sched[:, 3]
        a["lr"] = sched[:, 4]
        # dropout_seeds(step, rank) for all steps at once, in uint32 (products wrap mod 2^32, which is what the masks there keep)
        st2 = (np.arange(m, dtype=np.uint32) + np.uint32(step0 & 0xFFFFFFFF)) * np.uint32(2)
        r = (0xC2B2AE3D * int(self.rank)) & 0xFFFFFFFF
        a["seed_i"] = ((st2 + np.uint32(1)) * np.uint32(0x9E3779B1)) ^ np.uint32(r)
        a["seed_v"] = ((st2 + np.uint32(2)) * np.uint32(0x85EBCA77)) ^ np.uint32((r * 0x27D4EB2F) & 0xFFFFFFFF)
        return hy

    def _feeds(self, feeds, k0: int):
        """ta3n_feed structs (and the id tables that must outlive the enqueued gathers) of a multi-step call."""
        fd, keep = [None, None], []
        if feeds is not None:
            for i, (store, ids) in enumerate(feeds):
                if store is None:
                    continue
                ids = ids[k0:].to(device=self.device, dtype=torch.int32).contiguous()
                keep.append(ids)
                f = _lib.Feed()
                f.store = store.store.data_ptr(); f.bf16 = int(store.bf16); f.ids_per_step = ids.shape[1]
                f.first_row = store.first_row.data_ptr(); f.num_frames = store.num_frames.data_ptr()
                f.labels = store.labels.data_ptr(); f.video_ids = ids.data_ptr()
                fd[i] = f
        return fd, keep

    def train_steps(self, schedule: Sequence[Sequence], feeds=None, fused_update: Optional[bool] = None, _split: bool = True) -> None:
        """len(schedule) pipelined steps enqueued by ONE call into the library (ta3n_train_steps): schedule[k] = (beta, gamma, lr)
        of step k (main.py:350-352, 620-621 evaluated ahead of time).  Same launches, same results as calling
        train_step_pipelined once per entry; the host leaves the step's critical path (on a slow core the per-step ctypes call +
        9 launches cost as much wall time as the GPU needs for the step).  feeds: optional (source, target) pairs of
        (FeatureStore, int32 device tensor [len(schedule), n]) - the batch of step k is then assembled on the device before it.
        With a process group and the library's RCCL communicator the step's all-reduce is part of the same call.
        fused_update (TA3N_FUSED_UPDATE=1; single rank): the optimiser runs INSIDE the gradient launches
        (ta3n_train_steps_fused_update) - bit-identical whenever no step clips, within fp32 rounding of the clip correction otherwise.
        Measured time-neutral (profiles/r03_fused_update_ab.txt: the 63 MB of optimiser traffic cost ~8 us wherever they run), so the
        default stays the update as launches of its own (ta3n_train_steps), which is also what the data-parallel path needs."""
        n = len(schedule)
        if n == 0:
            return
        if _split:      # a caller's call (not a piece of one): what the pieces enqueue must outlive the enqueued gathers - every piece
            # APPENDS its id tables / scalar arrays (a piece used to replace the previous piece's), and the previous call's
            # are dropped only when the call after it starts
            self._feed_keep_prev, self._feed_keep = getattr(self, "_feed_keep", None), []
        ddp = self.world > 1 or self._ddp_selftest
        if fused_update is None:
            fused_update = os.environ.get("TA3N_FUSED_UPDATE", "0") == "1"
        if fused_update and not ddp and self.fused and self.bn_running is None and self._L.ta3n_has_fused_update(self.plan.handle) == 1:
            # the optimiser inside the gradient launches (ta3n_train_steps_fused_update): no separate update launches at all
            self.flush()
            if self._P2 is None:
                self._P2 = torch.empty_like(self.P)
            hy = self.hyper_array(schedule, self.step_count)
            fd, keep = self._feeds(feeds, 0)
            _lib.check(self._L.ta3n_train_steps_fused_update(
                self.plan.handle, self.X.data_ptr(), self.P.data_ptr(), self._P2.data_ptr(), self.G.data_ptr(), self.M.data_ptr(),
                self.ws.data_ptr(), float(self.momentum), float(self.weight_decay), float(self.clip) if self.clip is not None else 0.0,
                hy, n, C.byref(fd[0]) if fd[0] is not None else None, C.byref(fd[1]) if fd[1] is not None else None, self._stream()),
                "ta3n_train_steps_fused_update")
            if keep:
                self._feed_keep.append(keep)
            self._hyper = _lib.Hyper.from_buffer_copy(hy[n - 1])
            self.step_count += n
            return
        if not self.can_batch_steps():
            # no single library call for this configuration (torch.distributed fallback, two buckets, TA3N_SIDE_UPDATE=0, unfused):
            # the same steps one by one, the batch of each assembled on the device before it
            for k, (beta, gamma, lr) in enumerate(schedule):
                if feeds is not None:
                    for (store, ids), first in zip(feeds, (0, self.Bs)):
                        if store is not None:
                            store.gather_into(self, ids[k], first, labels_out=self._labels[: self.Bs] if first == 0 else None)
                (self.train_step_pipelined if self.fused else self.train_step)(beta, gamma, lr)
            return
        if _split and n >= 4 and not self._sharded and not _LEGACY_HOST_PREP:
            # The first step goes out on its own and the rest of the schedule is prepared while the GPU runs it (a long schedule in two
            # further pieces: 16 steps, then the rest): the host work in front of a call's first launch (the ta3n_hyper array: ~7 us +
            # ~0.5 us per step) then costs the GPU one step's preparation, whatever the length of the schedule.  Same launches in the
            # same order on the same stream; every later piece opens with the update the piece before it left pending, exactly as the
            # next step of a single call would.
            rows = (lambda lo, hi: None) if feeds is None else (lambda lo, hi: tuple((st, ids if st is None else ids[lo:hi]) for st, ids in feeds))
            lo = 0
            for hi in ((1, n) if n <= 33 else (1, 17, n)):
                self.train_steps(schedule[lo:hi], feeds=rows(lo, hi), fused_update=fused_update, _split=False)
                lo = hi
            return
        job, keep, n_run = self._steps_job(schedule, feeds)
        if job is None:
            return
        if self._sharded:      # the same steps with the sharded update; region B's collectives on a second stream
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(self.device)
            two = os.environ.get("TA3N_DDP_SHARDED_STREAMS", "1") == "2"
            _lib.check(self._L.ta3n_train_steps_sharded(job.plan, job.comm, job.x, job.params, job.grads, job.momentum, job.ws, job.lr_pending,
                                                        job.momentum_coef, job.weight_decay, job.clip, job.hypers, n_run, job.source, job.target,
                                                        job.scratch_bf16, job.stream,
                                                        C.c_void_p(self._comm_stream.cuda_stream) if two else None), "ta3n_train_steps_sharded")
            self._steps_done(schedule, keep, n_run)
            return
        _lib.check(self._L.ta3n_train_steps(job.plan, job.x, job.params, job.grads, job.momentum, job.ws, job.fused_norm, job.lr_pending,
                                            job.momentum_coef, job.weight_decay, job.clip, job.hypers, n_run, job.source, job.target,
                                            job.comm, job.scratch_bf16, job.stream), "ta3n_train_steps")
        self._steps_done(schedule, keep, n_run)

    def can_batch_steps(self) -> bool:
        """True when train_steps enqueues its steps through ONE library call (ta3n_train_steps / ta3n_train_steps_multi) - also the
        condition under which device-side batch feeds are accepted."""
        ddp = self.world > 1 or self._ddp_selftest
        if self._sharded:
            return bool(self.comm is not None and not self.skip_collective)
        return bool(self.fused and self._side_update and
                    not (ddp and (self.comm is None or self._ddp_buckets == 2 or self.skip_collective)))

    def _steps_job(self, schedule, feeds=None):
        """The ta3n_steps_job of `schedule` on this engine's buffers and the CURRENT stream (+ what must outlive the enqueued work, and
        the number of steps it covers).  The very first step of an engine has no update to open with: it is enqueued here, on its
        own; (None, ...) when nothing is left for the library call."""
        n, k0 = len(schedule), 0
        ddp = self.world > 1 or self._ddp_selftest
        if self._pending is None:
            if feeds is not None:
                for (store, ids), first in zip(feeds, (0, self.Bs)):
                    if store is not None:
                        store.gather_into(self, ids[0], first, labels_out=self._labels[: self.Bs] if first == 0 else None)
            self.train_step_pipelined(*schedule[0])
            k0 = 1
            if n == 1:
                return None, [], 0
        hy = self.hyper_array(schedule[k0:], self.step_count)
        self._job_last_hyper = _lib.Hyper.from_buffer_copy(hy[n - k0 - 1])      # (for _steps_done: the scalars of the last step enqueued)
        lr_p, mu, wd, clip = self._pending
        fd, keep = self._feeds(feeds, k0)
        job = _lib.StepsJob()
        job.plan, job.x, job.params, job.grads = self.plan.handle, self.X.data_ptr(), self.P.data_ptr(), self.G.data_ptr()
        job.momentum, job.ws, job.fused_norm = self.M.data_ptr(), self.ws.data_ptr(), 0 if ddp else 1
        job.lr_pending, job.momentum_coef, job.weight_decay, job.clip = lr_p, mu, wd, clip
        job.hypers = C.cast(hy, C.POINTER(_lib.Hyper))
        job.source = C.pointer(fd[0]) if fd[0] is not None else None
        job.target = C.pointer(fd[1]) if fd[1] is not None else None
        job.comm = self.comm.handle if ddp else None
        job.scratch_bf16 = self._g16.data_ptr() if (ddp and self._g16 is not None) else None
        job.stream = self._stream()
        return job, keep + [hy, fd], n - k0

    def _steps_done(self, schedule, keep, n_run: int) -> None:
        if keep:                              # the id tables must outlive the enqueued gathers
            if not isinstance(getattr(self, "_feed_keep", None), list):
                self._feed_keep = []
            self._feed_keep.append(keep)
        last = schedule[-1]
        self._pending = (float(last[2]), float(self.momentum), float(self.weight_decay), float(self.clip) if self.clip is not None else 0.0)
        self._hyper = self._job_last_hyper if self._job_last_hyper is not None else self.hyper_for(*last, step=self.step_count + n_run - 1)
        self._job_last_hyper = None
        self.step_count += n_run
        self._bn_track(n_run)

    def chain_status(self) -> None:
        """Raises if a chained launch enqueued so far left a hand-off unserved (synchronises; tests / end of a run)."""
        if self._L.ta3n_chain_status(self.plan.handle, self.ws.data_ptr(), self._stream()) != 0:
            raise _lib.Ta3nError(self._L.ta3n_last_error().decode())

    def capture(self) -> None:
        """Capture forward+loss+backward(+all-reduce)+update into one hipGraph (shapes
        are static).  set_hyper / set_batch stay outside: they only write device buffers."""
        torch.cuda.synchronize(self.device)
        keep_p, keep_m = self.P.clone(), self.M.clone()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self._enqueue_step()      # warm-up outside capture (lazy RCCL init etc.); state restored below
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.P.copy_(keep_p)
        self.M.copy_(keep_m)
        self.refresh_bf16(params=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enqueue_step()
        self.graph = g

    def train_step(self, beta: Sequence[float], gamma: float, lr: float, **hyper_kw) -> None:
        """One optimisation step on the batch already in the static buffers."""
        self.set_hyper(beta, gamma, lr, train=True, **hyper_kw)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue_step()
        self.step_count += 1

    def time_phases(self, reps: int = 20, all_groups: bool = False):
        """[(kind, tile, n_tasks, ms)] per launch of one train step as this engine runs it (the fused
        sequence + optimiser, or forward/loss/backward + optimiser), HIP events on the launch stream."""
        n = len(self.plan.description["phases"])
        ms = (C.c_float * n)()
        kinds = (C.c_int32 * n)()
        groups = (C.c_int32 * n)()
        _lib.check(self._L.ta3n_time_phases(self.plan.handle, self.X.data_ptr(), self.P.data_ptr(), self.G.data_ptr(),
                                            self.M.data_ptr(), self.ws.data_ptr(), self._stream(), reps, ms, kinds, groups, n),
                   "ta3n_time_phases")
        want = (4, 3) if self.fused else (0, 1, 2, 3)
        skip_norm = self.fused and self.world == 1      # ta3n_sgd_step_fused has no grad-norm launch
        return [(int(kinds[i]), ph["tile"], ph["task_count"], float(ms[i]))
                for i, ph in enumerate(self.plan.description["phases"])
                if all_groups or (int(groups[i]) in want and not (skip_norm and int(kinds[i]) == 4))]

    def time_update_launches(self, reps: int = 20, lr: float = 1e-3):
        """(ms of the optimiser launch that opens a pipelined step, ms of the step's first GEMM launch carrying the rest of the
        update as side workgroups) - what train_step_pipelined runs instead of the plain first launch time_phases reports.
        Applies the update `reps` times: a measurement aid for the END of a benchmark run."""
        out = (C.c_float * 2)()
        fused_norm = int(self.world == 1 and not self._ddp_selftest)
        _lib.check(self._L.ta3n_time_update_launches(self.plan.handle, self.X.data_ptr(), self.P.data_ptr(), self.G.data_ptr(),
                                                     self.M.data_ptr(), self.ws.data_ptr(), fused_norm, float(lr), float(self.momentum),
                                                     float(self.weight_decay), float(self.clip) if self.clip is not None else 0.0,
                                                     self._stream(), reps, out), "ta3n_time_update_launches")
        return float(out[0]), float(out[1])

    def gemm_phase_times(self, reps: int = 20):
        """ms of every GEMM launch of the plan, in plan order (the index space of phase_tiles)."""
        return [ms for kind, _, _, ms in self.time_phases(reps, all_groups=True) if kind == 0]

    # ---- validation (main.validate, main.py:669-761) ----
    def evaluate_batch(self, val_data: torch.Tensor, val_label: torch.Tensor, reset: bool = False) -> None:
        """Forward in eval mode (no dropout, beta = 0: main.py:707) on up to batch_source videos and accumulate
        loss / top-1 / top-5 / confusion matrix on the device; read them with eval_results()."""
        n = val_data.shape[0]
        if n > self.Bs:
            raise ValueError(f"at most batch_source = {self.Bs} validation videos per call")
        self.X[: n * self.T].copy_(val_data.reshape(-1, self.D), non_blocking=True)
        self.refresh_bf16(x=True)      # (the forward launches of a twin plan read the input's bf16 twin - since round 6 the unfused lists do too)
        self._labels[:n].copy_(val_label.to(torch.int32), non_blocking=True)
        self.set_hyper([0.0, 0.0, 0.0], 0.0, 0.0, train=False, valid_source=n, valid_target=0)
        self.forward()
        _lib.check(self._L.ta3n_eval_metrics(self.plan.handle, self.ws.data_ptr(), n, int(reset), self._stream()), "ta3n_eval_metrics")

    def eval_results(self) -> Dict[str, object]:
        m = self.region("metrics")[:4].tolist()
        n = max(m[3], 1.0)
        off, cnt = self.plan.region("confusion")
        conf = self.ws[off:off + cnt].view(torch.int32).view(self.C, self.C).cpu()
        return dict(loss=m[0] / n, prec1=100.0 * m[1] / n, prec5=100.0 * m[2] / n, n=int(m[3]), confusion=conf)

    # ---- results ----
    def outputs(self) -> Dict[str, torch.Tensor]:
        B, T, NR = self.B, self.T, self.T - 1
        if self.aggregation == "avgpool":      # attn is a placeholder column in the reference (models.py:627-628)
            v = self.region("V", (B, -1))
            out = dict(out=self.region("Y", (B, self.C)), attn=v[:, 0], feat_v=v, feat_f1=self.region("F1", (B, T, self.F)))
            if "Pv" in self.plan.regions:      # general variant: video- / frame-level domain logits (the relation slot repeats the video's)
                out.update(pred_vid=self.region("Pv", (B, 2)), pred_frm=self.region("Pf", (B, T, 2)))
            return out
        return dict(out=self.region("Y", (B, self.C)), attn=self.region("attn", (B, NR)),
                    pred_rel=self.region("Pr", (B, NR, 2)), pred_vid=self.region("Pv", (B, 2)),
                    pred_frm=self.region("Pf", (B, T, 2)), feat_v=self.region("V", (B, -1)),
                    feat_f1=self.region("F1", (B, T, self.F)))

    def losses(self) -> Dict[str, float]:
        v = self.region("losses")[:6].tolist()
        return dict(loss=v[0], loss_c=v[1], loss_adv_rel=v[2], loss_adv_vid=v[3], loss_adv_frm=v[4], loss_e=v[5])


def autotune_phase_tiles(batch_source: int, batch_target: int, num_segments: int, feature_dim: int, fc_dim: int,
                         num_class: int, flags: int = ALL_FLAGS, device=None, reps: int = 10,
                         candidates: Sequence[int] = (114, 118, 214, 124, 221, 222), verbose: bool = False):
    """Pick the fastest GEMM tile shape per launch by measuring each candidate on this
    GPU (HIP events on the launch stream).  Returns (phase_tiles, table)."""
    if flags & _lib.FLAG_F32_SPLIT and all(c < 1000 for c in candidates):
        candidates = [s * 1000 + c for c in candidates for s in (2, 3)]      # split arithmetic: 2 or 3 LDS stages of fp32 images
    if flags & _lib.FLAG_BF16_MFMA and all(c < 1000 for c in candidates):
        candidates = [s * 1000 + c for c in candidates for s in (2, 3)]      # bf16 kernels: 2 or 3 LDS stages
        if flags & _lib.FLAG_BF16_STORE:      # register-blocked tiles of the twin kernel: 128x64, 64x128 (2 / 3 stages), 128x128
            candidates = candidates + [12222, 13222, 22222, 23222, 32222, 32221]
    table = {}
    for cand in candidates:
        eng = TrainEngine(batch_source, batch_target, num_segments, feature_dim, fc_dim, num_class, flags=flags,
                          device=device, tile_config=cand)
        eng.X.uniform_(0, 1)
        for v in eng.param_views().values():
            v.normal_(0, 0.02)
        eng.refresh_bf16(x=True, params=True)
        eng.set_hyper([0.75, 0.75, 0.5], 0.003, 1e-3)
        eng.gemm_phase_times(2)
        table[cand] = eng.gemm_phase_times(reps)
        del eng
    n = len(next(iter(table.values())))
    best = [min(candidates, key=lambda c: table[c][i]) for i in range(n)]
    if verbose:
        for i in range(n):
            print(f"[autotune] gemm phase {i}: " + "  ".join(f"{c}:{1e3 * table[c][i]:.1f}us" for c in candidates) +
                  f"  -> {best[i]}", flush=True)
    return best, table
