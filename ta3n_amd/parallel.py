"""Data parallelism by video: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests).

Replaces the reference's single-process nn.DataParallel (main.py:79: per-step
parameter broadcast, input scatter, output gather, gradient reduce to GPU 0).
Every op of the TA3N step is row-independent per video except the loss means, so
each rank computes its shard with losses divided by the GLOBAL row counts and the
only exchange per step is ONE sum all-reduce of the flat live-gradient buffer
(13.9 MB at the headline configuration).  With that normalisation the summed
gradients equal the reference's global-batch gradients exactly, also for uneven
shards, and every rank applies the identical clip + SGD update, so
parameters never need to be broadcast after initialisation."""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; initialises the default
    process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `n` videos for `rank`; the first n % world ranks get one more."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def padded_shard_size(n: int, world: int) -> int:
    """Static per-rank batch: ceil(n / world).  Ranks with fewer real videos zero-pad, exactly
    like the reference pads a batch to a multiple of gpu_count (main.py:366-372) and trims the
    dummy rows from the loss (main.py:421-422)."""
    return -(-n // world)


def loss_normalisers(global_source: int, global_target: int, num_segments: int) -> Dict[str, float]:
    """1/N of every loss mean over the GLOBAL batch (main.py:446 CE over labelled source
    videos; main.py:533 adversarial CE over (src+tgt) x {T-1, 1, T} rows; loss.py:24)."""
    tot = max(global_source + global_target, 1)
    return dict(inv_n_cls=1.0 / max(global_source, 1), inv_n_rel=1.0 / (tot * (num_segments - 1)),
                inv_n_vid=1.0 / tot, inv_n_frm=1.0 / (tot * num_segments), inv_n_ent=1.0 / tot)


def global_counts(valid_source: int, valid_target: int, group=None, device=None) -> Tuple[int, int]:
    """Job-wide numbers of real (non-dummy) source / target videos of this step."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return valid_source, valid_target
    t = torch.tensor([valid_source, valid_target], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t[0]), int(t[1])


def all_reduce_sum_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """The step's single collective: in-place SUM all-reduce of the flat live-gradient buffer."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def all_reduce_sum_async(flat: torch.Tensor, group=None, even_if_single: bool = False):
    """Asynchronous SUM all-reduce of one gradient bucket; returns the work handle (None when not distributed).
    RCCL runs it on its own stream, ordered after everything already enqueued on the current stream, so kernels
    enqueued afterwards overlap with it; handle.wait() orders the current stream after the collective."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or even_if_single):
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def broadcast_(flat: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Initial parameter synchronisation (once, not per step)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat



def discrepancy_over_ranks(dis_DA: str, place_dis: Sequence[str], alpha: float, y: torch.Tensor, v: torch.Tensor, batch_source: int,
                           valid_source: int, valid_target: int, group=None, world: Optional[int] = None, rank: Optional[int] = None):
    """The discrepancy loss of main.py:452-505 (DAN: mmd_rbf per selected feature in chunks of <= 256 videos; JAN: the joint kernel of
    logits and video feature) on the GLOBAL batch, and the gradient of alpha * loss with respect to THIS rank's rows.

    y [B, C], v [B, Fv]: this rank's logits and video features, source rows first (batch_source of them, valid_source real), then the
    target rows.  The reference computes the loss after nn.DataParallel has gathered every replica's outputs in replica order, on the
    first min(#source, #target) videos of each domain (main.py:467, 482): with more than one rank the valid rows of all ranks are
    gathered in rank order (one collective), every rank evaluates the same global loss and keeps
    the gradient rows that are its own - the parameter gradients the ranks then SUM (the step's all-reduce) are the global-batch ones.
    Returns (loss, gy, gv): the loss value (detached, identical on every rank) and [B, .] gradients (zero rows where a video takes no part)."""
    from . import loss as L
    B = y.size(0)
    # world / rank: the CALLER's (TrainEngine passes its own: an engine that is single-rank while some default process group with more
    # ranks is initialised must not reduce over that group).  None: taken from `group` (the default group if None).
    if world is None:
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if rank is None:
        rank = dist.get_rank(group) if world > 1 else 0
    feat = torch.cat((y.detach(), v.detach()), 1)
    counts = [(int(valid_source), int(valid_target))]
    feats = [feat]
    if world > 1:
        # gathered as a SUM over slots that are zero except the owner's (the counts ride in one extra row): all_reduce is the one
        # collective every backend has for device tensors (gloo cannot all_gather CUDA tensors; the shared-GPU tests run on gloo), and
        # the features are small - B x (C + Fv) floats per rank
        slots = feat.new_zeros((world, B + 1, feat.size(1)))
        slots[rank, :B] = feat
        slots[rank, B, 0], slots[rank, B, 1] = float(valid_source), float(valid_target)
        dist.all_reduce(slots, op=dist.ReduceOp.SUM, group=group)
        cnt = slots[:, B, :2].round().to(torch.int64).tolist()              # one host sync for all ranks' counts
        counts = [(int(a), int(b)) for a, b in cnt]
        feats = [slots[r, :B] for r in range(world)]
    src = torch.cat([f[:ns] for f, (ns, _) in zip(feats, counts)]).requires_grad_(True)
    tgt = torch.cat([f[batch_source:batch_source + nt] for f, (_, nt) in zip(feats, counts)]).requires_grad_(True)
    size = min(src.size(0), tgt.size(0))
    C = y.size(1)
    feat_s, feat_t = [src[:size, :C], src[:size, C:]], [tgt[:size, :C], tgt[:size, C:]]
    muls, nums = [2.0, 2.0], [2, 5]
    loss = src.new_zeros(())
    if size > 0:
        if dis_DA == "JAN":
            loss = L.JAN(feat_s, feat_t, kernel_muls=muls, kernel_nums=nums, fix_sigma_list=[None, None], ver=2)
        else:
            for l in range(2):
                if place_dis[l] != "Y":
                    continue
                sb = min(256, size)
                fs = feat_s[l].reshape((-1, sb) + feat_s[l].shape[1:])
                ft = feat_t[l].reshape((-1, sb) + feat_t[l].shape[1:])
                parts = [L.mmd_rbf(fs[t], ft[t], kernel_mul=muls[l], kernel_num=nums[l], fix_sigma=None, ver=2) for t in range(fs.size(0))]
                loss = loss + sum(parts) / len(parts)
    gy, gv = torch.zeros_like(y), torch.zeros_like(v)
    if loss.requires_grad:
        gs, gt = torch.autograd.grad(alpha * loss, (src, tgt), allow_unused=True)
        s0 = sum(ns for ns, _ in counts[:rank])
        t0 = sum(nt for _, nt in counts[:rank])
        ns, nt = counts[rank]
        if gs is not None and ns:
            gy[:ns], gv[:ns] = gs[s0:s0 + ns, :C], gs[s0:s0 + ns, C:]
        if gt is not None and nt:
            gy[batch_source:batch_source + nt], gv[batch_source:batch_source + nt] = gt[t0:t0 + nt, :C], gt[t0:t0 + nt, C:]
    return loss.detach(), gy, gv


class NativeComm:
    """RCCL communicator owned by libta3n_hip.so (include/ta3n_hip.h: ta3n_comm_*): the step's collectives are enqueued by
    the library on the step's own HIP streams.  The 128-byte id travels over the existing torch.distributed group (any
    backend); without a group (world 1: the single-rank self-test) it stays local."""

    def __init__(self, group=None, device=None):
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank(group) if world > 1 else 0
        if device is not None:
            torch.cuda.set_device(device)

        def agree(ok: bool, what: str, detail: str = "") -> None:
            """Every rank learns whether EVERY rank succeeded; if not, all of them raise (the caller falls back together)."""
            if world > 1:
                flags = [None] * world
                dist.all_gather_object(flags, (bool(ok), detail), group=group)
            else:
                flags = [(bool(ok), detail)]
            bad = [(r, d) for r, (o, d) in enumerate(flags) if not o]
            if bad:
                raise RuntimeError(f"{what} failed on rank(s) {[r for r, _ in bad]}: {bad[0][1]}")

        # (1) can every rank load RCCL at all?  A rank that cannot must not leave the others waiting inside ncclCommInitRank.
        buf = C.create_string_buffer(128)
        ok, detail = True, ""
        try:
            _lib.check(L.ta3n_comm_unique_id(buf), "ta3n_comm_unique_id")
        except Exception as ex:      # noqa: BLE001
            ok, detail = False, str(ex)
        agree(ok, "loading RCCL (ta3n_comm_unique_id)", detail)
        # (2) rank 0's id to everybody, (3) the collective init, (4) did it succeed everywhere?
        ids = [bytes(buf.raw)]
        if world > 1:
            dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        h = C.c_void_p()
        ok, detail = True, ""
        try:
            _lib.check(L.ta3n_comm_create(C.c_char_p(ids[0]), rank, world, C.byref(h)), "ta3n_comm_create")
        except Exception as ex:      # noqa: BLE001
            ok, detail = False, str(ex)
        try:
            agree(ok, "ta3n_comm_create (ncclCommInitRank)", detail)
        except Exception:
            if ok and h:
                L.ta3n_comm_destroy(h)
            raise
        self.handle, self.world, self.rank, self._L = h, world, rank, L

    shared = False      # shared_native_comm: not destroyed with the engine that asked for it

    def close(self):
        if getattr(self, "handle", None) and not self.shared:
            self._L.ta3n_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_SHARED_COMMS: Dict[tuple, "NativeComm"] = {}


def shared_native_comm(group=None, device=None) -> "NativeComm":
    """ONE RCCL communicator of the library per (process group, device) for engines that are built one after another on the same ranks
    (bench.py probes every gradient exchange with an engine of its own and then builds the chosen one: five engines, one ncclCommInitRank).
    Every rank must ask in the same order - creation is collective.  The communicator lives until the process exits; whoever uses it
    attaches / detaches its own peer transport (ta3n_comm_attach_peer)."""
    key = (id(group) if group is not None else None, str(device))
    c = _SHARED_COMMS.get(key)
    if c is None or getattr(c, "handle", None) is None:
        c = NativeComm(group, device)
        c.shared = True
        _SHARED_COMMS[key] = c
    return c


class PeerComm:
    """Two-shot all-reduce over peer-mapped buffers (include/ta3n_hip.h: ta3n_peer_*; csrc/ta3n_peer.hip): staging buffers in
    fine-grained device memory, their HIP IPC handles exchanged over the existing torch.distributed group (any backend), every
    rank maps every peer.  Creation is agreed on by all ranks: if any step fails anywhere, every rank raises and the caller keeps
    the default exchange."""

    def __init__(self, group, device, max_count: int, bf16: bool = False):
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank(group) if world > 1 else 0
        if device is not None:
            torch.cuda.set_device(device)
        self._L, self.handle, self.world, self.rank, self.bf16 = L, None, world, rank, bool(bf16)

        def agree(ok, what, detail=""):
            flags = [None] * world
            if world > 1:
                dist.all_gather_object(flags, (bool(ok), detail), group=group)
            else:
                flags = [(bool(ok), detail)]
            bad = [(r, d) for r, (o, d) in enumerate(flags) if not o]
            if bad:
                self.close()
                raise RuntimeError(f"{what} failed on rank(s) {[r for r, _ in bad]}: {bad[0][1]}")

        h = C.c_void_p()
        buf = C.create_string_buffer(128)
        ok, detail = True, ""
        try:
            _lib.check(L.ta3n_peer_create(rank, world, int(max_count), int(bool(bf16)), C.byref(h)), "ta3n_peer_create")
            self.handle = h
            _lib.check(L.ta3n_peer_handle(h, buf), "ta3n_peer_handle")
        except Exception as ex:      # noqa: BLE001
            ok, detail = False, str(ex)
        agree(ok, "peer buffer allocation / export", detail)
        handles = [None] * world
        if world > 1:
            dist.all_gather_object(handles, bytes(buf.raw), group=group)
        else:
            handles = [bytes(buf.raw)]
        ok, detail = True, ""
        try:
            _lib.check(L.ta3n_peer_connect(h, C.c_char_p(b"".join(handles))), "ta3n_peer_connect")
        except Exception as ex:      # noqa: BLE001
            ok, detail = False, str(ex)
        agree(ok, "mapping the peers' buffers (hipIpcOpenMemHandle)", detail)

    def all_reduce_sum_(self, flat: torch.Tensor) -> torch.Tensor:
        import ctypes as C
        from . import _lib
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        stream = C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
        _lib.check(self._L.ta3n_peer_all_reduce_sum(self.handle, flat.data_ptr(), flat.numel(), stream), "ta3n_peer_all_reduce_sum")
        return flat

    def status(self, device=None) -> None:
        import ctypes as C
        from . import _lib
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        if self._L.ta3n_peer_status(self.handle, stream) != 0:
            raise _lib.Ta3nError(self._L.ta3n_last_error().decode())

    def close(self):
        if getattr(self, "handle", None):
            self._L.ta3n_peer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
