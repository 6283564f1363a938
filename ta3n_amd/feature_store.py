"""Packed feature store: the dataset resident in HBM, batches assembled on the device.

The reference keeps one torch-saved 1-D tensor per frame (`<video>/img_00001.t7` ..., written by
dataset_preparation/video2feature.py:206-217) and `TSNDataSet.__getitem__` (dataset.py:118-144) does one
`torch.load` per selected frame - 1 010 tiny file reads per 128+74 step, the reference's real bottleneck
(its own "Data" column, main.py:592).  Here a dataset is packed ONCE into

    <prefix>.f32      raw little-endian fp32 [total_frames, feature_dim], videos back to back, frames in order
                      (or <prefix>.bf16: the same rows rounded to bf16, nearest even - half the bytes on disk and in HBM, and
                      what the bf16 arithmetic's twin of the input holds anyway)
    <prefix>.idx.npy  int64 [n_videos, 3] = (first_row, num_frames, label)

(UCF-HMDB_full: ~3.2 k videos x ~100 frames x 8 KiB = 2.5 GB - a sliver of the 288 GB of HBM), loaded to
the GPU once, and a batch is one kernel: `ta3n_gather_segments` computes the reference's test-mode segment
indices (dataset.py:103-116, float64, bit-exact) and copies the selected rows straight into the train
step's input buffer.  List parsing and the list repeat/truncate rule (dataset.py:69-74) stay on the host.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib


def pack(list_file: str, prefix: str, image_tmpl: str = "img_{:05d}.t7", root_path: str = "", dtype: str = "f32") -> Tuple[int, int]:
    """Pack the videos of a reference list file (`<dir> <num_frames> <label>` per line, README.md:90-95) into
    `<prefix>.f32` (or `<prefix>.bf16` with dtype="bf16") + `<prefix>.idx.npy`.  Returns (n_videos, feature_dim)."""
    if dtype not in ("f32", "bf16"):
        raise ValueError("dtype must be 'f32' or 'bf16'")
    rows = [line.strip().split(" ") for line in open(list_file) if line.strip()]
    idx = np.zeros((len(rows), 3), dtype=np.int64)
    first, dim = 0, None
    with open(prefix + "." + dtype, "wb") as out:
        for i, (path, n, label) in enumerate(rows):
            n = int(n)
            if n < 1:
                raise ValueError(f"{path}: a video needs at least one frame")
            idx[i] = (first, n, int(label))
            for f in range(1, n + 1):
                t = torch.load(os.path.join(root_path, path, image_tmpl.format(f))).reshape(-1).to(torch.float32)
                if dim is None:
                    dim = t.numel()
                if t.numel() != dim:
                    raise ValueError(f"{path} frame {f}: feature dim {t.numel()} != {dim}")
                out.write(t.numpy().tobytes() if dtype == "f32" else t.to(torch.bfloat16).view(torch.int16).numpy().tobytes())
            first += n
    np.save(prefix + ".idx.npy", idx)
    return len(rows), int(dim or 0)


class FeatureStore:
    """A packed dataset on one GPU.  `gather` is TSNDataSet.__getitem__ + DataLoader collation for a batch of
    video ids, on the device."""

    def __init__(self, prefix: str, feature_dim: int, device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise _lib.Ta3nError("FeatureStore needs a HIP device (the host path is ta3n_amd.dataset.TSNDataSet)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        idx = np.load(prefix + ".idx.npy")
        self.bf16 = not os.path.exists(prefix + ".f32")
        blob = np.memmap(prefix + (".bf16" if self.bf16 else ".f32"), dtype=np.int16 if self.bf16 else np.float32, mode="r")
        total = int(idx[:, 1].sum())
        if blob.size != total * feature_dim:
            raise ValueError(f"{prefix} holds {blob.size} elements, index says {total} x {feature_dim}")
        self.feature_dim, self.n_videos = feature_dim, idx.shape[0]
        self.store = torch.from_numpy(np.array(blob)).to(self.device).view(total, feature_dim)   # one host copy, then HBM (int16 bits for bf16)
        self.first_row = torch.from_numpy(idx[:, 0].copy()).to(self.device)
        self.num_frames = torch.from_numpy(idx[:, 1].astype(np.int32)).to(self.device)
        self.labels = torch.from_numpy(idx[:, 2].astype(np.int32)).to(self.device)
        self._L = _lib.lib()

    @classmethod
    def from_tensors(cls, store: torch.Tensor, num_frames: torch.Tensor, labels: torch.Tensor) -> "FeatureStore":
        """A store from tensors already on the device (synthetic datasets of bench.py / tests): `store` [total_frames, D] fp32 (or
        int16 bit patterns of bf16 rows), videos back to back; `num_frames`, `labels` one entry per video."""
        self = cls.__new__(cls)
        self.device = store.device
        self.bf16 = store.dtype == torch.int16
        assert store.is_cuda and store.dim() == 2 and store.is_contiguous() and store.dtype in (torch.float32, torch.int16)
        nf = num_frames.to(device=self.device, dtype=torch.int64)
        assert int(nf.sum().item()) == store.shape[0] and int(nf.min().item()) >= 1
        self.feature_dim, self.n_videos = int(store.shape[1]), int(nf.numel())
        self.store = store
        self.first_row = (torch.cumsum(nf, 0) - nf).contiguous()
        self.num_frames = nf.to(torch.int32).contiguous()
        self.labels = labels.to(device=self.device, dtype=torch.int32).contiguous()
        self._L = _lib.lib()
        return self

    def __len__(self) -> int:
        return self.n_videos

    def gather(self, video_ids: torch.Tensor, num_segments: int, out: Optional[torch.Tensor] = None,
               labels_out: Optional[torch.Tensor] = None, segment_ids_out: Optional[torch.Tensor] = None):
        """video_ids: int32 tensor on this device -> (features [n, T, D] written into `out`, int32 labels [n])."""
        ids = video_ids.to(device=self.device, dtype=torch.int32).contiguous()
        n, T, D = ids.numel(), num_segments, self.feature_dim
        if self.bf16:        # (validation / host-side use: widen the selected rows with torch; the train step uses gather_into)
            seg = torch.tensor([[s - 1 for s in _lib.segment_indices(int(nf), T)] for nf in self.num_frames[ids.long()].tolist()],
                               device=self.device, dtype=torch.long)
            rows = self.first_row[ids.long()].unsqueeze(1) + seg
            feats = self.store[rows.reshape(-1)].view(torch.bfloat16).to(torch.float32).view(n, T, D)
            lab = self.labels[ids.long()]
            if out is not None:
                out.view(-1)[: n * T * D].copy_(feats.reshape(-1))
            if labels_out is not None:
                labels_out[:n].copy_(lab)
            return feats, lab
        if out is None:
            out = torch.empty(n * T, D, dtype=torch.float32, device=self.device)
        if labels_out is None:
            labels_out = torch.empty(n, dtype=torch.int32, device=self.device)
        assert out.is_contiguous() and out.numel() >= n * T * D and out.dtype == torch.float32
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(self._L.ta3n_gather_segments(p(self.store), p(self.first_row), p(self.num_frames), p(self.labels), p(ids), n, T, D,
                                                p(out), p(labels_out), p(segment_ids_out), stream), "ta3n_gather_segments")
        return out.view(-1)[: n * T * D].view(n, T, D), labels_out

    def gather_into(self, engine, video_ids: torch.Tensor, first_video: int, labels_out: Optional[torch.Tensor] = None) -> None:
        """Assemble videos [first_video, first_video + n) of a TrainEngine's input batch on the device: the fp32 rows of
        engine.X and - when the engine reads bf16 twins - the input twin, in one pass (ta3n_gather_segments_into)."""
        ids = video_ids.to(device=self.device, dtype=torch.int32).contiguous()
        assert engine.D == self.feature_dim
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        if self.bf16:      # the rows go into the input's bf16 twin as they are; fp32 rows only for an engine that does not read twins
            _lib.check(self._L.ta3n_gather_segments_bf16_into(engine.plan.handle, p(self.store), p(self.first_row), p(self.num_frames),
                                                              p(self.labels), p(ids), ids.numel(), int(first_video),
                                                              None if engine.bf16_store else p(engine.X), p(engine.ws), p(labels_out), stream),
                       "ta3n_gather_segments_bf16_into")
            return
        _lib.check(self._L.ta3n_gather_segments_into(engine.plan.handle, p(self.store), p(self.first_row), p(self.num_frames),
                                                     p(self.labels), p(ids), ids.numel(), int(first_video), p(engine.X), p(engine.ws),
                                                     p(labels_out), stream), "ta3n_gather_segments_into")
