"""Deterministic synthetic weights / features for parity tests and benchmarks.

The reference trains on pre-extracted ResNet-101 pool5 features (dataset.py:53-60,
README.md:62-85), which are non-negative, so synthetic features are half-normal
(SURVEY.md 8d).  Weights come from a numpy Generator so that the golden
generator (tests/golden/make_golden.py, which runs the reference), the CPU
oracle and the HIP path all see bit-identical inputs without shipping tensors.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch


def synth_state(shapes: Dict[str, Tuple[int, ...]], seed: int = 7, scale: str = "trained",
                dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """scale='trained': weight ~ N(0, 1/fan_in), bias ~ N(0, 0.01) so logits are O(1)
    and the 1e-3 parity bound is meaningful; scale='init': the reference's
    0.001-std normal init with zero bias (models.py:128, 142-283)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in sorted(shapes.items()):
        if len(shp) == 0 or "running_" in name or "num_batches" in name:
            continue
        if name == "alpha":           # use_bn='AutoDIAL' (models.py:314-316): initialised to one and never trained (it is read with
            arr = np.ones(shp)        # .item(), so it has no gradient); no draw, so the other tensors do not depend on the option
        elif name.startswith("bn_"):
            arr = np.ones(shp) if name.endswith("weight") else np.zeros(shp)
        elif name.endswith(".weight"):
            std = 1.0 / math.sqrt(shp[1]) if scale == "trained" else 0.001
            arr = rng.standard_normal(shp) * std
        else:
            arr = rng.standard_normal(shp) * (0.01 if scale == "trained" else 0.0)
        out[name] = torch.tensor(arr, dtype=dtype)
    return out


def synth_batch(num_class: int, num_segments: int, feature_dim: int, batch_source: int,
                batch_target: int, seed: int = 1234, dtype=torch.float32):
    """Half-normal features [B,T,D] for source and target plus integer labels."""
    rng = np.random.default_rng(seed)
    xs = np.abs(rng.standard_normal((batch_source, num_segments, feature_dim)))
    xt = np.abs(rng.standard_normal((batch_target, num_segments, feature_dim)))
    ys = rng.integers(0, num_class, size=(batch_source,))
    yt = rng.integers(0, num_class, size=(batch_target,))
    return (torch.tensor(xs, dtype=dtype), torch.tensor(xt, dtype=dtype),
            torch.tensor(ys, dtype=torch.long), torch.tensor(yt, dtype=torch.long))


def task_batch(num_class: int, num_segments: int, feature_dim: int, batch_source: int, batch_target: int, step: int,
               task_seed: int = 0, dtype=torch.float32):
    """A LEARNABLE synthetic domain-adaptation task (tests/test_gpu_training_equivalence.py): every class has a mean pattern over the
    feature channels (fixed by task_seed); a video's frames are |mean * (0.5 + t / T) + noise| (non-negative, like ResNet pool5
    features, with a mild temporal trend for the relation module to see); the target domain is the same classes seen through a fixed
    per-channel gain and offset.  `step` selects the batch.  Returns (xs [Bs,T,D], xt [Bt,T,D], ys, yt)."""
    task = np.random.default_rng(1_000_003 * (task_seed + 1))
    means = task.standard_normal((num_class, feature_dim)) * 0.6
    gain = 1.0 + 0.3 * task.standard_normal(feature_dim)
    offset = 0.2 * task.standard_normal(feature_dim)
    rng = np.random.default_rng([task_seed, step])
    trend = (0.5 + np.arange(num_segments) / num_segments)[None, :, None]

    def draw(n, target):
        y = rng.integers(0, num_class, size=(n,))
        x = means[y][:, None, :] * trend + rng.standard_normal((n, num_segments, feature_dim))
        if target:
            x = x * gain + offset
        return np.abs(x), y
    xs, ys = draw(batch_source, False)
    xt, yt = draw(batch_target, True)
    return (torch.tensor(xs, dtype=dtype), torch.tensor(xt, dtype=dtype), torch.tensor(ys, dtype=torch.long), torch.tensor(yt, dtype=torch.long))
