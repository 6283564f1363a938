"""Integer fixtures produced by the REFERENCE ITSELF (SURVEY.md 8 rows a1 and the "bit-exact segment indices" of north_star).

Run in the build container only (it imports /root/reference, unmodified, through tests/golden/ref_shim.py):

    python tests/golden/make_index_golden.py        # writes tests/golden/index_golden.npz

What is recorded - nothing here is restated, every number comes out of the reference's own objects:

* ``tuples_T{T}`` for T = 2..16: the frame tuples ``TRNmodule.RelationModuleMultiScale(F, 256, T).forward`` actually GATHERS
  (TRNmodule.py:58-82), read back from a probe input whose feature value is the frame index (``x[b, t, :] = t``) through a
  forward-pre-hook on every ``fc_fusion_scales[i]``: row r = (scale id, frames..., -1 padding), in the order the forward visits them.
  This pins ``relations_scales`` (itertools order), ``subsample_scales`` and the ``int(ceil(i * n / k))`` pick of :71 together.
* ``segidx_S{S}_L{L}`` for num_segments S in {3, 5, 9, 12, 25} x new_length L in {1, 2, 5}: ``TSNDataSet._get_test_indices``
  (dataset.py:103-116) for num_frames = 1..400, one row per num_frames ([num_frames, idx_0..idx_{S-1}]); a row of -1 where the
  reference raises (num_frames - new_length + 1 <= 0: id_select is empty).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
import TRNmodule as ref_trn  # noqa: E402  (the reference's, /root/reference/TRNmodule.py)
import dataset as ref_dataset  # noqa: E402

assert os.path.realpath(ref_trn.__file__).startswith("/root/reference/"), ref_trn.__file__
assert os.path.realpath(ref_dataset.__file__).startswith("/root/reference/"), ref_dataset.__file__


def gathered_tuples(T, F=4):
    mod = ref_trn.RelationModuleMultiScale(F, 256, T)
    seen = []

    def hook_for(scale_id, scale):
        def hook(_m, args):
            flat = args[0]                       # [B, scale * F], value = frame index of the gathered row
            frames = flat[0].view(scale, F)[:, 0].round().to(torch.int64).tolist()
            assert all(torch.equal(flat[0], flat[b]) for b in range(flat.shape[0]))
            seen.append((scale_id, frames))
        return hook

    for i, fc in enumerate(mod.fc_fusion_scales):
        fc.register_forward_pre_hook(hook_for(i, mod.scales[i]))
    x = torch.arange(T, dtype=torch.float32).view(1, T, 1).expand(2, T, F).contiguous()
    with torch.no_grad():
        out = mod(x)
    assert out.shape == (2, T - 1, 256)
    rows = np.full((len(seen), T + 1), -1, dtype=np.int32)
    for r, (sid, frames) in enumerate(seen):
        rows[r, 0] = sid
        rows[r, 1:1 + len(frames)] = frames
    return rows


def test_indices(S, L, max_frames=400):
    ds = object.__new__(ref_dataset.TSNDataSet)          # the method reads two attributes only; __init__ wants list files
    ds.num_segments, ds.new_length = S, L
    rows = np.full((max_frames, S + 1), -1, dtype=np.int64)
    for n in range(1, max_frames + 1):
        rows[n - 1, 0] = n
        rec = types.SimpleNamespace(num_frames=n)
        try:
            idx = ref_dataset.TSNDataSet._get_test_indices(ds, rec)
        except (IndexError, ValueError):
            continue                                      # the reference raises: row stays -1
        rows[n - 1, 1:] = np.asarray(idx, dtype=np.int64)
    return rows


def main():
    store = {}
    for T in range(2, 17):
        store[f"tuples_T{T}"] = gathered_tuples(T)
    for S in (3, 5, 9, 12, 25):
        for L in (1, 2, 5):
            store[f"segidx_S{S}_L{L}"] = test_indices(S, L)
    path = os.path.join(HERE, "index_golden.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path), "bytes;", len(store), "arrays")


if __name__ == "__main__":
    main()
