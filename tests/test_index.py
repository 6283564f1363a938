"""Integer layer: TRN frame tuples and test-mode segment indices must be
BIT-EXACT with the reference (TRNmodule.py:30-41, 60, 68-71, 84-86;
dataset.py:103-116).  Three implementations are compared: the product's
combinatorial unranking (C ABI), the oracle's C enumeration and the oracle's
Python restatement (itertools, as the reference does) - and all of them against
tests/golden/index_golden.npz, which the REFERENCE ITSELF produced (tests/golden/make_index_golden.py: the tuples
RelationModuleMultiScale.forward gathers for T = 2..16, TSNDataSet._get_test_indices for 1..400 frames)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import ta3n_oracle as orc
from ta3n_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def c_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    L = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libta3n_index_oracle.so"))
    L.ta3n_oracle_relation_table.argtypes = [C.c_int] + [C.POINTER(C.c_int32)] * 3
    L.ta3n_oracle_test_indices.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    return L


def test_abi_exports_every_declared_symbol():
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "ta3n_hip.h")).read()
    import re
    declared = set(re.findall(r"\b(ta3n_[a-z_0-9]+)\s*\(", header))
    declared -= {"ta3n_set_hyper_sync"}        # mentioned in a comment only
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_known_tuples_T5():
    # the ten tuples of the headline configuration
    assert _lib.relation_table(5) == [[(0, 1, 2, 3, 4)], [(0, 1, 2, 3), (0, 1, 3, 4), (1, 2, 3, 4)],
                                      [(0, 1, 2), (0, 2, 4), (1, 2, 4)], [(0, 1), (1, 2), (2, 3)]]


@pytest.mark.parametrize("T", list(range(2, 26)))
def test_relation_tuples_bit_exact(T, c_oracle):
    mine = _lib.relation_table(T)
    ref = orc.selected_relations(T)                       # itertools enumeration, like the reference
    assert mine == [[tuple(t) for t in s] for s in ref]
    n = _lib.lib().ta3n_num_relation_tuples(T)
    tup = (C.c_int32 * (n * T))(); sl = (C.c_int32 * n)(); sid = (C.c_int32 * n)()
    assert c_oracle.ta3n_oracle_relation_table(T, tup, sl, sid) == n
    flat = [tuple(tup[r * T + j] for j in range(sl[r])) for r in range(n)]
    assert flat == [t for s in mine for t in s]
    assert sum(len(t) for t in flat) == {5: 32, 9: 114, 12: 207}.get(T, sum(len(t) for t in flat))


def test_relation_tuples_large_T_unranking_only():
    # T = 40: C(40,20) ~ 1.4e11 tuples per scale - enumeration is impossible, unranking is instant
    rel = _lib.relation_table(40)
    assert len(rel) == 39 and all(len(s) <= 3 for s in rel)
    for s in rel:
        for t in s:
            assert list(t) == sorted(set(t)) and 0 <= t[0] and t[-1] < 40
        assert s == sorted(s)
    assert rel[-1][0] == (0, 1)


@pytest.mark.parametrize("T,new_length", [(5, 1), (3, 1), (9, 1), (25, 1), (5, 5), (12, 2)])
def test_segment_indices_bit_exact(T, new_length, c_oracle):
    for num_frames in range(1, 400):
        out = (C.c_int64 * T)()
        rc = c_oracle.ta3n_oracle_test_indices(num_frames, T, new_length, out)
        if num_frames - new_length + 1 <= 0:
            assert rc == -1
            with pytest.raises(ValueError):
                _lib.segment_indices(num_frames, T, new_length)
            continue
        mine = _lib.segment_indices(num_frames, T, new_length)
        assert mine == list(out)
        assert mine == [int(v) for v in orc.segment_indices_test_mode(num_frames, T, new_length)]
        assert all(1 <= v <= num_frames for v in mine)


# ---- against fixtures produced by the reference itself (tests/golden/make_index_golden.py) ----
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "index_golden.npz"))


@pytest.mark.parametrize("T", list(range(2, 17)))
def test_relation_tuples_equal_what_the_reference_forward_gathers(T):
    """Rows of tuples_T{T}: (scale id, frames.., -1 padding) in the order TRNmodule.py:58-82 visits them, read back from the
    reference module's own forward through a probe input (feature value = frame index)."""
    rows = GOLD[f"tuples_T{T}"]
    ref = [[] for _ in range(T - 1)]
    for r in rows:
        ref[int(r[0])].append(tuple(int(v) for v in r[1:] if v >= 0))
    assert _lib.relation_table(T) == ref                                        # the product (C ABI, combinatorial unranking)
    assert [[tuple(t) for t in s] for s in orc.selected_relations(T)] == ref    # the oracle's restatement
    assert [len(t) for s in ref for t in s] == [T - i for i, s in enumerate(ref) for _ in s]


@pytest.mark.parametrize("S", [3, 5, 9, 12, 25])
@pytest.mark.parametrize("L", [1, 2, 5])
def test_segment_indices_equal_the_reference_dataset(S, L):
    """Rows of segidx_S{S}_L{L}: [num_frames, idx_0 .. idx_{S-1}] from the reference's TSNDataSet._get_test_indices
    (dataset.py:103-116), num_frames = 1..400; -1 where the reference raises (no selectable frame)."""
    rows = GOLD[f"segidx_S{S}_L{L}"]
    assert rows.shape == (400, S + 1)
    for r in rows:
        n = int(r[0])
        if r[1] < 0:
            assert n - L + 1 <= 0
            with pytest.raises(ValueError):
                _lib.segment_indices(n, S, L)
            continue
        want = [int(v) for v in r[1:]]
        assert _lib.segment_indices(n, S, L) == want, (n, S, L)
        assert [int(v) for v in orc.segment_indices_test_mode(n, S, L)] == want, (n, S, L)


def test_host_dataset_mirror_equals_the_reference_dataset():
    """ta3n_amd.dataset.TSNDataSet._get_test_indices (what main.py's loaders call) against the same fixture."""
    import types
    from ta3n_amd.dataset import TSNDataSet
    for S in (3, 5, 9, 12, 25):
        rows = GOLD[f"segidx_S{S}_L1"]
        ds = object.__new__(TSNDataSet)
        ds.num_segments, ds.new_length = S, 1
        for r in rows:
            got = ds._get_test_indices(types.SimpleNamespace(num_frames=int(r[0])))
            assert [int(v) for v in got] == [int(v) for v in r[1:]], (S, int(r[0]))
