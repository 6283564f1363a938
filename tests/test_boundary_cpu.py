"""Host-side boundary (no GPU): the module/flag surface the reference's scripts bind to."""
import os
import sys

import pytest
import torch

from oracle import ta3n_oracle as orc
from ta3n_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def test_state_dict_keys_match_reference_layout():
    from ta3n_amd.models import VideoModel
    m = VideoModel(12, 'video', 'trn-m', 'RGB', train_segments=5, val_segments=5, base_model='resnet101', fc_dim=512,
                   verbose=False)
    sd = m.state_dict()
    want = orc.param_shapes(orc.Config())
    got = {k: tuple(v.shape) for k, v in sd.items() if "running_" not in k and "num_batches" not in k}
    assert got == want
    # BatchNorm buffers of bn_trn_S/T are part of the checkpoint format (strict load in test_models.py:90)
    assert {"bn_trn_S.running_mean", "bn_trn_S.running_var", "bn_trn_S.num_batches_tracked",
            "bn_trn_T.running_mean", "bn_trn_T.running_var", "bn_trn_T.num_batches_tracked"} <= set(sd)
    assert "alpha" not in sd                                   # models.py:314: plain attribute
    assert sum(p.numel() for p in m.parameters()) == 3884836
    # 0.001-std init for the fc layers, default nn.Linear init for TRN / relation discriminators
    assert abs(m.fc_feature_shared_source.weight.std().item() - 1e-3) < 1e-4
    assert m.fc_feature_shared_source.bias.abs().max().item() == 0
    assert m.TRN.fc_fusion_scales[0][1].weight.std().item() > 5e-3
    # load_state_dict round trip with the DataParallel 'module.' prefix stripped as test_models.py:89 does
    ck = {"module." + k: v.clone() for k, v in sd.items()}
    m2 = VideoModel(12, 'video', 'trn-m', 'RGB', train_segments=5, val_segments=5, fc_dim=512, verbose=False)
    m2.load_state_dict({'.'.join(k.split('.')[1:]): v for k, v in ck.items()}, strict=True)


def test_constructor_rejects_what_is_not_the_hot_path():
    from ta3n_amd.models import VideoModel
    with pytest.raises(NotImplementedError):
        VideoModel(12, 'video', 'avgpool', 'RGB', verbose=False)
    with pytest.raises(NotImplementedError):
        VideoModel(12, 'frame', 'trn-m', 'RGB', verbose=False)
    with pytest.raises(NotImplementedError):
        VideoModel(12, 'video', 'trn-m', 'RGB', use_bn='BN', verbose=False)
    assert "bn_shared_S.weight" in VideoModel(12, 'video', 'avgpool', 'RGB', use_attn='none', use_bn='AdaBN', ens_DA='MCD', fc_dim=64,
                                              base_model='resnet18', verbose=False).state_dict()      # the TemPooling + X rows
    with pytest.raises(NotImplementedError):
        VideoModel(12, 'video', 'rnn', 'RGB', verbose=False)
    # built options of the DA tables construct and carry the reference's extra state_dict entries
    m = VideoModel(12, 'video', 'trn-m', 'RGB', use_bn='AutoDIAL', ens_DA='MCD', fc_dim=64, base_model='resnet18', verbose=False)
    keys = set(m.state_dict())
    assert {"bn_shared_S.running_var", "bn_shared_T.weight", "bn_source_video_2_T.bias", "alpha",
            "fc_classifier_video_source_2.weight"} <= keys
    with pytest.raises(ValueError):
        VideoModel(12, 'video', 'trn-m', 'RGB', add_fc=0, verbose=False)       # models.py:137-138


def test_forward_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ta3n_amd.models import VideoModel
    m = VideoModel(12, 'video', 'trn-m', 'RGB', train_segments=5, val_segments=5, fc_dim=64, base_model='resnet18',
                   verbose=False)
    with pytest.raises(_lib.Ta3nError):
        m(torch.zeros(2, 5, 512), torch.zeros(2, 5, 512), [0, 0, 0], 0, True, False)
    from ta3n_amd.engine import TrainEngine
    with pytest.raises(_lib.Ta3nError):
        TrainEngine(2, 2, 5, 512, 64, 12)


def test_compat_import_names():
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    try:
        for name in ("models", "TRNmodule", "loss", "opts", "dataset"):
            sys.modules.pop(name, None)
        import models, TRNmodule, loss, opts, dataset           # noqa: E401  (the names main.py:12-16 imports)
        from utils.utils import randSelectBatch                 # noqa: F401
        assert models.VideoModel.__module__ == "ta3n_amd.models"
        assert hasattr(TRNmodule, "RelationModuleMultiScale") and hasattr(loss, "attentive_entropy")
        assert opts.parser is not None and hasattr(dataset, "TSNDataSet")
    finally:
        sys.path.remove(os.path.join(ROOT, "compat"))
        for name in ("models", "TRNmodule", "loss", "opts", "dataset", "utils", "utils.utils"):
            sys.modules.pop(name, None)


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout only exists in the build container")
def test_flag_surface_identical_to_reference_parser():
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_opts", os.path.join(REF, "opts.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from ta3n_amd import opts

    def table(p):
        return {a.dest: (tuple(sorted(a.option_strings)), a.default, tuple(a.choices) if a.choices else None, a.nargs,
                         getattr(a, "type", None), type(a).__name__)
                for a in p._actions if not isinstance(a, argparse._HelpAction)}
    assert table(ref.parser) == table(opts.parser)


def test_script_command_line_parses():
    """The argument vector script_train_val.sh:144-155 builds for the UCF->HMDB_full TA3N run."""
    from ta3n_amd import opts
    argv = ("c.txt RGB s.txt t.txt v.txt --exp_path x --arch resnet101 --pretrained none --baseline_type video "
            "--frame_aggregation trn-m --num_segments 5 --val_segments 5 --add_fc 1 --fc_dim 512 --dropout_i 0.5 "
            "--dropout_v 0.5 --use_target uSv --share_params Y --dis_DA none --alpha 0 --place_dis N Y N --adv_DA RevGrad "
            "--beta 0.75 0.75 0.5 --place_adv Y Y Y --use_bn none --add_loss_DA attentive_entropy --gamma 0.003 "
            "--ens_DA none --mu 0 --use_attn TransAttn --n_attn 1 --use_attn_frame none --gd 20 --lr 3e-2 --lr_decay 10 "
            "--lr_adaptive dann --lr_steps 10 20 --epochs 30 --optimizer SGD --n_rnn 1 --rnn_cell LSTM --n_directions 1 "
            "--n_ts 5 -b 128 74 128 -j 4 -ef 1 -pf 50 -sf 50 --copy_list N N --save_model").split()
    a = opts.parser.parse_args(argv)
    assert a.beta == [0.75, 0.75, 0.5] and a.batch_size == [128, 74, 128] and a.clip_gradient == 20
    assert a.frame_aggregation == "trn-m" and a.lr_adaptive == "dann" and a.no_partialbn is True
    from ta3n_amd.engine import flags_from_options
    f = flags_from_options(a.place_adv, a.add_loss_DA, a.use_attn, a.adv_DA, a.use_target)
    assert f == (_lib.FLAG_ADV_RELATION | _lib.FLAG_ADV_VIDEO | _lib.FLAG_ADV_FRAME | _lib.FLAG_ATTN_ENTROPY |
                 _lib.FLAG_TRANS_ATTN)
    assert flags_from_options(use_target="none") == _lib.FLAG_TRANS_ATTN


def test_dataset_items_match_reference_format(tmp_path):
    """One torch-saved 1-D tensor per frame, list line '<dir> <num_frames> <label>' (README.md:90-95)."""
    from ta3n_amd.dataset import TSNDataSet
    lines = []
    for v, (n, lab) in enumerate([(11, 3), (4, 7), (30, 0)]):
        d = tmp_path / f"vid{v}"
        d.mkdir()
        for f in range(1, n + 1):
            torch.save(torch.full((8,), float(100 * v + f)), d / f"img_{f:05d}.t7")
        lines.append(f"{d}/ {n} {lab}\n")
    lst = tmp_path / "list.txt"
    lst.write_text("".join(lines))
    ds = TSNDataSet("", str(lst), num_dataload=5, num_segments=5, new_length=1, modality="RGB", random_shift=False,
                    test_mode=True)
    assert len(ds) == 5                                           # list repeated to num_dataload (dataset.py:69-74)
    x, y = ds[0]
    assert x.shape == (5, 8) and y == 3
    assert [int(v) for v in x[:, 0]] == [int(i) for i in orc.segment_indices_test_mode(11, 5, 1)]
    x, y = ds[1]                                                  # short video: last frame repeated (dataset.py:110-114)
    assert [int(v) - 100 for v in x[:, 0]] == [1, 2, 3, 4, 4] and y == 7
    assert ds[3][1] == 3                                          # wrapped around


def test_workspace_pool_release_is_tied_to_the_checkout():
    """the finalizer of an OLD autograd node fires after its entry was released and checked out again by the
    next iteration's first forward; it must not mark that in-use buffer free (the MCD step's second forward would then take it
    and overwrite the first forward's activations before backward)."""
    import gc
    import torch
    from ta3n_amd.models import VideoModel

    m = VideoModel(12, 'video', 'trn-m', 'RGB', train_segments=5, val_segments=5, base_model='resnet101', fc_dim=64,
                   verbose=False, ens_DA='MCD')
    plan = object()
    m._ws_template = lambda p: torch.zeros(8)

    class Ctx:        # stands in for the autograd context (weakref-able)
        pass

    it1 = Ctx()
    a = m._ws_checkout(plan, it1, True)
    m._ws_release(it1)                     # backward of iteration 1 ran
    it2_first = Ctx()
    b = m._ws_checkout(plan, it2_first, True)
    assert b.data_ptr() == a.data_ptr()    # one loop, one buffer
    del it1                                # `loss` rebound: iteration 1's graph dies only now
    gc.collect()
    it2_second = Ctx()                     # ens_DA MCD: forward(..., reverse=True) before iteration 2's backward
    c = m._ws_checkout(plan, it2_second, True)
    assert c.data_ptr() != b.data_ptr(), "the stale finalizer freed a workspace that is in use"
    m._ws_release(it2_second)
    m._ws_release(it2_first)
    d = m._ws_checkout(plan, Ctx(), False)
    assert d.data_ptr() in (b.data_ptr(), c.data_ptr())
    dropped = Ctx()
    e = m._ws_checkout(plan, dropped, True)
    del dropped                            # a graph dropped without backward gives its buffer back
    gc.collect()
    f = m._ws_checkout(plan, Ctx(), True)
    assert f.data_ptr() == e.data_ptr()


def test_da_entry_points_validate_their_arguments_before_touching_a_device():
    """ta3n_discrepancy / ta3n_mcd_source_loss / ta3n_mcd_second_loss (round 6: the DA options' loss assembly from the library): argument
    errors come back as TA3N_ERR_INVALID with a message - no HIP call is made for them, so this runs without a GPU."""
    import ctypes as C
    from ta3n_amd import _lib
    L = _lib.lib()
    buf = (C.c_float * 4096)()
    p = C.cast(buf, C.c_void_p)
    n = L.ta3n_discrepancy_scratch_floats(128, 74, 12, 256)
    assert n >= 2 * 148 * (12 + 256) + 4 * 148 * 148                       # two stacks + gradients, two kernel matrices + derivatives
    assert L.ta3n_discrepancy_scratch_floats(0, 74, 12, 256) > 0          # (an empty domain still gets a non-empty buffer)
    args = dict(ws=p, o_y=0, c=12, o_v=64, f=256, o_gy=128, o_gv=192, bs=8, bt=6, ns=8, nt=6, kind=1, pl=1, pf=1, alpha=C.c_float(0.5), scr=p, n=4096, loss=p, st=None)

    def call(**kw):
        a = dict(args, **kw)
        return L.ta3n_discrepancy(a["ws"], a["o_y"], a["c"], a["o_v"], a["f"], a["o_gy"], a["o_gv"], a["bs"], a["bt"], a["ns"], a["nt"], a["kind"],
                                  a["pl"], a["pf"], a["alpha"], a["scr"], a["n"], a["loss"], a["st"])
    for bad in (dict(ws=None), dict(scr=None), dict(loss=None), dict(kind=0), dict(kind=3), dict(ns=9), dict(nt=-1), dict(c=0)):
        assert call(**bad) == -1, bad
        assert L.ta3n_last_error()
    plan = _lib.Plan(4, 4, 5, 512, 64, 12, 0x1F)                            # no TA3N_FLAG_MCD
    assert L.ta3n_mcd_source_loss(plan.handle, p, p, p, None) == -1 and b"TA3N_FLAG_MCD" in L.ta3n_last_error()
    assert L.ta3n_mcd_second_loss(plan.handle, p, p, 4, p, p, None) == -1      # one workspace for both passes
    assert L.ta3n_mcd_source_loss(None, p, p, p, None) == -1
