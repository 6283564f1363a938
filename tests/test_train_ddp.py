"""train_ddp.py's host logic: option validation (nothing behaviour-changing is silently ignored), the reference's
list-repeat rule, and the reference's checkpoint format (main.py:266-274; consumers: main.py:94-106, test_models.py:85-90)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import train_ddp  # noqa: E402
from ta3n_amd.opts import parser  # noqa: E402

BASE = ["classInd.txt", "RGB", "s.txt", "t.txt", "v.txt", "--baseline_type", "video", "--frame_aggregation", "trn-m",
        "--use_target", "uSv", "--adv_DA", "RevGrad", "--use_attn", "TransAttn", "--add_loss_DA", "attentive_entropy",
        "--lr_adaptive", "dann", "--fc_dim", "512"]


def test_headline_command_line_is_accepted():
    train_ddp.validate_options(parser.parse_args(BASE))
    train_ddp.validate_options(parser.parse_args(BASE + ["--dis_DA", "DAN"]))      # round 3: discrepancy losses on the engine path (round 4: on any number of ranks)
    train_ddp.validate_options(parser.parse_args(BASE + ["--dis_DA", "JAN"]))
    train_ddp.validate_options(parser.parse_args(BASE + ["--use_bn", "AdaBN"]))     # ... and the domain BatchNorm (per-replica batch statistics, as under nn.DataParallel)
    train_ddp.validate_options(parser.parse_args(BASE + ["--use_bn", "AutoDIAL"]))
    train_ddp.validate_options(parser.parse_args(BASE + ["--ens_DA", "MCD", "--mu", "0.5"]))      # ... and MCD's second classifier / reversed pass
    avg = ["c", "RGB", "s", "t", "v", "--baseline_type", "video", "--frame_aggregation", "avgpool", "--use_attn", "none", "--add_loss_DA", "none"]
    for extra in (["--dis_DA", "DAN"], ["--dis_DA", "JAN"], ["--ens_DA", "MCD", "--mu", "0.5"], ["--use_bn", "AdaBN"]):      # the "TemPooling + X" rows
        train_ddp.validate_options(parser.parse_args(avg + extra))
    train_ddp.validate_options(parser.parse_args(["c", "RGB", "s", "t", "v", "--baseline_type", "video", "--frame_aggregation", "avgpool"]))


@pytest.mark.parametrize("extra", [["--optimizer", "Adam"], ["--dis_DA", "CORAL"],
                                   ["--add_loss_DA", "target_entropy"],
                                   ["--use_target", "Sv"], ["--weighted_class_loss", "Y"], ["--weighted_class_loss_DA", "Y"],
                                   ["--pred_normalize", "Y"], ["--pretrain_source"], ["--lr_adaptive", "loss"], ["--ens_DA", "MCD", "--use_bn", "AdaBN"], ["--mu", "0.5"],
                                   ["--share_params", "N"], ["--frame_aggregation", "rnn"],
                                   ["--baseline_type", "frame"], ["--use_attn", "general"], ["--place_adv", "N", "Y", "Y"]])
def test_unimplemented_option_values_are_rejected_not_ignored(extra):
    with pytest.raises(SystemExit) as e:
        train_ddp.validate_options(parser.parse_args(BASE + extra))
    assert "unsupported option" in str(e.value)


@pytest.mark.parametrize("ns,nt,bs,copy", [(1438, 840, [128, 74, 128], ["N", "Y"]), (840, 1438, [74, 128, 64], ["Y", "Y"]),
                                           (100, 37, [32, 28, 64], ["N", "N"]), (1438, 840, [128, 74, 128], ["Y", "N"])])
def test_list_repeat_rule_matches_the_reference_formula(ns, nt, bs, copy):
    # main.py:145-153 restated literally
    num_iter_source, num_iter_target = ns / bs[0], nt / bs[1]
    num_max_iter = max(num_iter_source, num_iter_target)
    want = (round(num_max_iter * bs[0]) if copy[0] == 'Y' else ns, round(num_max_iter * bs[1]) if copy[1] == 'Y' else nt)
    assert train_ddp.train_list_sizes(ns, nt, bs, copy) == want
    if copy == ["N", "Y"] and ns == 1438:       # the headline run: 11.23 source vs 11.35 target iterations - 12 batches each, nothing repeated
        assert want == (1438, 840) and train_ddp.n_batches(want[0], bs[0]) == 12 and train_ddp.n_batches(want[1], bs[1]) == 12


def test_optimizer_entry_loads_into_torch_sgd():
    from ta3n_amd.checkpoint import optimizer_state_dict
    lin = torch.nn.Linear(4, 3)
    names = ["weight", "bias"]
    sd = optimizer_state_dict(names, {"weight": torch.ones(3, 4)}, lr=0.01, mu=0.9, weight_decay=1e-4)
    opt = torch.optim.SGD(lin.parameters(), 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["lr"] == 0.01 and opt.param_groups[0]["nesterov"]
    assert torch.equal(opt.state[lin.weight]["momentum_buffer"], torch.ones(3, 4)) and lin.bias not in opt.state


@pytest.mark.gpu
def test_engine_checkpoint_has_the_reference_format_and_resumes(tmp_path):
    from ta3n_amd import checkpoint as ckpt
    from ta3n_amd.engine import TrainEngine
    from ta3n_amd.models import VideoModel
    from ta3n_amd.synthetic import synth_batch
    T, C = 5, 12
    model = VideoModel(C, "video", "trn-m", "RGB", train_segments=T, val_segments=T, base_model="resnet18", fc_dim=64, verbose=False)
    eng = TrainEngine(6, 4, T, 512, 64, C, dropout_i=0.0, dropout_v=0.0)
    eng.load_state(model.state_dict())
    xs, xt, ys, yt = synth_batch(C, T, 512, 6, 4, seed=3)
    eng.set_batch(xs.cuda(), xt.cuda(), ys.cuda())
    for _ in range(2):
        eng.train_step([0.75, 0.75, 0.5], 0.003, 1e-2)
    path = ckpt.save_checkpoint(ckpt.engine_checkpoint(eng, model, 3, "resnet18", 7e-3, 55.0, 50.0), True, str(tmp_path / "RGB"))
    assert os.path.exists(str(tmp_path / "RGB" / "model_best.pth.tar"))
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "arch", "state_dict", "optimizer", "best_prec1", "prec1"}          # main.py:266-274
    assert ck["epoch"] == 3 and ck["prec1"] == 50.0 and ck["best_prec1"] == 55.0
    # test_models.py:85-90: strip 'module.', strict load into a fresh model
    base_dict = {'.'.join(k.split('.')[1:]): v for k, v in list(ck['state_dict'].items())}
    fresh = VideoModel(C, "video", "trn-m", "RGB", train_segments=T, val_segments=T, base_model="resnet18", fc_dim=64, verbose=False)
    fresh.load_state_dict(base_dict)
    # main.py:104: optimizer.load_state_dict(checkpoint['optimizer']) with an optimizer built from model.parameters()
    opt = torch.optim.SGD(fresh.parameters(), 0.1, momentum=0.9, weight_decay=1e-4, nesterov=True)
    opt.load_state_dict(ck["optimizer"])
    n_buf = sum(1 for p in fresh.parameters() if p in opt.state)
    assert n_buf == len(eng.live_names()) and opt.param_groups[0]["lr"] == 7e-3
    # resume into a second engine: parameters, momentum and epoch / best score come back
    eng2 = TrainEngine(6, 4, T, 512, 64, C, dropout_i=0.0, dropout_v=0.0)
    st = ckpt.load_into_engine(eng2, fresh, ck, resume_hp=True)
    assert st == {"start_epoch": 4, "best_prec1": 55.0, "lr": 7e-3}
    torch.cuda.synchronize()
    assert torch.equal(eng2.P, eng.P) and torch.equal(eng2.M, eng.M) and eng.M.abs().max().item() > 0


def test_da_options_are_accepted_on_more_than_one_rank(monkeypatch):
    """Round 4: dis_DA / ens_DA / use_bn under WORLD_SIZE > 1 follow nn.DataParallel's semantics (global-batch discrepancy losses,
    per-replica BatchNorm statistics) instead of being refused."""
    monkeypatch.setenv("WORLD_SIZE", "8")
    for extra in (["--dis_DA", "DAN"], ["--dis_DA", "JAN"], ["--ens_DA", "MCD", "--mu", "0.5"], ["--use_bn", "AdaBN"], ["--use_bn", "AutoDIAL"]):
        train_ddp.validate_options(parser.parse_args(BASE + extra))
