"""Command-line surface of the reference (opts.py:2-118): 5 positionals and the same
flags, defaults and choices, so script_train_val.sh's command lines parse unchanged.
Built from a table instead of ~60 add_argument calls.  Flags the HIP path does not
implement still parse (the model constructor rejects unsupported configurations)."""
import argparse

parser = argparse.ArgumentParser(description="TA3N temporal-adversarial training (MI355X-native train step)")
for _name in ("class_file",):
    parser.add_argument(_name, type=str, default="classInd.txt")
parser.add_argument("modality", type=str, choices=["RGB", "Flow", "RGBDiff", "RGBDiff2", "RGBDiffplus"])
for _name in ("train_source_list", "train_target_list", "val_list"):
    parser.add_argument(_name, type=str)

YN = ["Y", "N"]
_FLAGS = [
    # (names, kwargs)                                                                   reference line
    (("--arch",), dict(type=str, default="resnet101")),                                 # opts.py:10
    (("--pretrained",), dict(type=str, default="none")),
    (("--num_segments",), dict(type=int, default=5)),
    (("--val_segments",), dict(type=int, default=-1)),
    (("--add_fc",), dict(type=int, default=1, metavar="M")),
    (("--fc_dim",), dict(type=int, default=1024)),
    (("--baseline_type",), dict(type=str, default="frame", choices=["frame", "video", "tsn"])),
    (("--frame_aggregation",), dict(type=str, default="avgpool",
                                    choices=["avgpool", "rnn", "temconv", "trn", "trn-m", "none"])),
    (("--optimizer",), dict(type=str, default="SGD", choices=["SGD", "Adam"])),
    (("--use_opencv",), dict(default=False, action="store_true")),
    (("--dropout_i", "--doi"), dict(type=float, default=0.8, metavar="DOI")),          # opts.py:24
    (("--dropout_v", "--dov"), dict(type=float, default=0.8, metavar="DOV")),
    (("--loss_type",), dict(type=str, default="nll", choices=["nll"])),
    (("--weighted_class_loss",), dict(type=str, default="N", choices=YN)),
    (("--n_rnn",), dict(type=int, default=1, metavar="M")),
    (("--rnn_cell",), dict(type=str, default="LSTM", choices=["LSTM", "GRU"])),
    (("--n_directions",), dict(type=int, default=1, choices=[1, 2])),
    (("--n_ts",), dict(type=int, default=5)),
    (("--share_params",), dict(type=str, default="Y", choices=YN)),                      # opts.py:43
    (("--use_target",), dict(type=str, default="none", choices=["none", "Sv", "uSv"])),
    (("--dis_DA",), dict(type=str, default="none", choices=["none", "DAN", "JAN", "CORAL"])),
    (("--adv_DA",), dict(type=str, default="none", choices=["none", "RevGrad"])),
    (("--use_bn",), dict(type=str, default="none", choices=["none", "AdaBN", "AutoDIAL"])),
    (("--ens_DA",), dict(type=str, default="none", choices=["none", "MCD"])),
    (("--use_attn_frame",), dict(type=str, default="none", choices=["none", "TransAttn", "general", "DotProduct"])),
    (("--use_attn",), dict(type=str, default="none", choices=["none", "TransAttn", "general", "DotProduct"])),
    (("--n_attn",), dict(type=int, default=1)),
    (("--add_loss_DA",), dict(type=str, default="none", choices=["none", "target_entropy", "attentive_entropy"])),
    (("--pred_normalize",), dict(type=str, default="N", choices=YN)),
    (("--alpha",), dict(type=float, default=1, metavar="M")),
    (("--beta",), dict(type=float, default=[1, 1, 1], nargs="+", metavar="M")),          # [relation, video, frame]
    (("--gamma",), dict(type=float, default=1, metavar="M")),
    (("--mu",), dict(type=float, default=0, metavar="M")),
    (("--weighted_class_loss_DA",), dict(type=str, default="N", choices=YN)),
    (("--place_dis",), dict(type=str, default=["Y", "Y", "N"], nargs="+", metavar="N")),
    (("--place_adv",), dict(type=str, default=["Y", "Y", "Y"], nargs="+", metavar="N")),
    (("--pretrain_source",), dict(default=False, action="store_true")),                  # opts.py:72
    (("--epochs",), dict(type=int, default=100, metavar="N")),
    (("-b", "--batch_size"), dict(type=int, default=[32, 28, 64], nargs="+", metavar="N")),
    (("--lr", "--learning_rate"), dict(type=float, default=0.0001, metavar="LR")),
    (("--lr_decay",), dict(type=float, default=10, metavar="LRDecay")),
    (("--lr_adaptive",), dict(type=str, default="none", choices=["none", "loss", "dann"])),
    (("--lr_steps",), dict(type=float, default=[60, 100], nargs="+", metavar="LRSteps")),
    (("--momentum",), dict(type=float, default=0.9, metavar="M")),
    (("--weight_decay", "--wd"), dict(type=float, default=1e-4, metavar="W")),
    (("--clip_gradient", "--gd"), dict(type=float, default=20, metavar="W")),
    (("--no_partialbn", "--npb"), dict(default=True, action="store_true")),
    (("--copy_list",), dict(type=str, default=["N", "Y"], nargs="+", metavar="N")),
    (("--print_freq", "-pf"), dict(type=int, default=10, metavar="N")),                  # opts.py:94
    (("--show_freq", "-sf"), dict(type=int, default=10, metavar="N")),
    (("--eval_freq", "-ef"), dict(type=int, default=1, metavar="N")),
    (("--verbose",), dict(default=False, action="store_true")),
    (("-j", "--workers"), dict(type=int, default=8, metavar="N")),
    (("--resume",), dict(type=str, default="", metavar="PATH")),
    (("--resume_hp",), dict(default=False, action="store_true")),
    (("-e", "--evaluate"), dict(dest="evaluate", action="store_true")),
    (("--exp_path",), dict(type=str, default="")),
    (("--gpus",), dict(nargs="+", type=int, default=None)),
    (("--flow_prefix",), dict(type=str, default="")),
    (("--save_model",), dict(default=False, action="store_true")),
    (("--save_best_log",), dict(type=str, default="best.log")),
    (("--save_attention",), dict(type=int, default=-1)),
    (("--tensorboard",), dict(dest="tensorboard", action="store_true")),
]
for _names, _kw in _FLAGS:
    parser.add_argument(*_names, **_kw)
