"""The reference's checkpoint file (main.py:266-274, 764-770), written and read by the engine path.

Format: torch.save({'epoch', 'arch', 'state_dict', 'optimizer', 'best_prec1', 'prec1'}) to
<exp_path>/checkpoint.pth.tar, copied to model_best.pth.tar when it is the best so far.  `state_dict` keys carry
nn.DataParallel's `module.` prefix (main.py:79, 270; test_models.py:89 strips it again), `optimizer` is a
torch.optim.SGD state_dict whose parameter order is VideoModel.parameters() (main.py:83) with the momentum buffers of the
parameters that have received a gradient - so the reference's own `test_models.py` and `main.py --resume [--resume_hp]`
read a file written here, and a file written by the reference resumes here."""
from __future__ import annotations

import os
import shutil
from typing import Dict, List, Optional

import torch


def optimizer_state_dict(param_names: List[str], momentum: Dict[str, torch.Tensor], lr: float, mu: float, weight_decay: float) -> dict:
    """torch.optim.SGD(nesterov=True).state_dict() for parameters in `param_names` order (= model.parameters() order)."""
    state = {i: {"momentum_buffer": momentum[n].detach().cpu().clone()} for i, n in enumerate(param_names) if n in momentum}
    group = dict(lr=lr, momentum=mu, dampening=0, weight_decay=weight_decay, nesterov=True, maximize=False, foreach=None,
                 differentiable=False, fused=None, params=list(range(len(param_names))))
    return {"state": state, "param_groups": [group]}


def save_checkpoint(state: dict, is_best: bool, path_exp: str, filename: str = "checkpoint.pth.tar") -> str:
    """main.py:764-770."""
    os.makedirs(path_exp, exist_ok=True)
    path_file = os.path.join(path_exp, filename)
    torch.save(state, path_file)
    if is_best:
        shutil.copyfile(path_file, os.path.join(path_exp, "model_best.pth.tar"))
    return path_file


def engine_checkpoint(eng, model, epoch: int, arch: str, lr: float, best_prec1: float, prec1: float) -> dict:
    """Checkpoint dict of a TrainEngine run.  `model` is the VideoModel the engine was initialised from: it supplies the
    parameter order of the optimizer entry and the BatchNorm buffers that are part of the reference's state_dict."""
    eng.flush()
    sd = {"module." + k: v.cpu() for k, v in eng.state_dict().items()}
    for k, v in model.state_dict().items():
        sd.setdefault("module." + k, v.detach().cpu())
    names = [n for n, _ in model.named_parameters()]
    mom = eng.momentum_views() if eng.step_count > 0 else {}      # torch.optim.SGD creates a buffer at a parameter's first update
    return {"epoch": epoch, "arch": arch, "state_dict": sd,
            "optimizer": optimizer_state_dict(names, mom, lr, eng.momentum, eng.weight_decay),
            "best_prec1": float(best_prec1), "prec1": float(prec1)}


def load_into_engine(eng, model, checkpoint: dict, resume_hp: bool = False) -> Dict[str, float]:
    """main.py:94-106: parameters always, optimizer state (momentum buffers, lr) only with --resume_hp.
    Returns {'start_epoch', 'best_prec1', 'lr' (None unless resume_hp)}."""
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in checkpoint["state_dict"].items()}
    eng.load_state(sd)
    out = {"start_epoch": int(checkpoint["epoch"]) + 1, "best_prec1": float(checkpoint.get("best_prec1", 0.0)), "lr": None}
    if resume_hp and "optimizer" in checkpoint:
        names = [n for n, _ in model.named_parameters()]
        views = eng.momentum_views()
        for i, st in checkpoint["optimizer"].get("state", {}).items():
            n = names[int(i)]
            if n in views and "momentum_buffer" in st and st["momentum_buffer"] is not None:
                views[n].copy_(st["momentum_buffer"].to(views[n].device, torch.float32))
        out["lr"] = float(checkpoint["optimizer"]["param_groups"][0]["lr"])
    return out
