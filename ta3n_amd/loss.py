"""Loss helpers with the reference's names (loss.py) for code that assembles the loss
itself, e.g. the reference's main.py (main.py:439-562) running on top of
ta3n_amd.models.VideoModel.  They are a few elementwise ops on [B,C] / [B,2] logits.
The fused train step (ta3n_amd.engine.TrainEngine) does NOT use them: there the
whole loss assembly and its gradients are one HIP kernel (csrc/ta3n_pointwise.hip)."""
import torch
import torch.nn.functional as F


def cross_entropy_soft(pred):
    """loss.py:8-12."""
    return torch.mean(torch.sum(-F.softmax(pred, 1) * F.log_softmax(pred, 1), 1))


def attentive_entropy(pred, pred_domain):
    """loss.py:15-25: mean((1 + H(softmax(pred_domain))) * H(softmax(pred)))."""
    weights = 1 + torch.sum(-F.softmax(pred_domain, 1) * F.log_softmax(pred_domain, 1), 1)
    return torch.mean(weights * torch.sum(-F.softmax(pred, 1) * F.log_softmax(pred, 1), 1))


def dis_MCD(out1, out2):
    """loss.py:29-30."""
    return torch.mean(torch.abs(F.softmax(out1, dim=1) - F.softmax(out2, dim=1)))
