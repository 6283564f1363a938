"""utils/utils.py of the reference: `randSelectBatch` (imported by main.py, never
called) and `plot_confusion_matrix` (evaluation plotting, out of the hot path)."""
import torch


def randSelectBatch(input, num):
    """utils/utils.py:8-11."""
    id_all = torch.randperm(input.size(0))
    if input.is_cuda:
        id_all = id_all.to(input.device)
    return input[id_all[:num]]


def plot_confusion_matrix(*args, **kwargs):
    raise NotImplementedError("plotting is outside the TA3N train-step path (SURVEY.md 2 row 12)")
