"""utils/utils.py of the reference: `randSelectBatch` (imported by main.py, never
called) and `plot_confusion_matrix` (evaluation plotting, out of the hot path)."""
import torch


def randSelectBatch(input, num):
    """utils/utils.py:8-11: (ids, input[ids]) for `num` random rows."""
    id_all = torch.randperm(input.size(0)).to(input.device)
    id = id_all[:num]
    return id, input[id]


def plot_confusion_matrix(*args, **kwargs):
    raise NotImplementedError("plotting is outside the TA3N train-step path (SURVEY.md 2 row 12)")
