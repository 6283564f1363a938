"""utils/utils.py of the reference: `randSelectBatch` (imported by main.py, never called) and `plot_confusion_matrix`
(evaluation plotting for test_models.py)."""
import torch


def randSelectBatch(input, num):
    """utils/utils.py:8-11: (ids, input[ids]) for `num` random rows."""
    id_all = torch.randperm(input.size(0)).to(input.device)
    id = id_all[:num]
    return id, input[id]


def plot_confusion_matrix(path, cm, classes, normalize=False, title='Confusion matrix', cmap=None):
    """utils/utils.py:13-43: the confusion matrix `cm` ([true class][predicted class] counts) as an annotated heat map written
    to `path`; normalize=True divides every row by the number of videos of that class (rows without videos stay zero) and
    annotates in percent.  matplotlib is imported here, not at module import: training never needs it."""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    import numpy as np
    cm = np.asarray(cm)
    per_class = np.maximum(cm.sum(axis=1), 1)
    shown = cm.astype(float) / per_class[:, None] if normalize else cm
    print("Normalized confusion matrix" if normalize else "Confusion matrix, without normalization")
    fig, ax = plt.subplots(figsize=(13, 10))
    im = ax.imshow(shown, interpolation='nearest', cmap=cmap if cmap is not None else plt.cm.Blues)
    ax.set_title(title)
    fig.colorbar(im, ax=ax)
    ticks = np.arange(len(classes))
    ax.set_xticks(ticks); ax.set_xticklabels(classes, rotation=90)
    ax.set_yticks(ticks); ax.set_yticklabels(classes)
    half = shown.max() / 2.0 if shown.size else 0.0
    for (i, j), val in np.ndenumerate(shown):
        text = format(val * 100, '.0f') if normalize else format(int(val), 'd')
        ax.text(j, i, text, ha="center", va="center", color="white" if val > half else "black")
    ax.set_ylabel('True label'); ax.set_xlabel('Predicted label')
    fig.tight_layout()
    fig.savefig(path)
    plt.close(fig)
