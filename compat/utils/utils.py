from ta3n_amd.utils.utils import plot_confusion_matrix, randSelectBatch  # noqa: F401
