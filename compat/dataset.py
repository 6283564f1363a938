"""Import-name shim: the reference's scripts do `import dataset` / `from dataset import ...`
(main.py:12-16, models.py:8).  Put this directory first on sys.path and they pick up
the MI355X implementation unchanged."""
from ta3n_amd.dataset import *  # noqa: F401,F403
from ta3n_amd.dataset import TSNDataSet, VideoRecord  # noqa: F401
