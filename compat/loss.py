"""Import-name shim: the reference's scripts do `import loss` / `from loss import ...`
(main.py:12-16, models.py:8).  Put this directory first on sys.path and they pick up
the MI355X implementation unchanged."""
from ta3n_amd.loss import *  # noqa: F401,F403
